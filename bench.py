#!/usr/bin/env python3
"""Headline benchmark: pose-candidates x beams scored per second (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|cfg2|cfg4]

A "step" is one pass of the hot path -- ScanMatcherNDT::matchScan's search (reference
src/scan_matcher_ndt.cpp:103-143) -- over one lattice, inputs resident in HBM.

N = 1 (workload "auto" -> cfg-2, BASELINE.json configs[1]): one 720-beam scan against
the 41x41 NDT @0.25 m, +-1.0 m / 0.02 m x +-0.5 rad / 0.005 rad = 100 x 100 x 200
candidates = 1.44e9 candidate-beam units per step.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL; workload "auto" ->
cfg-4, BASELINE.json configs[3]): the global loop-closure search +-5 m / 0.02 m x
+-pi / 0.005 rad = 501 x 501 x 1257 = 315,508,257 candidates = 2.27e11 units per step,
STRONG scaling: the theta steps are dealt round-robin to the ranks, every rank reduces
its share on its GPU to one 12-double record and the step ends with the single
all-reduce of the [N, 12] record table (it runs on RCCL's stream while the next step's
search runs; all of them have completed when the timed region ends).  The same line
carries `particle_filter` = BASELINE.json configs[4] (cfg-5): 1,000,000 particles x 720
beams on the 801x801 NDT, particles sharded contiguously, one all-reduce of the [N, 8]
moment sums (the total particle weight of ParticleFilter::updateStatistics) -- the ONE
collective of a step; the statistics then run on the device, and a rank's share of the theta
variance (the reference's second pass, src/particle_filter.cpp:213-217) stays with its
weights: whoever collects the weights (the resampler) adds the N shares in rank order.

Rank 0 prints one JSON line: the contract's fields and one summary per side leg (compact_line);
the full record -- per-class issue table, per-share arrays, probe output -- goes to
bench_detail.json (--detail-file) and stderr.  `roofline` is the resource that binds the dominant kernel
-- VALU issue -- as a fraction <= 1; `roofline_hbm` holds the measured HBM-side traffic
against the 8 TB/s peak and, separately, the declared algorithmic 64 B/unit figure.
"""
import argparse
import json
import os
import statistics
import sys
import time

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)

BYTES_PER_UNIT = 64.0        # 16 B beam endpoint + 48 B cell record
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MAX_CLOCK_HZ = 2.4e9         # MI355X_MICROARCH.md chip table
VALU_CYCLES_PER_INST = 4.0   # FP64 / CVT / compare / select / VOP3 wave64 instructions: 4 cycles measured (profiles/r05_ubench_issue.json)
ISSUE_FILE = os.path.join(_ROOT, "profiles", "r05_ubench_issue.json")   # experiments/ubench_issue.hip
MIX_FILE = os.path.join(_ROOT, "profiles", "r06_valu_mix.json")         # experiments/asm_loop_mix.py (static, per kernel)
# the headline kernel's DYNAMIC mix: path counts x the paths' instruction lists (experiments/lane_path_mix.py)
PATH_MIX_FILE = os.path.join(_ROOT, "profiles", "r06_lane_path_mix.json")
FP64_PEAK_TFLOPS = 78.6      # vector FP64
PMC_CANDIDATES = ("r06_pmc.json", "r05_pmc.json", "r04_pmc.json", "r03_pmc.json", "r02_pmc.json")   # the newest committed counter passes win
PMC_NAME = next((n for n in PMC_CANDIDATES if os.path.exists(os.path.join(_ROOT, "profiles", n))),
                PMC_CANDIDATES[0])
PMC_FILE = os.path.join(_ROOT, "profiles", PMC_NAME)
CPU_BASELINE_FILE = os.path.join(_ROOT, "profiles", "r02_cpu_baselines.json")
PREWARM_SECONDS = 0.5        # untimed launches before --warmup: the chip reaches its sustained clocks


# --------------------------------------------------------------------------------------
# CPU baseline (the oracle = CPU restatement of the reference; checker code, used here as
# the reported baseline only -- never on the product path)
# --------------------------------------------------------------------------------------

def granted_cpus():
    """CPUs this process may actually use: the cgroup quota when there is one (the GPU
    box shows 256 logical CPUs and grants 16; 256 OpenMP threads then run slower than 16)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(params, scans, guess, pts):
    """SURVEY.md 8(d) "CPU path timing" on this box's host cores: the oracle's matchScan
    on the cfg-2 lattice.  All cores the container grants (cgroup quota): the FULL lattice
    (1.44e9 units), 1 warm-up + median of 5, the (theta, dx) strips dealt to OpenMP threads.  Single thread (the reference's
    own execution model): every 4th theta of the same lattice (3.6e8 units), 1 warm-up +
    median of 3 -- the full-lattice single-thread medians (8 s per run) are in the
    committed profiles/r02_cpu_baselines.json, quoted under `committed`."""
    sys.path.insert(0, os.path.join(_ROOT, "tests"))
    import oracle_lib as O

    def matcher(p):
        ref = O.ScanMatcherNDT()
        ref.initialize(**p)
        ref.addScans(scans)
        return ref

    def units(p):
        n_th = len(O.search_offsets(p["search_angular_size"], p["search_angular_resolution"]))
        n_lin = len(O.search_offsets(p["search_linear_size"], p["search_linear_resolution"]))
        return n_th * n_lin * n_lin * min(p["laser_max_beams"], len(pts)), (n_th, n_lin)

    cores = granted_cpus()
    ref = matcher(params)
    u_all, (n_th, n_lin) = units(params)
    times, used = [], 0
    for i in range(6):
        t0 = time.perf_counter()
        r = ref.matchScan(guess, pts, omp_threads=cores)
        if i > 0:
            times.append(time.perf_counter() - t0)
        used = r["threads_used"]
    t_all = statistics.median(times)

    p1 = dict(params)
    p1["search_angular_resolution"] = params["search_angular_resolution"] * 4
    ref1 = matcher(p1)
    u_one, (n_th1, _) = units(p1)
    times1 = []
    for i in range(4):
        t0 = time.perf_counter()
        ref1.matchScan(guess, pts)
        if i > 0:
            times1.append(time.perf_counter() - t0)
    t_one = statistics.median(times1)
    out = {
        "value": u_all / t_all, "unit": "candidate-beams/s", "cores": used, "kind": "port",
        "sample": "full cfg-2 lattice %dx%dx%d x %d beams = %.3g units, oracle matchScan, "
                  "(theta, dx) strips over %d OpenMP threads, 1 warm-up + median of 5 (%.3f s)"
                  % (n_th, n_lin, n_lin, min(params["laser_max_beams"], len(pts)), u_all, used, t_all),
        "single_thread_value": u_one / t_one,
        "single_thread_sample": "%dx%dx%d candidates (every 4th theta), 1 thread -- the reference's own "
                                "execution model -- 1 warm-up + median of 3 (%.3f s)"
                                % (n_th1, n_lin, n_lin, t_one),
        "host_logical_cpus": os.cpu_count(), "cpus_granted_by_cgroup": cores,
    }
    try:
        with open(CPU_BASELINE_FILE) as f:
            out["committed"] = {"file": "profiles/r02_cpu_baselines.json", "cfg2": json.load(f)["cfg2"]}
    except (OSError, KeyError, ValueError):
        pass
    return out


# --------------------------------------------------------------------------------------
# Roofline from the committed PMC passes + the live kernel time
# --------------------------------------------------------------------------------------

def load_pmc():
    try:
        with open(PMC_FILE) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def source_hash():
    """sha256 over the kernel sources (ndt_2d_amd/csrc/*.hip, *.h, sorted by name): the profile
    script stores it in pmc.json, so that counters committed for other kernels than the ones
    being timed are noticed (`pmc_matches_source`)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_ROOT, "ndt_2d_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h"))):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def library_identity():
    """Which sources the loaded libndt2d_hip.so was compiled from (ndt2d_build_info) and whether
    that is the tree this script runs from (ndt_2d_amd/build.py source_sha256)."""
    try:
        from ndt_2d_amd import _capi
        from ndt_2d_amd import build as _build
        have, want, info = _capi.lib_source_sha256(), _build.source_sha256(), _capi.build_info()
        # (hooks=1 marks the test build with fault-injection hooks: same sources, never the product)
        return {"path": _capi.LIB_PATH, "build_info": info, "source_sha256_of_tree": want,
                "lib_matches_source": have == want and " hooks=1" not in info}
    except Exception as exc:   # noqa: BLE001 -- a missing figure, not a failed bench
        return {"error": str(exc), "lib_matches_source": None}


def pmc_matches_source(pmc):
    return bool(pmc) and pmc.get("source_sha256") == source_hash()


SHARE_COUNTERS = ("SQ_INSTS_VALU", "SQ_BUSY_CU_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU",
                  "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                  "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32",
                  "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT",
                  "FETCH_SIZE", "WRITE_SIZE")


def share_counters(pmc, workload, rank, world):
    """Counters of ONE rank's share of an 8-GPU workload, from the committed per-share passes
    (pmc["shares"][workload] = 8 dicts, share r of 8 = theta steps r, r + 8, ... of cfg-4 /
    particle range r of cfg-5, each measured as one launch on one GPU by
    experiments/profile_r04.sh).  A rank of a world that divides 8 holds the shares rank,
    rank + world, ...: instruction counts add (the per-launch set-up, repeated in each of the
    summed launches, is < 0.1 % of them)."""
    shares = (pmc or {}).get("shares", {}).get(workload)
    if not shares or len(shares) != 8 or 8 % world != 0:
        return None
    mine = [shares[r] for r in range(rank, 8, world)]
    out = {}
    for c in SHARE_COUNTERS:
        if all(c in sh for sh in mine):
            out[c] = sum(sh[c] for sh in mine)
    out["shares_summed"] = list(range(rank, 8, world))
    return out


_ISSUE_CACHE = {}


def _load_json(path):
    if path not in _ISSUE_CACHE:
        try:
            with open(path) as f:
                _ISSUE_CACHE[path] = json.load(f)
        except (OSError, ValueError):
            _ISSUE_CACHE[path] = None
    return _ISSUE_CACHE[path]


# counter -> (class of profiles/r05_valu_mix.json, the ubench instruction whose cycles price it
# when the kernel's own mix is unknown)
_CLASS_COUNTERS = (("SQ_INSTS_VALU_ADD_F64", "ADD_F64", "v_add_f64"), ("SQ_INSTS_VALU_MUL_F64", "MUL_F64", "v_mul_f64"),
                   ("SQ_INSTS_VALU_FMA_F64", "FMA_F64", "v_fma_f64"), ("SQ_INSTS_VALU_TRANS_F64", "TRANS_F64", "v_rcp_f64"),
                   ("SQ_INSTS_VALU_CVT", "CVT", "v_cvt_i32_f64"), ("SQ_INSTS_VALU_INT64", "INT64", "v_mad_u64_u32"),
                   ("SQ_INSTS_VALU_INT32", "INT32", None), ("SQ_INSTS_VALU_ADD_F32", "ADD_F32", "v_add_f32"),
                   ("SQ_INSTS_VALU_MUL_F32", "MUL_F32", "v_mul_f32"), ("SQ_INSTS_VALU_FMA_F32", "FMA_F32", "v_pk_fma_f32"),
                   ("SQ_INSTS_VALU_TRANS_F32", "TRANS_F32", "v_rcp_f32"))


def class_pricing(kernel, k):
    """Issue cycles of one launch from its per-class instruction counts: {"cycles", "cycles_low",
    "cycles_high", "table"}.  A class's cycles per instruction: the kernel's own static mean
    (profiles/r05_valu_mix.json) when it lists the class, else the measured cycles of the
    class's instruction (profiles/r05_ubench_issue.json), else 4.  The two classes that mix 2-
    and 4-cycle instructions (INT32, and "other" = SQ_INSTS_VALU minus all classes) span 2 .. 4
    in the bracket."""
    issue = (_load_json(ISSUE_FILE) or {}).get("cycles", {})
    mix = ((_load_json(MIX_FILE) or {}).get("kernels", {}).get(kernel, {}) or {}).get("classes", {})
    # Round 6, the headline kernel: the two mixed classes are no longer priced by a STATIC count of the
    # kernel's instructions -- a build with path counters says how often a wave takes each path of the
    # search (profiles/r06_lane_paths.json), the paths' instruction lists say which forms those are, and
    # only the per-item remainder no counter covers (12 % of the launch) is bracketed 2 .. 4 cycles.
    dyn = {}
    if kernel == "match_lane_compact_kernel" and counters_are_whole_cfg2(k):
        dyn = (_load_json(PATH_MIX_FILE) or {}).get("mixed_classes", {})
    total = k["SQ_INSTS_VALU"]
    table, cycles, low, high, classified = [], 0.0, 0.0, 0.0, 0.0
    for counter, cls, rep in _CLASS_COUNTERS:
        n = k.get(counter)
        if not n:
            continue
        n = min(n, total - classified)
        classified += n
        c = mix.get(cls, {}).get("mean_cycles") or issue.get(rep or "", VALU_CYCLES_PER_INST)
        mixed = cls == "INT32"
        c_low, c_high, source = (2.0 if mixed else c), (4.0 if mixed else c), ("kernel's static mix" if cls in mix else "ubench")
        if mixed and "INT32" in dyn:
            c, c_low, c_high = dyn["INT32"]["mean_cycles"], dyn["INT32"]["mean_cycles_low"], dyn["INT32"]["mean_cycles_high"]
            source = "path counts x path instruction lists (profiles/r06_lane_path_mix.json)"
        cycles += n * c
        low += n * c_low
        high += n * c_high
        table.append({"class": cls, "instructions": n, "share": n / total, "cycles_per_instruction": c,
                      "source": source})
    rest = max(total - classified, 0.0)
    if rest > 0:
        c = mix.get("other", {}).get("mean_cycles") or VALU_CYCLES_PER_INST
        c_low, c_high, source = 2.0, 4.0, ("kernel's static mix" if "other" in mix else "assumed")
        if "other" in dyn:
            c, c_low, c_high = dyn["other"]["mean_cycles"], dyn["other"]["mean_cycles_low"], dyn["other"]["mean_cycles_high"]
            source = "path counts x path instruction lists (profiles/r06_lane_path_mix.json)"
        cycles += rest * c
        low += rest * c_low
        high += rest * c_high
        table.append({"class": "other (no class counter: moves, selects, compares, permutes, v_ldexp_f64 ...)",
                      "instructions": rest, "share": rest / total, "cycles_per_instruction": c,
                      "source": source})
    return {"cycles": cycles, "cycles_low": low, "cycles_high": high, "table": table}


def counters_are_whole_cfg2(k):
    """The path mix was taken on the whole cfg-2 lattice: it prices THAT launch's counters (the file
    names the instruction count it was checked against), not a share of cfg-4."""
    doc = _load_json(PATH_MIX_FILE) or {}
    want = (doc.get("pmc") or {}).get("SQ_INSTS_VALU")
    return bool(want) and abs(k.get("SQ_INSTS_VALU", 0.0) - want) <= 1e-6 * want


def roofline(kernel, kernel_ms, units, n_cu, pmc, expected_dispatch_note, counters=None):
    """(roofline, roofline_hbm) for `kernel`.

    VALU issue: achieved = VALU wave-instructions per launch (SQ_INSTS_VALU, committed PMC
    pass of this same workload; the count is a property of the workload, not of the run)
    / the kernel's average duration measured LIVE with HIP events on the launch stream;
    peak = the chip's fixed issue peak: SIMDs x 2.4 GHz (MI355X_MICROARCH.md chip table)
    / 4 cycles per FP64 / VOP3 wave-instruction = 614.4 G wave-instr/s for 256 CUs.  So
    `frac` moves with the live kernel time.  `issue_slot_occupancy_pmc` = SQ_INSTS_VALU x 4
    / (4 x SQ_BUSY_CU_CYCLES) is the run-invariant share of the kernel's own SIMD cycles
    in which a VALU instruction issues (reproducible from profiles/ alone); the two differ
    by sustained clock / 2.4 GHz."""
    t = kernel_ms * 1e-3
    alg_bytes = units * BYTES_PER_UNIT
    hbm = {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "achieved": None, "frac": None,
           "traffic": None,
           "algorithmic_bytes_per_launch": alg_bytes,
           "algorithmic_GBps": alg_bytes / t / 1e9,
           "algorithmic_over_peak": alg_bytes / t / 1e9 / HBM_PEAK_GBPS,
           "note": "achieved / frac = MEASURED HBM-side bytes (PMC) per launch / live kernel time; the "
                   "declared algorithmic 64 B/unit (SURVEY.md 8d) exceeds the HBM peak because beams "
                   "and grid are SGPR / LDS resident and empty cells are skipped -- not a roofline"}
    roof = {"bound": "valu_issue", "achieved": None, "peak": None, "unit": "G wave-instr/s",
            "frac": None, "traffic": None, "kernel": kernel, "kernel_ms_avg": kernel_ms}
    k = counters if counters is not None else (pmc or {}).get("kernels", {}).get(kernel)
    roof["pmc_matches_source"] = hbm["pmc_matches_source"] = pmc_matches_source(pmc)
    if not k or "SQ_INSTS_VALU" not in k:
        roof["note"] = "profiles/%s has no counters for this kernel" % PMC_NAME
        return roof, hbm
    n_simd = 4 * n_cu
    # cycles are the workload's (counted under the profiler), time is this run's: the clock
    # the chip sustained here is their ratio (it clocks to its power budget; the profiled
    # passes run ~6 % slower, MI355X_MICROARCH.md "DVFS")
    busy_cycles = k["SQ_BUSY_CU_CYCLES"] / n_cu            # per-CU busy cycles of one launch
    clock = busy_cycles / t
    dur_pmc = k.get("avg_duration_ns", {}).get("sq1")
    achieved = k["SQ_INSTS_VALU"] / t / 1e9
    # The peak is that of THIS kernel's instruction mix: every class the counters tell apart is
    # priced with the issue cycles measured for it on this chip (experiments/ubench_issue.hip ->
    # profiles/r05_ubench_issue.json: FP64 add / mul / fma, conversions, 64-bit integer, compares,
    # selects, v_perm_b32, DPP moves ... 4 cycles per wave64 instruction; v_mov_b32 / v_add_u32 /
    # v_sub_u32 / v_and|or|xor_b32 / v_lshrrev_b32 and FP32 add / mul / fmac 2; FP32 and FP64
    # transcendentals 8 / 16).  Two classes MIX 2- and 4-cycle instructions and no counter splits
    # them -- INT32 and the unclassified rest: priced with the mean over the kernel's own
    # instructions of that class inside loops (profiles/r05_valu_mix.json, static), and bracketed
    # by "all of them at 2 cycles" .. "all at 4" (frac_bracket).
    pricing = class_pricing(kernel, k)
    cycles_per_inst = pricing["cycles"] / k["SQ_INSTS_VALU"]
    peak = n_simd * MAX_CLOCK_HZ / cycles_per_inst / 1e9
    roof.update(
        achieved=achieved, peak=peak, frac=achieved / peak,
        frac_bracket=[k["SQ_INSTS_VALU"] * (pricing["cycles_low"] / k["SQ_INSTS_VALU"]) / t / (n_simd * MAX_CLOCK_HZ),
                      k["SQ_INSTS_VALU"] * (pricing["cycles_high"] / k["SQ_INSTS_VALU"]) / t / (n_simd * MAX_CLOCK_HZ)],
        peak_note="%d SIMDs x %.1f GHz / %.3f cycles per wave-instruction, priced per counter class with the issue cycles "
                  "measured in profiles/r05_ubench_issue.json; the two classes that mix 2- and 4-cycle forms by the "
                  "kernel's own mix (headline kernel: measured path counts, profiles/r06_lane_path_mix.json; others: static, "
                  "profiles/r06_valu_mix.json); frac_bracket = what of those classes is not pinned down at 2 cycles .. at 4"
                  % (n_simd, MAX_CLOCK_HZ / 1e9, cycles_per_inst),
        issue_pricing=pricing["table"],
        valu_mix_matches_source=(_load_json(MIX_FILE) or {}).get("source_sha256") == source_hash(),
        issue_slot_occupancy_pmc=k["SQ_INSTS_VALU"] * cycles_per_inst / (4.0 * k["SQ_BUSY_CU_CYCLES"]),
        sustained_clock_GHz_est=clock / 1e9,
        clock_in_pmc_pass_GHz=(busy_cycles / (dur_pmc * 1e-9) / 1e9) if dur_pmc else None,
        valu_insts_per_launch=k["SQ_INSTS_VALU"],
        valu_insts_per_unit=k["SQ_INSTS_VALU"] * 64.0 / units,
        source="profiles/%s (experiments/profile_r0N.sh) + live HIP-event kernel time; " % PMC_NAME
               + expected_dispatch_note)
    if "SQ_ACTIVE_INST_VALU" in k and "SQ_BUSY_CYCLES" in k:
        # SQ_ACTIVE_INST_VALU counts quad-cycles a SIMD spends issuing VALU work
        roof["valu_busy_pmc"] = k["SQ_ACTIVE_INST_VALU"] * 4.0 / (4.0 * k["SQ_BUSY_CU_CYCLES"])
    f64 = [k.get("SQ_INSTS_VALU_ADD_F64"), k.get("SQ_INSTS_VALU_MUL_F64"),
           k.get("SQ_INSTS_VALU_FMA_F64"), k.get("SQ_INSTS_VALU_TRANS_F64")]
    if None not in f64:
        flops = (f64[0] + f64[1] + 2.0 * f64[2] + f64[3]) * 64.0
        roof["fp64_flops_per_launch"] = flops
        roof["fp64_flops_frac"] = flops / t / 1e12 / FP64_PEAK_TFLOPS
        roof["fp64_share_of_valu_insts"] = sum(f64) / k["SQ_INSTS_VALU"]
    if "SQ_THREAD_CYCLES_VALU" in k and "SQ_ACTIVE_INST_VALU" in k and k["SQ_ACTIVE_INST_VALU"] > 0:
        roof["avg_active_lanes"] = k["SQ_THREAD_CYCLES_VALU"] / k["SQ_ACTIVE_INST_VALU"]
    if "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        # FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)
        traffic = (2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
        roof["traffic"] = traffic
        hbm.update(traffic=traffic, achieved=traffic / t / 1e9, frac=traffic / t / 1e9 / HBM_PEAK_GBPS,
                   traffic_bytes_per_unit=traffic / units,
                   traffic_source="(2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes, profiles/%s" % PMC_NAME)
    return roof, hbm


# --------------------------------------------------------------------------------------
# Secondary legs
# --------------------------------------------------------------------------------------

def particle_bench_1gpu(matcher_cls, synth, torch, device_index, pmc, n_cu, reps=10):
    """ParticleFilter::measure's scoring on BASELINE.json configs[2] ("cfg-3"): 100k
    particles x 720 beams, 201x201 NDT, one GPU."""
    m = matcher_cls(device_index)
    m.initialize("global_scan_matcher", **synth.matcher_params(3))
    m.addScans(synth.map_scans(3))
    _, pts, _ = synth.query_scan(3)
    parts = synth.particles(3)
    n_beams = m.prepare_beams(pts)
    dev = torch.device("cuda", device_index)
    d_parts = torch.from_numpy(parts).to(dev)
    d_scores = torch.zeros(len(parts), dtype=torch.float64, device=dev)
    d_stats = torch.zeros(8, dtype=torch.float64, device=dev)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(dev)
    m.set_stream(stream.cuda_stream)
    ms = []
    for i in range(reps + 2):
        m.score_poses_launch(d_parts.data_ptr(), len(parts), d_scores.data_ptr(), d_stats.data_ptr())
        t, _ = m.last_launch_ms()
        if i > 1:
            ms.append(t)
    units = len(parts) * n_beams
    avg = sum(ms) / len(ms)
    m.set_stream(None)
    out = {"workload": "cfg-3 (BASELINE.json configs[2]): 100000 particles x 720 beams, 201x201 NDT @0.25 m",
           "units_per_launch": units, "kernel_ms": avg, "value": units / (avg * 1e-3),
           "unit": "candidate-beams/s", "variant": m.last_variant()}
    roof, hbm = roofline("score_poses_compact_kernel", avg, units, n_cu, pmc,
                         "counters of the cfg-3 launch of the same bench command")
    out["roofline"], out["roofline_hbm"] = roof, hbm
    # PCIe-inclusive: the whole ParticleFilter::measure call, host particles in, host
    # weights out.  Pageable host memory (what a std::vector holds) and pinned host
    # memory (ndt2d_host_alloc; what the C++ mirror's particle store uses).
    w = None
    for label, pinned in (("pageable", False), ("pinned", True)):
        buf_p = m.host_alloc((len(parts), 3)) if pinned and hasattr(m, "host_alloc") else None
        if pinned and buf_p is None:
            continue
        src = parts
        dst = None
        if pinned:
            buf_p[:] = parts
            src = buf_p
            dst = m.host_alloc((len(parts),))
        calls = []
        for _ in range(12):
            t0 = time.perf_counter()
            w = m.scorePoses(pts, src, out=dst) if dst is not None else m.scorePoses(pts, src)
            calls.append(time.perf_counter() - t0)
        calls = sorted(calls[2:])
        out["host_call_%s_ms" % label] = calls[len(calls) // 2] * 1e3
    out["host_call_ms"] = out.get("host_call_pinned_ms", out["host_call_pageable_ms"])
    out["host_call_over_kernel"] = out["host_call_ms"] / avg
    out["host_call_value"] = units / (out["host_call_ms"] * 1e-3)
    del w
    m.close()
    return out


def cfg1_search_bench(matcher_cls, synth, device_index, with_cpu):
    """BASELINE.json configs[0], the reference's own CPU-runnable case (17,640 candidates x
    720 beams): the whole matchScan call on the GPU (host buffers in, results out, medians)
    and -- the reference's execution model -- the CPU oracle single-threaded on this box's
    host, in full, 1 warm-up + median of 5; same winner asserted."""
    params = synth.matcher_params(1)
    scans = synth.map_scans(1)
    guess, pts, _ = synth.query_scan(1)
    m = matcher_cls(device_index)
    m.initialize("cfg1", **params)
    m.addScans(scans)
    for _ in range(5):
        got = m.matchScan(guess, pts)
    kernel_ms, _ = m.last_launch_ms()
    variant = m.last_variant()
    m.set_timing(False)
    ts = []
    for _ in range(200):
        t0 = time.perf_counter()
        got = m.matchScan(guess, pts)
        ts.append(time.perf_counter() - t0)
    call_ms = statistics.median(ts) * 1e3
    m.close()
    units = 17640 * 720
    out = {"workload": "cfg-1 (BASELINE.json configs[0]): 21 x 21 x 40 = 17,640 candidates x 720 beams, "
                       "41x41 NDT @0.25 m", "units": units, "gpu_match_scan_ms": call_ms,
           "gpu_kernel_ms": kernel_ms, "gpu_value": units / (call_ms * 1e-3), "kernel_variant": variant,
           "score": got["score"], "pose": [float(v) for v in got["pose"]]}
    if with_cpu:
        sys.path.insert(0, os.path.join(_ROOT, "tests"))
        import oracle_lib as O
        ref = O.ScanMatcherNDT()
        ref.initialize(**params)
        ref.addScans(scans)
        tc = []
        for i in range(6):
            t0 = time.perf_counter()
            exp = ref.matchScan(guess, pts)
            if i > 0:
                tc.append(time.perf_counter() - t0)
        cpu_ms = statistics.median(tc) * 1e3
        if tuple(exp["pose"]) != tuple(got["pose"]) or abs(exp["score"] - got["score"]) > 1e-9:
            raise SystemExit("bench.py: cfg-1 result differs from the oracle's")
        out.update({"cpu_single_thread_ms": cpu_ms, "cpu_value": units / (cpu_ms * 1e-3),
                    "cpu_kind": "port (oracle/ndt2d_oracle.c, 1 thread: the reference's execution model)",
                    "gpu_over_cpu": cpu_ms / call_ms})
    return out


def default_search_bench(matcher_cls, synth, device_index, reps=300, with_cpu=True):
    """The node's actual default workload (reference src/scan_matcher_ndt.cpp:37-44): 100 of
    720 beams, 21 x 21 x 80 = 35,280 candidates, per accepted scan the mapper runs
    reset + addScans + scoreScan + matchScan (src/ndt_mapper.cpp:508-515).  Host-call
    latencies (host buffers in, results out), medians."""
    scans = synth.map_scans(1)
    params = synth.matcher_params(1, search_linear_size=0.05, search_linear_resolution=0.005,
                                  search_angular_size=0.1, search_angular_resolution=0.0025,
                                  laser_max_beams=100)
    guess, pts, _ = synth.query_scan(1)
    m = matcher_cls(device_index)
    m.initialize("local_scan_matcher", **params)
    m.addScans(scans)

    def med(fn, n=reps):
        ts = []
        for i in range(n + 20):
            t0 = time.perf_counter()
            fn()
            if i >= 20:
                ts.append(time.perf_counter() - t0)
        ts.sort()
        return ts[len(ts) // 2] * 1e3, ts[int(len(ts) * 0.99)] * 1e3

    for _ in range(5):
        m.matchScan(guess, pts)
    kernel_ms, n_kernels = m.last_launch_ms()   # HIP events around the search kernel
    variant = m.last_variant()
    m.set_timing(False)                         # as the pluginlib shim runs: no event pairs
    match_ms, match_p99 = med(lambda: m.matchScan(guess, pts))
    score_ms, _ = med(lambda: m.scoreScan(guess, pts))
    points_ms, _ = med(lambda: m.scorePoints(pts, guess))
    add_ms, _ = med(lambda: (m.reset(), m.addScans(scans)), n=100)

    def cycle():
        m.reset()
        m.addScans(scans)
        m.scoreScan(guess, pts)
        m.matchScan(guess, pts)
    cycle_ms, cycle_p99 = med(cycle, n=100)
    # ParticleFilter::measure through the UNCHANGED per-particle loop (reference
    # src/particle_filter.cpp:81-87): 500 scorePoints calls with the same points
    poses = synth.particles(3, 500) * [4.0 / 23.0, 4.0 / 23.0, 1.0]

    def loop():
        for p in poses:
            m.scorePoints(pts, p)
    loop_ms, _ = med(loop, n=10)
    batch_ms, _ = med(lambda: m.scorePoses(pts, poses), n=100)
    units = 35280 * 100
    m.close()
    # the same calls from a plain-C host (what the pluginlib shim pays: no interpreter)
    c_host = None
    probe = os.path.join(_ROOT, "ndt_2d_amd", "ndt2d_latency_probe")
    if os.path.exists(probe):
        import subprocess
        r = subprocess.run([probe], capture_output=True, text=True, timeout=300)
        if r.returncode == 0:
            c_host = json.loads(r.stdout.strip().splitlines()[-1])
    out = default_search_record(units, match_ms, match_p99, kernel_ms, n_kernels, variant, score_ms,
                                points_ms, add_ms, cycle_ms, cycle_p99, loop_ms, batch_ms)
    if with_cpu:
        # the reference's own execution of the same calls: the oracle, one thread, this box
        sys.path.insert(0, os.path.join(_ROOT, "tests"))
        import oracle_lib as O
        ref = O.ScanMatcherNDT()
        ref.initialize(**params)
        ref.addScans(scans)

        def med_cpu(fn, n):
            ts = []
            for i in range(n + 1):
                t0 = time.perf_counter()
                r = fn()
                if i > 0:
                    ts.append(time.perf_counter() - t0)
            return statistics.median(ts) * 1e3, r
        cpu_match_ms, exp = med_cpu(lambda: ref.matchScan(guess, pts), 5)
        cpu_score_ms, _ = med_cpu(lambda: ref.scoreScan(guess, pts), 50)
        cpu_add_ms, _ = med_cpu(lambda: (ref.reset(), ref.addScans(scans)), 20)
        # ParticleFilter::measure's loop (src/particle_filter.cpp:81-87) incl. the reference's
        # per-particle copy of the point vector (src/scan.cpp:67-70)
        cpu_loop_ms, _ = med_cpu(lambda: O.pf_measure(ref, poses, pts, copy_points=True), 20)
        out["cpu_single_thread"] = {
            "match_scan_ms": cpu_match_ms, "score_scan_ms": cpu_score_ms, "add_scans_ms": cpu_add_ms,
            "mapper_cycle_ms": cpu_match_ms + cpu_score_ms + cpu_add_ms,
            "measure_500_particles_ms": cpu_loop_ms,
            "kind": "port (oracle/ndt2d_oracle.c, 1 thread: the reference's execution model)",
            "match_scan_best_index": exp["best_index"],
            "note": "through the UNCHANGED per-particle loop the GPU plugin is slower than this CPU "
                    "path (one launch per particle); the batched entry point needs the one-line node "
                    "change of INTEGRATION.md section 4"}
    if with_cpu and c_host is not None and "real_lidar_map" in c_host:
        # the probe's second scenario -- a 30 m lidar's 245 x 245 local map -- on the CPU oracle
        import numpy as np
        w5 = synth.world_of(5)
        g5, pts5, true5 = synth.query_scan(5)
        scans5 = []
        for j in range(3):
            for i in range(3):
                x, y = true5[0] + 0.5 * (i - 1), true5[1] + 0.5 * (j - 1)
                if not synth.pose_blocked(w5, x, y):
                    scans5.append(((x, y, 0.0), synth.scan(w5, (x, y, 0.0), 77 + 10 * j + i)))
        p5 = dict(params, range_max=30.0)
        ref5 = O.ScanMatcherNDT()
        ref5.initialize(**p5)
        ref5.addScans(scans5)
        pose5 = true5 + np.array([0.02, -0.02, 0.01])
        cm, exp5 = med_cpu(lambda: ref5.matchScan(pose5, pts5), 3)
        cs, _ = med_cpu(lambda: ref5.scoreScan(pose5, pts5), 20)
        ca, _ = med_cpu(lambda: (ref5.reset(), ref5.addScans(scans5)), 10)
        out["cpu_single_thread"]["real_lidar_map"] = {
            "match_scan_ms": cm, "score_scan_ms": cs, "add_scans_ms": ca, "mapper_cycle_ms": cm + cs + ca,
            "match_scan_pose": [float(v) for v in exp5["pose"]], "match_scan_score": exp5["score"]}
        got5 = c_host["real_lidar_map"]
        if [float(v) for v in exp5["pose"]] != got5["check_pose"] or abs(exp5["score"] - got5["check_score"]) > 1e-9:
            raise SystemExit("bench.py: the real-lidar-map search differs from the oracle's")
    if c_host is not None:
        out["c_host"] = dict(c_host, what="ndt_2d_amd/tools/latency_probe.c: the same calls through the C-ABI "
                                          "from a C program, medians of 2000 (500 for addScans / the cycle); "
                                          "in the cycle scoreScan queues the scan's search behind itself once the "
                                          "library has seen the mapper's scoreScan / matchScan pair "
                                          "(mapper_cycle_no_search_ahead_us: with that turned off)")
        out["match_scan_ms"] = c_host["match_scan_us"] * 1e-3
        out["mapper_cycle_ms"] = c_host["mapper_cycle_us"] * 1e-3
        out["match_scan_value"] = units / (c_host["match_scan_us"] * 1e-6)
        out["headline_source"] = "c_host (python_host holds the ctypes figures)"
    return out


def default_search_record(units, match_ms, match_p99, kernel_ms, n_kernels, variant, score_ms, points_ms,
                          add_ms, cycle_ms, cycle_p99, loop_ms, batch_ms):
    python_host = {"match_scan_ms": match_ms, "match_scan_p99_ms": match_p99, "score_scan_ms": score_ms,
                   "score_points_call_us": points_ms * 1e3, "add_scans_ms": add_ms,
                   "mapper_cycle_ms": cycle_ms, "mapper_cycle_p99_ms": cycle_p99,
                   "measure_500_particles_unchanged_loop_ms": loop_ms,
                   "measure_500_particles_batched_ms": batch_ms,
                   "note": "the calls made from Python: ctypes overhead (~3-5 us per call) included"}
    return {"workload": "plugin defaults: 100 of 720 beams, 21x21x80 = 35,280 candidates, 41x41 NDT from 9 scans",
            "match_scan_ms": match_ms, "mapper_cycle_ms": cycle_ms,
            "match_scan_value": units / (match_ms * 1e-3),
            "match_scan_kernel_ms": kernel_ms, "kernels_per_call": n_kernels, "kernel_variant": variant,
            "headline_source": "python_host", "python_host": python_host}


def in_grid_share(np, synth, params, guess, pts, grid, samples=2000000):
    """Share of the (candidate, beam) pairs of a lattice whose point lies inside the NDT
    grid (the others are exact zeros in the reference too, src/ndt_model.cpp:165-169)."""
    from ndt_2d_amd.scan_matcher import search_offsets
    dth = search_offsets(params["search_angular_size"], params["search_angular_resolution"])
    dlin = search_offsets(params["search_linear_size"], params["search_linear_resolution"])
    rng = np.random.default_rng(4)
    t = dth[rng.integers(0, len(dth), samples)] + guess[2]
    dx = dlin[rng.integers(0, len(dlin), samples)]
    dy = dlin[rng.integers(0, len(dlin), samples)]
    b = pts[rng.integers(0, len(pts), samples)]
    x = b[:, 0] * np.cos(t) - b[:, 1] * np.sin(t) + guess[0] + dx
    y = b[:, 0] * np.sin(t) + b[:, 1] * np.cos(t) + guess[1] + dy
    _, sx, sy, cs, ox, oy = grid
    inside = (x >= ox) & (y >= oy) & ((x - ox) / cs < sx) & ((y - oy) / cs < sy)
    return float(inside.mean())


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher: the same job the driver starts with
    torch.distributed.run, as a child process (one rank per GPU, rendezvous on 127.0.0.1,
    a free port); its stdout -- the one JSON line of rank 0 -- is relayed, its return
    code becomes ours."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    sys.stdout.write(proc.stdout.decode(errors="replace"))
    sys.stdout.flush()
    return proc.returncode


def gpu_clock_mhz(torch, device_index=0):
    """The shader clock the driver reports for THIS GPU right now (sysfs pp_dpm_sclk of its
    PCI function, the level marked '*'), or None.  A host shows many cards: the device is
    found by its PCI address."""
    try:
        pr = torch.cuda.get_device_properties(device_index)
        addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        with open("/sys/bus/pci/devices/%s/pp_dpm_sclk" % addr) as f:
            for ln in f:
                if "*" in ln:
                    return float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
    except (OSError, ValueError, IndexError, AttributeError, RuntimeError):
        pass
    return None


def cfg4_single_gpu_bench(matcher_cls, synth, shard, np, torch, dev_index, steps=3):
    """BASELINE.json configs[3] (cfg-4, the 8-GPU loop-closure lattice) on ONE GPU: the anchor
    a 1 -> N curve of `--gpus N` lines is read against.  Whole lattice per step, `steps`
    timed steps after one warm-up; the oracle's whole-lattice winner is asserted."""
    params = synth.matcher_params(4)
    guess, pts, _ = synth.query_scan(4)
    m = matcher_cls(dev_index)
    m.initialize("global_scan_matcher", **params)
    m.addScans(synth.map_scans(4))
    n_th, n_lin, n_beams = m.prepare_search(guess, pts)
    units = n_th * n_lin * n_lin * n_beams
    ms, kernel_ms = [], []
    for i in range(steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.match_launch(0, n_th)
        rec = m.match_fetch()
        if i > 0:
            ms.append((time.perf_counter() - t0) * 1e3)
            kernel_ms.append(m.last_launch_ms()[0])
    res = m.finish_match(rec)
    variant = m.last_variant()
    best_index = int(rec[1])
    # The eight 1-of-8 shares an 8-GPU run deals out (theta steps r, r + 8, ...), each run ALONE on this
    # GPU: what every rank of `--gpus 8` would spend in its search -- the step of such a run is the
    # slowest share plus the exchange of the records (<= 768 bytes).  No 8-GPU node has been available:
    # this is the measured part of the scaling claim.
    share_ms, share_kernel_ms = [], []
    for r in range(8):
        first, stride, count = shard.shard_strided(n_th, r, 8)
        best = None
        for _ in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.match_launch_strided(first, stride, count)
            m.match_fetch()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        share_ms.append(best)
        share_kernel_ms.append(m.last_launch_ms()[0])
    expect = 80443810   # tests/golden/big_winners.json cfg4: the oracle over all 315,508,257 candidates
    if best_index != expect:
        raise SystemExit("bench.py: cfg-4 winner %d differs from the oracle's %d" % (best_index, expect))
    share = in_grid_share(np, synth, params, guess, pts, m.grid())
    m.close()
    t = statistics.median(ms)
    return {"workload": "cfg-4 (BASELINE.json configs[3]) on one GPU: 501 x 501 x 1257 = 315,508,257 candidates "
                        "x 720 beams, the whole lattice per step",
            "units_per_step": units, "steps": steps, "ms_per_step": t, "kernel_ms": statistics.median(kernel_ms),
            "value": units / (t * 1e-3), "unit": "candidate-beams/s",
            "in_grid_share_of_units": share, "value_in_grid_units": units * share / (t * 1e-3),
            "best_index": best_index, "score": res["score"], "pose": [float(v) for v in res["pose"]],
            "kernel_variant": variant,
            "eight_shares_alone_on_this_gpu": {
                "ms": share_ms, "kernel_ms": share_kernel_ms, "max_ms": max(share_ms),
                "max_over_mean": max(share_ms) / (sum(share_ms) / 8.0),
                "whole_over_slowest_share": t / max(share_ms),
                "what": "each 1-of-8 share (theta steps r, r + 8, ...) searched alone on this GPU, launch to result, "
                        "best of 2: an 8-GPU step is the slowest share plus the record exchange"},
            "what": "the N = 1 point of the strong-scaling curve `bench.py --gpus N` (N > 1) reports"}


# --------------------------------------------------------------------------------------
# The stdout line: the driver's contract and one summary per side leg, nothing else
# --------------------------------------------------------------------------------------

MAX_LINE_BYTES = 12288       # the driver stops parsing somewhere below 32 KiB (round 5: a 33 KB line, parsed: null)
DETAIL_NAME = "bench_detail.json"


def _pick(src, keys):
    return {k: src[k] for k in keys if isinstance(src, dict) and k in src}


def _roof_short(r):
    """A roofline object cut to its numbers (the per-class issue table and the notes stay in
    the detail file)."""
    if not isinstance(r, dict):
        return None
    out = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "frac_bracket", "traffic", "kernel",
                    "kernel_ms_avg", "pmc_matches_source", "valu_insts_per_launch", "valu_insts_per_unit",
                    "fp64_share_of_valu_insts", "issue_slot_occupancy_pmc"))
    if "source" in r:
        out["source"] = str(r["source"])[:160]
    return out


def _hbm_short(h):
    if not isinstance(h, dict):
        return None
    return _pick(h, ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                     "algorithmic_GBps", "algorithmic_over_peak"))


def _cpu_short(c):
    if not isinstance(c, dict):
        return None
    out = _pick(c, ("value", "unit", "cores", "kind", "single_thread_value"))
    if "sample" in c:
        out["sample"] = str(c["sample"])[:240]
    return out


def _leg(src, keys):
    """One side leg: the named scalars + the roofline fraction, no text."""
    if not isinstance(src, dict):
        return None
    out = _pick(src, keys)
    if isinstance(src.get("roofline"), dict):
        out["roofline_frac"] = src["roofline"].get("frac")
        out["roofline_bound"] = src["roofline"].get("bound")
    if isinstance(src.get("roofline_hbm"), dict):
        out["hbm_frac"] = src["roofline_hbm"].get("frac")
    if "error" in src:
        out["error"] = str(src["error"])[:120]
    return out


def compact_line(d, detail_paths=()):
    """The ONE line of stdout from the full record `d`: every field of the driver's contract,
    `roofline` / `roofline_hbm` / `cpu_baseline` as numbers, and a few scalars per side leg
    (cfg-1/3/4/5, the node's default search, the C-ABI multi-device leg).  Pure function of `d`
    (tests/test_bench_artefacts.py builds it from a committed record); guaranteed below
    MAX_LINE_BYTES: side legs are dropped, last first, if a future field ever pushes it over."""
    line = _pick(d, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                     "scaling", "vs_baseline", "dtype", "data"))
    cfg = d.get("config", {})
    line["config"] = _pick(cfg, ("workload", "candidates", "n_theta", "n_linear", "beams", "units_per_step",
                                 "kernel_variant", "in_grid_share_of_units"))
    line["roofline"] = _roof_short(d.get("roofline"))
    line["roofline_hbm"] = _hbm_short(d.get("roofline_hbm"))
    line["cpu_baseline"] = _cpu_short(d.get("cpu_baseline"))
    line["match_result"] = d.get("match_result")
    line["library"] = _pick(d.get("library", {}), ("lib_matches_source", "build_info", "error"))
    for k in ("shader_clock_MHz_sysfs", "rccl_world_size", "collective_backend", "value_in_grid_units",
              "speedup_vs_single_gpu_same_workload"):
        if k in d:
            line[k] = d[k]
    if isinstance(d.get("rank_kernel_ms"), dict):
        line["rank_kernel_ms"] = _pick(d["rank_kernel_ms"], ("max", "mean"))
    if isinstance(d.get("host_call"), dict):
        line["host_call_pcie_inclusive"] = _pick(d["host_call"], ("ms", "value"))
    if isinstance(d.get("single_gpu_same_workload"), dict):
        line["single_gpu_same_workload"] = _pick(d["single_gpu_same_workload"], ("ms_per_step", "value"))
    # ---- side legs, most important first (dropped from the END when over the limit) ----
    legs = []
    ds = d.get("default_search")
    if isinstance(ds, dict):
        ch = ds.get("c_host") or {}
        leg = _pick(ds, ("match_scan_ms", "mapper_cycle_ms", "match_scan_kernel_ms", "kernel_variant", "headline_source"))
        leg.update(_pick(ch, ("match_scan_us", "match_scan_p99_us", "score_scan_us", "add_scans_us",
                              "mapper_cycle_us", "mapper_cycle_p99_us", "mapper_cycle_no_search_ahead_us",
                              "measure_500_particles_unchanged_loop_us", "pf_measure_500_particles_us")))
        rl = ch.get("real_lidar_map")
        if isinstance(rl, dict):
            leg["real_lidar_map"] = _pick(rl, ("match_scan_us", "add_scans_us", "mapper_cycle_us",
                                               "mapper_cycle_no_search_ahead_us"))
        cpu = ds.get("cpu_single_thread")
        if isinstance(cpu, dict):
            leg["cpu_single_thread"] = _pick(cpu, ("match_scan_ms", "add_scans_ms", "mapper_cycle_ms",
                                                   "measure_500_particles_ms"))
        legs.append(("default_search", leg))
    pf = d.get("particle_filter")
    if isinstance(pf, dict):
        legs.append(("particle_filter", _leg(pf, ("workload", "n_gpus", "ms_per_step", "kernel_ms", "value", "unit",
                                                    "units_per_launch", "units_per_step", "host_call_ms",
                                                    "collectives_per_step", "variant"))))
        if isinstance(pf.get("cpu_baseline"), dict):
            legs[-1][1]["cpu_baseline_value"] = pf["cpu_baseline"].get("value")
    for key, keys in (("cfg4_single_gpu", ("ms_per_step", "kernel_ms", "value", "unit", "units_per_step",
                                           "in_grid_share_of_units", "best_index")),
                      ("cfg5_single_gpu", ("ms_per_step", "kernel_ms", "value", "unit", "units_per_step",
                                           "collectives_per_step")),
                      ("particle_filter_cfg5", ("ms_per_step", "kernel_ms", "value", "unit", "units_per_step",
                                                "collectives_per_step"))):
        src = d.get(key)
        if isinstance(src, dict):
            leg = _leg(src, keys)
            sh = src.get("eight_shares_alone_on_this_gpu")
            if isinstance(sh, dict):
                leg["slowest_of_8_shares_alone_ms"] = sh.get("max_ms")
            if isinstance(src.get("cpu_baseline"), dict):
                leg["cpu_baseline_value"] = src["cpu_baseline"].get("value")
            legs.append((key, leg))
    c1 = d.get("cfg1_search")
    if isinstance(c1, dict):
        legs.append(("cfg1_search", _pick(c1, ("units", "gpu_match_scan_ms", "gpu_kernel_ms", "gpu_value",
                                               "cpu_single_thread_ms", "cpu_value", "gpu_over_cpu"))))
    mh = d.get("c_host_multi_device")
    if isinstance(mh, dict):
        def probe(src):
            out = {}
            for key in ("cfg2", "cfg4", "cfg5"):
                if isinstance(src.get(key), dict):
                    out[key] = _pick(src[key], ("step_ms", "call_ms", "units_per_s", "best_index", "exchange"))
            if "error" in src:
                out["error"] = str(src["error"])[:120]
            return out
        leg = probe(mh)
        leg["devices"] = mh.get("devices")
        if isinstance(mh.get("rccl"), dict):
            leg["rccl"] = probe(mh["rccl"])
        legs.append(("c_host_multi_device", leg))
    for key, leg in legs:
        line[key] = leg
    line["detail"] = {"files": list(detail_paths), "what": "the full record (per-class issue table, per-share arrays, "
                      "probe output, notes); also on stderr"}
    while len(json.dumps(line, separators=(",", ":"))) >= MAX_LINE_BYTES and legs:
        key, _ = legs.pop()
        line.pop(key, None)
        line.setdefault("dropped_for_size", []).append(key)
    return line


def write_detail(d, where=None):
    """The full record: `where`, or bench_detail.json beside this script (and under gpurun_out/
    when that exists, so that it comes back from the GPU box), and one line on stderr.  Returns
    the paths written (relative to the repository unless `where` was given)."""
    text = json.dumps(d)
    paths = []
    for rel in ((where,) if where else (DETAIL_NAME, os.path.join("gpurun_out", DETAIL_NAME))):
        path = os.path.join(_ROOT, rel)
        if not os.path.isdir(os.path.dirname(path) or "."):
            continue
        try:
            with open(path, "w") as f:
                f.write(text + "\n")
            paths.append(rel)
        except OSError:
            pass
    sys.stderr.write("bench.py detail: " + text + "\n")
    sys.stderr.flush()
    return paths


# --------------------------------------------------------------------------------------

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=["auto", "cfg2", "cfg4"], default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-particles", action="store_true")
    ap.add_argument("--no-default-search", action="store_true")
    ap.add_argument("--no-anchors", action="store_true",
                    help="N = 1: skip the cfg-4 / cfg-5 single-GPU anchors of the 8-GPU workloads")
    ap.add_argument("--host", choices=["python", "c"], default="python",
                    help="c: the headline step is ndt2d_matcher_match_scan on ONE multi-device matcher "
                         "(ndt2d_matcher_create_multi over the N GPUs) driven by rank 0 through the plain-C "
                         "probe; python (default): one rank per GPU, torch.distributed")
    ap.add_argument("--no-c-host", action="store_true", help="skip the C-ABI multi-device leg")
    ap.add_argument("--profile-shares", action="store_true",
                    help="(profiling aid, N = 1) one launch per 1-of-8 share of cfg-4 and cfg-5 on this GPU, "
                         "in rank order; prints the dispatch plan (experiments/profile_r04.sh)")
    ap.add_argument("--print-source-hash", action="store_true")
    ap.add_argument("--detail-file", default=None,
                    help="where the FULL record goes (default: bench_detail.json beside this script, and "
                         "gpurun_out/bench_detail.json when that directory exists); stdout carries the short line")
    ap.add_argument("--prewarm", type=float, default=PREWARM_SECONDS,
                    help="seconds of untimed launches before --warmup (profiler passes use 0)")
    args = ap.parse_args()

    if args.print_source_hash:
        print(source_hash())
        return
    if args.profile_shares:
        return profile_shares()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # run directly (`python bench.py --gpus N`): start the N ranks as a CHILD job before
        # anything here has touched the GPU (never exec: see gpurun's rules), relay its line
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    # Native libraries write to stdout as well (RCCL flushes a version banner at
    # exit): keep the real stdout for the one JSON line, send the rest to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    cfg = {"auto": 2 if world == 1 else 4, "cfg2": 2, "cfg4": 4}[args.workload]
    if args.steps is None:
        args.steps = 200 if cfg == 2 else 20
    if args.warmup is None:
        args.warmup = 10 if cfg == 2 else 3

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # NDT2D_BENCH_BACKEND=gloo is a debugging / test aid only (several ranks sharing one
    # GPU on a 1-GPU box, records exchanged through host memory); the driver's runs use
    # the default: one GPU per rank, RCCL.
    backend = os.environ.get("NDT2D_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    # NDT2D_BENCH_FORCE_COLLECTIVE=1: take the multi-rank code path (process group,
    # all-reduces, barrier) with a single rank too -- the only way to exercise the RCCL
    # path on a 1-GPU box
    collective = world > 1 or os.environ.get("NDT2D_BENCH_FORCE_COLLECTIVE") == "1"
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo":
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    store = dist.distributed_c10d._get_default_store() if collective else None

    def all_reduce(tensor, op, async_op=False):
        if backend == "gloo":
            host = tensor.cpu()
            dist.all_reduce(host, op=op)
            tensor.copy_(host)
            return None
        return dist.all_reduce(tensor, op=op, async_op=async_op)

    def fence():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    from ndt_2d_amd import ScanMatcherNDT, synth
    from ndt_2d_amd import dist as shard

    # ---- headline: the matchScan search, theta steps dealt round-robin to the ranks ----
    params = synth.matcher_params(cfg)
    scans = synth.map_scans(cfg)
    guess, pts, _ = synth.query_scan(cfg)

    m = ScanMatcherNDT(dev_index)
    m.initialize("global_scan_matcher", **params)
    m.addScans(scans)
    n_th, n_lin, n_beams = m.prepare_search(guess, pts)
    # rank r searches the theta steps r, r + N, r + 2N, ...: equal shares of every part
    # of the angular range (contiguous slabs differ by up to 1.4x in cost at N = 8)
    th_first, th_stride, th_count = shard.shard_strided(n_th, rank, world)
    my_units = th_count * n_lin * n_lin * n_beams
    total_units = n_th * n_lin * n_lin * n_beams

    # One explicit (non-null) torch stream carries everything: torch ops, the
    # library's kernels (ndt2d_set_stream) and, through torch.distributed's
    # current-stream hand-off, the RCCL all-reduce.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    m.set_stream(stream.cuda_stream)
    # Two record tables: the all-reduce of one search runs on RCCL's stream while
    # the next search's kernels run on this one; a table is reused only after its
    # own all-reduce has been waited for (a stream-level wait, the host runs on).
    tables = [torch.zeros((world, shard.MATCH_RECORD), dtype=torch.float64, device=dev)
              for _ in range(2)]
    pending = [None, None]
    n_steps_run = [0]

    # HIP events around the search kernel: on every launch when there are few steps, else on
    # every 8th (a recorded event holds the stream up for ~5 us; the pluginlib shim runs
    # with none at all) -- the kernel's average duration is taken over the timed launches
    # of the timed region
    # (the driver's --steps 20: every 4th step, five samples -- a pair on every step made the 20-step
    # line read 1.7 % below the 200-step one)
    time_every = 1 if args.steps < 8 else (4 if args.steps < 64 else 8)
    n_timed = [0]

    def step():
        slot = n_steps_run[0] & 1
        timed = n_steps_run[0] % time_every == 0
        m.set_timing(timed)
        n_timed[0] += 1 if timed else 0
        n_steps_run[0] += 1
        table = tables[slot]
        if collective:
            if pending[slot] is not None:
                pending[slot].wait()
                pending[slot] = None
            table.zero_()
        m.match_launch_strided(th_first, th_stride, th_count, record_ptr=table[rank].data_ptr())
        if collective:
            pending[slot] = all_reduce(table, dist.ReduceOp.SUM, async_op=True)

    def drain():
        for slot in (0, 1):
            if pending[slot] is not None:
                pending[slot].wait()
                pending[slot] = None

    # untimed pre-warm: launches for --prewarm seconds, so that the timed region -- 10 ms
    # under the driver's --steps 20 -- runs at the clocks the chip sustains, not at the
    # ones it idles at
    # (every rank must run the SAME number of steps -- each carries a collective -- so the
    # count comes from a short calibration whose slowest rank's time all ranks share)
    n_prewarm = 0
    if args.prewarm > 0:
        for _ in range(2):
            step()
        drain()
        fence()
        t_cal = time.perf_counter()
        for _ in range(4):
            step()
        drain()
        fence()
        cal = torch.tensor([(time.perf_counter() - t_cal) / 4.0], dtype=torch.float64, device=dev)
        if collective:
            all_reduce(cal, dist.ReduceOp.MAX)
        n_prewarm = int(min(max(args.prewarm / max(float(cal[0]), 1e-6), 1.0), 20000.0))
        for i in range(n_prewarm):
            step()
            if (i & 63) == 63:
                drain()
                torch.cuda.synchronize()
        n_prewarm += 6
    for _ in range(args.warmup):
        step()
    drain()
    fence()
    n_timed[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    clock_mhz = gpu_clock_mhz(torch, dev_index)   # sampled while the queued steps run
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    m.set_timing(True)
    # HIP events the library recorded around the search kernel of the timed launches among
    # those, on the launch stream (it keeps the last 256 pairs)
    kernel_ms = m.launch_history_ms(min(max(n_timed[0], 1), 256))
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if collective:
        all_reduce(t, dist.ReduceOp.MAX)
    elapsed = float(t[0])
    ms_per_step = elapsed / args.steps * 1e3
    # every rank's average search-kernel time (slot per rank, summed = gathered)
    rank_kernel = torch.zeros(world, dtype=torch.float64, device=dev)
    rank_kernel[rank] = sum(kernel_ms) / max(len(kernel_ms), 1)
    if collective:
        all_reduce(rank_kernel, dist.ReduceOp.SUM)
    rank_kernel_ms = [float(v) for v in rank_kernel.cpu()]

    # the result the search produced
    rec = tables[(n_steps_run[0] - 1) & 1].cpu().numpy()
    best_score, best_index, acc = shard.combine_match_records(rec)
    result = m.finish_match(np.concatenate([[best_score, -1.0 if best_index is None else best_index], acc]))
    variant = m.last_variant()

    # ---- N > 1: the same whole lattice on ONE GPU, measured by rank 0 in this job ----
    single = None
    if world > 1 and rank == 0:
        ms1 = []
        for i in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            m.match_launch(0, n_th)
            torch.cuda.synchronize()
            if i > 0:
                ms1.append((time.perf_counter() - t0) * 1e3)
        single = {"ms_per_step": min(ms1), "value": total_units / (min(ms1) * 1e-3),
                  "what": "the whole lattice searched by rank 0's GPU alone, same job (untimed region), "
                          "best of 3: the 1-GPU figure for THIS workload"}
    if collective:
        dist.barrier()

    # ---- cfg-5: sharded ParticleFilter::measure (N > 1 or forced collective) ----
    pf5 = None
    if collective and not args.no_particles:
        pf5 = particle_bench_sharded(ScanMatcherNDT, synth, shard, torch, dist, dev, dev_index, rank,
                                     world, backend, all_reduce, fence)

    if rank == 0:
        avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
        pmc = load_pmc()
        kname = ("match_lane_compact_kernel" if "compact-records" in variant else
                 "match_lane_kernel" if "lane" in variant else "match_kernel")
        if cfg == 2 and world == 1:
            roof, hbm = roofline(kname, avg_kernel_ms, my_units, n_cu, pmc,
                                 "counters of the cfg-2 launch of the same bench command")
        else:
            # cfg-4: the counters of this rank's share of the lattice (rank 0: theta steps 0, N, 2N, ...),
            # from the committed per-share passes; the kernel time is this rank's, live
            sc = share_counters(pmc, "cfg4", rank, world) if cfg == 4 else None
            roof, hbm = roofline(kname, avg_kernel_ms, my_units, n_cu, pmc,
                                 "counters of rank 0's share of the cfg-4 lattice (sum of the 1-of-8 shares %s, "
                                 "each one launch on one GPU under rocprofv3)" % (sc or {}).get("shares_summed"),
                                 counters=sc if sc is not None else {})
            roof["kernel_launches_per_step"] = "one per step on this rank (larger shares are cut into theta slabs)"
        grid = m.grid()
        line = {
            "metric": "pose-candidates x beams scored per second",
            "value": total_units / (ms_per_step * 1e-3),
            "unit": "candidate-beams/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if cfg == 4 else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": ("cfg-2 (BASELINE.json configs[1]): one 720-beam scan vs 41x41 NDT @0.25 m, "
                             "exhaustive search +-1.0 m/0.02 m x +-0.5 rad/0.005 rad") if cfg == 2 else
                            ("cfg-4 (BASELINE.json configs[3]): global loop-closure search, one 720-beam scan "
                             "vs 41x41 NDT @0.25 m, +-5 m/0.02 m x +-pi/0.005 rad, pose grid sharded over "
                             "%d GPU(s) + argmin/covariance all-reduce" % world),
                "candidates": n_th * n_lin * n_lin, "n_theta": n_th, "n_linear": n_lin,
                "beams": n_beams, "units_per_step": total_units,
                "sharding": "theta steps dealt round-robin to the ranks (total work fixed), one all-reduce "
                            "of the [N,12] record table per search, overlapped with the next search",
                "kernel_variant": variant,
            },
            "roofline": roof,
            "roofline_hbm": hbm,
            "match_result": {"score": result["score"], "pose": [float(v) for v in result["pose"]],
                             "best_index": best_index},
            "prewarm": {"seconds": args.prewarm, "steps": n_prewarm,
                        "what": "untimed launches before --warmup (clocks reach their sustained level)"},
            "shader_clock_MHz_sysfs": clock_mhz,
            "library": library_identity(),
            "rank_kernel_ms": {"max": max(rank_kernel_ms), "mean": sum(rank_kernel_ms) / len(rank_kernel_ms),
                               "per_rank": rank_kernel_ms},
        }
        if collective:
            line["rccl_world_size"] = dist.get_world_size() if backend == "nccl" else None
            line["collective_backend"] = backend
        if cfg == 4:
            share = in_grid_share(np, synth, params, guess, pts, grid)
            line["config"]["in_grid_share_of_units"] = share
            line["value_in_grid_units"] = line["value"] * share
            if single is not None:
                line["single_gpu_same_workload"] = single
                line["speedup_vs_single_gpu_same_workload"] = single["ms_per_step"] / ms_per_step
        if world == 1 and cfg == 2:
            # PCIe-inclusive figure (never `value`): the whole matchScan call with host
            # buffers in and out (subsample, tables, H2D, search, D2H of the 12-double record)
            m.set_stream(None)
            e2e = []
            for _ in range(5):
                t0 = time.perf_counter()
                m.matchScan(guess, pts)
                e2e.append(time.perf_counter() - t0)
            line["host_call"] = {"ms": min(e2e) * 1e3, "value": total_units / min(e2e),
                                 "what": "ndt2d_matcher_match_scan, host buffers in/out"}
        if world == 1 and cfg == 2 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(synth.matcher_params(2), synth.map_scans(2),
                                                *synth.query_scan(2)[:2])
        elif cfg == 4:
            line["cpu_baseline"] = committed_cpu_baseline("cfg4_match")
            if pf5 is not None:
                pf5["cpu_baseline"] = committed_cpu_baseline("cfg5_particles")
        if world == 1 and not args.no_particles:
            m.set_stream(None)
            line["particle_filter"] = particle_bench_1gpu(ScanMatcherNDT, synth, torch, dev_index, pmc, n_cu)
        if pf5 is not None:
            line["particle_filter_cfg5" if world == 1 else "particle_filter"] = pf5
        if world == 1 and cfg == 2 and not args.no_anchors:
            # the single-GPU points of the two 8-GPU workloads (BASELINE.json configs[3], [4])
            m.set_stream(None)
            a4 = cfg4_single_gpu_bench(ScanMatcherNDT, synth, shard, np, torch, dev_index)
            sc4 = share_counters(pmc, "cfg4", 0, 1)
            a4["roofline"], a4["roofline_hbm"] = roofline(
                "match_lane_compact_kernel", a4["kernel_ms"], a4["units_per_step"], n_cu, pmc,
                "counters of the whole cfg-4 lattice = the sum of its eight 1-of-8 shares (one launch each on one GPU "
                "under rocprofv3); kernel_ms = the step's search launches (theta slabs) together",
                counters=sc4 if sc4 is not None else {})
            a4["cpu_baseline"] = committed_cpu_baseline("cfg4_match")
            line["cfg4_single_gpu"] = a4
            if pf5 is None and not args.no_particles:
                a5 = particle_bench_sharded(
                    ScanMatcherNDT, synth, shard, torch, dist, dev, dev_index, 0, 1, backend, None,
                    torch.cuda.synchronize)
                a5["cpu_baseline"] = committed_cpu_baseline("cfg5_particles")
                line["cfg5_single_gpu"] = a5
        if world == 1 and not args.no_default_search:
            line["default_search"] = default_search_bench(ScanMatcherNDT, synth, dev_index,
                                                          with_cpu=not args.no_cpu_baseline)
            line["cfg1_search"] = cfg1_search_bench(ScanMatcherNDT, synth, dev_index,
                                                    not args.no_cpu_baseline)
        if not args.no_c_host:
            # ONE matcher over all N GPUs through the plain-C host (the unchanged node's
            # global_scan_matcher_->matchScan with the plugin's device_ids parameter); the other
            # ranks idle at the barrier below meanwhile
            ids = [dev_index] * world if backend == "gloo" else list(range(world))
            line["c_host_multi_device"] = c_host_multi_device(ids)
            if args.host == "c" and line["c_host_multi_device"].get("cfg4"):
                best = line["c_host_multi_device"]
                if best.get("rccl", {}).get("cfg4"):     # north_star's exchange when it ran
                    best = best["rccl"]
                c4 = best["cfg4"]
                line.update(value=c4["units_per_s"], ms_per_step=c4["step_ms"], steps=c4.get("steps", 7),
                            scaling="strong")
                line["config"]["workload"] = ("cfg-4 (BASELINE.json configs[3]) through ndt2d_matcher_match_scan on one "
                                              "multi-device matcher over %d GPU(s), C host" % world)
                line["config"]["kernel_variant"] = c4["variant"]
        # `line` is the FULL record (33 KB in round 5 -- past what the driver keeps of a line): it goes to
        # bench_detail.json and stderr; stdout gets the contract's fields + one summary per side leg
        detail_paths = write_detail(line, args.detail_file)
        short = compact_line(line, detail_paths)
        os.write(json_fd, (json.dumps(short, separators=(",", ":")) + "\n").encode())
        if collective:
            store.set("ndt2d_bench_rank0_done", "1")
    elif collective:
        # rank 0 is still measuring (the C-host leg drives every GPU from its own process): wait on
        # the host, not in a collective -- a pending RCCL kernel would spin on this rank's GPU
        from datetime import timedelta
        store.wait(["ndt2d_bench_rank0_done"], timedelta(seconds=3600))

    m.set_stream(None)
    m.close()
    if collective:
        dist.barrier()
        dist.destroy_process_group()


def particle_bench_sharded(matcher_cls, synth, shard, torch, dist, dev, dev_index, rank, world, backend,
                           all_reduce, fence, steps=20, warmup=3):
    """BASELINE.json configs[4] ("cfg-5"): 1,000,000 particles x 720 beams on the 801x801
    NDT, particles sharded contiguously over the ranks.  One step = ParticleFilter::measure
    (reference src/particle_filter.cpp:78-89 + updateStatistics :163-218): every rank scores
    its range and reduces its 8 moment sums on the device, ONE all-reduce of the [N, 8]
    table gives every rank the total particle weight and the moments -- the only collective
    of the step -- the rank normalises its weights and forms mean / covariance
    on the device.  The theta variance is the reference's second pass (:213-217: it needs the
    circular mean): every rank reduces ITS particles' share on the device and keeps it beside
    its weights; cov(2,2) = the N shares added in rank order by whoever collects the weights
    (here: once, after the timed region, for the result check).  KLD resampling is host code
    outside the timed region, as BASELINE.json says."""
    import numpy as np
    m = matcher_cls(dev_index)
    m.initialize("global_scan_matcher", **synth.matcher_params(5))
    m.addScans(synth.map_scans(5))
    _, pts, _ = synth.query_scan(5)
    parts = synth.particles(5)
    n_total = len(parts)
    begin, end = shard.shard_range(n_total, rank, world)
    n_beams = m.prepare_beams(pts)
    stream = torch.cuda.current_stream(dev)
    m.set_stream(stream.cuda_stream)
    d_parts = torch.from_numpy(parts[begin:end].copy()).to(dev)
    n_local = end - begin
    d_w = torch.zeros(max(n_local, 1), dtype=torch.float64, device=dev)
    table = torch.zeros((world, shard.POSE_STATS), dtype=torch.float64, device=dev)
    d_sum = torch.zeros(shard.POSE_STATS, dtype=torch.float64, device=dev)
    d_out = torch.zeros(8, dtype=torch.float64, device=dev)
    d_var = d_out[7:8]          # the theta-variance increment where pf_finalize leaves it

    def step():
        table.zero_()
        if n_local:
            m.score_poses_launch(d_parts.data_ptr(), n_local, d_w.data_ptr(), table[rank].data_ptr())
        if all_reduce is not None:
            all_reduce(table, dist.ReduceOp.SUM)               # total particle weight + moments
        torch.sum(table, dim=0, out=d_sum)                      # rank order, every rank the same
        if n_local:
            m.pf_finalize_launch(d_parts.data_ptr(), n_local, d_w.data_ptr(), d_sum.data_ptr(),
                                 d_out.data_ptr())
        # (d_out[7] = this rank's share of the theta variance: it stays here, with the weights)

    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if all_reduce is not None:
        all_reduce(t, dist.ReduceOp.MAX)
    ms = float(t[0]) / steps * 1e3
    out_h = d_out.cpu().numpy()
    # outside the timed region: the N shares of the theta variance, one slot per rank, added in rank order
    shares = torch.zeros(world, dtype=torch.float64, device=dev)
    shares[rank] = d_var[0]
    if all_reduce is not None:
        all_reduce(shares, dist.ReduceOp.SUM)
    theta_variance = 0.0
    for v in shares.cpu().numpy():
        theta_variance += float(v)
    units = n_total * n_beams
    res = {"workload": "cfg-5 (BASELINE.json configs[4]): 1000000 particles x 720 beams, 801x801 NDT @0.25 m, "
                       "particles sharded over %d GPU(s) + weight-sum all-reduce" % world,
           "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms,
           "value": units / (ms * 1e-3), "unit": "candidate-beams/s", "scaling": "strong",
           "units_per_step": units, "variant": m.last_variant(),
           "collectives_per_step": 1 if all_reduce is not None else 0,
           "collective": ("one all-reduce of the [N,8] moment sums per step; the theta-variance shares stay "
                          "with the weights" if all_reduce is not None else "none (one GPU, no process group)"),
           "result": {"sum_w": float(out_h[0]), "mean": [float(v) for v in out_h[1:4]],
                      "cov_xx_xy_yy": [float(v) for v in out_h[4:7]],
                      "theta_variance": theta_variance}}
    # roofline of this rank's scoring kernel: live HIP-event time of its launches, counters of its
    # share of the particle set from the committed per-share passes
    try:
        hist = m.launch_history_ms(steps)      # (the scoring launches: the statistics kernel records no events)
        score_ms = statistics.median(hist) if hist else None
    except Exception:   # noqa: BLE001 -- a missing figure, not a failed bench
        score_ms = None
    if score_ms:
        pmc = load_pmc()
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        sc = share_counters(pmc, "cfg5", rank, world)
        res["kernel_ms"] = score_ms
        res["roofline"], res["roofline_hbm"] = roofline(
            "score_poses_compact_kernel", score_ms, n_local * n_beams, n_cu, pmc,
            "counters of rank %d's particle range (sum of the 1-of-8 shares, one launch each on one GPU under "
            "rocprofv3)" % rank, counters=sc if sc is not None else {})
    if world == 1 and all_reduce is None:
        # the eight particle ranges an 8-GPU run deals out, each scored alone on this GPU (kernel time)
        sh = []
        for r in range(8):
            b, e = shard.shard_range(n_total, r, 8)
            ks = []
            for _ in range(3):
                m.score_poses_launch(d_parts[b:e].data_ptr(), e - b, d_w[b:e].data_ptr(), table[0].data_ptr())
                ks.append(m.last_launch_ms()[0])
            sh.append(min(ks))
        res["eight_shares_alone_on_this_gpu"] = {
            "scoring_kernel_ms": sh, "max_ms": max(sh),
            "what": "each 1-of-8 particle range scored alone on this GPU (scoring kernel, best of 3): an 8-GPU "
                    "step is the slowest range plus the statistics kernels and the two small all-reduces"}
    m.set_stream(None)
    m.close()
    return res


def committed_cpu_baseline(key):
    """The CPU oracle on an 8-GPU workload: the medians committed in profiles/r02_cpu_baselines.json
    (measured on the GPU box's host by experiments/cpu_baselines.py on a bounded sample -- the
    full lattice would take 16 minutes single-threaded), not re-timed while N ranks wait."""
    try:
        with open(CPU_BASELINE_FILE) as f:
            doc = json.load(f)
        c = doc[key]
        return {"value": c["all_cores"]["units_per_s"], "unit": "candidate-beams/s",
                "cores": c["all_cores"]["threads"], "kind": "port",
                "sample": "COMMITTED figure (profiles/r02_cpu_baselines.json, %s on %s): oracle, %s subset = %.3g units, "
                          "%d OpenMP threads, %.3f s; not re-timed in this run"
                          % (key, doc["host"]["cpu_model"],
                             c["all_cores"].get("theta_subset", c["all_cores"].get("particle_subset")),
                             c["all_cores"]["sample_units"], c["all_cores"]["threads"], c["all_cores"]["seconds"]),
                "single_thread_value": c["single"]["units_per_s"],
                "single_thread_sample": "%s subset, 1 thread (the reference's execution model), %.3f s"
                                        % (c["single"].get("theta_subset", c["single"].get("particle_subset")),
                                           c["single"]["seconds"])}
    except (OSError, KeyError, ValueError) as exc:
        return {"value": None, "unit": "candidate-beams/s", "cores": None, "kind": "port",
                "sample": "profiles/r02_cpu_baselines.json unreadable: %s" % exc}


def c_host_multi_device(ids):
    """ndt_2d_amd/tools/latency_probe.c --devices ids: cfg-2 and cfg-4 through
    ndt2d_matcher_match_scan of ONE multi-device matcher (whole call: host buffers in,
    result out), run as a child process -- with the host exchange (no collective) and, when
    the devices are distinct, with the RCCL exchange as well (`rccl`).  A failure or a
    time-out of the child is reported in the record, never raised: this leg must not cost the
    line its headline."""
    import subprocess
    probe = os.path.join(_ROOT, "ndt_2d_amd", "ndt2d_latency_probe")
    if not os.path.exists(probe):
        return {"error": "ndt2d_latency_probe not built"}
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK"):
        env.pop(k, None)

    def run(exchange):
        # (the probe takes 2 s with eight contexts on one GPU, 8 s with RCCL set up; the RCCL exchange over more than one
        # device has never run on hardware -- no multi-GPU node in any round -- so its leg gets the
        # shorter leash: a hang there must not cost the `--gpus N` line its place in the driver's budget)
        limit = 150 if exchange == "rccl" else 240
        try:
            r = subprocess.run([probe, "--devices", ",".join(str(i) for i in ids), "--exchange", exchange,
                                "--workload", "all"], capture_output=True, text=True, timeout=limit, env=env)
        except subprocess.TimeoutExpired as exc:
            tail = exc.stdout.decode(errors="replace") if isinstance(exc.stdout, bytes) else (exc.stdout or "")
            return {"error": "probe timed out (%d s)" % limit, "exchange_requested": exchange, "stdout_tail": tail[-1500:]}
        if r.returncode != 0:
            return {"error": "probe exit %d: %s" % (r.returncode, r.stderr[-500:]), "exchange_requested": exchange}
        try:   # (RCCL writes its own lines to stdout: the probe's is the one that opens the object)
            out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"devices"')][-1])
        except (ValueError, IndexError):
            return {"error": "no JSON from the probe", "exchange_requested": exchange, "stdout_tail": r.stdout[-500:]}
        for key in ("cfg2", "cfg4"):
            if key in out:
                out[key]["steps"] = 40 if key == "cfg2" else 7
        if out.get("cfg4", {}).get("best_index") != 80443810:
            return {"error": "cfg-4 winner differs from the oracle's", "got": out}
        return out

    out = run("host")
    if len(set(ids)) == len(ids):
        out["rccl"] = run("rccl")
    out["what"] = ("one ndt2d_matcher over devices %s (ndt2d_matcher_create_multi), plain-C host: median "
                   "wall time of the whole ndt2d_matcher_match_scan call (cfg2, cfg4) and of the whole "
                   "ndt2d_matcher_pf_measure call with the DEFAULT thresholds (cfg5: BASELINE.json configs[4], host "
                   "particles in, host weights out); fanout_us = when each device's launch had been queued; "
                   "dealing_overhead = the dealt call against the one-device call on workloads too small to matter; "
                   "host exchange at the top level, the RCCL exchange under `rccl`" % ids)
    return out


def profile_shares():
    """One launch per 1-of-8 share of the two 8-GPU workloads on this GPU, preceded by one
    warm-up launch of share 0 each: under `rocprofv3 --pmc` the LAST eight dispatches of
    match_lane_compact_kernel are cfg-4's shares 0..7 and the last eight of
    score_poses_compact_kernel cfg-5's."""
    import torch
    from ndt_2d_amd import ScanMatcherNDT, synth
    from ndt_2d_amd import dist as shard
    dev = torch.device("cuda", 0)
    plan = {}
    m = ScanMatcherNDT(0)
    m.initialize("global_scan_matcher", **synth.matcher_params(4))
    m.addScans(synth.map_scans(4))
    guess, pts, _ = synth.query_scan(4)
    n_th, n_lin, n_beams = m.prepare_search(guess, pts)
    units = []
    for r in [0] + list(range(8)):
        first, stride, count = shard.shard_strided(n_th, r, 8)
        m.match_launch_strided(first, stride, count)
        rec = m.match_fetch()
        units.append(count * n_lin * n_lin * n_beams)
    plan["cfg4"] = {"kernel": m.last_variant(), "units_per_share": units[1:], "last_record_index": int(rec[1])}
    m.close()
    m = ScanMatcherNDT(0)
    m.initialize("global_scan_matcher", **synth.matcher_params(5))
    m.addScans(synth.map_scans(5))
    _, pts, _ = synth.query_scan(5)
    parts = synth.particles(5)
    nb = m.prepare_beams(pts)
    d_parts = torch.from_numpy(parts).to(dev)
    d_w = torch.zeros(len(parts), dtype=torch.float64, device=dev)
    d_st = torch.zeros(8, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    units = []
    for r in [0] + list(range(8)):
        b, e = shard.shard_range(len(parts), r, 8)
        m.score_poses_launch(d_parts[b:e].data_ptr(), e - b, d_w[b:e].data_ptr(), d_st.data_ptr())
        m.synchronize()
        units.append((e - b) * nb)
    plan["cfg5"] = {"kernel": m.last_variant(), "units_per_share": units[1:]}
    m.close()
    print(json.dumps(plan))


if __name__ == "__main__":
    main()
