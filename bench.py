#!/usr/bin/env python3
"""Headline benchmark: pose-candidates x beams scored per second (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1 runs BASELINE.json configs[1] ("cfg-2"): one 720-beam scan against the
41x41 NDT @0.25 m, exhaustive search +-1.0 m / 0.02 m x +-0.5 rad / 0.005 rad =
100 x 100 x 200 candidate poses = 1.44e9 candidate-beam units per step, inputs
resident in HBM when the timed region starts.  For N > 1 (launched by
torch.distributed.run, one rank per GPU, RCCL) the angular resolution is refined
to 0.005/N rad and the theta steps are dealt round-robin, so every rank owns a cfg-2
sized share with the same mix of cheap and expensive headings (weak scaling) and
each step ends with the single all-reduce of the [N, 12] result table; that
all-reduce runs on RCCL's stream while the next step's search runs (two tables),
and all of them have completed when the timed region ends.

A "step" is one pass of the hot path (ScanMatcherNDT::matchScan's search,
reference src/scan_matcher_ndt.cpp:103-143) over that lattice.  Rank 0 prints
one JSON line.
"""
import argparse
import json
import os
import sys
import time

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)

BYTES_PER_UNIT = 64.0        # 16 B beam endpoint + 48 B cell record
HBM_PEAK_GBPS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(params, scans, guess, pts, seconds_hint=20.0):
    """The oracle (CPU restatement of the reference) on a bounded sample of the
    same workload: the cfg-2 lattice with every 4th theta (angular resolution
    0.02 rad -> 50 x 100 x 100 candidates x 720 beams = 3.6e8 units).  Checker
    code used as the reported CPU baseline only -- never on the product path."""
    sys.path.insert(0, os.path.join(_ROOT, "tests"))
    import oracle_lib as O

    p = dict(params)
    p["search_angular_resolution"] = 0.02
    ref = O.ScanMatcherNDT()
    ref.initialize(**p)
    ref.addScans(scans)
    n_th = len(O.search_offsets(p["search_angular_size"], p["search_angular_resolution"]))
    n_lin = len(O.search_offsets(p["search_linear_size"], p["search_linear_resolution"]))
    units = n_th * n_lin * n_lin * min(p["laser_max_beams"], len(pts))
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    ref.matchScan(guess, pts, omp_threads=cores)
    t_all = time.perf_counter() - t0
    # single thread, the reference's own execution model, on a quarter of the sample
    p1 = dict(p)
    p1["search_angular_resolution"] = 0.08
    ref1 = O.ScanMatcherNDT()
    ref1.initialize(**p1)
    ref1.addScans(scans)
    n_th1 = len(O.search_offsets(p1["search_angular_size"], p1["search_angular_resolution"]))
    units1 = n_th1 * n_lin * n_lin * min(p["laser_max_beams"], len(pts))
    t0 = time.perf_counter()
    ref1.matchScan(guess, pts)
    t_one = time.perf_counter() - t0
    return {
        "value": units / t_all, "unit": "candidate-beams/s", "cores": cores, "kind": "port",
        "sample": "cfg-2 lattice at angular resolution 0.02 rad: %dx%dx%d candidates x %d beams"
                  " = %.3g units, oracle matchScan, OpenMP over theta on %d threads"
                  % (n_th, n_lin, n_lin, min(p["laser_max_beams"], len(pts)), units, cores),
        "single_thread_value": units1 / t_one,
        "single_thread_sample": "%dx%dx%d candidates (angular resolution 0.08 rad), 1 thread,"
                                " the reference's own execution model" % (n_th1, n_lin, n_lin),
    }


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC
    passes (profiles/r01_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in
    separate --pmc runs of this same command, corrected as MI355X_MICROARCH.md
    prescribes).  None if the file is absent."""
    path = os.path.join(_ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["match"]
    except (OSError, KeyError, ValueError):
        return None


def particle_bench(matcher_cls, synth, torch, device_index, reps=5):
    """Secondary figure (not the headline): ParticleFilter::measure's scoring on
    BASELINE.json configs[2] ("cfg-3"): 100k particles x 720 beams, 201x201 NDT."""
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
    # an explicit (non-null) torch stream shared with the library: torch ops and
    # the kernels are ordered on it
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize(dev)
    m.set_stream(stream.cuda_stream)
    ms = []
    for i in range(reps + 1):
        m.score_poses_launch(d_parts.data_ptr(), len(parts), d_scores.data_ptr(),
                             d_stats.data_ptr())
        t, _ = m.last_launch_ms()
        if i > 0:
            ms.append(t)
    units = len(parts) * n_beams
    avg = sum(ms) / len(ms)
    m.set_stream(None)
    # PCIe-inclusive: host particles in, host weights out (ndt2d_matcher_score_poses)
    e2e = []
    for _ in range(3):
        t0 = time.perf_counter()
        m.scorePoses(pts, parts)
        e2e.append(time.perf_counter() - t0)
    out = {"workload": "cfg-3: 100000 particles x 720 beams, 201x201 NDT @0.25 m",
           "host_call_ms": min(e2e) * 1e3, "host_call_value": units / min(e2e),
           "units_per_launch": units, "kernel_ms": avg, "value": units / (avg * 1e-3),
           "unit": "candidate-beams/s", "variant": m.last_variant(),
           "achieved_GBps": units * BYTES_PER_UNIT / (avg * 1e-3) / 1e9}
    m.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-particles", action="store_true")
    args = ap.parse_args()

    # Native libraries write to stdout as well (RCCL flushes a version banner at
    # exit): keep the real stdout for the one JSON line, send the rest to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run "
                             "--nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible (there is no CPU fallback)")
    # NDT2D_BENCH_BACKEND=gloo is a debugging aid only (several ranks sharing one
    # GPU on a 1-GPU box, records exchanged through host memory); the driver's
    # runs use the default: one GPU per rank, RCCL.
    backend = os.environ.get("NDT2D_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    local_rank = dev_index
    # NDT2D_BENCH_FORCE_COLLECTIVE=1: take the multi-rank code path (process group,
    # all-reduce, barrier) with a single rank too -- the only way to exercise the
    # RCCL path on a 1-GPU box
    collective = world > 1 or os.environ.get("NDT2D_BENCH_FORCE_COLLECTIVE") == "1"
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "gloo":
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    def all_reduce(tensor, op):
        if backend == "gloo":
            host = tensor.cpu()
            dist.all_reduce(host, op=op)
            tensor.copy_(host)
        else:
            dist.all_reduce(tensor, op=op)

    from ndt_2d_amd import ScanMatcherNDT, synth
    from ndt_2d_amd import dist as shard

    # ---- workload: cfg-2, theta axis refined N-fold for weak scaling ----
    params = synth.matcher_params(2)
    params["search_angular_resolution"] = params["search_angular_resolution"] / world
    scans = synth.map_scans(2)
    guess, pts, _ = synth.query_scan(2)

    m = ScanMatcherNDT(local_rank)
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

    def step():
        slot = n_steps_run[0] & 1
        n_steps_run[0] += 1
        table = tables[slot]
        if collective:
            if pending[slot] is not None:
                pending[slot].wait()
                pending[slot] = None
            table.zero_()
        m.match_launch_strided(th_first, th_stride, th_count, record_ptr=table[rank].data_ptr())
        if collective:
            if backend == "gloo":
                all_reduce(table, dist.ReduceOp.SUM)
            else:
                pending[slot] = dist.all_reduce(table, op=dist.ReduceOp.SUM, async_op=True)

    def fence():
        for slot in (0, 1):
            if pending[slot] is not None:
                pending[slot].wait()
                pending[slot] = None
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # HIP events the library recorded around the search kernel of each of those
    # launches, on the launch stream (it keeps the last 256 pairs)
    kernel_ms = m.launch_history_ms(min(args.steps, 256))
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if collective:
        all_reduce(t, dist.ReduceOp.MAX)
    elapsed = float(t[0])
    ms_per_step = elapsed / args.steps * 1e3

    # the result the search produced (sanity: finite, a winner was found)
    rec = tables[(n_steps_run[0] - 1) & 1].cpu().numpy()
    best_score, best_index, acc = shard.combine_match_records(rec)
    result = m.finish_match(np.concatenate([[best_score, -1.0 if best_index is None else best_index], acc]))
    variant = m.last_variant()

    if rank == 0:
        avg_kernel_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = my_units * BYTES_PER_UNIT / (avg_kernel_ms * 1e-3) / 1e9
        line = {
            "metric": "pose-candidates x beams scored per second",
            "value": total_units / (ms_per_step * 1e-3),
            "unit": "candidate-beams/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "cfg-2 (BASELINE.json configs[1]): one 720-beam scan vs 41x41 NDT "
                            "@0.25 m, exhaustive search +-1.0 m/0.02 m x +-0.5 rad/%g rad"
                            % params["search_angular_resolution"],
                "candidates": n_th * n_lin * n_lin, "n_theta": n_th, "n_linear": n_lin,
                "beams": n_beams, "units_per_step": total_units,
                "sharding": "theta steps dealt round-robin to the ranks, one all-reduce of the [N,12] record table per search, overlapped with the next search",
                "kernel_variant": variant,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                "kernel": "match_lane_kernel (+ its 7 us rotated-beam table pre-pass)"
                          if "lane" in variant else "match_kernel",
                "kernel_ms_avg": avg_kernel_ms,
                "algorithmic_bytes_per_launch": my_units * BYTES_PER_UNIT,
                "note": "algorithmic 64 B/unit; compulsory HBM traffic is ~0.02 B/unit "
                        "(grid + beams are LDS/register resident) -- see DESIGN.md",
            },
            "match_result": {"score": result["score"], "pose": [float(v) for v in result["pose"]],
                             "best_index": best_index},
        }
        traffic = pmc_traffic()
        if traffic is not None:
            line["roofline"]["traffic"] = traffic["bytes_per_launch"]
            line["roofline"]["traffic_source"] = traffic["source"]
        if world == 1:
            # PCIe-inclusive figure (never `value`): the whole matchScan call with host
            # buffers in and out (subsample, tables, H2D, search, D2H of the 12-double record)
            m.set_stream(None)
            e2e = []
            for _ in range(3):
                t0 = time.perf_counter()
                m.matchScan(guess, pts)
                e2e.append(time.perf_counter() - t0)
            line["host_call"] = {"ms": min(e2e) * 1e3, "value": total_units / min(e2e),
                                 "what": "ndt2d_matcher_match_scan, host buffers in/out"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(synth.matcher_params(2), scans, guess, pts)
        if world == 1 and not args.no_particles:
            m.set_stream(None)
            line["particle_filter"] = particle_bench(ScanMatcherNDT, synth, torch, local_rank)
        os.write(json_fd, (json.dumps(line) + "\n").encode())

    m.set_stream(None)
    m.close()
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
