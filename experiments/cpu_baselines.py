#!/usr/bin/env python3
"""SURVEY.md 8(d) "CPU path timing": the CPU oracle (gcc -O3, the reference's algorithm)
on the host cores of the GPU box, (1) single-threaded -- the reference's own execution
model -- and (2) OpenMP on all cores ((theta, dx) strips / particles dealt to threads).
cfg-2 in FULL both ways (1 warm-up + median of 5); cfg-1 fully; cfg-3 fully on all cores
and on a 1/8 subset single-threaded; cfg-4 and cfg-5 on subsets (stated), extrapolation
left to the reader.  Also records what the container actually grants: os.cpu_count() is
the machine's logical CPUs, the cgroup quota and the scheduler affinity may be far less
-- a thread-count sweep of the cfg-2 search shows where the all-core figure saturates.
Prints one JSON object (committed as profiles/r02_cpu_baselines.json)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from ndt_2d_amd import synth  # noqa: E402

sys.path.insert(0, ROOT)
from bench import granted_cpus as _granted  # noqa: E402

CORES = _granted()   # what the cgroup grants, not os.cpu_count()


def med(f, reps=5):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2]


def match_case(cfg, theta_coarsen, threads, reps=5):
    p = synth.matcher_params(cfg)
    p["search_angular_resolution"] *= theta_coarsen
    m = O.ScanMatcherNDT()
    m.initialize(**p)
    m.addScans(synth.map_scans(cfg))
    guess, pts, _ = synth.query_scan(cfg)
    n_th = len(O.search_offsets(p["search_angular_size"], p["search_angular_resolution"]))
    n_lin = len(O.search_offsets(p["search_linear_size"], p["search_linear_resolution"]))
    units = n_th * n_lin * n_lin * min(len(pts), p["laser_max_beams"])
    s = med(lambda: m.matchScan(guess, pts, omp_threads=threads), reps=reps)
    return dict(sample_units=units, seconds=s, units_per_s=units / s, threads=threads or 1,
                theta_subset="1/%d" % theta_coarsen, runs="1 warm-up + median of %d" % reps)


def particle_case(cfg, keep, threads):
    p = synth.matcher_params(cfg)
    m = O.ScanMatcherNDT()
    m.initialize(**p)
    m.addScans(synth.map_scans(cfg))
    _, pts, _ = synth.query_scan(cfg)
    pa = synth.particles(cfg)[::keep]
    units = len(pa) * len(pts)
    if threads:
        s = med(lambda: O.pf_measure(m, pa, pts, omp_threads=threads), reps=3)
    else:
        # the reference copies the scan's point vector for every particle (src/scan.cpp:67-70)
        s = med(lambda: O.pf_measure(m, pa, pts, copy_points=True), reps=3)
    return dict(sample_units=units, seconds=s, units_per_s=units / s, threads=threads or 1,
                particle_subset="1/%d" % keep)


def granted_cpus():
    out = dict(os_cpu_count=os.cpu_count(), threads_used_for_all_cores=CORES)
    try:
        out["sched_affinity"] = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                out[path] = f.read().strip()
        except OSError:
            pass
    try:
        with open("/proc/cpuinfo") as f:
            models = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")]
        out["cpu_model"] = models[0] if models else ""
    except OSError:
        pass
    return out


out = dict(host=granted_cpus(),
           note="CPU oracle = in-repo restatement of the reference (it cannot be built here); "
                "units = candidate poses x beams")
out["cfg2"] = dict(single_thread=match_case(2, 1, None), all_cores=match_case(2, 1, CORES))
out["cfg2"]["thread_sweep"] = [dict(threads=t, units_per_s=match_case(2, 1, t, reps=1)["units_per_s"])
                               for t in (2, 4, 8, 16, 32, 64, 128, 256) if t <= (os.cpu_count() or 1)]
out["cfg1_match"] = dict(single=match_case(1, 1, None), all_cores=match_case(1, 1, CORES))
out["cfg3_particles"] = dict(single=particle_case(3, 8, None), all_cores=particle_case(3, 1, CORES))
out["cfg4_match"] = dict(single=match_case(4, 64 * 8, None, reps=2), all_cores=match_case(4, 64, CORES, reps=2))
out["cfg5_particles"] = dict(single=particle_case(5, 64, None), all_cores=particle_case(5, 8, CORES))
print(json.dumps(out, indent=1))
