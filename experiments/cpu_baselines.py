#!/usr/bin/env python3
"""SURVEY.md 8(d) "CPU path timing": the CPU oracle (g++ -O3, the reference's algorithm)
on the host cores of the GPU box, (1) single-threaded -- the reference's own execution
model -- and (2) OpenMP over theta / particles on all cores.  cfg-1 fully; cfg-2 and
cfg-3 fully on all cores, on a 1/8 subset single-threaded; cfg-4 and cfg-5 on a 1/64
subset, extrapolated.  Median of 5 runs after one warm-up.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from ndt_2d_amd import synth  # noqa: E402

CORES = os.cpu_count()


def med(f, reps=5):
    f()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return sorted(t)[len(t) // 2]


def match_case(cfg, theta_coarsen, threads):
    p = synth.matcher_params(cfg)
    p["search_angular_resolution"] *= theta_coarsen
    m = O.ScanMatcherNDT()
    m.initialize(**p)
    m.addScans(synth.map_scans(cfg))
    guess, pts, _ = synth.query_scan(cfg)
    n_th = len(O.search_offsets(p["search_angular_size"], p["search_angular_resolution"]))
    n_lin = len(O.search_offsets(p["search_linear_size"], p["search_linear_resolution"]))
    units = n_th * n_lin * n_lin * min(len(pts), p["laser_max_beams"])
    s = med(lambda: m.matchScan(guess, pts, omp_threads=threads), reps=5 if units < 5e9 else 2)
    return dict(sample_units=units, seconds=s, units_per_s=units / s, threads=threads or 1,
                theta_subset="1/%d" % theta_coarsen)


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


out = dict(cores=CORES, note="CPU oracle = in-repo restatement of the reference (it cannot be built here)")
out["cfg1_match"] = dict(single=match_case(1, 1, None), all_cores=match_case(1, 1, CORES))
out["cfg2_match"] = dict(single=match_case(2, 8, None), all_cores=match_case(2, 1, CORES))
out["cfg3_particles"] = dict(single=particle_case(3, 8, None), all_cores=particle_case(3, 1, CORES))
out["cfg4_match"] = dict(single=match_case(4, 64 * 8, None), all_cores=match_case(4, 64, CORES))
out["cfg5_particles"] = dict(single=particle_case(5, 64, None), all_cores=particle_case(5, 8, CORES))
print(json.dumps(out, indent=1))
