"""A/B of two builds of the library over the lattices of the mapper and of loop closure: the
plugin default (100 beams), cfg-1, three mid-size lattices on cfg-2's map and scan (1,352 /
6,760 / 23,660 work items) and cfg-2 itself.  Per lattice: the search's kernel time (HIP events,
median of 20), the whole matchScan call (median of 40), winner, score and a hash of ALL candidate
scores -- the scores must be the same bits whatever the build.  NDT2D_HIP_LIB selects the build:
    NDT2D_HIP_LIB=$PWD/experiments/bin/<variant>.so python experiments/lattice_ab.py
"""
import json
import os
import statistics
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402

from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

CASES = [
    ("default", dict(search_linear_size=0.05, search_linear_resolution=0.005, search_angular_size=0.1,
                     search_angular_resolution=0.0025, laser_max_beams=100)),
    ("cfg1", dict(search_linear_size=0.5, search_linear_resolution=0.05, search_angular_size=0.2,
                  search_angular_resolution=0.01)),
    ("mid_1352", dict(search_linear_size=1.0, search_linear_resolution=0.02, search_angular_size=0.02,
                      search_angular_resolution=0.005)),
    ("mid_6760", dict(search_linear_size=1.0, search_linear_resolution=0.02, search_angular_size=0.1,
                      search_angular_resolution=0.005)),
    ("mid_23660", dict(search_linear_size=1.0, search_linear_resolution=0.02, search_angular_size=0.35,
                       search_angular_resolution=0.005)),
    ("cfg2", dict(search_linear_size=1.0, search_linear_resolution=0.02, search_angular_size=0.5,
                  search_angular_resolution=0.005)),
    ("cfg4", dict(search_linear_size=5.0, search_linear_resolution=0.02, search_angular_size=3.141592653589793,
                  search_angular_resolution=0.005)),
]
guess, pts, _ = synth.query_scan(2)
scans = synth.map_scans(2)
out = {"lib": os.environ.get("NDT2D_HIP_LIB", "in-tree")}
for name, search in CASES:
    m = ScanMatcherNDT(0)
    m.initialize(name, **synth.matcher_params(2, **search))
    if os.environ.get("NDT2D_AB_BUILD_MODE"):   # "device": the grid as a loop closure's addScans of many scans builds it
        m.set_build_mode(os.environ["NDT2D_AB_BUILD_MODE"])
    m.addScans(scans)
    big = name == "cfg4"
    for _ in range(1 if big else 5):
        r = m.matchScan(guess, pts)
    kernel = []
    for _ in range(3 if big else 20):
        r = m.matchScan(guess, pts)
        kernel.append(m.last_launch_ms()[0])
    variant = m.last_variant()
    m.set_timing(False)
    calls = []
    for _ in range(3 if big else 40):
        t0 = time.perf_counter()
        r = m.matchScan(guess, pts)
        calls.append((time.perf_counter() - t0) * 1e3)
    full = r if big else m.matchScan(guess, pts, want_scores=True)
    if big:
        full = {"scores": np.array([r["score"], float(r["best_index"])])}
    out[name] = {"kernel_ms": statistics.median(kernel), "call_ms": statistics.median(calls), "best_index": r["best_index"],
                 "score": r["score"], "variant": variant,
                 "scores_sha": __import__("hashlib").sha256(np.ascontiguousarray(full["scores"]).tobytes()).hexdigest()[:16]}
    m.close()
print(json.dumps(out))
