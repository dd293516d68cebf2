#!/usr/bin/env python3
"""Latency of what the mapper does per laser scan in mapping mode (reference
src/ndt_mapper.cpp:508-515): reset() + addScans(last K scans) + scoreScan + matchScan,
at the plugin's default search, through the C-ABI -- and the same cycle on the CPU
oracle (one thread, the reference's execution model)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402
import oracle_lib as O  # noqa: E402  (checker / CPU baseline only)

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = 200
scans = synth.map_scans(1)
while len(scans) < K:
    scans = scans + scans
scans = scans[:K]
guess, pts, _ = synth.query_scan(1)
g = np.array([0.11, -0.05, 0.02])
params = dict(ndt_resolution=0.25, search_angular_resolution=0.0025, search_angular_size=0.1,
              search_linear_resolution=0.005, search_linear_size=0.05, laser_max_beams=100)


def cycle(m, stages):
    t0 = time.perf_counter()
    m.reset()
    m.addScans(scans)
    t1 = time.perf_counter()
    m.scoreScan(g, pts)
    t2 = time.perf_counter()
    r = m.matchScan(g, pts)
    t3 = time.perf_counter()
    stages[0].append(t1 - t0)
    stages[1].append(t2 - t1)
    stages[2].append(t3 - t2)
    return r


def med(v):
    return sorted(v)[len(v) // 2] * 1e6


for mode in ("host", "device", "auto"):
    m = ScanMatcherNDT(0)
    m.initialize("local_scan_matcher", range_max=synth.matcher_params(1)["range_max"], **params)
    m.set_build_mode(mode)
    st = ([], [], [])
    for _ in range(20):
        cycle(m, ([], [], []))
    for _ in range(N):
        r = cycle(m, st)
    print("GPU build=%-6s K=%d: addScans %.0f us, scoreScan %.0f us, matchScan %.0f us, cycle %.0f us"
          % (mode, K, med(st[0]), med(st[1]), med(st[2]), med([a + b + c for a, b, c in zip(*st)])))

om = O.ScanMatcherNDT()
om.initialize(range_max=synth.matcher_params(1)["range_max"], **params)
st = ([], [], [])
for _ in range(5):
    t0 = time.perf_counter()
    om.reset()
    om.addScans(scans)
    t1 = time.perf_counter()
    om.scoreScan(g, pts)
    t2 = time.perf_counter()
    ro = om.matchScan(g, pts)
    t3 = time.perf_counter()
    st[0].append(t1 - t0)
    st[1].append(t2 - t1)
    st[2].append(t3 - t2)
print("CPU oracle (1 thread)  : addScans %.0f us, scoreScan %.0f us, matchScan %.0f us, cycle %.0f us"
      % (med(st[0]), med(st[1]), med(st[2]), med([a + b + c for a, b, c in zip(*st)])))
print("same pose:", np.array_equal(np.asarray(r["pose"]), np.asarray(ro["pose"])),
      " score diff %.2e" % abs(r["score"] - ro["score"]))
