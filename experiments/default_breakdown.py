#!/usr/bin/env python3
"""Where the time of one matchScan at the plugin defaults goes (host wall time of the
three stages of the call, and the kernels inside)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

scans = synth.map_scans(1)
p = synth.matcher_params(1, search_linear_size=0.05, search_linear_resolution=0.005,
                         search_angular_size=0.1, search_angular_resolution=0.0025,
                         laser_max_beams=100)
guess, pts, _ = synth.query_scan(1)
m = ScanMatcherNDT(0)
m.initialize("local_scan_matcher", **p)
m.addScans(scans)
g = np.array([0.11, -0.05, 0.02])
for _ in range(20):
    m.matchScan(g, pts)
N = 200
t = {k: 0.0 for k in ("prepare", "launch", "fetch", "finish", "whole")}
for _ in range(N):
    t0 = time.perf_counter()
    n_th, n_lin, nb = m.prepare_search(g, pts)
    t1 = time.perf_counter()
    m.match_launch(0, n_th)
    t2 = time.perf_counter()
    rec = m.match_fetch()
    t3 = time.perf_counter()
    m.finish_match(rec)
    t4 = time.perf_counter()
    t["prepare"] += t1 - t0
    t["launch"] += t2 - t1
    t["fetch"] += t3 - t2
    t["finish"] += t4 - t3
whole = []
for _ in range(N):
    t0 = time.perf_counter()
    m.matchScan(g, pts)
    whole.append(time.perf_counter() - t0)
t["whole"] = sum(whole)
whole.sort()
print("whole: min %.1f median %.1f max %.1f us" % (whole[0] * 1e6, whole[N // 2] * 1e6, whole[-1] * 1e6))
print({k: round(v / N * 1e6, 1) for k, v in t.items()}, "us;  kernel", m.last_launch_ms())
