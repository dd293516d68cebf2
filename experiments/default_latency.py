#!/usr/bin/env python3
"""Latency of the node's default workload, call by call (medians over many calls), and --
under rocprofv3 --kernel-trace --stats -- the kernels behind it.

    python experiments/default_latency.py [n_calls]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
scans = synth.map_scans(1)
guess, pts, _ = synth.query_scan(1)


def med(fn, reps=n):
    ts = []
    for i in range(reps + 20):
        t0 = time.perf_counter()
        fn()
        if i >= 20:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6


for name, over in [("defaults 35,280 x 100", dict(search_linear_size=0.05, search_linear_resolution=0.005,
                                                   search_angular_size=0.1, search_angular_resolution=0.0025,
                                                   laser_max_beams=100)),
                   ("35,280 x 720", dict(search_linear_size=0.05, search_linear_resolution=0.005,
                                         search_angular_size=0.1, search_angular_resolution=0.0025)),
                   ("cfg-1 17,640 x 720", {})]:
    m = ScanMatcherNDT(0)
    m.initialize("m", **synth.matcher_params(1, **over))
    m.addScans(scans)
    out = {}
    for variant in ("auto", "wave", "lane"):
        m.set_variant(variant)
        m.set_timing(True)
        m.matchScan(guess, pts)
        m.matchScan(guess, pts)
        k_ms, _ = m.last_launch_ms()
        m.set_timing(False)
        out[variant] = (med(lambda: m.matchScan(guess, pts)), k_ms * 1e3, m.last_variant())
    m.set_variant("auto")
    print(name)
    for v, (call, k, var) in out.items():
        print("   %-5s matchScan call %7.1f us   search kernel %7.1f us   %s" % (v, call, k, var))
    print("   scoreScan %.1f us  scorePoints %.1f us  addScans %.1f us" % (
        med(lambda: m.scoreScan(guess, pts)), med(lambda: m.scorePoints(pts, guess)),
        med(lambda: (m.reset(), m.addScans(scans)), 100)))
    m.close()
