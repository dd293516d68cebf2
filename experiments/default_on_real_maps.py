#!/usr/bin/env python3
"""The node's default search (21 x 21 x 80 candidates x 100 beams, reference
src/scan_matcher_ndt.cpp:37-44) on maps of the size a real lidar gives: the local NDT spans
the scan poses +- range_max (src/scan_matcher_ndt.cpp:52-66), 241 x 241 cells for a 30 m
lidar at 0.25 m -- far beyond the 41 x 41 toy map of cfg-1, whose records fit LDS.
matchScan / scoreScan call latency (medians) and the search kernel's time."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

DEFAULTS = dict(search_linear_size=0.05, search_linear_resolution=0.005, search_angular_size=0.1,
                search_angular_resolution=0.0025, laser_max_beams=100)


def med(fn, reps=300):
    ts = []
    for i in range(reps + 20):
        t0 = time.perf_counter()
        fn()
        if i >= 20:
            ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e6


def local_scans(cfg, centre, k=3, pitch=0.5, seed=77):
    """k x k scans around `centre` in the world of `cfg`: what the mapper's local matcher holds."""
    w = synth.world_of(cfg)
    out = []
    for j in range(k):
        for i in range(k):
            x = centre[0] + (i - (k - 1) / 2.0) * pitch
            y = centre[1] + (j - (k - 1) / 2.0) * pitch
            if not synth.pose_blocked(w, x, y):
                out.append(((x, y, 0.0), synth.scan(w, (x, y, 0.0), seed + 10 * j + i)))
    return out


cases = [("cfg-1 toy map 41x41", 1, synth.map_scans(1), synth.matcher_params(1)["range_max"], synth.query_scan(1))]
for cfg, rmax in ((3, 12.0), (3, 30.0), (5, 30.0), (5, 60.0)):
    q = synth.query_scan(cfg)
    cases.append(("cfg-%d world, 9 local scans, range_max %.0f m" % (cfg, rmax), cfg,
                  local_scans(cfg, q[2][:2]), rmax, q))
cases.append(("cfg-3 global map 201x201", 3, synth.map_scans(3), synth.matcher_params(3)["range_max"], synth.query_scan(3)))
for name, cfg, scans, rmax, (guess, pts, true) in cases:
    m = ScanMatcherNDT(0)
    p = dict(synth.matcher_params(cfg, **DEFAULTS), range_max=rmax)
    m.initialize("m", **p)
    m.addScans(scans)
    _, sx, sy, _, _, _ = m.grid()
    pose = true + np.array([0.02, -0.02, 0.01])
    m.set_timing(True)
    for _ in range(3):
        r = m.matchScan(pose, pts)
    k_ms, _ = m.last_launch_ms()
    variant = m.last_variant()
    m.set_timing(False)
    t_match = med(lambda: m.matchScan(pose, pts))
    t_score = med(lambda: m.scoreScan(pose, pts))
    t_add = med(lambda: (m.reset(), m.addScans(scans)), reps=50)
    print("%-44s grid %4dx%-4d  matchScan %6.1f us (kernel %5.1f)  scoreScan %5.1f us  addScans %7.1f us  %s  score %.4f"
          % (name, sx, sy, t_match, k_ms * 1e3, t_score, t_add, variant.replace("match/lane-per-candidate/", ""), r["score"]),
          flush=True)
    m.close()
