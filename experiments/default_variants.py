#!/usr/bin/env python3
"""Kernel time of the plugin-default search (35,280 candidates x 100 beams) per variant."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

scans = synth.map_scans(1)
p = synth.matcher_params(1, search_linear_size=0.05, search_linear_resolution=0.005,
                         search_angular_size=0.1, search_angular_resolution=0.0025,
                         laser_max_beams=100)
guess, pts, _ = synth.query_scan(1)
g = np.array([0.11, -0.05, 0.02])
for variant in ("auto", "lane-noskip", "wave"):
    m = ScanMatcherNDT(0)
    m.initialize("local_scan_matcher", **p)
    m.addScans(scans)
    m.set_variant(variant)
    for _ in range(30):
        r = m.matchScan(g, pts)
    ms = m.launch_history_ms(20)
    print("%-12s kernel %.4f ms (min %.4f)  %s  score %.17g" % (variant, float(np.median(ms)), min(ms),
                                                              m.last_variant(), r["score"]))
