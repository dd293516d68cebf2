#!/usr/bin/env python3
"""The mapper's local matching step with the plugin defaults, a few times (for rocprofv3)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

scans = synth.map_scans(1) + [synth.map_scans(1)[0]]
p = synth.matcher_params(1, search_linear_size=0.05, search_linear_resolution=0.005,
                         search_angular_size=0.1, search_angular_resolution=0.0025,
                         laser_max_beams=100)
guess, pts, _ = synth.query_scan(1)
m = ScanMatcherNDT(0)
m.initialize("local_scan_matcher", **p)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    m.reset()
    m.addScans(scans)
    r = m.matchScan(np.array([0.11, -0.05, 0.02]), pts)
print(r["score"], r["pose"], m.last_launch_ms())
