#!/usr/bin/env python3
"""OccupancyGrid::getMsg on the cfg-3 / cfg-5 maps (0.05 m cells), a few calls each, for a
rocprofv3 kernel trace of the ray-tracing kernels; prints the wall time of a call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402
from ndt_2d_amd.occupancy_grid import OccupancyGrid  # noqa: E402

for cfg in (3, 5):
    scans = synth.map_scans(cfg)
    dev = ScanMatcherNDT(0)
    og = OccupancyGrid(0.05, 0.25, dev)
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        msg = og.getMsg(scans)
        ts.append(time.perf_counter() - t0)
    print("cfg-%d: %d scans, grid %d x %d, getMsg %.2f ms (min %.2f)" % (cfg, len(scans), msg["width"], msg["height"],
                                                                      float(np.median(ts[1:])) * 1e3, min(ts) * 1e3))
