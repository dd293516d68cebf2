"""Wall time of OccupancyGrid::getMsg: GPU (ndt2d_occupancy_grid, host buffers in / out)
against the CPU oracle, on the synthetic maps."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402
from ndt_2d_amd.occupancy_grid import OccupancyGrid  # noqa: E402

dev = ScanMatcherNDT(0)
dev.initialize("occ", **synth.matcher_params(1))
for cfg in (1, 3, 5):
    scans = synth.map_scans(cfg)
    rays = sum(len(s[1]) for s in scans)
    g = OccupancyGrid(0.05, 0.25, dev)
    m = g.getMsg(scans)
    t = time.perf_counter()
    for _ in range(5):
        m = g.getMsg(scans)
    gpu = (time.perf_counter() - t) / 5
    t = time.perf_counter()
    ref = O.OccupancyGrid(0.05, 0.25).getMsg(scans)
    cpu = time.perf_counter() - t
    assert np.array_equal(ref["data"], m["data"])
    print("cfg-%d map: %d scans, %d rays, %d x %d cells: GPU %.2f ms (two calls: size + render, "
          "H2D %d KB each, D2H %d KB), CPU oracle %.1f ms" %
          (cfg, len(scans), rays, m["width"], m["height"], gpu * 1e3, rays * 16 // 1024,
           m["width"] * m["height"] // 1024, cpu * 1e3))
