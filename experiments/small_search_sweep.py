#!/usr/bin/env python3
"""Lane-per-candidate vs wave-per-candidate kernel time over small search lattices:
where is the crossover?  (cfg-1 map and scan; items = theta steps x 8x8 patches.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

scans = synth.map_scans(1)
guess, pts, _ = synth.query_scan(1)
g = np.array([0.11, -0.05, 0.02])
print("beams n_lin n_th candidates items   lane_ms  wave_ms")
for beams in (100, 720):
    for lin_size, lin_res, ang_size, ang_res in ((0.05, 0.005, 0.1, 0.0025), (0.05, 0.005, 0.2, 0.0025),
                                                 (0.1, 0.005, 0.1, 0.0025), (0.1, 0.005, 0.15, 0.0025),
                                                 (0.1, 0.005, 0.2, 0.0025), (0.1, 0.005, 0.25, 0.0025),
                                                 (0.1, 0.005, 0.3, 0.0025), (0.2, 0.005, 0.1, 0.0025),
                                                 (0.2, 0.005, 0.15, 0.0025), (0.2, 0.005, 0.2, 0.0025),
                                                 (0.2, 0.005, 0.4, 0.0025)):
        p = synth.matcher_params(1, search_linear_size=lin_size, search_linear_resolution=lin_res,
                                 search_angular_size=ang_size, search_angular_resolution=ang_res,
                                 laser_max_beams=beams)
        out = []
        for variant in ("lane", "wave"):
            m = ScanMatcherNDT(0)
            m.initialize("m", **p)
            m.addScans(scans)
            m.set_variant(variant)
            for _ in range(12):
                r = m.matchScan(g, pts)
            out.append(float(np.median(m.launch_history_ms(8))))
        n_th, n_lin, nb = m.prepare_search(g, pts)
        items = n_th * ((n_lin + 7) // 8) ** 2
        print("%5d %5d %4d %10d %6d  %8.4f %8.4f" % (nb, n_lin, n_th, n_th * n_lin * n_lin, items, out[0], out[1]))
