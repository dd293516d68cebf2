#!/usr/bin/env python3
"""cfg-2 search at several NDT cell sizes: power-of-two sizes take the multiply path of
NDT::getIndex, the others the true IEEE divide (a different kernel instantiation)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

scans = synth.map_scans(2)
guess, pts, _ = synth.query_scan(2)
for res in (0.125, 0.2, 0.25, 0.3, 0.5):
    m = ScanMatcherNDT(0)
    m.initialize("m", **synth.matcher_params(2, ndt_resolution=res))
    m.addScans(scans)
    for _ in range(8):
        r = m.matchScan(guess, pts)
    ms = float(np.median(m.launch_history_ms(6)))
    print("cell %.3f m: %.4f ms  %.3e units/s  %s  pose %s" % (res, ms, r["n_candidates"] * 720 / (ms * 1e-3),
                                                            m.last_variant(), np.round(r["pose"], 3)))
