#!/usr/bin/env python3
"""Is the lane kernel's time a step function of ceil(items / wave slots)?  cfg-2 with
the number of theta steps varied: items = n_th x 169 patches, 4096 wave slots."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

scans = synth.map_scans(2)
guess, pts, _ = synth.query_scan(2)
params = synth.matcher_params(2)
m = ScanMatcherNDT(0)
m.initialize("m", **params)
m.addScans(scans)
n_th, n_lin, nb = m.prepare_search(guess, pts)
patches = ((n_lin + 7) // 8) ** 2
print("n_th items items/slot  kernel_ms  us_per_item_per_slot")
for k in (97, 121, 122, 145, 146, 160, 169, 170, 182, 193, 194, 195, 200):
    lo = (n_th - k) // 2
    for _ in range(6):
        m.match_launch(lo, lo + k)
    m.synchronize()
    ms = float(np.median(m.launch_history_ms(5)))
    items = k * patches
    print("%4d %6d %8.3f  %9.4f  %8.2f" % (k, items, items / 4096.0, ms, ms * 1e3 / (items / 4096.0)))
