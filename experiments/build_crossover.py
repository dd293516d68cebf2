#!/usr/bin/env python3
"""addScans of k scans (720 points each) built on the host against built on the device: where the
crossover lies (the library took the device from 32,768 points until round 5; 73,728 since)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

all_scans = synth.map_scans(3)
# neighbouring scans, as a loop closure takes them: the first rows of the pose lattice
for k in (9, 27, 45, 90, 120, 150, 200, 300, 525):
    scans = all_scans[:k]
    out = []
    for mode in ("host", "device"):
        m = ScanMatcherNDT(0)
        m.initialize("x", **synth.matcher_params(3))
        m.set_build_mode(mode)
        ts = []
        for _ in range(12):
            m.reset()
            t0 = time.perf_counter()
            m.addScans(scans)
            m.synchronize()
            ts.append(time.perf_counter() - t0)
        out.append(float(np.median(ts[2:])) * 1e3)
        m.close()
    print("%3d scans (%6d points): host %.3f ms, device %.3f ms" % (k, 720 * k, out[0], out[1]), flush=True)
