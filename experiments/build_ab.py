#!/usr/bin/env python3
"""Device NDT build (addScans on the GPU, csrc/ndt2d_build.hip) at cfg-3 / cfg-5: wall time of the
whole addScans call (median of 7) and a hash of the resulting grid -- for an A/B of two builds
of the library (NDT2D_HIP_LIB)."""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

for cfg in (3, 5):
    scans = synth.map_scans(cfg)
    m = ScanMatcherNDT(0)
    m.initialize("b", **synth.matcher_params(cfg))
    m.set_build_mode("device")
    ts = []
    for _ in range(8):
        m.reset()
        t0 = time.perf_counter()
        m.addScans(scans)
        m.synchronize()
        ts.append(time.perf_counter() - t0)
    cells = m.grid()[0]
    print("cfg-%d: addScans %.3f ms (min %.3f), grid sha %s" % (cfg, float(np.median(ts[1:])) * 1e3, min(ts) * 1e3,
                                                               hashlib.sha256(np.ascontiguousarray(cells).tobytes()).hexdigest()[:12]))
