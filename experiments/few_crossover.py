#!/usr/bin/env python3
"""Block-per-pose ("few") against the batched particle kernel for small particle sets: the whole
scorePoses / pf_measure call from host memory, event pairs off, 100 and 720 beams."""
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, pf_measure, synth  # noqa: E402

for beams in (100, 720):
    m = ScanMatcherNDT(0)
    m.initialize("x", **synth.matcher_params(1, laser_max_beams=beams))
    m.addScans(synth.map_scans(1))
    _, pts, _ = synth.query_scan(1)
    m.set_timing(False)
    for n in (64, 256, 512, 1024, 2048, 4096):
        pa = synth.particles(3, n)
        pa[:, :2] *= 4.0 / 23.0
        row = []
        for variant in ("auto", "batched"):
            m.set_variant(variant)
            for f, name in ((lambda: m.scorePoses(pts, pa), "score"), (lambda: pf_measure(m, pa, pts), "measure")):
                for _ in range(5):
                    f()
                ts = []
                for _ in range(40):
                    t0 = time.perf_counter()
                    f()
                    ts.append((time.perf_counter() - t0) * 1e6)
                row.append("%s %s %.1f us [%s]" % (variant, name, statistics.median(ts), m.last_variant().split("/")[1][:10]))
        m.set_variant("auto")
        print("%d beams, %4d poses: %s" % (beams, n, " | ".join(row)), flush=True)
    m.close()
