#!/usr/bin/env python3
"""The whole ParticleFilter::measure / scorePoses CALL from ordinary host memory (BASELINE
configs[4]: 10^6 particles x 720 beams on the 801 x 801 map; and cfg-3) with the pose batch in one
piece against cut into overlapped upload / scoring / download pieces (ndt2d_set_pipeline_pieces):
wall time of the call, raw scores bit for bit, weights and statistics relative difference."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, pf_measure, synth  # noqa: E402


def timed(f, reps=12):
    f()
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, float(min(ts)) * 1e3, out


for cfg in (5, 3):
    m = ScanMatcherNDT(0)
    m.initialize("pf", **synth.matcher_params(cfg))
    m.addScans(synth.map_scans(cfg))
    _, pts, _ = synth.query_scan(cfg)
    pa = synth.particles(cfg)
    res = {}
    for pieces in (1, 2, 4, 8, 16, 0):
        m.set_pipeline_pieces(pieces)
        ms, mn, (w, mean, cov) = timed(lambda: pf_measure(m, pa, pts))
        used = m.last_pipeline_pieces()
        ms2, mn2, s = timed(lambda: m.scorePoses(pts, pa))
        res[pieces] = (w, mean, cov, s)
        w1, mean1, cov1, s1 = res[1]
        print("cfg-%d pieces=%d (used %d): measure call %.3f ms (min %.3f) | scorePoses call %.3f ms (min %.3f, used %d) | "
              "raw scores equal %s, weights max rel diff %.1e, mean diff %.1e, cov rel diff %.1e"
              % (cfg, pieces, used, ms, mn, ms2, mn2, m.last_pipeline_pieces(), np.array_equal(s, s1),
                 float(np.max(np.abs(w - w1) / np.abs(w1).max())), float(np.max(np.abs(mean - mean1))),
                 float(np.max(np.abs(cov - cov1)) / np.abs(cov1).max())), flush=True)
