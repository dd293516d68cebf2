#!/usr/bin/env python3
"""Particle scoring on a grid whose occupancy bitmap does not fit LDS (cfg-5's world at 0.125 m:
1601 x 1601 cells): the compacted kernel screening with one bit per 2 x 2 cells against the dense
kernel it used to fall back to.  1 M particles x 720 beams, kernel time (HIP events)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

for res in (0.25, 0.125, 0.1):
    m = ScanMatcherNDT(0)
    m.initialize("pf", **synth.matcher_params(5, ndt_resolution=res))
    m.addScans(synth.map_scans(5))
    _, pts, _ = synth.query_scan(5)
    pa = synth.particles(5)
    nb = m.prepare_beams(pts)
    d_p = torch.from_numpy(pa).cuda()
    d_w = torch.empty(len(pa), dtype=torch.float64, device="cuda")
    d_s = torch.empty(8, dtype=torch.float64, device="cuda")
    row = "resolution %.3f:" % res
    for variant in ("auto", "dense"):
        m.set_variant(variant)
        torch.cuda.synchronize()
        for _ in range(12):
            m.score_poses_launch(d_p.data_ptr(), len(pa), d_w.data_ptr(), d_s.data_ptr())
        m.synchronize()
        ms = m.launch_history_ms(8)
        row += "  %s %.3f ms (%.2e units/s) [%s]" % (variant, float(np.median(ms)), len(pa) * nb / (float(np.median(ms)) * 1e-3),
                                                     m.last_variant().replace("poses/lane-per-pose/", ""))
    print(row, flush=True)
    m.close()
