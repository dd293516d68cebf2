#!/usr/bin/env python3
"""Kernel time of the particle scoring launch (cfg-3 and cfg-5) for the library
NDT2D_HIP_LIB points at; run once per build for an A/B in one GPU session."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

for cfg in (3, 5):
    m = ScanMatcherNDT(0)
    m.initialize("pf", **synth.matcher_params(cfg))
    m.addScans(synth.map_scans(cfg))
    _, pts, _ = synth.query_scan(cfg)
    pa = synth.particles(cfg)
    nb = m.prepare_beams(pts)
    d_p = torch.from_numpy(pa).cuda()
    d_w = torch.empty(len(pa), dtype=torch.float64, device="cuda")
    d_s = torch.empty(8, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(30):
        m.score_poses_launch(d_p.data_ptr(), len(pa), d_w.data_ptr(), d_s.data_ptr())
    m.synchronize()
    ms = m.launch_history_ms(20)
    w = d_w.cpu().numpy()
    print("cfg-%d %s: kernel %.4f ms (min %.4f), %.3e units/s, checksum %.17g"
          % (cfg, m.last_variant(), float(np.median(ms)), min(ms), len(pa) * nb / (float(np.median(ms)) * 1e-3),
             float(np.sum(w))))
