#!/usr/bin/env python3
"""cfg-3 / cfg-5 particle scoring in the three-launch form, 20 launches each, for
rocprofv3 --kernel-trace --stats (per-kernel durations of prepare / screen / drain)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "split"
cfgs = [int(c) for c in sys.argv[2:]] or [3, 5]
for cfg in cfgs:
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
    m.set_variant(variant)
    for _ in range(20):
        m.score_poses_launch(d_p.data_ptr(), len(pa), d_w.data_ptr(), d_s.data_ptr())
    m.synchronize()
    print(cfg, m.last_variant(), float(np.median(m.launch_history_ms(10))))
