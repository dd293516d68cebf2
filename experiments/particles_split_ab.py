"""Particle scoring kernel at cfg-3's size and at a 1-of-8 share of cfg-5 (125,000 particles on the
801 x 801 map), kernel ms by HIP events -- run under NDT2D_POSES_EIGHT_WAVES=0 / 1 / unset to compare
four- and eight-wave groups (a wave then walks two chunks of the beams or one)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ndt_2d_amd import ScanMatcherNDT, synth
print("NDT2D_POSES_EIGHT_WAVES =", os.environ.get("NDT2D_POSES_EIGHT_WAVES"))
for cfg, sizes in ((3, (64, 16384, 65536, 100000, 125000)), (5, (64, 125000, 1000000))):
    m = ScanMatcherNDT(0)
    m.initialize("g", **synth.matcher_params(cfg))
    m.addScans(synth.map_scans(cfg))
    _, pts, _ = synth.query_scan(cfg)
    nb = m.prepare_beams(pts)
    allp = synth.particles(cfg, max(sizes))
    for n in sizes:
        d_parts = torch.from_numpy(allp[:n].copy()).cuda()
        d_scores = torch.zeros(n, dtype=torch.float64, device="cuda")
        d_stats = torch.zeros(8, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        ms = []
        for i in range(14):
            m.score_poses_launch(d_parts.data_ptr(), n, d_scores.data_ptr(), d_stats.data_ptr())
            t, _ = m.last_launch_ms()
            if i > 3:
                ms.append(t)
        print("cfg-%d map %8d particles: kernel %.4f ms (min %.4f)  checksum %.17g  %s"
              % (cfg, n, float(np.median(ms)), min(ms), float(d_scores.sum().cpu()), m.last_variant()))
    m.close()
