#!/usr/bin/env python3
"""The particle scoring launch as one fused kernel ("fused") against the three-launch form
("split": prepare + screen + drain, ndt2d_poses_split.hip): kernel time (HIP events around the
whole launch sequence), scores and moment sums compared bit for bit, cfg-3 and cfg-5 and a
sweep of particle counts on the cfg-3 map."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402


def run(m, variant, d_p, n, d_w, d_s, reps=30):
    m.set_variant(variant)
    for _ in range(reps):
        m.score_poses_launch(d_p.data_ptr(), n, d_w.data_ptr(), d_s.data_ptr())
    m.synchronize()
    ms = m.launch_history_ms(20)
    return float(np.median(ms)), float(min(ms)), d_w.cpu().numpy().copy(), d_s.cpu().numpy().copy(), m.last_variant()


counts = [int(c) for c in sys.argv[1:]] or None
for cfg in (3, 5):
    m = ScanMatcherNDT(0)
    m.initialize("pf", **synth.matcher_params(cfg))
    m.addScans(synth.map_scans(cfg))
    _, pts, _ = synth.query_scan(cfg)
    pa = synth.particles(cfg)
    nb = m.prepare_beams(pts)
    sizes = [len(pa)]
    if cfg == 3:
        sizes += [1, 63, 64, 65, 1000, 16384, 65536] + (counts or [])
    for n in sizes:
        reps = -(-n // len(pa))
        pn = np.tile(pa, (reps, 1))[:n].copy()
        d_p = torch.from_numpy(pn).cuda()
        d_w = torch.empty(n, dtype=torch.float64, device="cuda")
        d_s = torch.empty(8, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        res = {}
        for v in ("fused", "split", "compact-exact"):
            res[v] = run(m, v, d_p, n, d_w, d_s)
        f, s, x = res["fused"], res["split"], res["compact-exact"]
        print("cfg-%d n=%d beams=%d: fused %.4f ms (min %.4f) | split %.4f ms (min %.4f) | scores equal: split==fused %s, "
              "split==exact %s; sums rel diff %.2e  [%s]"
              % (cfg, n, nb, f[0], f[1], s[0], s[1], np.array_equal(f[2], s[2]), np.array_equal(x[2], s[2]),
                 float(np.max(np.abs(f[3] - s[3]) / (np.abs(f[3]) + 1e-300))), s[4]), flush=True)
