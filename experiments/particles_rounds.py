"""Particle kernel time against the number of particles on the cfg-3 map: is the cfg-3 launch
(1,563 groups of 64 x 4 waves over 4,096 wave slots = 1.53 rounds) paying for a half-empty second
round?  Prints kernel ms and units/s per size."""
import json, sys
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
from ndt_2d_amd import ScanMatcherNDT, synth
m = ScanMatcherNDT(0)
m.initialize("g", **synth.matcher_params(3))
m.addScans(synth.map_scans(3))
_, pts, _ = synth.query_scan(3)
nb = m.prepare_beams(pts)
out = {}
for n in (16384, 32768, 65536, 81920, 98304, 100000, 114688, 131072, 196608, 262144, 1000000):
    parts = synth.particles(3, n)
    d_parts = torch.from_numpy(parts).cuda()
    d_scores = torch.zeros(n, dtype=torch.float64, device="cuda")
    d_stats = torch.zeros(8, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ms = []
    for i in range(12):
        m.score_poses_launch(d_parts.data_ptr(), n, d_scores.data_ptr(), d_stats.data_ptr())
        t, _ = m.last_launch_ms()
        if i > 1:
            ms.append(t)
    k = float(np.median(ms))
    out[n] = {"kernel_ms": k, "units_per_s": n * nb / (k * 1e-3), "groups": (n + 63) // 64}
    print(n, out[n], m.last_variant())
