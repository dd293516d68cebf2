"""Kernel time of the batched particle scoring against the number of particles (cfg-3 map):
does a launch pay for whole rounds of resident blocks?"""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402

from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

m = ScanMatcherNDT(0)
m.initialize("rounds", **synth.matcher_params(3))
m.addScans(synth.map_scans(3))
_, pts, _ = synth.query_scan(3)
parts = synth.particles(3, 262144)
m.prepare_beams(pts)
import torch  # noqa: E402
dev = torch.device("cuda:0")
d_parts = torch.from_numpy(np.ascontiguousarray(parts)).to(dev)
d_w = torch.zeros(len(parts), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for n in (16384, 32768, 49152, 65536, 66000, 81920, 100000, 114688, 131072, 196608, 262144):
    ts = []
    for i in range(25):
        m.score_poses_launch(d_parts.data_ptr(), n, d_w.data_ptr(), None)
        m.synchronize()
        if i >= 5:
            ts.append(m.last_launch_ms()[0])
    t = float(np.median(ts))
    print("%7d particles (%5d groups): %.4f ms  %.3e units/s  %s" % (n, (n + 63) // 64, t, n * 720 / t * 1e3, m.last_variant()))
