"""Kernel time of the large search against the share of the lattice it is given (strided
theta shares: every share has the same mix of cheap and expensive steps): the intercept
of the fit is what does not shrink with the work -- launch, LDS image, the tail in which
the last items finish."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402

from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m = ScanMatcherNDT(0)
m.initialize("scaling", **synth.matcher_params(cfg))
m.addScans(synth.map_scans(cfg))
guess, pts, _ = synth.query_scan(cfg)
n_th, n_lin, n_b = m.prepare_search(guess, pts)
rows = []
for stride in (1, 2, 4, 8, 16):
    count = (n_th + stride - 1) // stride
    for _ in range(5):
        m.match_launch_strided(0, stride, count)
    ts = []
    for _ in range(20):
        m.match_launch_strided(0, stride, count)
        m.match_fetch()
        ts.append(m.last_launch_ms()[0])
    rows.append((count, float(np.median(ts)), m.last_launch_ms()[1]))
    print("theta steps %4d: %.4f ms (%d kernels) %s" % (count, rows[-1][1], rows[-1][2], m.last_variant()))
x = np.array([r[0] for r in rows], dtype=float)
y = np.array([r[1] for r in rows])
b, a = np.polyfit(x[:3], y[:3], 1)
print("fit over the three largest: %.4f ms + %.5f ms per theta step (x %d = %.4f ms)" % (a, b, n_th, b * n_th))
