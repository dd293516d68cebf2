#!/usr/bin/env python3
"""The default search as a CALL (launch, host waits on the flag, next call): the host's wait against
the GPU's own span first block start -> flag left (trace build, NDT2D_HIP_LIB=experiments/bin/trace.so)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

m = ScanMatcherNDT(0)
m.initialize("m", **synth.matcher_params(1, search_linear_size=0.05, search_linear_resolution=0.005,
                                         search_angular_size=0.1, search_angular_resolution=0.0025,
                                         laser_max_beams=100))
m.addScans(synth.map_scans(1))
guess, pts, _ = synth.query_scan(1)
m.set_timing(False)
buf = torch.zeros(8192 * 16 * 8, dtype=torch.float64, device="cuda:0")
spans, waits, firsts = [], [], []
for i in range(300):
    n_th, n_lin, n_b = m.prepare_search(guess, pts)
    t0 = time.perf_counter()
    m.match_launch(0, n_th, scores_ptr=buf.data_ptr())
    m.match_fetch()
    waits.append((time.perf_counter() - t0) * 1e6)
    raw = buf.cpu().numpy()
    tail = raw[-8:].copy()
    raw[-8:] = 0
    t = raw.reshape(-1, 16, 8)[:240]
    act = t[:, :, 6] > 0
    starts = np.array([t[b, :, 5][act[b]].min() for b in range(len(t))]) / 100.0
    ends = np.array([t[b, :, 7][act[b]].max() for b in range(len(t))]) / 100.0
    spans.append(tail[0] / 100.0 - starts.min())
    firsts.append(ends.max() - starts.min())
    buf.zero_()
    torch.cuda.synchronize()
print("host: launch + wait for the flag (Python): median %.1f us" % np.median(waits[50:]))
print("GPU: first block start -> last record written: median %.1f us; -> flag left: median %.1f us"
      % (np.median(firsts[50:]), np.median(spans[50:])))
