#!/usr/bin/env python3
"""How full is an exact evaluation of the large-lattice search?  Histogram of the lanes an
evaluation is run for (live and occupied / near a boundary) and of the lanes whose term
can change their sum, over one cfg-2 (or cfg-4) search.  Needs the histogram build:

    bash experiments/build_variant_lib.sh lane_hist ndt2d_match_lane.hip -DNDT2D_LANE_HIST
    NDT2D_HIP_LIB=$PWD/experiments/bin/lane_hist.so python experiments/lane_useful_hist.py [cfg]

(the __device__ pointer is set by block 0 of the first launch: the histogram is taken from
the SECOND launch)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m = ScanMatcherNDT(0)
m.initialize("hist", **synth.matcher_params(cfg))
m.addScans(synth.map_scans(cfg))
guess, pts, _ = synth.query_scan(cfg)
n_th, n_lin, n_b = m.prepare_search(guess, pts)
hist = torch.zeros(256, dtype=torch.float64, device="cuda")
m.match_launch(0, n_th, scores_ptr=hist.data_ptr())
m.synchronize()
hist.zero_()
torch.cuda.synchronize()
m.match_launch(0, n_th, scores_ptr=hist.data_ptr())
m.synchronize()
h = hist.cpu().numpy()
flagged, needed = h[0:65], h[70:135]
n_eval = flagged.sum()
p1 = (n_lin + 7) // 8
wave_beams = n_th * p1 * p1 * n_b
print("cfg-%d %s: %d exact evaluations = %.3f of %d wave-beams" % (cfg, m.last_variant(), n_eval, n_eval / wave_beams, wave_beams))
k = np.arange(65)
print("lanes flagged per evaluation: mean %.1f; lanes whose term matters: mean %.1f; exp needed in %.3f of the evaluations"
      % ((flagged * k).sum() / n_eval, (needed * k).sum() / n_eval, h[140] / (h[140] + h[141])))
for name, arr in (("flagged", flagged), ("needed", needed)):
    c = np.cumsum(arr) / arr.sum()
    print("  %s: share of evaluations with <= 8 / 16 / 32 / 48 / 63 lanes: %s; exactly 64: %.3f"
          % (name, " ".join("%.3f" % c[i] for i in (8, 16, 32, 48, 63)), arr[64] / arr.sum()))
# greedy packing of consecutive evaluations into batches of <= 64 lanes (whole evaluations,
# in order -- the register form of a FIFO): batches per evaluation if counts were i.i.d.
rng = np.random.default_rng(0)
for name, arr in (("flagged", flagged), ("needed", needed)):
    sample = rng.choice(65, size=200000, p=arr / arr.sum())
    sample = sample[sample > 0]
    batches, fill = 0, 0
    for v in sample:
        if fill + v > 64:
            batches += 1
            fill = 0
        fill += v
    split = np.ceil(sample.sum() / 64.0)
    print("  %s: whole-evaluation packing -> %.3f batches per evaluation; with splitting (ideal FIFO) %.3f"
          % (name, batches / len(sample), split / len(sample)))
