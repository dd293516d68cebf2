"""cfg-4 (315.5 M candidates) on one GPU: ms per whole-lattice search, winner (experiments/r06_defer_ab.sh)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ndt_2d_amd import ScanMatcherNDT, synth
m = ScanMatcherNDT(0)
m.initialize("g", **synth.matcher_params(4))
m.addScans(synth.map_scans(4))
guess, pts, _ = synth.query_scan(4)
n_th, n_lin, nb = m.prepare_search(guess, pts)
ms = []
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.match_launch(0, n_th)
    rec = m.match_fetch()
    if i:
        ms.append((time.perf_counter() - t0) * 1e3)
res = m.finish_match(rec)
print("cfg-4 NDT2D_LANE_DEFER=%s: %.2f ms per search (min %.2f), kernel %.2f ms, winner %d score %.15g cov00 %.15g  %s"
      % (os.environ.get("NDT2D_LANE_DEFER"), float(np.median(ms)), min(ms), m.last_launch_ms()[0], int(rec[1]), res["score"],
         res["covariance"][0, 0], m.last_variant()))
