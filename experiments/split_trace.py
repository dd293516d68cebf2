"""Split items: who walks the parts, and how long a split item takes from its split to its last part.
Needs the trace build (see lane_wave_trace.py); argv: linear size, angular size, knob."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

lin, ang, knob = float(sys.argv[1]), float(sys.argv[2]), sys.argv[3]
os.environ["NDT2D_LANE_SPLIT"] = knob
m = ScanMatcherNDT(0)
m.initialize("trace", **synth.matcher_params(2, search_linear_size=lin, search_linear_resolution=0.02,
                                             search_angular_size=ang, search_angular_resolution=0.005))
m.addScans(synth.map_scans(2))
guess, pts, _ = synth.query_scan(2)
n_th, n_lin, n_b = m.prepare_search(guess, pts)
buf = torch.zeros(100 + 2 * 70000, dtype=torch.float64, device="cuda")
m.set_timing(True)
for _ in range(3):
    buf.zero_()
    torch.cuda.synchronize()
    m.match_launch(0, n_th, scores_ptr=buf.data_ptr())
    m.synchronize()
ms = m.last_launch_ms()[0]
t = buf.cpu().numpy()
pairs = t[100:].reshape(-1, 2)
sel = pairs[:, 0] > 0
took = (pairs[sel, 1] - pairs[sel, 0]) / 100.0
pushed_at = (pairs[sel, 0] - t[5]) / 100.0
done_at = (pairs[sel, 1] - t[5]) / 100.0
print("knob %s, lin %d theta %d: %.1f us; items split %d (parts queued) + %d (parts kept); parts walked by the splitting wave %d, taken from the queue %d (of which by the wave that walked the previous part %d)"
      % (knob, n_lin, n_th, ms * 1e3, t[0], t[1], t[2], t[3], t[4]))
q = [0, 10, 50, 90, 100]
print("  split at us %s; done at us %s; split -> done us %s" % (np.percentile(pushed_at, q).round(0), np.percentile(done_at, q).round(0), np.percentile(took, q).round(0)))
