"""Split items of the large search (NDT2D_LANE_SPLIT): scores against the unsplit search, kernel time per
threshold.  cfg-2 by default; argv: linear size, angular size."""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402

from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

lin = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
ang = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
settings = sys.argv[3].split(":") if len(sys.argv) > 3 else ["0", "270,3", "360,3", "450,3", "540,3", "360,2", "450,2", "270,4"]
guess, pts, _ = synth.query_scan(2)
m = ScanMatcherNDT(0)
m.initialize("split", **synth.matcher_params(2, search_linear_size=lin, search_linear_resolution=0.02,
                                             search_angular_size=ang, search_angular_resolution=0.005))
m.addScans(synth.map_scans(2))
m.set_timing(True)
ref = None
for knob in settings:
    os.environ["NDT2D_LANE_SPLIT"] = knob
    t0 = time.time()
    r = m.matchScan(guess, pts, want_scores=True)
    first = time.time() - t0
    if first > 0.5:
        print("knob %s: first search took %.2f s -- giving up" % (knob, first), flush=True)
        break
    ts = []
    for i in range(12):
        r2 = m.matchScan(guess, pts)
        ts.append(m.last_launch_ms()[0])
    if ref is None:
        ref = r
    d = np.abs(r["scores"] - ref["scores"])
    print("knob %-6s %s: kernel %.1f us (min %.1f); vs unsplit: max |d| %.3e, %d of %d scores differ, winner %d (%s), score %.17g, repeat same winner %s"
          % (knob, m.last_variant().split("/")[-2], 1e3 * float(np.median(ts)), 1e3 * min(ts), d.max(), int((d > 0).sum()), d.size,
             r["best_index"], "same" if r["best_index"] == ref["best_index"] else "DIFFERENT", r["score"],
             r2["best_index"] == r["best_index"] and r2["score"] == r["score"]), flush=True)
