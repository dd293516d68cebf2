"""cfg-2 with the control variant `lane-noskip`: EVERY beam of every candidate takes the exact path (no pre-test,
no look-up skip, exp always) -- the exact path as a workload of its own (experiments/pmc_noskip.sh)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

variant = sys.argv[1] if len(sys.argv) > 1 else "lane-noskip"
guess, pts, _ = synth.query_scan(2)
m = ScanMatcherNDT(0)
m.initialize("noskip", **synth.matcher_params(2))
m.addScans(synth.map_scans(2))
m.set_variant(variant)
m.set_timing(True)
ts = []
for i in range(6):
    r = m.matchScan(guess, pts)
    ts.append(m.last_launch_ms()[0])
print("%s: %s, kernel %.3f ms (min %.3f), best %d" % (variant, m.last_variant(), float(np.median(ts[1:])), min(ts), r["best_index"]))
