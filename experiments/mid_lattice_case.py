"""One mid-size lattice on cfg-2's map and scan, 12 searches (for rocprofv3: experiments/pmc_midsize.sh).
argv: linear size, angular size [variant]."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

lin_size, ang_size = float(sys.argv[1]), float(sys.argv[2])
variant = sys.argv[3] if len(sys.argv) > 3 else "auto"
guess, pts, _ = synth.query_scan(2)
m = ScanMatcherNDT(0)
m.initialize("mid", **synth.matcher_params(2, search_linear_size=lin_size, search_linear_resolution=0.02,
                                           search_angular_size=ang_size, search_angular_resolution=0.005))
m.addScans(synth.map_scans(2))
m.set_variant(variant)
n_th, n_lin, nb = m.prepare_search(guess, pts)
for i in range(12):
    r = m.matchScan(guess, pts)
print("lin %d theta %d beams %d units %d variant %s best %d" %
      (n_lin, n_th, nb, n_th * n_lin * n_lin * nb, m.last_variant(), r["best_index"]))
