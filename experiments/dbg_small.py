import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from ndt_2d_amd import ScanMatcherNDT, synth
scans = synth.map_scans(3)
for ov in [dict(search_angular_size=0.05, search_linear_size=0.5), dict(search_angular_size=0.05, search_linear_size=0.5, laser_max_beams=100), dict(search_angular_size=0.05, search_linear_size=0.1)]:
    params = synth.matcher_params(3, **ov)
    gpu = ScanMatcherNDT(0); gpu.initialize("t", **params); gpu.addScans(scans)
    guess, pts, _ = synth.query_scan(3)
    print(ov, "rmax", np.hypot(pts[:,0],pts[:,1]).max(), "guess", guess)
    for v in ("small","lane","auto"):
        gpu.set_variant(v)
        try:
            r = gpu.matchScan(guess, pts); print(v, gpu.last_variant(), r["best_index"], r["score"])
        except Exception as e: print(v, "ERR", e)
