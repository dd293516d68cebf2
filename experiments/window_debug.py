"""Window-compacted records against the gather form on the cfg-3 map with a 12 m range cut:
kernel ms per lattice, both forms (NDT2D_LANE_WINDOW=0/1), optional NDT2D_LANE_DEBUG geometry."""
import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from ndt_2d_amd import ScanMatcherNDT, synth
scans = synth.map_scans(3)
pts = synth.scan(synth.world_of(3), (1.0, 0.5, 0.3), 4242); pts = pts[np.hypot(pts[:, 0], pts[:, 1]) <= float(os.environ.get("CUT", "12"))]
print("beams", len(pts))
for lin, ang in ((0.6, 0.1), (0.6, 0.5), (1.0, 0.3), (1.0, 0.5)):
    p = synth.matcher_params(3, search_linear_size=lin, search_linear_resolution=0.02, search_angular_size=ang, search_angular_resolution=0.005)
    m = ScanMatcherNDT(0); m.initialize("g", **p); m.addScans(scans)
    out = []
    for w in ("1", "0"):
        os.environ["NDT2D_LANE_WINDOW"] = w
        ms = []
        for i in range(8):
            r = m.matchScan((1.06, 0.46, 0.31), pts); ms.append(m.last_launch_ms()[0])
        out.append((w, round(float(np.median(ms[2:])), 4), m.last_variant().split("/")[2:], r["best_index"]))
    n = r["n_candidates"]
    print(lin, ang, n, "units %.3g" % (n * len(pts)), out)
