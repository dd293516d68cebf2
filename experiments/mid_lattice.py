"""Small-lattice vs large-lattice search on mid-size lattices (cfg-2's map and scan, 720 beams,
the translation lattice of cfg-2, a varying number of theta steps): where is the crossover?"""
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402

from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

guess, pts, _ = synth.query_scan(2)
for lin_size, lin_res in ((1.0, 0.02), (0.5, 0.02), (0.3, 0.02)):
    for ang_size in (0.02, 0.05, 0.1, 0.2):
        m = ScanMatcherNDT(0)
        m.initialize("mid", **synth.matcher_params(2, search_linear_size=lin_size, search_linear_resolution=lin_res,
                                                   search_angular_size=ang_size, search_angular_resolution=0.005))
        m.addScans(synth.map_scans(2))
        n_th, n_lin, nb = m.prepare_search(guess, pts)
        p1 = (n_lin + 7) // 8
        row = "lin %d theta %3d items %5d:" % (n_lin, n_th, n_th * p1 * p1)
        for variant in ("small", "lane", "auto"):
            try:
                m.set_variant(variant)
                m.set_timing(True)
                ts = []
                for i in range(15):
                    m.matchScan(guess, pts)
                    if i >= 5:
                        ts.append(m.last_launch_ms()[0])
                row += "  %s %.1f us" % (variant, 1e3 * float(np.median(ts)))
                if variant == "auto":
                    row += " (" + m.last_variant().split("/")[2] + ")"
            except Exception as e:
                row += "  %s n/a" % variant
        print(row)
        m.close()
