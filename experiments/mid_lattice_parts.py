"""Mid-size lattices (cfg-2's map and scan, 720 beams): the large search with a candidate's
beams cut into 1 / 2 / 4 / 8 parts (NDT2D_LANE_PARTS) next to the small-lattice search.
Kernel time incl. reductions (HIP events), median of 10."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402

from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

guess, pts, _ = synth.query_scan(2)
cases = [(0.3, 0.05), (0.3, 0.1), (0.3, 0.2), (0.5, 0.05), (0.5, 0.1), (0.5, 0.2), (1.0, 0.02), (1.0, 0.05),
         (1.0, 0.1), (1.0, 0.2), (1.0, 0.35)]
for lin_size, ang_size in cases:
    m = ScanMatcherNDT(0)
    m.initialize("mid", **synth.matcher_params(2, search_linear_size=lin_size, search_linear_resolution=0.02,
                                               search_angular_size=ang_size, search_angular_resolution=0.005))
    m.addScans(synth.map_scans(2))
    n_th, n_lin, nb = m.prepare_search(guess, pts)
    p1 = (n_lin + 7) // 8
    row = "lin %3d theta %3d items %5d:" % (n_lin, n_th, n_th * p1 * p1)
    ref = None
    for label, variant, parts in (("small", "small", None), ("lane/1", "lane", "1"), ("lane/2", "lane", "2"),
                                  ("lane/4", "lane", "4"), ("lane/8", "lane", "8"), ("auto", "auto", None)):
        if parts is None:
            os.environ.pop("NDT2D_LANE_PARTS", None)
        else:
            os.environ["NDT2D_LANE_PARTS"] = parts
        try:
            m.set_variant(variant)
            m.set_timing(True)
            ts = []
            for i in range(15):
                r = m.matchScan(guess, pts)
                if i >= 5:
                    ts.append(m.last_launch_ms()[0])
            if ref is None:
                ref = r
            ok = r["best_index"] == ref["best_index"] and abs(r["score"] - ref["score"]) < 1e-12
            row += "  %s %.1f%s" % (label, 1e3 * float(np.median(ts)), "" if ok else "(!)")
            if variant == "auto":
                row += " us (" + "/".join(m.last_variant().split("/")[2:]) + ")"
        except Exception:
            row += "  %s n/a" % label
    os.environ.pop("NDT2D_LANE_PARTS", None)
    print(row, flush=True)
    m.close()
