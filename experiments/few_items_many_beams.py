"""Few work items, many beams (the plugin's default lattice with all 720 beams; cfg-1): the
small-lattice search against the large search with a candidate's beams cut into 4 / 8 / 12 parts."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np  # noqa: E402

from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

guess, pts, _ = synth.query_scan(1)
cases = [("defaults x 720 beams", dict(search_linear_size=0.05, search_linear_resolution=0.005, search_angular_size=0.1,
                                       search_angular_resolution=0.0025)),
         ("cfg-1", {}),
         ("cfg-1 lattice x 360 beams", dict(laser_max_beams=360))]
for name, over in cases:
    m = ScanMatcherNDT(0)
    m.initialize("m", **synth.matcher_params(1, **over))
    m.addScans(synth.map_scans(1))
    n_th, n_lin, nb = m.prepare_search(guess, pts)
    p1 = (n_lin + 7) // 8
    row = "%-28s items %4d beams %3d:" % (name, n_th * p1 * p1, nb)
    ref = None
    for label, variant, parts in (("small", "small", None), ("lane/1", "lane", "1"), ("lane/4", "lane", "4"),
                                  ("lane/8", "lane", "8"), ("wave", "wave", None), ("auto", "auto", None)):
        if parts is None:
            os.environ.pop("NDT2D_LANE_PARTS", None)
        else:
            os.environ["NDT2D_LANE_PARTS"] = parts
        m.set_variant(variant)
        m.set_timing(True)
        ts = []
        for i in range(25):
            r = m.matchScan(guess, pts)
            if i >= 5:
                ts.append(m.last_launch_ms()[0])
        if ref is None:
            ref = r
        ok = r["best_index"] == ref["best_index"] and abs(r["score"] - ref["score"]) < 1e-12
        row += "  %s %.1f%s" % (label, 1e3 * float(np.median(ts)), "" if ok else "(!)")
        if variant == "auto":
            row += " us (" + "/".join(m.last_variant().split("/")[2:]) + ")"
    os.environ.pop("NDT2D_LANE_PARTS", None)
    print(row, flush=True)
    m.close()
