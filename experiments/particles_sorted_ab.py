"""cfg-5 (1,000,000 particles, 801 x 801 map): the scoring kernel on the particles as generated against
the same particles sorted on the HOST by 8 x 8-cell tile of the grid (outside the timed region): the
upper bound of what a device-side sort by tile could give the phase-B record gathers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ndt_2d_amd import ScanMatcherNDT, synth
m = ScanMatcherNDT(0)
m.initialize("g", **synth.matcher_params(5))
m.addScans(synth.map_scans(5))
_, pts, _ = synth.query_scan(5)
nb = m.prepare_beams(pts)
parts = synth.particles(5)
_, sx, sy, cs, ox, oy = m.grid()
tile = (np.floor((parts[:, 1] - oy) / cs / 8).astype(np.int64) * ((sx + 7) // 8)
        + np.floor((parts[:, 0] - ox) / cs / 8).astype(np.int64))
order = np.argsort(tile, kind="stable")
sets = {"as generated": parts, "sorted by 8x8-cell tile": parts[order].copy(),
        "sorted by tile, then by heading": parts[np.lexsort((parts[:, 2], tile))].copy()}
ref = None
for name, p in sets.items():
    d_parts = torch.from_numpy(p).cuda()
    d_scores = torch.zeros(len(p), dtype=torch.float64, device="cuda")
    d_stats = torch.zeros(8, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ms = []
    for i in range(14):
        m.score_poses_launch(d_parts.data_ptr(), len(p), d_scores.data_ptr(), d_stats.data_ptr())
        t, _ = m.last_launch_ms()
        if i > 3:
            ms.append(t)
    s = d_scores.cpu().numpy()
    if ref is None:
        ref = s
        same = True
    elif name.startswith("sorted by 8x8"):
        same = bool(np.array_equal(s, ref[order]))
    else:
        same = None
    print("%-34s kernel %.4f ms (min %.4f)  scores equal to the unsorted ones (permuted): %s" % (name, float(np.median(ms)), min(ms), same))
