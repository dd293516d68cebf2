#!/usr/bin/env python3
"""When do the waves of the large-lattice search start and end their work, how many items does a
wave run, and how long is its longest item?  (Is there a tail?)  Needs the trace build:

    bash experiments/build_variant_lib.sh lane_trace ndt2d_match_lane.hip -DNDT2D_LANE_TRACE
    NDT2D_HIP_LIB=$PWD/experiments/bin/lane_trace.so python experiments/lane_wave_trace.py [lin ang]

wall_clock64() ticks at 100 MHz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

lin = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
ang = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
m = ScanMatcherNDT(0)
m.initialize("trace", **synth.matcher_params(2, search_linear_size=lin, search_linear_resolution=0.02,
                                             search_angular_size=ang, search_angular_resolution=0.005))
m.addScans(synth.map_scans(2))
guess, pts, _ = synth.query_scan(2)
n_th, n_lin, n_b = m.prepare_search(guess, pts)
buf = torch.zeros(5 * 8192 + 450000, dtype=torch.float64, device="cuda")
for _ in range(3):
    buf.zero_()
    torch.cuda.synchronize()
    m.set_timing(True)
    m.match_launch(0, n_th, scores_ptr=buf.data_ptr())
    m.synchronize()
ms = m.last_launch_ms()[0]
raw = buf.cpu().numpy()
t = raw[:5 * 8192].reshape(-1, 5)
t = t[t[:, 1] > 0]
item_start = raw[5 * 8192:]
t0 = t[:, 0].min()
start, end, items, longest, last_start = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, t[:, 2], t[:, 3] / 100.0, (t[:, 4] - t0) / 100.0
print("lin %d theta %d: %s, %.1f us (events, all kernels); %d waves" % (n_lin, n_th, m.last_variant(), ms * 1e3, len(t)))
q = [0, 10, 50, 90, 99, 100]
print("  wave start us  (percentiles %s): %s" % (q, np.percentile(start, q).round(1)))
print("  wave end us    : %s" % np.percentile(end, q).round(1))
print("  last item start: %s" % np.percentile(last_start, q).round(1))
print("  items per wave : %s" % np.percentile(items, q).round(0))
print("  longest item us: %s" % np.percentile(longest, q).round(1))
print("  mean wave life %.1f us of %.1f us = %.3f" % ((end - start).mean(), end.max(), (end - start).mean() / end.max()))
late = end > np.percentile(end, 50) + 0.5 * (end.max() - np.percentile(end, 50))
print("  waves ending in the last half of [median end, last end]: %d; their last item took %s us"
      % (late.sum(), np.percentile((end - last_start)[late], [0, 50, 100]).round(1)))

# per item: start stamps; an item's duration = the same wave's next stamp (or the wave's end) - its start.
# Without the wave id per item, use: duration of item = (next event of the wave); reconstruct from the waves'
# sorted starts is not possible here, so report WHEN items of each rank (theta step order) start instead.
n_items = int((item_start[:150000] > 0).sum())
st = (item_start[:n_items] - t0) / 100.0
p1 = (n_lin + 7) // 8
per_rank = p1 * p1 * (n_items // (n_th * p1 * p1))
print("  items %d; start time of the items by tenth of the item order (us, median / max): %s" % (
    n_items, " ".join("%.0f/%.0f" % (np.median(c), c.max()) for c in np.array_split(st, 10))))
dur = item_start[150000:150000 + n_items] / 100.0
st_all = (item_start[:n_items] - t0) / 100.0
print("  item duration us (percentiles %s): %s; sum %.0f wave-us = %.1f us x %d waves" % (
    q, np.percentile(dur, q).round(1), dur.sum(), dur.sum() / len(t), len(t)))
print("  duration by tenth of the item order (median / max): %s" % " ".join(
    "%.0f/%.0f" % (np.median(c), c.max()) for c in np.array_split(dur, 10)))
late_items = np.argsort(st_all + dur)[-10:]
print("  the ten items that end last: %s" % ", ".join("#%d start %.0f dur %.0f" % (i, st_all[i], dur[i]) for i in late_items))
# an LPT bound: makespan if the items (with these durations) were dealt longest-first to the waves
import heapq
h = [0.0] * len(t)
for d in np.sort(dur)[::-1]:
    heapq.heapreplace(h, h[0] + d)
print("  longest-first schedule of these durations on %d waves: %.1f us; in today's order: %.1f us" % (len(t), max(h), end.max()))
fl = item_start[300000:300000 + n_items]
print("  flagged beams (patch pre-test) against item duration:")
for lo, hi in ((0, 90), (90, 180), (180, 270), (270, 360), (360, 450), (450, 540), (540, 630), (630, 690), (690, 721)):
    sel = (fl >= lo) & (fl < hi)
    if sel.any():
        print("    flagged %3d..%3d: %6d items, duration us median %.0f, 90%% %.0f, max %.0f; share of all wave-us %.3f"
              % (lo, hi - 1, sel.sum(), np.median(dur[sel]), np.percentile(dur[sel], 90), dur[sel].max(), dur[sel].sum() / dur.sum()))
heavy = dur > 200
print("  items over 200 us: %d, their flagged beams: min %d median %d" % (heavy.sum(), fl[heavy].min() if heavy.any() else -1, np.median(fl[heavy]) if heavy.any() else -1))

out = os.path.join(ROOT, "gpurun_out", "lane_trace_items_%d_%d.npz" % (n_lin, n_th))
os.makedirs(os.path.dirname(out), exist_ok=True)
np.savez_compressed(out, start=st_all, dur=dur, flagged=fl, n_lin=n_lin, n_th=n_th, wave_start=start, wave_end=end)
print("  saved", out)
