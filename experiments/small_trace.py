#!/usr/bin/env python3
"""Where a block of the small-lattice search spends its time: per-wave shader-clock stamps
(block start, setup done, beams done, barrier passed, records written) from a library
built with -DNDT2D_SMALL_TRACE (experiments/build_trace_lib.sh -> experiments/bin/trace.so;
run with NDT2D_HIP_LIB pointing at it).

    NDT2D_HIP_LIB=experiments/bin/trace.so python experiments/small_trace.py [defaults|d720|cfg1|mid1352]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "defaults"
DEFAULTS = dict(search_linear_size=0.05, search_linear_resolution=0.005, search_angular_size=0.1,
                search_angular_resolution=0.0025, laser_max_beams=100)
over = {"defaults": DEFAULTS, "real30": DEFAULTS,
        "d720": dict(search_linear_size=0.05, search_linear_resolution=0.005, search_angular_size=0.1,
                     search_angular_resolution=0.0025),
        "mid1352": dict(search_linear_size=1.0, search_linear_resolution=0.02, search_angular_size=0.02,
                        search_angular_resolution=0.005),
        "cfg1": {}}[which]
m = ScanMatcherNDT(0)
if which == "real30":
    # a 30 m lidar's local map (245 x 245 cells): nine scans around the pose in cfg-5's world
    w = synth.world_of(5)
    guess, pts, true = synth.query_scan(5)
    scans = []
    for j in range(3):
        for i in range(3):
            x, y = true[0] + (i - 1) * 0.5, true[1] + (j - 1) * 0.5
            if not synth.pose_blocked(w, x, y):
                scans.append(((x, y, 0.0), synth.scan(w, (x, y, 0.0), 77 + 10 * j + i)))
    m.initialize("m", **dict(synth.matcher_params(5, **over), range_max=30.0))
    m.addScans(scans)
    guess = true + np.array([0.02, -0.02, 0.01])
else:
    m.initialize("m", **synth.matcher_params(1, **over))
    m.addScans(synth.map_scans(1))
    guess, pts, _ = synth.query_scan(1)
n_th, n_lin, n_b = m.prepare_search(guess, pts)
buf = torch.zeros(8192 * 16 * 8, dtype=torch.float64, device="cuda:0")
isolated = len(sys.argv) > 2 and sys.argv[2] == "isolated"
for _ in range(5):
    m.match_launch(0, n_th, scores_ptr=buf.data_ptr())
    if isolated:
        # as the node calls it: one launch, the host waits for it, does something else, calls again
        m.synchronize()
        import time
        time.sleep(0.002)
m.synchronize()
raw = buf.cpu().numpy()
tail = raw[-8:].copy()      # the last block's stamp after its final reduction and host publish
raw[-8:] = 0
t = raw.reshape(-1, 16, 8)
used = t[:, :, 0].max(axis=1) > 0
t = t[used]
act = t[:, :, 6] > 0
clk = 2.1e9   # shader clock (s_memtime ticks; MI355X_MICROARCH.md), per-XCD time bases
print("%s%s: %d blocks, %d active waves per block (max)" % (which, " (isolated launches)" if isolated else "", len(t), int(act.sum(axis=1).max())))
# block start / end on the chip-wide 100 MHz clock (the shader clock counters above have
# a time base per CU)
starts = np.array([t[b, :, 5][act[b]].min() for b in range(len(t))]) / 100.0
ends = np.array([t[b, :, 7][act[b]].max() for b in range(len(t))]) / 100.0
t0 = starts.min()
late = starts - t0
print("  block start after the launch's first block: median %.2f  p90 %.2f  max %.2f us; blocks starting > 5 us late: %d"
      % (np.median(late), np.percentile(late, 90), late.max(), int((late > 5.0).sum())))
print("  first block start -> last record written: %.1f us; block duration median %.1f  max %.1f us"
      % (ends.max() - t0, np.median(ends - starts), (ends - starts).max()))
print("  first block start -> the last block's flag has left (final reduction + publish done): %.1f us (block %d)"
      % (tail[0] / 100.0 - t0, int(tail[1])))
d_setup = (t[:, :, 1] - t[:, :, 0])[act] / clk * 1e6
d_main = (t[:, :, 2] - t[:, :, 1])[act] / clk * 1e6
d_wait = (t[:, :, 3] - t[:, :, 2])[act] / clk * 1e6
d_rec = (t[:, :, 4] - t[:, :, 3])[act] / clk * 1e6
print("  per wave: setup %.2f (max %.2f)  beams %.2f (min %.2f max %.2f)  waiting at the barrier %.2f (max %.2f)"
      "  combine + record %.2f (max %.2f) us"
      % (d_setup.mean(), d_setup.max(), d_main.mean(), d_main.min(), d_main.max(), d_wait.mean(), d_wait.max(),
         d_rec.mean(), d_rec.max()))
# when the blocks start and how long they run: the launch's shape (a mid-size lattice needs more
# than one round of block slots; its time is the rounds' critical path)
dur = ends - starts
order = np.argsort(starts)
print("  blocks started by: 25 %% %.1f  50 %% %.1f  75 %% %.1f  100 %% %.1f us; durations: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us"
      % tuple(list(np.percentile(late, [25, 50, 75, 100])) + list(np.percentile(dur, [10, 50, 90, 100]))))
slow = np.argsort(dur)[-8:]
print("  the eight longest blocks: " + ", ".join("start %.1f dur %.1f" % (late[b], dur[b]) for b in slow))
last = np.argsort(ends)[-8:]
print("  the eight blocks that end last: " + ", ".join("start %.1f dur %.1f end %.1f" % (late[b], dur[b], ends[b] - t0) for b in last))
running = [(int(((starts <= t0 + x) & (ends > t0 + x)).sum())) for x in np.arange(0.0, ends.max() - t0, 5.0)]
print("  blocks in flight every 5 us: " + " ".join(str(r) for r in running))
