#!/usr/bin/env python3
"""Kernel time of the small-lattice search against its block plan (beam chunks C per patch,
patches P per block), read from HIP events and -- with rocprofv3 --kernel-trace -- from
the trace.  NDT2D_SMALL_CHUNKS / NDT2D_SMALL_PATCHES are read by the library per launch.

    python experiments/small_plan_sweep.py [defaults|d720|cfg1] [C,P ...]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "defaults"
over = {"defaults": dict(search_linear_size=0.05, search_linear_resolution=0.005, search_angular_size=0.1,
                         search_angular_resolution=0.0025, laser_max_beams=100),
        "d720": dict(search_linear_size=0.05, search_linear_resolution=0.005, search_angular_size=0.1,
                     search_angular_resolution=0.0025),
        "cfg1": {}}[which]
plans = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]] or [(0, 0)]
m = ScanMatcherNDT(0)
m.initialize("m", **synth.matcher_params(1, **over))
m.addScans(synth.map_scans(1))
guess, pts, _ = synth.query_scan(1)
n_th, n_lin, n_b = m.prepare_search(guess, pts)
for c, p in plans:
    for k, v in (("NDT2D_SMALL_CHUNKS", c), ("NDT2D_SMALL_PATCHES", p)):
        if v:
            os.environ[k] = str(v)
        else:
            os.environ.pop(k, None)
    for _ in range(30):
        m.match_launch(0, n_th)
    m.synchronize()
    ms = m.launch_history_ms(20)
    t0 = time.perf_counter()
    for _ in range(200):
        m.match_launch(0, n_th)
    m.synchronize()
    wall = (time.perf_counter() - t0) / 200
    print("%s C=%d P=%d: events %.1f us (min %.1f)  back-to-back %.1f us per launch  %s" % (
        which, c, p, 1e3 * sum(ms) / len(ms), 1e3 * min(ms), wall * 1e6, m.last_variant()))
