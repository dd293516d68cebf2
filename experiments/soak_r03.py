#!/usr/bin/env python3
"""Soak of the paths round 3 touched: every repetition must return the first one's bits.
  - the node's default search (small-lattice kernel: records -> ticket -> last block's reduction),
  - a mid-size lattice in beam parts, a forced multi-slab launch,
  - the mapper's cycle on a 245 x 245 grid (sparse install + search), map changing every cycle,
  - ParticleFilter::measure of 500 particles in one launch.
    python experiments/soak_r03.py [seconds per leg]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, pf_measure, synth  # noqa: E402

T = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
DEFAULTS = dict(search_linear_size=0.05, search_linear_resolution=0.005, search_angular_size=0.1,
                search_angular_resolution=0.0025, laser_max_beams=100)


def leg(name, fn):
    ref = fn()
    n, t0 = 1, time.time()
    while time.time() - t0 < T:
        got = fn()
        n += 1
        if got != ref:
            print(name, "repetition", n, "differs")
            sys.exit(1)
    print("%s: %d repetitions, identical" % (name, n), flush=True)


def blob(r):
    return r["pose"].tobytes() + np.float64(r["score"]).tobytes() + r["covariance"].tobytes()


guess, pts, _ = synth.query_scan(1)
m = ScanMatcherNDT(0)
m.initialize("d", **synth.matcher_params(1, **DEFAULTS))
m.addScans(synth.map_scans(1))
m.set_timing(False)
leg("default search", lambda: blob(m.matchScan(guess, pts)))
parts = synth.particles(3, 500) * [4.0 / 23.0, 4.0 / 23.0, 1.0]
leg("measure 500", lambda: b"".join(a.tobytes() for a in pf_measure(m, parts, pts)))
m.close()

m = ScanMatcherNDT(0)
m.initialize("mid", **synth.matcher_params(2, search_angular_size=0.1))
m.addScans(synth.map_scans(2))
g2, p2, _ = synth.query_scan(2)
leg("beam parts", lambda: blob(m.matchScan(g2, p2)))
os.environ["NDT2D_LANE_SLAB_ITEMS"] = "3000"
leg("beam parts", lambda: blob(m.matchScan(g2, p2)))
m.close()
m = ScanMatcherNDT(0)
m.initialize("slab", **synth.matcher_params(2))
m.addScans(synth.map_scans(2))
leg("slabs", lambda: blob(m.matchScan(g2, p2)))
del os.environ["NDT2D_LANE_SLAB_ITEMS"]
m.close()

w = synth.world_of(5)
g5, p5, t5 = synth.query_scan(5)
maps = []
for k in range(4):
    scans = []
    for j in range(3):
        for i in range(3):
            x, y = t5[0] + 0.5 * (i - 1) + 0.1 * k, t5[1] + 0.5 * (j - 1) - 0.07 * k
            if not synth.pose_blocked(w, x, y):
                scans.append(((x, y, 0.0), synth.scan(w, (x, y, 0.0), 77 + 10 * j + i + 100 * k)))
    maps.append(scans)
m = ScanMatcherNDT(0)
m.initialize("real", **dict(synth.matcher_params(5, **DEFAULTS), range_max=30.0))
m.set_timing(False)
pose = t5 + np.array([0.02, -0.02, 0.01])
state = [0]


def cycle():
    out = b""
    for scans in maps:          # four different maps, whose extents differ by a cell now and then
        m.reset()
        m.addScans(scans)
        out += np.float64(m.scoreScan(pose, p5)).tobytes() + blob(m.matchScan(pose, p5))
    return out


leg("mapper cycle on 245 x 245 grids", cycle)
m.close()
print("soak ok")
