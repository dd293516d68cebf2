#!/usr/bin/env python3
"""Weak-scaling layout of bench.py --gpus 8 on one GPU: cfg-2 with the angular
resolution refined 8-fold, the eight contiguous theta slabs timed one after the other.
How unequal are they?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402
from ndt_2d_amd import dist as shard  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
params = synth.matcher_params(2)
params["search_angular_resolution"] /= world
m = ScanMatcherNDT(0)
m.initialize("m", **params)
m.addScans(synth.map_scans(2))
guess, pts, _ = synth.query_scan(2)
n_th, n_lin, nb = m.prepare_search(guess, pts)
times = []
for r in range(world):
    b, e = shard.shard_range(n_th, r, world)
    for _ in range(6):
        m.match_launch(b, e)
    m.synchronize()
    times.append(float(np.median(m.launch_history_ms(5))))
print("n_th", n_th, "slab kernel ms:", np.round(times, 4), " max/mean = %.3f" % (max(times) / np.mean(times)))
times = []
for r in range(world):
    first, stride, count = shard.shard_strided(n_th, r, world)
    for _ in range(6):
        m.match_launch_strided(first, stride, count)
    m.synchronize()
    times.append(float(np.median(m.launch_history_ms(5))))
print("interleaved shares     ms:", np.round(times, 4), " max/mean = %.3f" % (max(times) / np.mean(times)))
