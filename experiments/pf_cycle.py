#!/usr/bin/env python3
"""Latency of the particle filter's per-scan cycle (reference src/ndt_mapper.cpp:473-477:
update + measure + resample + getMean) with the particles resident on the GPU, for
particle counts a localisation run uses, next to the CPU oracle's measure()."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402
from ndt_2d_amd.particle_filter import MotionModel, ParticleFilter  # noqa: E402
import oracle_lib as O  # noqa: E402  (CPU baseline only)

params = synth.matcher_params(1, laser_max_beams=100)
m = ScanMatcherNDT(0)
m.initialize("global_scan_matcher", **params)
m.addScans(synth.map_scans(1))
om = O.ScanMatcherNDT()
om.initialize(**params)
om.addScans(synth.map_scans(1))
_, pts, true_pose = synth.query_scan(1)

for n in (500, 2000, 10000, 100000):
    pf = ParticleFilter(n, n, MotionModel(0.1, 0.1, 0.1, 0.1, 0.05), m, seed=3)
    pf.init(true_pose[0], true_pose[1], true_pose[2], 0.2, 0.2, 0.1)
    t_up, t_me = [], []
    for i in range(60):
        t0 = time.perf_counter()
        pf.update(0.01, 0.0, 0.002)
        t1 = time.perf_counter()
        pf.measure(m, pts)
        mean = pf.getMean()
        t2 = time.perf_counter()
        if i >= 10:
            t_up.append(t1 - t0)
            t_me.append(t2 - t1)
    t0 = time.perf_counter()
    pf.resample(0.01, 0.99)
    t_rs = time.perf_counter() - t0
    pa = np.random.default_rng(0).normal(true_pose, [0.2, 0.2, 0.1], size=(min(n, 2000), 3))
    t0 = time.perf_counter()
    O.pf_measure(om, pa, pts, copy_points=True)
    cpu = (time.perf_counter() - t0) / len(pa) * n
    print("n=%6d: update %.0f us, measure %.0f us, resample(host KLD) %.0f us; CPU measure %.0f us; mean %s"
          % (n, np.median(t_up) * 1e6, np.median(t_me) * 1e6, t_rs * 1e6, cpu * 1e6, np.round(mean, 3)))
