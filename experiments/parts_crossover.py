#!/usr/bin/env python3
"""Beam parts of the large search (NDT2D_LANE_PARTS = 1 / 2 / 4 against the library's choice) on
lattices of n_th x 169 work items, 720 beams: whole matchScan call with the event pairs off."""
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1:
    from ndt_2d_amd import ScanMatcherNDT, synth
    n_th = int(sys.argv[1])
    guess, pts, _ = synth.query_scan(2)
    m = ScanMatcherNDT(0)
    m.initialize("x", **synth.matcher_params(2, search_linear_size=1.0, search_linear_resolution=0.02,
                                             search_angular_size=0.0025 * n_th, search_angular_resolution=0.005))
    m.addScans(synth.map_scans(2))
    for _ in range(5):
        m.matchScan(guess, pts)
    m.set_timing(False)
    ts = []
    for _ in range(25):
        t0 = time.perf_counter()
        m.matchScan(guess, pts)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("%.4f %s" % (statistics.median(ts), m.last_variant().split("/")[-1]))
    sys.exit(0)

for n_th in (28, 40, 60, 80, 100, 120):
    row = []
    for parts in ("", "1", "2", "4"):
        env = dict(os.environ)
        if parts:
            env["NDT2D_LANE_PARTS"] = parts
        else:
            env.pop("NDT2D_LANE_PARTS", None)
        out = subprocess.run([sys.executable, __file__, str(n_th)], env=env, capture_output=True, text=True).stdout.strip().splitlines()
        row.append("%s: %s" % (parts or "auto", out[-1] if out else "?"))
    print("n_th %3d = %6d items | %s" % (n_th, n_th * 169, " | ".join(row)), flush=True)
