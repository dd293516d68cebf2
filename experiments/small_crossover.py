#!/usr/bin/env python3
"""Where the small-lattice search (one launch, blocks per tile) hands over to the large one
(persistent waves, beam parts, reduction launches): lattices of n_th x 169 work items on cfg-2's
map and scan, main kernel time of "auto", "small" and "lane" and the whole matchScan call with the
event pairs off (as the plugin runs), medians.
    python experiments/small_crossover.py [beams]"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

guess, pts, _ = synth.query_scan(2)
scans = synth.map_scans(2)
beams = int(sys.argv[1]) if len(sys.argv) > 1 else 720
for n_th in (12, 16, 20, 24, 30, 36):
    row = []
    for variant in ("auto", "small", "lane"):
        m = ScanMatcherNDT(0)
        m.initialize("x", **synth.matcher_params(2, search_linear_size=1.0, search_linear_resolution=0.02,
                                                 search_angular_size=0.0025 * n_th, search_angular_resolution=0.005, laser_max_beams=beams))
        m.addScans(scans)
        try:
            m.set_variant(variant)
            for _ in range(4):
                r = m.matchScan(guess, pts)
            ks, calls = [], []
            import time
            for _ in range(9):
                r = m.matchScan(guess, pts)
                ks.append(m.last_launch_ms()[0])
            m.set_timing(False)      # as the plugin runs: no event pairs around the kernels
            for _ in range(25):
                t0 = time.perf_counter()
                r = m.matchScan(guess, pts)
                calls.append((time.perf_counter() - t0) * 1e3)
            row.append("%s %.4f/%.4f (%s)" % (variant, statistics.median(ks), statistics.median(calls), m.last_variant().split("/")[2][:14]))
        except Exception as e:  # noqa: BLE001
            row.append("%s n/a (%s)" % (variant, str(e)[:30]))
        m.close()
    print("n_th %2d = %5d items: %s" % (n_th, r["n_candidates"] // 64 if False else n_th * 169, " | ".join(row)), flush=True)
