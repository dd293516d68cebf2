#!/usr/bin/env python3
"""A few pipelined ParticleFilter::measure calls at cfg-5 for a rocprofv3 --kernel-trace
--memory-copy-trace timeline:  python3 experiments/pipeline_trace.py <pieces> [measure|score]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, pf_measure, synth  # noqa: E402

pieces = int(sys.argv[1]) if len(sys.argv) > 1 else 4
what = sys.argv[2] if len(sys.argv) > 2 else "measure"
m = ScanMatcherNDT(0)
m.initialize("pf", **synth.matcher_params(5))
m.addScans(synth.map_scans(5))
_, pts, _ = synth.query_scan(5)
pa = synth.particles(5)
m.set_pipeline_pieces(pieces)
for _ in range(4):
    if what == "measure":
        pf_measure(m, pa, pts)
    else:
        m.scorePoses(pts, pa)
print("done", m.last_pipeline_pieces())
