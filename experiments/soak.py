#!/usr/bin/env python3
"""Soak: the cfg-2 search launched many times; every launch must return the same
12-double record bit for bit (the work distribution is dynamic, the records must
not depend on it), and likewise a mid-size and a strided launch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402

m = ScanMatcherNDT(0)
m.initialize("m", **synth.matcher_params(2))
m.addScans(synth.map_scans(2))
guess, pts, _ = synth.query_scan(2)
n_th, n_lin, nb = m.prepare_search(guess, pts)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for name, launch in (("whole", lambda: m.match_launch(0, n_th)),
                     ("slab", lambda: m.match_launch(37, 111)),
                     ("strided", lambda: m.match_launch_strided(3, 8, 24))):
    ref = None
    for i in range(n if name == "whole" else n // 4):
        launch()
        rec = m.match_fetch()
        if ref is None:
            ref = rec.copy()
        elif not np.array_equal(rec, ref):
            print(name, "launch", i, "differs:", rec, ref)
            sys.exit(1)
    print(name, "ok:", ref[:2])
