#!/usr/bin/env python3
"""How often a wave of the large-lattice search takes each of its paths over one cfg-2 (or cfg-4
share) search -- the dynamic side of the instruction table of DESIGN.md section 3.1
(experiments/lane_path_mix.py multiplies it with the paths' static instruction lists).

    bash experiments/build_variant_lib.sh lane_paths ndt2d_match_lane.hip -DNDT2D_LANE_PATHS
    NDT2D_HIP_LIB=$PWD/experiments/bin/lane_paths.so python experiments/lane_paths.py [cfg] > profiles/r06_lane_paths.json

(the __device__ pointer is set by block 0 of the first launch: the counts are the SECOND launch's)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth, _capi  # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
m = ScanMatcherNDT(0)
m.initialize("paths", **synth.matcher_params(cfg))
m.addScans(synth.map_scans(cfg))
guess, pts, _ = synth.query_scan(cfg)
n_th, n_lin, n_b = m.prepare_search(guess, pts)
if cfg == 4:
    n_th_run = (n_th + 7) // 8            # share 0 of 8 (theta steps 0, 8, 16, ...)
    launch = lambda ptr: m.match_launch_strided(0, 8, n_th_run, scores_ptr=ptr)   # noqa: E731
else:
    n_th_run = n_th
    launch = lambda ptr: m.match_launch(0, n_th, scores_ptr=ptr)   # noqa: E731
hist = torch.zeros(256, dtype=torch.float64, device="cuda")
launch(hist.data_ptr())
m.synchronize()
hist.zero_()
torch.cuda.synchronize()
launch(hist.data_ptr())
m.synchronize()
h = hist.cpu().numpy()
names = ["items", "chunks_pretested", "beams_in_lookup_groups", "groups_with_a_live_lane", "beams_with_a_live_lane",
         "exact_evaluations", "evaluations_by_reference_index", "evaluations_with_exp", "skip_refreshes",
         "items_reduced", "single_beams"]
p1 = (n_lin + 7) // 8
out = {"what": "path counts of one search, wave-level events (experiments/lane_paths.py, -DNDT2D_LANE_PATHS build)",
       "cfg": cfg, "variant": m.last_variant(), "n_theta_run": n_th_run, "n_lin": n_lin, "beams": n_b,
       "wave_beams": n_th_run * p1 * p1 * n_b, "units": n_th_run * n_lin * n_lin * n_b,
       "build_info": _capi.build_info(),
       "counts": {n: float(h[200 + i]) for i, n in enumerate(names)}}
print(json.dumps(out, indent=1))
