#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

The reference cannot be built or imported here (C++ needing Eigen3/rclcpp/...),
and its own tests hold vectors only for Cell / NDT::likelihood -- those are
checked in tests/test_oracle_reference_vectors.py and stored in
reference_ndt_model_tests.json below.  The remaining fixtures are outputs of the
oracle (oracle/ndt2d_oracle.c, the CPU restatement of the reference) on the
synthetic configs of BASELINE.md section 3, so they are "parity unpinned" by the
reference but pin the GPU path to the restatement across boxes and rounds.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib as O  # noqa: E402
from ndt_2d_amd import synth  # noqa: E402


def reference_vectors():
    """Inputs and expected values of the reference's test/ndt_model_tests.cpp (data only)."""
    return {
        "source": "mikeferguson/ndt_2d test/ndt_model_tests.cpp:32-230",
        "test_ndt_cell": {
            "points_first": [[3.5, 3.5], [3.5, 3.5], [3.4, 3.45], [3.6, 3.55]],
            "points_more": [[3.6, 3.45], [3.4, 3.55]],
            "mean": [3.5, 3.5], "cov": [0.008, 0.0, 0.002], "cov_tol": 0.001,
            "score_at_mean": 1.0, "score_1sigma": 0.6065, "score_2sigma": 0.1353,
            "score_tol": 0.001},
        "test_ndt_cell_no_x_variation": {
            "points": [[3.5, 3.5], [3.5, 3.45], [3.5, 3.45], [3.5, 3.55], [3.5, 3.55]],
            "information_00": 400000.0, "tol": 1e-6, "cov_11": 0.0025},
        "test_ndt_cell_no_y_variation": {
            "points": [[3.5, 3.5], [3.45, 3.5], [3.45, 3.5], [3.55, 3.5], [3.55, 3.5]],
            "information_11": 400000.0, "tol": 1e-6, "cov_00": 0.0025},
        "test_ndt": {
            "ndt": [1.0, 10.0, 10.0, -5.0, -5.0],
            "scan_pose": [0.0, 0.0, 0.0],
            "scan_points": [[3.5, 3.5], [3.45, 3.4], [3.55, 3.6], [3.45, 3.6], [3.45, 3.6]],
            "query": [[3.5, 3.5]], "likelihood": 0.7659, "tol": 0.001},
    }


def cfg1_match():
    scans = synth.map_scans(1)
    params = synth.matcher_params(1)
    guess, pts, true_pose = synth.query_scan(1)
    m = O.ScanMatcherNDT()
    m.initialize(**params)
    m.addScans(scans)
    r = m.matchScan(guess, pts, want_scores=True)
    ndt = m.ndt
    out = dict(
        cells6=ndt.cells6(), size_x=ndt.size_x, size_y=ndt.size_y, cell_size=ndt.cell_size,
        origin=np.array(ndt.origin), scan_pose=guess, points=pts, true_pose=true_pose,
        params_json=json.dumps(params), scores=r["scores"], best_index=r["best_index"],
        score=r["score"], pose=r["pose"], covariance=r["covariance"],
        n_candidates=r["n_candidates"],
        score_scan=m.scoreScan(guess, pts),
    )
    # plugin defaults (100 of the 720 beams, 21 x 21 x 80 lattice) on the same map
    pd = dict(params)
    pd.update(search_linear_size=0.05, search_linear_resolution=0.005,
              search_angular_size=0.1, search_angular_resolution=0.0025, laser_max_beams=100)
    md = O.ScanMatcherNDT()
    md.initialize(**pd)
    md.addScans(scans)
    true_guess = np.array([0.11, -0.05, 0.02])
    rd = md.matchScan(true_guess, pts, want_scores=True)
    out.update(default_params_json=json.dumps(pd), default_scan_pose=true_guess,
               default_scores=rd["scores"], default_best_index=rd["best_index"],
               default_score=rd["score"], default_pose=rd["pose"],
               default_covariance=rd["covariance"])
    return out


def cfg3_poses(n=256):
    scans = synth.map_scans(3)
    params = synth.matcher_params(3)
    _, pts, _ = synth.query_scan(3)
    m = O.ScanMatcherNDT()
    m.initialize(**params)
    m.addScans(scans)
    parts = synth.particles(3, n)
    # half of the batch near the true pose so that the weights are not all ~0
    parts[: n // 2, 0] = 1.0 + (parts[: n // 2, 0] / 23.0) * 0.3
    parts[: n // 2, 1] = 0.5 + (parts[: n // 2, 1] / 23.0) * 0.3
    parts[: n // 2, 2] = 0.3 + (parts[: n // 2, 2] / np.pi) * 0.1
    w_raw = O.pf_measure(m, parts, pts)
    w, mean, cov = O.pf_update_statistics(parts, w_raw)
    ndt = m.ndt
    cells = ndt.cells6()
    occupied = np.nonzero(cells[:, 5] > 0)[0]
    return dict(
        occupied_index=occupied.astype(np.int64), occupied_cells6=cells[occupied],
        size_x=ndt.size_x, size_y=ndt.size_y, cell_size=ndt.cell_size,
        origin=np.array(ndt.origin), points=pts, particles=parts, params_json=json.dumps(params),
        weights_raw=w_raw, weights=w, mean=mean, cov=cov)


def main():
    with open(os.path.join(HERE, "reference_ndt_model_tests.json"), "w") as f:
        json.dump(reference_vectors(), f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "cfg1_match.npz"), **cfg1_match())
    np.savez_compressed(os.path.join(HERE, "cfg3_poses256.npz"), **cfg3_poses())
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
