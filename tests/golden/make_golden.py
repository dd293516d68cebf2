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


def next_rows():
    """The steps either side of the hot path: LaserScan
    conversion, motion model / filter init, occupancy grid -- oracle outputs."""
    out = {}
    # N2: the cfg-1 query scan as a LaserScan (ranges float32), with drop-outs
    _, pts, _ = synth.query_scan(1)
    rng = np.random.default_rng(11)
    ranges = np.hypot(pts[:, 0], pts[:, 1]).astype(np.float32)
    ranges[rng.random(len(ranges)) < 0.04] = np.nan
    ranges[rng.random(len(ranges)) < 0.02] = 1e6
    conv = dict(angle_min=-np.pi, angle_increment=2.0 * np.pi / len(ranges), range_max=4.75,
                laser=(0.05, -0.02, 0.01), motion=(0.02, -0.01, 0.015))
    out.update(scan_ranges=ranges, scan_params_json=json.dumps(conv),
               scan_points=O.convert_scan(ranges, inverted=False, **conv),
               scan_points_inverted=O.convert_scan(ranges, inverted=True, **conv))
    # N3: MotionModel::sample and ParticleFilter::init on given standard normals
    poses = np.stack([rng.uniform(-3, 3, 256), rng.uniform(-3, 3, 256),
                      rng.uniform(-np.pi, np.pi, 256)], axis=1)
    z = rng.standard_normal((256, 3)).astype(np.float32)
    alphas = [0.2, 0.05, 0.15, 0.02, 0.0]
    moved, mparams = O.motion_sample(0.3, -0.1, 0.4, alphas, poses, z)
    out.update(pf_poses=poses, pf_noise=z, pf_alphas=np.array(alphas), pf_motion=np.array([0.3, -0.1, 0.4]),
               pf_moved=moved, pf_motion_params=mparams,
               pf_init_args=np.array([1.25, -3.5, 3.0, 0.3, 0.2, 0.5]),
               pf_init=O.pf_init(1.25, -3.5, 3.0, 0.3, 0.2, 0.5, z))
    # N4: the cfg-1 map rendered at 5 cm, hit ratio 0.25
    m = O.OccupancyGrid(0.05, 0.25).getMsg(synth.map_scans(1))
    out.update(occ_info=np.array([m["resolution"], m["width"], m["height"], m["origin_x"], m["origin_y"]]),
               occ_data=m["data"])
    return out


def big_winners():
    """Winners of the full cfg-2 and cfg-4 lattices (BASELINE.json configs[1] and [3]):
    the oracle over all 2,000,000 / 315,508,257 candidates (about 4 minutes on 8 cores
    for cfg-4).  The cfg-4 search spans +-pi over a 4-fold symmetric room; range noise
    makes one of the four basins the oracle's definite winner, and the GPU must pick it."""
    out = {"source": "oracle/ndt2d_oracle.c orc_matcher_match_scan_omp_ex over the whole lattice"}
    for cfg in (2, 4):
        params = synth.matcher_params(cfg)
        guess, pts, _ = synth.query_scan(cfg)
        m = O.ScanMatcherNDT()
        m.initialize(**params)
        m.addScans(synth.map_scans(cfg))
        r = m.matchScan(guess, pts, omp_threads=os.cpu_count())
        n_th = len(O.search_offsets(params["search_angular_size"], params["search_angular_resolution"]))
        n_lin = len(O.search_offsets(params["search_linear_size"], params["search_linear_resolution"]))
        out["cfg%d" % cfg] = {
            "n_theta": n_th, "n_linear": n_lin, "n_candidates": n_th * n_lin * n_lin,
            "best_index": int(r["best_index"]), "score": float(r["score"]),
            "score_hex": float(r["score"]).hex(),
            "pose": [float(v) for v in r["pose"]], "pose_hex": [float(v).hex() for v in r["pose"]],
            "covariance": [[float(v) for v in row] for row in r["covariance"]],
        }
    return out


FIXTURES = {"cfg1_match.npz": cfg1_match, "cfg3_poses256.npz": cfg3_poses, "next_rows.npz": next_rows}
JSON_FIXTURES = {"reference_ndt_model_tests.json": reference_vectors, "big_winners.json": big_winners}


def main():
    """python make_golden.py [fixture.npz ...]   (default: all)"""
    names = sys.argv[1:] or sorted(JSON_FIXTURES) + sorted(FIXTURES)
    for name in names:
        if name in JSON_FIXTURES:
            with open(os.path.join(HERE, name), "w") as f:
                json.dump(JSON_FIXTURES[name](), f, indent=1, sort_keys=True)
        else:
            np.savez_compressed(os.path.join(HERE, name), **FIXTURES[name]())
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
