#!/usr/bin/env python3
"""Known-answer vectors for the OUTER loops of the hot path, derived WITHOUT the oracle.

The reference's own tests pin Cell / NDT::likelihood only (test/ndt_model_tests.cpp);
nothing there pins matchScan's candidate order, argmin, accumulators and covariance
formula, scorePoints' transform, or addScans' extent.  This script states those from
the mathematics of the reference's lines in 60-digit arithmetic (mpmath) -- sample
mean, unbiased sample covariance, 2x2 inverse, exp(-q^T Sigma^-1 q / 2), rigid
transforms, sums -- with no code shared with oracle/ndt2d_oracle.c or the kernels, on
a scenario small enough to follow by hand:

  map   two scans (poses (0,0,0) and (0.25,0.5,0), range_max 2, resolution 1.0):
        extent [-2, 2.25] x [-2, 2.5] -> origin (-2,-2), 5 x 5 cells
        (src/scan_matcher_ndt.cpp:52-66, src/ndt_model.cpp:118-126); three cells get
        points: A (6 points), B (5 points), C (4 points: n < 5 scores 0, :107)
  scan  six beams; scan pose (0.1, -0.05, 0.3): one rotated pose
  search  +-0.375 step 0.25 on every axis: offsets {-0.375, -0.125, 0.125} (binary
        exact, so the reference's accumulated loop values are these), 3 x 3 x 3 = 27
        candidates in the reference's loop order theta, x, y (:103,117,119)

Expected (JSON, 17 significant digits): the packed cells, all 27 raw scores
(-sum of likelihoods, :127), the winner's flat index / pose, matchScan's return value
best/N (:148), the accumulators k, u, s and covariance = k/s + u u^T/s^2 (:137-146),
scorePoints at three poses (:156-178), and ParticleFilter::measure with its
updateStatistics (src/particle_filter.cpp:78-89,163-218) for eight particles whose headings
straddle +-pi: raw and normalised weights, weighted mean, circular mean, x/y covariance and
the accumulated theta variance.  Differences between this and an IEEE
double evaluation in the reference's order are rounding only: the tests compare at
1e-12.  Every point is checked to lie > 1e-6 from a cell boundary, so the cell a
point falls in does not depend on rounding.

    python tests/golden/make_known_answers.py      # rewrites known_answers.json
"""
import json
import os

from mpmath import mp, mpf, atan2, cos, sin, exp, floor, pi

mp.dps = 60
HERE = os.path.dirname(os.path.abspath(__file__))

RESOLUTION = mpf(1)
RANGE_MAX = mpf(2)
MAP_SCANS = [
    # (pose x, y, theta), robot-frame points
    ((mpf(0), mpf(0), mpf(0)), [
        # cell A: world [0,1) x [0,1)
        ("0.25", "0.25"), ("0.75", "0.375"), ("0.25", "0.75"), ("0.75", "0.875"), ("0.5", "0.5"),
        # cell B: world [1,2) x [0,1)
        ("1.25", "0.5"), ("1.5", "0.25"), ("1.75", "0.625"), ("1.5", "0.75"),
        # cell C: world [-1,0) x [0,1): four points only
        ("-0.5", "0.25"), ("-0.25", "0.5"), ("-0.75", "0.5"), ("-0.5", "0.75"),
    ]),
    ((mpf("0.25"), mpf("0.5"), mpf(0)), [
        ("0.375", "0.125"),    # world (0.625, 0.625): cell A's sixth point
        ("1.125", "-0.125"),   # world (1.375, 0.375): cell B's fifth point
    ]),
]
SCAN_POSE = (mpf("0.1"), mpf("-0.05"), mpf("0.3"))
# robot-frame beams: under the search they visit A, B, C, an empty cell and the outside
BEAMS = [("0.45", "0.35"), ("1.4", "0.05"), ("0.55", "0.6"), ("-0.5", "0.7"), ("0.3", "-0.9"),
         ("3.6", "0.2")]
OFFSETS = [mpf("-0.375"), mpf("-0.125"), mpf("0.125")]   # `for (v = -0.375; v < 0.375; v += 0.25)`
SCORE_POSES = [("0", "0", "0"), ("0.1", "-0.05", "0.3"), ("0.4", "0.2", "-1.1")]
# ParticleFilter::measure + updateStatistics (src/particle_filter.cpp:78-89,163-218): a
# particle set whose headings straddle +-pi (circular mean, shortest angular distance)
PARTICLES = [("0.1", "-0.05", "0.3"), ("0.2", "0.1", "0.1"), ("-0.1", "0.05", "0.5"),
             ("0.35", "-0.2", "-0.2"), ("1.0", "0.9", "3.05"), ("1.1", "0.8", "-3.1"),
             ("0.6", "0.3", "1.2"), ("0.15", "-0.1", "0.25")]
COV_THETA_BEFORE = mpf("0.125")   # cov_(2,2) is accumulated into, never zeroed (:216)


def transform(pose, p):
    """T(pose) * (x, y, 1): rotation by theta then translation (conversions.hpp:64-68)."""
    c, s = cos(pose[2]), sin(pose[2])
    return (pose[0] + (c * p[0] - s * p[1]), pose[1] + (s * p[0] + c * p[1]))


def build_map():
    xs = [pose[0] for pose, _ in MAP_SCANS]
    ys = [pose[1] for pose, _ in MAP_SCANS]
    min_x, max_x = min(xs) - RANGE_MAX, max(xs) + RANGE_MAX
    min_y, max_y = min(ys) - RANGE_MAX, max(ys) + RANGE_MAX
    size_x = int(floor((max_x - min_x) / RESOLUTION + 1))
    size_y = int(floor((max_y - min_y) / RESOLUTION + 1))
    cells = {}
    for pose, pts in MAP_SCANS:
        for p in pts:
            w = transform(pose, (mpf(p[0]), mpf(p[1])))
            idx = cell_of(w, (min_x, min_y), size_x, size_y)
            assert idx is not None
            cells.setdefault(idx, []).append(w)
    packed = {}
    for idx, pts in cells.items():
        n = len(pts)
        mx = sum(p[0] for p in pts) / n
        my = sum(p[1] for p in pts) / n
        rec = {"n": n, "mean": (mx, my), "information": None}
        if n >= 3:
            # unbiased sample covariance and its inverse (src/ndt_model.cpp:65-103;
            # none of these cells is near the eigenvalue clamp of :88-96)
            cxx = sum((p[0] - mx) ** 2 for p in pts) / (n - 1)
            cxy = sum((p[0] - mx) * (p[1] - my) for p in pts) / (n - 1)
            cyy = sum((p[1] - my) ** 2 for p in pts) / (n - 1)
            tr, det = cxx + cyy, cxx * cyy - cxy * cxy
            small = tr / 2 - ((tr / 2) ** 2 - det) ** mpf("0.5")
            large = tr / 2 + ((tr / 2) ** 2 - det) ** mpf("0.5")
            assert small > mpf("0.01") * large, "scenario must stay off the clamp branch"
            rec["information"] = (cyy / det, -cxy / det, cxx / det)
        packed[idx] = rec
    return (min_x, min_y), size_x, size_y, packed


def cell_of(p, origin, size_x, size_y):
    """NDT::getIndex (src/ndt_model.cpp:203-218); asserts p is not at a rounding-sensitive spot."""
    fx = (p[0] - origin[0]) / RESOLUTION
    fy = (p[1] - origin[1]) / RESOLUTION
    for f in (fx, fy):
        assert abs(f - floor(f + mpf("0.5"))) > mpf("1e-6"), "point too close to a cell boundary"
    if fx < 0 or fy < 0:
        return None
    gx, gy = int(floor(fx)), int(floor(fy))
    if gx >= size_x or gy >= size_y:
        return None
    return gy * size_x + gx


def likelihood(p, origin, size_x, size_y, packed):
    """NDT::likelihood(point) (:162-170) with Cell::score (:105-116)."""
    idx = cell_of(p, origin, size_x, size_y)
    if idx is None or idx not in packed:
        return mpf(0)
    c = packed[idx]
    if c["n"] < 5:
        return mpf(0)
    q0, q1 = p[0] - c["mean"][0], p[1] - c["mean"][1]
    i00, i01, i11 = c["information"]
    return exp(-(q0 * q0 * i00 + 2 * q0 * q1 * i01 + q1 * q1 * i11) / 2)


def f17(v):
    return float(mp.nstr(v, 25))


def main():
    origin, size_x, size_y, packed = build_map()
    beams = [(mpf(x), mpf(y)) for x, y in BEAMS]
    n = len(beams)

    scores, k, u, s = [], [[mpf(0)] * 3 for _ in range(3)], [mpf(0)] * 3, mpf(0)
    best, best_idx, best_pose = mpf(0), None, None
    flat = 0
    for dth in OFFSETS:
        outer = [transform((SCAN_POSE[0], SCAN_POSE[1], SCAN_POSE[2] + dth), b) for b in beams]
        for dx in OFFSETS:
            for dy in OFFSETS:
                sc = -sum(likelihood((o[0] + dx, o[1] + dy), origin, size_x, size_y, packed)
                          for o in outer)
                scores.append(sc)
                if sc < best:
                    best, best_idx, best_pose = sc, flat, (dx, dy, dth)
                x = (dx, dy, dth)
                for r in range(3):
                    for c in range(3):
                        k[r][c] += x[r] * x[c] * sc
                    u[r] += x[r] * sc
                s += sc
                flat += 1
    # winners must be decided by more than rounding
    ordered = sorted(scores)
    assert ordered[1] - ordered[0] > mpf("1e-6")
    cov = [[k[r][c] / s + u[r] * u[c] / (s * s) for c in range(3)] for r in range(3)]

    score_points = []
    for pose in SCORE_POSES:
        ps = tuple(mpf(v) for v in pose)
        total = -sum(likelihood(transform(ps, b), origin, size_x, size_y, packed) for b in beams)
        score_points.append({"pose": [float(v) for v in pose], "score": f17(total / n)})

    # measure: weights_[i] = scorePoints(points, particle_i) (:81-87), then updateStatistics
    parts = [tuple(mpf(v) for v in q) for q in PARTICLES]
    raw = [-sum(likelihood(transform(q, b), origin, size_x, size_y, packed) for b in beams) / n
           for q in parts]
    assert sum(1 for w in raw if w != 0) >= 5
    sum_w = sum(raw)
    w = [v / sum_w for v in raw]                                   # (:166-174)
    mean_x = sum(wi * q[0] for wi, q in zip(w, parts))             # (:182-186)
    mean_y = sum(wi * q[1] for wi, q in zip(w, parts))
    mean_th = atan2(sum(wi * sin(q[2]) for wi, q in zip(w, parts)),
                    sum(wi * cos(q[2]) for wi, q in zip(w, parts)))  # (:187-188,200)
    cov_xx = sum(wi * q[0] * q[0] for wi, q in zip(w, parts)) - mean_x * mean_x   # (:190-210)
    cov_xy = sum(wi * q[0] * q[1] for wi, q in zip(w, parts)) - mean_x * mean_y
    cov_yy = sum(wi * q[1] * q[1] for wi, q in zip(w, parts)) - mean_y * mean_y

    def shortest(frm, to):
        """angles::shortest_angular_distance: to - from wrapped into [-pi, pi]."""
        d = to - frm
        d = d - 2 * pi * floor((d + pi) / (2 * pi))
        assert abs(abs(d) - pi) > mpf("1e-6")
        return d

    theta_inc = sum(wi * shortest(q[2], mean_th) ** 2 for wi, q in zip(w, parts))   # (:213-217)
    particles = {
        "poses": [[float(v) for v in q] for q in PARTICLES],
        "raw_weights": [f17(v) for v in raw], "sum_weight": f17(sum_w),
        "weights": [f17(v) for v in w],
        "mean": [f17(mean_x), f17(mean_y), f17(mean_th)],
        "cov_before": [[0.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, float(COV_THETA_BEFORE)]],
        "cov": [[f17(cov_xx), f17(cov_xy), 0.0], [f17(cov_xy), f17(cov_yy), 0.0],
                [0.0, 0.0, f17(COV_THETA_BEFORE + theta_inc)]],
    }

    out = {
        "derivation": "mpmath, 60 digits, tests/golden/make_known_answers.py; no oracle code",
        "params": {"ndt_resolution": 1.0, "range_max": 2.0, "laser_max_beams": 100,
                   "search_linear_size": 0.375, "search_linear_resolution": 0.25,
                   "search_angular_size": 0.375, "search_angular_resolution": 0.25},
        "map_scans": [{"pose": [float(v) for v in pose],
                       "points": [[float(x), float(y)] for x, y in pts]} for pose, pts in MAP_SCANS],
        "grid": {"origin": [f17(origin[0]), f17(origin[1])], "size_x": size_x, "size_y": size_y,
                 "cells": [{"index": idx, "n": c["n"], "mean": [f17(c["mean"][0]), f17(c["mean"][1])],
                            "information": [f17(v) for v in c["information"]]}
                           for idx, c in sorted(packed.items())]},
        "scan_pose": [float(v) for v in SCAN_POSE],
        "beams": [[float(x), float(y)] for x, y in BEAMS],
        "offsets": [float(v) for v in OFFSETS],
        "match": {"n_candidates": len(scores), "scores": [f17(v) for v in scores],
                  "best_index": best_idx, "pose": [float(v) for v in best_pose],
                  "score": f17(best / n),
                  "k": [[f17(v) for v in row] for row in k], "u": [f17(v) for v in u], "s": f17(s),
                  "covariance": [[f17(v) for v in row] for row in cov]},
        "score_points": score_points,
        "particles": particles,
    }
    with open(os.path.join(HERE, "known_answers.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("best", best_idx, [float(v) for v in best_pose], f17(best / n))
    print("scores", [round(float(v), 6) for v in scores])
    print("particle weights", [round(float(v), 6) for v in w], "mean", [round(float(v), 6) for v in (mean_x, mean_y, mean_th)])


if __name__ == "__main__":
    main()
