#!/usr/bin/env python3
"""Known-answer vectors for two of the rows next to the hot path, derived WITHOUT the oracle:

  N2  LaserScan -> Scan conversion with de-skew (reference src/ndt_mapper.cpp:385-453)
  N3  MotionModel::sample (reference src/motion_model.cpp:45-83)

Both are stated from the mathematics of the reference's lines in 60-digit arithmetic
(mpmath); where the reference computes in float -- the beam angle `angle_min + i *
angle_increment` of a sensor_msgs/LaserScan, and std::normal_distribution<float>'s
`z * stddev + mean` -- the inputs are either binary-exact (so the float expression IS the
real value) or the float roundings are modelled explicitly (round-to-nearest at 24 bits).
No code is shared with oracle/ndt2d_oracle.c or the kernels.  The tests compare at 1e-12.

    python tests/golden/make_known_answers_next.py     # rewrites known_answers_next.json
"""
import json
import os

from mpmath import mp, mpf, atan2, cos, sin, sqrt, floor, pi, workprec

mp.dps = 60
HERE = os.path.dirname(os.path.abspath(__file__))


def f17(v):
    return float(mp.nstr(v, 25))


def fl32(v):
    """Round to the nearest float (24-bit significand; no value here is near the range limits)."""
    with workprec(24):
        return +mpf(v)


# ---------------------------------------------------------------------------------- N2
RANGES = ["1.5", "2.25", "nan", "3.0", "12.0", "0.75", "inf", "4.5"]   # float-exact
ANGLE_MIN, ANGLE_INC = mpf("-0.75"), mpf("0.25")                       # float-exact; i * inc too
RANGE_MAX = mpf("10")
LASER = (mpf("0.125"), mpf("-0.0625"), mpf("0.2"))        # laser_transform_ (x, y, theta)
MOTION = (mpf("0.08"), mpf("-0.04"), mpf("0.06"))         # `translation` (:386-389)


def dbl(v):
    """The double nearest to a decimal literal (what the C++ holds)."""
    with workprec(53):
        return +mpf(v)


def convert(inverted):
    n = len(RANGES)
    laser = tuple(dbl(v) for v in LASER)
    motion = tuple(dbl(v) for v in MOTION)
    per = tuple(v / n for v in motion)                     # trans_per_meas (:392-395)
    c_lt, s_lt = cos(laser[2]), sin(laser[2])
    order = range(n - 1, 0, -1) if inverted else range(n)  # (:410 `i > 0`: beam 0 is never used) / (:433)
    out = []
    for i in order:
        r = RANGES[i]
        if r == "nan" or (r != "inf" and mpf(r) > RANGE_MAX) or r == "inf":
            continue                                       # (:413,436)
        r = mpf(r)
        angle = ANGLE_MIN + i * ANGLE_INC
        if inverted:
            angle = -angle                                 # (:415)
        lx, ly = cos(angle) * r, sin(angle) * r            # (:416-417,439-440)
        px = c_lt * lx - s_lt * ly + laser[0]              # (:419-420,442-443)
        py = s_lt * lx + c_lt * ly + laser[1]
        if inverted:
            th, tx, ty = motion[2] - per[2] * i, motion[0] - per[0] * i, motion[1] - per[1] * i   # (:422-426)
        else:
            th, tx, ty = per[2] * i, per[0] * i, per[1] * i                                       # (:445-448)
        out.append((cos(th) * px - sin(th) * py + tx, sin(th) * px + cos(th) * py + ty))
    return out


# ---------------------------------------------------------------------------------- N3
ALPHAS = [mpf("0.2"), mpf("0.1"), mpf("0.15"), mpf("0.05"), mpf("0.0")]
MOTIONS = [("0.3", "0.1", "0.2"),          # forward, rot1 from atan2
           ("-0.25", "0.05", "-0.1"),      # reverse motion: rot1 near pi (:55-58)
           ("0.004", "0.003", "0.3")]      # trans <= 0.01: rot1 = 0 (:50)
POSES = [("0.5", "-0.25", "0.3"), ("-1.0", "2.0", "3.0"), ("0.0", "0.0", "-3.1"), ("4.0", "1.0", "-1.5")]
Z = [("0.5", "-1.25", "0.75"), ("-2.0", "1.5", "3.0"), ("0.25", "0.0", "-1.0"), ("1.0", "2.0", "-0.5")]


def wrap(a, magnitude_only=False):
    """angles::normalize_angle: into [-pi, pi].  At +-pi the sign depends on rounding: only
    callers that take the magnitude may land there."""
    a = a - 2 * pi * floor((a + pi) / (2 * pi))
    assert magnitude_only or abs(abs(a) - pi) > mpf("1e-6")
    return a


def motion_case(dx, dy, dth):
    dx, dy, dth = dbl(dx), dbl(dy), dbl(dth)
    a1, a2, a3, a4 = (dbl(v) for v in ALPHAS[:4])
    trans = sqrt(dx * dx + dy * dy)
    assert abs(trans - mpf("0.01")) > mpf("1e-6")
    rot1 = atan2(dy, dx) if trans > mpf("0.01") else mpf(0)
    rot2 = wrap(dth - rot1)                                            # angle_diff(rot1, dth)
    rot1_ = min(abs(wrap(-rot1, True)), abs(wrap(pi - rot1, True)))
    rot2_ = min(abs(wrap(-rot2, True)), abs(wrap(pi - rot2, True)))
    s_rot1 = sqrt(a1 * rot1_ ** 2 + a2 * trans ** 2)
    s_trans = sqrt(a3 * trans ** 2 + a4 * rot1_ ** 2 + a4 * rot2_ ** 2)
    s_rot2 = sqrt(a1 * rot2_ ** 2 + a2 * trans ** 2)
    # std::normal_distribution<float>(mean, stddev): both held as float, z * stddev + mean in float
    m1, s1, mt, st, m2, s2 = (fl32(v) for v in (rot1, s_rot1, trans, s_trans, rot2, s_rot2))
    poses = []
    for pose, z in zip(POSES, Z):
        x, y, th = (dbl(v) for v in pose)
        z1, z2, z3 = (mpf(v) for v in z)                   # float-exact
        r1 = fl32(fl32(z1 * s1) + m1)
        t = fl32(fl32(z2 * st) + mt)
        r2 = fl32(fl32(z3 * s2) + m2)
        poses.append((x + t * cos(th + r1), y + t * sin(th + r1), wrap(th + r1 + r2)))
    return {"motion": [float(dx), float(dy), float(dth)],
            "params": [f17(v) for v in (rot1, trans, rot2, s_rot1, s_trans, s_rot2)],
            "poses_after": [[f17(v) for v in p] for p in poses]}


def main():
    out = {
        "derivation": "mpmath, 60 digits, tests/golden/make_known_answers_next.py; no oracle code",
        "scan_conversion": {
            "ranges": [float(r) for r in RANGES], "angle_min": float(ANGLE_MIN),
            "angle_increment": float(ANGLE_INC), "range_max": float(RANGE_MAX),
            "laser": [float(v) for v in LASER], "motion": [float(v) for v in MOTION],
            "points": [[f17(x), f17(y)] for x, y in convert(False)],
            "points_inverted": [[f17(x), f17(y)] for x, y in convert(True)],
        },
        "motion_model": {
            "alphas": [float(v) for v in ALPHAS],
            "poses": [[float(v) for v in p] for p in POSES],
            "z": [[float(v) for v in z] for z in Z],
            "cases": [motion_case(*m) for m in MOTIONS],
        },
    }
    with open(os.path.join(HERE, "known_answers_next.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(len(out["scan_conversion"]["points"]), "+", len(out["scan_conversion"]["points_inverted"]),
          "points;", len(MOTIONS), "motions")


if __name__ == "__main__":
    main()
