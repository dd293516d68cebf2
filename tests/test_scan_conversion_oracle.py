"""CPU tests of the oracle's LaserScan -> Scan restatement (reference
src/ndt_mapper.cpp:385-453) against an independent numpy
statement of the same formulas and against the loop's documented quirks.  The
reference holds no test or fixture for this loop (parity unpinned)."""
import numpy as np
import pytest

import oracle_lib as O


def _numpy_convert(ranges, angle_min, angle_inc, range_max, inverted, laser, motion):
    r32 = np.asarray(ranges, dtype=np.float32)
    n = len(r32)
    idx = np.arange(n - 1, 0, -1) if inverted else np.arange(n)
    a = np.float32(angle_min) + idx.astype(np.float32) * np.float32(angle_inc)   # float32 math
    a = (-a if inverted else a).astype(np.float64)
    r = r32[idx].astype(np.float64)
    keep = ~(np.isnan(r) | (r > range_max))
    r = np.where(keep, r, 0.0)
    lx, ly = np.cos(a) * r, np.sin(a) * r
    cl, sl = np.cos(laser[2]), np.sin(laser[2])
    px = cl * lx - sl * ly + laser[0]
    py = sl * lx + cl * ly + laser[1]
    per = np.array(motion, dtype=np.float64) / n
    i = idx.astype(np.float64)
    if inverted:
        tt, tx, ty = motion[2] - per[2] * i, motion[0] - per[0] * i, motion[1] - per[1] * i
    else:
        tt, tx, ty = per[2] * i, per[0] * i, per[1] * i
    x = np.cos(tt) * px - np.sin(tt) * py + tx
    y = np.sin(tt) * px + np.cos(tt) * py + ty
    return np.stack([x, y], axis=1)[keep]


def _case(seed, n):
    rng = np.random.default_rng(seed)
    ranges = rng.uniform(0.3, 12.0, size=n).astype(np.float32)
    ranges[rng.random(n) < 0.05] = np.nan
    ranges[rng.random(n) < 0.05] = np.inf
    ranges[rng.random(n) < 0.05] = 40.0
    return dict(ranges=ranges, angle_min=float(rng.uniform(-3.2, -1.0)),
                angle_increment=float(rng.uniform(0.002, 0.01)), range_max=10.0,
                laser=tuple(rng.uniform(-0.3, 0.3, size=3)),
                motion=tuple(rng.uniform(-0.1, 0.1, size=3)))


@pytest.mark.parametrize("inverted", [False, True])
@pytest.mark.parametrize("n", [1, 2, 63, 720, 1081])
def test_oracle_conversion_matches_numpy_statement(n, inverted):
    c = _case(n, n)
    got = O.convert_scan(inverted=inverted, **c)
    want = _numpy_convert(c["ranges"], c["angle_min"], c["angle_increment"], c["range_max"],
                          inverted, c["laser"], c["motion"])
    assert got.shape == want.shape
    assert np.allclose(got, want, rtol=0, atol=1e-13)


def test_oracle_conversion_filters_and_order():
    r = np.array([1.0, np.nan, 2.0, 30.0, 3.0, np.inf, 10.0], dtype=np.float32)
    pts = O.convert_scan(r, 0.0, 0.0, 10.0)          # all beams along +x
    assert np.array_equal(pts, [[1, 0], [2, 0], [3, 0], [10, 0]])   # == range_max is kept (:436 `>`)
    inv = O.convert_scan(r, 0.0, 0.0, 10.0, inverted=True)
    # descending order and index 0 never visited (:410 `i > 0`)
    assert np.array_equal(inv, [[10, 0], [3, 0], [2, 0]])
    assert len(O.convert_scan(r[:1], 0.0, 0.0, 10.0, inverted=True)) == 0
    assert len(O.convert_scan(np.zeros(0, np.float32), 0.0, 0.1, 10.0)) == 0


def test_oracle_conversion_deskew_is_a_rigid_motion_per_beam():
    """Beam i of a forward sweep is moved by i/n of the sweep's odometry motion."""
    n = 360
    r = np.full(n, 5.0, dtype=np.float32)
    still = O.convert_scan(r, -np.pi, 2 * np.pi / n, 10.0, laser=(0.2, 0.0, 0.1))
    moved = O.convert_scan(r, -np.pi, 2 * np.pi / n, 10.0, laser=(0.2, 0.0, 0.1),
                           motion=(0.3, -0.1, 0.2))
    for i in (0, 1, 100, 359):
        f = i / n
        c, s = np.cos(0.2 * f), np.sin(0.2 * f)
        want = [c * still[i, 0] - s * still[i, 1] + 0.3 * f,
                s * still[i, 0] + c * still[i, 1] - 0.1 * f]
        assert np.allclose(moved[i], want, rtol=0, atol=1e-12)
    assert np.array_equal(moved[0], still[0])
