"""CPU tests of the oracle's OccupancyGrid restatement (reference
src/occupancy_grid.cpp:47-185) against a plain-Python
statement of the same loops on small cases, and against properties of the map.
The reference holds no test or fixture for this class (parity unpinned)."""
import math

import numpy as np
import pytest

import oracle_lib as O


def _python_occupancy(resolution, occ_thresh, scans):
    """getMsg for a fresh generator, statement by statement."""
    min_x = max_x = min_y = max_y = 0.0
    for pose, pts in scans:
        c, s = math.cos(pose[2]), math.sin(pose[2])
        for qx, qy in pts:
            px = pose[0] + (qx * c - qy * s)
            py = pose[1] + (qx * s + qy * c)
            min_x, max_x = min(px, min_x), max(px, max_x)
            min_y, max_y = min(py, min_y), max(py, max_y)
    min_x = math.floor(min_x / resolution) * resolution
    max_x = math.ceil(max_x / resolution) * resolution
    min_y = math.floor(min_y / resolution) * resolution
    max_y = math.ceil(max_y / resolution) * resolution
    pad = 5 * resolution
    width = int((max_x - min_x + 2 * pad) / resolution)
    height = int((max_y - min_y + 2 * pad) / resolution)
    ox, oy = min_x - pad, min_y - pad
    hit = np.zeros(width * height, dtype=np.int64)
    empty = np.zeros(width * height, dtype=np.int64)
    for pose, pts in scans:
        c, s = math.cos(pose[2]), math.sin(pose[2])
        start_x = int((pose[0] - ox) / resolution)
        start_y = int((pose[1] - oy) / resolution)
        for qx, qy in pts:
            end_x = int((qx * c - qy * s + pose[0] - ox) / resolution)
            end_y = int((qx * s + qy * c + pose[1] - oy) / resolution)
            dx, sx = abs(end_x - start_x), (1 if start_x < end_x else -1)
            dy, sy = -abs(end_y - start_y), (1 if start_y < end_y else -1)
            error = dx + dy
            x, y = start_x, start_y
            while True:
                inside = 0 <= x < width and 0 <= y < height
                index = x + y * width
                if x == end_x and y == end_y:
                    if inside:
                        hit[index] += 1
                    break
                if inside:
                    empty[index] += 1
                if 2 * error >= dy:
                    if x == end_x:
                        if inside:
                            hit[index] += 1
                        break
                    error += dy
                    x += sx
                if 2 * error <= dx:
                    if y == end_y:
                        if inside:
                            hit[index] += 1
                        break
                    error += dx
                    y += sy
    touches = (hit + empty).astype(np.float64)
    data = np.full(width * height, -1, dtype=np.int8)
    known = touches > 0.5
    ratio = np.divide(hit.astype(np.float64), touches, out=np.zeros_like(touches), where=known)
    data[known] = np.where(ratio[known] > occ_thresh, 100, 0)
    return dict(width=width, height=height, origin_x=ox, origin_y=oy, data=data.reshape(height, width))


@pytest.mark.parametrize("seed", range(6))
def test_oracle_matches_python_statement(seed):
    rng = np.random.default_rng(seed)
    scans = []
    for _ in range(int(rng.integers(1, 4))):
        pose = (float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), float(rng.uniform(-3.1, 3.1)))
        scans.append((pose, rng.uniform(-4, 4, size=(int(rng.integers(1, 60)), 2))))
    res = float(rng.choice([0.05, 0.1, 0.25]))
    thresh = float(rng.choice([0.1, 0.25, 0.5]))
    got = O.OccupancyGrid(res, thresh).getMsg(scans)
    want = _python_occupancy(res, thresh, scans)
    for k in ("width", "height", "origin_x", "origin_y"):
        assert got[k] == want[k]
    assert np.array_equal(got["data"], want["data"])


def test_oracle_map_properties():
    """A square room seen from its centre: walls occupied, interior free, outside unknown."""
    n = 720
    ang = -np.pi + np.arange(n) * 2 * np.pi / n
    r = 2.02 / np.maximum(np.abs(np.cos(ang)), np.abs(np.sin(ang)))     # 4.04 x 4.04 m room
    pts = np.stack([r * np.cos(ang), r * np.sin(ang)], axis=1)
    m = O.OccupancyGrid(0.05, 0.25).getMsg([((0.0, 0.0, 0.0), pts)])

    def cell(x, y):
        return m["data"][int((y - m["origin_y"]) / 0.05), int((x - m["origin_x"]) / 0.05)]

    assert cell(0.0, 0.0) == 0 and cell(1.0, -1.2) == 0          # interior: free
    assert cell(2.02, 0.31) == 100 and cell(-0.72, -2.02) == 100  # walls: occupied
    assert cell(2.22, 2.22) == -1                                # beyond the walls: unknown
    # Most beam ends are hits -- not all: the reference's "simplified Bresenham" stops
    # as soon as EITHER coordinate has reached its end value when that axis is about
    # to step (:111-117,121-127), so a beam's hit can land in the end column / row
    # short of its end cell.  The restatement keeps that.
    assert 0.7 < np.mean([cell(x, y) == 100 for x, y in pts]) < 1.0
    assert m["width"] == m["height"] == 92                       # 4.1 m + 2 x 0.25 m pad at 5 cm
    # the bounds always contain the origin (min / max start at 0, :37-40) and persist
    g = O.OccupancyGrid(0.1, 0.25)
    g.getMsg([((5.0, 5.0, 0.0), np.array([[1.0, 1.0]]))])
    assert g.bounds[0] == 0.0 and g.bounds[1] >= 6.0 and g.num_scans == 1
