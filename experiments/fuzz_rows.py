"""One-off: many random cases for the rows either side of the hot path (occupancy
grid: bit-exact; scan conversion: kept set exact, coordinates 1e-12)."""
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT, synth  # noqa: E402
from ndt_2d_amd.occupancy_grid import OccupancyGrid  # noqa: E402

dev = ScanMatcherNDT(0)
dev.initialize("fuzz", **synth.matcher_params(1))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for seed in range(n):
    rng = np.random.default_rng(50000 + seed)
    scans = []
    for _ in range(int(rng.integers(1, 8))):
        pose = (rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(-np.pi, np.pi))
        scans.append((pose, rng.uniform(-8, 8, size=(int(rng.integers(0, 500)), 2))))
    res = float(rng.choice([0.03, 0.05, 0.1, 0.25, 0.3]))
    th = float(rng.choice([0.1, 0.25, 0.5, 0.9]))
    ref, gpu = O.OccupancyGrid(res, th), OccupancyGrid(res, th, dev)
    for upto in sorted(set(int(v) for v in rng.integers(1, len(scans) + 1, 2))):
        a, b = gpu.getMsg(scans[:upto]), ref.getMsg(scans[:upto])
        ok = all(a[k] == b[k] for k in ("width", "height", "origin_x", "origin_y")) and \
            np.array_equal(a["data"], b["data"])
        if not ok:
            bad += 1
            print("occupancy mismatch seed", seed, upto)
    m = int(rng.integers(1, 3000))
    ranges = rng.uniform(0.1, 15.0, m).astype(np.float32)
    ranges[rng.random(m) < 0.05] = np.nan
    conv = dict(angle_min=float(rng.uniform(-3.2, 0)), angle_increment=float(rng.uniform(0.001, 0.02)),
                range_max=float(rng.uniform(2, 14)), inverted=bool(rng.integers(0, 2)),
                laser=tuple(rng.uniform(-0.5, 0.5, 3)), motion=tuple(rng.uniform(-0.2, 0.2, 3)))
    a, b = dev.convertScan(ranges, **conv), O.convert_scan(ranges, **conv)
    if a.shape != b.shape or (len(a) and np.max(np.abs(a - b)) > 1e-12):
        bad += 1
        print("conversion mismatch seed", seed)
print("%d cases, %d failures" % (n, bad))
