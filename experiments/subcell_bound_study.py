"""How many slow wave-iterations of the lane-per-candidate search would a per-sub-cell
upper bound on the exponent remove?  (cfg-2, 8x8 patches, random sample.)

For every sub-cell of the map (SUB x SUB per NDT cell) the bound is the maximum of
-0.5 q^T I q over the sub-cell's box, for the cell the box lies in.  A wave-iteration
(one beam, 64 candidates) needs the exact path only if some lane's bound reaches the
lane's skip threshold (~ ln(sum) - 38; a fixed -45 is used here)."""
import math
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np  # noqa: E402

import oracle_lib as O  # noqa: E402
from ndt_2d_amd import host_build_grid, synth  # noqa: E402

CELL = 0.25
THRESH = -45.0


def box_bounds(cells, sx, sy, ox, oy, sub):
    """E_ub[sy*sub, sx*sub]: max exponent over each sub-cell box (-inf if no distribution)."""
    n = sub
    out = np.full((sy * n, sx * n), -np.inf)
    step = CELL / n
    for cy in range(sy):
        for cx in range(sx):
            rec = cells[cy * sx + cx]
            if rec[5] < 5:
                continue
            mx, my, a, b, d = rec[0], rec[1], rec[2], rec[3], rec[4]
            for j in range(n):
                for i in range(n):
                    x0 = ox + cx * CELL + i * step
                    y0 = oy + cy * CELL + j * step
                    x1, y1 = x0 + step, y0 + step
                    if x0 <= mx <= x1 and y0 <= my <= y1:
                        best = 0.0
                    else:
                        best = np.inf
                        # edges: x fixed, y free in [y0, y1]; y fixed, x free
                        for xf in (x0, x1):
                            q0 = xf - mx
                            # minimise a q0^2 + 2 b q0 q1 + d q1^2 over q1
                            q1 = np.clip(-b * q0 / d if d > 0 else 0.0, y0 - my, y1 - my)
                            best = min(best, a * q0 * q0 + 2 * b * q0 * q1 + d * q1 * q1)
                        for yf in (y0, y1):
                            q1 = yf - my
                            q0 = np.clip(-b * q1 / a if a > 0 else 0.0, x0 - mx, x1 - mx)
                            best = min(best, a * q0 * q0 + 2 * b * q0 * q1 + d * q1 * q1)
                    out[cy * n + j, cx * n + i] = -0.5 * best
    return out


def main():
    scans = synth.map_scans(2)
    p = synth.matcher_params(2)
    cells, sx, sy, ox, oy = host_build_grid(CELL, p["range_max"], scans)
    occ = cells[:, 5] >= 5
    guess, pts, _ = synth.query_scan(2)
    dth = O.search_offsets(0.5, 0.005)
    dlin = O.search_offsets(1.0, 0.02)
    bounds = {s: box_bounds(cells, sx, sy, ox, oy, s) for s in (4, 8)}
    rng = np.random.default_rng(1)
    tot = 0
    act = 0
    sig_exact = 0
    keep = {s: 0 for s in bounds}
    live_lanes, sig_lanes, per_group = [], [], []
    for _ in range(200):
        ith = rng.integers(len(dth))
        ix0 = rng.integers(len(dlin) - 8)
        iy0 = rng.integers(len(dlin) - 8)
        th = guess[2] + dth[ith]
        c, s = math.cos(th), math.sin(th)
        oxp = pts[:, 0] * c - pts[:, 1] * s + guess[0]
        oyp = pts[:, 0] * s + pts[:, 1] * c + guess[1]
        PX = (oxp[:, None, None] + dlin[ix0:ix0 + 8][None, :, None] + 0 * dlin[None, None, iy0:iy0 + 8]).reshape(len(pts), 64)
        PY = (oyp[:, None, None] + 0 * dlin[ix0:ix0 + 8][None, :, None] + dlin[None, None, iy0:iy0 + 8]).reshape(len(pts), 64)
        gx = np.floor((PX - ox) / CELL).astype(int)
        gy = np.floor((PY - oy) / CELL).astype(int)
        inside = (gx >= 0) & (gx < sx) & (gy >= 0) & (gy < sy)
        idx = np.where(inside, gy * sx + gx, 0)
        o = inside & occ[idx]
        rec = cells[idx]
        q0 = PX - rec[..., 0]
        q1 = PY - rec[..., 1]
        e = -0.5 * (q0 * (rec[..., 2] * q0 + rec[..., 3] * q1) + q1 * (rec[..., 3] * q0 + rec[..., 4] * q1))
        e = np.where(o, e, -np.inf)
        tot += len(pts)
        act += o.any(axis=1).sum()
        sig_exact += (e.max(axis=1) >= THRESH).sum()
        for sub, B in bounds.items():
            hx = np.floor((PX - ox) / (CELL / sub)).astype(int)
            hy = np.floor((PY - oy) / (CELL / sub)).astype(int)
            ins = (hx >= 0) & (hx < sx * sub) & (hy >= 0) & (hy < sy * sub)
            ub = np.where(ins, B[np.clip(hy, 0, sy * sub - 1), np.clip(hx, 0, sx * sub - 1)], -np.inf)
            assert np.all(ub >= e - 1e-9)
            kept = ub.max(axis=1) >= THRESH
            keep[sub] += kept.sum()
            if sub == 4:
                live_lanes.append((ub[kept] >= THRESH).mean(axis=1))
                sig_lanes.append((e[kept] >= THRESH).mean(axis=1))
                k8 = kept[:len(kept) // 8 * 8].reshape(-1, 8).sum(axis=1)
                per_group.append(k8)
    print("wave-iterations with an occupied lane      : %.4f" % (act / tot))
    print("... with a lane whose exponent >= %.0f (ideal): %.4f" % (THRESH, sig_exact / tot))
    for sub in bounds:
        print("... kept by the %dx%d sub-cell bound           : %.4f" % (sub, sub, keep[sub] / tot))
    ll = np.concatenate(live_lanes)
    sl = np.concatenate(sig_lanes)
    pg = np.concatenate(per_group)
    print("kept iterations: mean fraction of lanes above the bound %.3f, truly significant %.3f"
          % (ll.mean(), sl.mean()))
    print("kept beams per group of 8: histogram", np.bincount(pg, minlength=9) / len(pg))


if __name__ == "__main__":
    main()
