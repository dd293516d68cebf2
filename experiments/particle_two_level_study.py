"""Study for VERDICT r02 #3 (particle kernel): a two-level screen whose FIRST level works
per RUN of consecutive beams (adaptive: a run ends when its bounding circle would exceed
r_max or it holds g_max beams) against a distance-to-occupied map, with the surviving
(particle, run) pairs COMPACTED across the wave before the per-beam FP32 screen runs on
them (a wave-level cull never fires: the 64 particles of a wave are unrelated, one of
them survives nearly every run).  Prints, per (r_max, g_max): the runs, the share of
(particle, run) pairs and of (particle, beam) pairs that survive level 1, and a VALU
model per wave of 64 particles next to today's 18 x 720 + queue rounds.

    python experiments/particle_two_level_study.py [cfg]      (CPU only)
"""
import math
import sys

sys.path.insert(0, '/root/repo')
import numpy as np

from ndt_2d_amd import host_build_grid, synth

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
scans = synth.map_scans(cfg)
p = synth.matcher_params(cfg)
cells, sx, sy, ox, oy = host_build_grid(0.25, p["range_max"], scans)
occ = (cells[:, 5] >= 5).reshape(sy, sx)
_, pts, _ = synth.query_scan(cfg)
parts = synth.particles(cfg, 20000)
print("cfg", cfg, "grid", sx, sy, "occupied fraction %.4f" % occ.mean())
R = 12
D = np.full(occ.shape, R + 1, dtype=np.int32)
cur = occ.copy()
D[occ] = 0
for d in range(1, R + 1):
    nxt = cur.copy()
    nxt[1:, :] |= cur[:-1, :]; nxt[:-1, :] |= cur[1:, :]
    cur = nxt.copy()
    nxt[:, 1:] |= cur[:, :-1]; nxt[:, :-1] |= cur[:, 1:]
    D[nxt & (D > R)] = d
    cur = nxt
c, s = np.cos(parts[:, 2]), np.sin(parts[:, 2])
n = len(pts)
chunks = 8
clen = (n + chunks - 1) // chunks
# today's kernel, VALU per wave of 64 particles: 18 per beam in the screen,
# ~3 per beam in the queue rounds, ~6 in phase B, ~3 block overhead
TODAY = 30.0 * n
for r_max in (0.25, 0.5, 0.75, 1.0):
    for g_max in (4, 8, 16):
        runs = []
        for ch in range(chunks):
            k0 = ch * clen; k1 = min(n, k0 + clen)
            b = k0
            while b < k1:
                e = b + 1
                lo = pts[b].copy(); hi = pts[b].copy()
                while e < k1 and e - b < g_max:
                    lo2 = np.minimum(lo, pts[e]); hi2 = np.maximum(hi, pts[e])
                    if 0.5 * math.hypot(*(hi2 - lo2)) > r_max:
                        break
                    lo, hi = lo2, hi2
                    e += 1
                ctr = (lo + hi) / 2
                q = pts[b:e]
                runs.append((b, e, ctr, float(np.max(np.hypot(q[:, 0] - ctr[0], q[:, 1] - ctr[1])))))
                b = e
        pairs = 0; live_pairs = 0; live_beams = 0
        for (b, e, ctr, r) in runs:
            X = parts[:, 0] + c * ctr[0] - s * ctr[1]; Y = parts[:, 1] + s * ctr[0] + c * ctr[1]
            gx = np.floor((X - ox) / 0.25).astype(int); gy = np.floor((Y - oy) / 0.25).astype(int)
            need = math.ceil(r / 0.25) + 2          # radius + FP32 slack + boundary neighbour
            d = D[np.clip(gy, 0, sy - 1), np.clip(gx, 0, sx - 1)]   # clamping only brings a point closer
            m = d <= need
            pairs += len(parts); live_pairs += int(m.sum()); live_beams += int(m.sum()) * (e - b)
        n_runs = len(runs)
        f_pair = live_pairs / pairs
        f_beam = live_beams / (len(parts) * n)
        # model, per wave: level 1 = n_runs x (13 test + 13 compacting push); level 2 = per
        # 64 surviving pairs: 12 fetch + g_eff x 20 (screen + bit) + 12 scan + ~30 write-out;
        # g_eff = the longest run of a batch ~ g_max; phase B (6 per beam) and block overhead stay
        lvl1 = n_runs * 26.0
        batches = 64.0 * n_runs * f_pair / 64.0
        lvl2 = batches * (12 + g_max * 20 + 42)
        lvl2_ideal = 64.0 * n * f_beam / 64.0 * 20 + batches * 54
        rest = 9.0 * n - 3.0 * n     # phase B + overhead, queue rounds replaced by the write-out above
        print("r_max %.2f g_max %2d: %3d runs (mean %.1f beams) | pairs alive %.3f beams alive %.3f | "
              "VALU/wave: lvl1 %5.0f lvl2 %5.0f (ideal %5.0f) total %6.0f vs today %6.0f -> %+.0f %% (ideal %+.0f %%)"
              % (r_max, g_max, n_runs, n / n_runs, f_pair, f_beam, lvl1, lvl2, lvl2_ideal,
                 lvl1 + lvl2 + rest, TODAY, 100 * ((lvl1 + lvl2 + rest) / TODAY - 1),
                 100 * ((lvl1 + lvl2_ideal + rest) / TODAY - 1)))
