"""Round-5 soak of the worker threads behind the multi-device matcher (ndt2d_workers.h,
multi_match / multi_score_poses in ndt2d_host.cpp): eight contexts on the one GPU, host exchange,
driven for `seconds` with a different piece of work every time -- particle counts from 300 to
300,000, beam subsets (the devices' beam copies go stale and are refreshed), searches of 3 ... 41
theta steps, a new map now and then, pauses of 0 ... 2 ms between calls (the threads spin for
200 us, then park: both ways of being woken are taken) -- and every result compared with a
single-device matcher given the same call: raw scores and search results bit for bit, weights
and statistics to rounding (the moment sums meet in device order, not in block order).

    python experiments/soak_r05.py [seconds] [seed]
"""
import sys
import time

sys.path.insert(0, "/root/repo")
import numpy as np
from ndt_2d_amd import ScanMatcherNDT, pf_measure, synth

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(seed)

multi = ScanMatcherNDT(device_ids=[0] * 8)
multi.set_exchange("host")
multi.set_multi_thresholds(0, 0)
one = ScanMatcherNDT(0)
maps = {1: synth.map_scans(1), 3: synth.map_scans(3)}
counts = {"maps": 0, "match": 0, "measure": 0, "score_poses": 0, "dealt": 0}
current = None
t0 = time.time()
while time.time() - t0 < seconds:
    if current is None or rng.random() < 0.03:
        cfg = int(rng.choice([1, 3]))
        n_th = int(rng.integers(1, 21))
        p = synth.matcher_params(cfg, search_angular_size=0.005 * n_th + 0.0025, search_angular_resolution=0.005,
                                 search_linear_size=0.05, search_linear_resolution=0.01)
        for m in (multi, one):
            m.initialize("soak", **p)
            m.reset()
            m.addScans(maps[cfg])
        guess, pts_all, _ = synth.query_scan(cfg)
        world = 23.0 if cfg == 3 else 4.0
        current = cfg
        counts["maps"] += 1
    keep = rng.random(len(pts_all)) < rng.uniform(0.3, 1.0)
    pts = np.ascontiguousarray(pts_all[keep]) if keep.sum() >= 8 else pts_all
    what = rng.random()
    if what < 0.35:
        a, b = multi.matchScan(guess, pts), one.matchScan(guess, pts)
        assert a["best_index"] == b["best_index"] and a["score"] == b["score"], (counts, a, b)
        assert np.array_equal(a["pose"], b["pose"])
        assert np.allclose(a["covariance"], b["covariance"], rtol=1e-9, atol=1e-300)
        counts["match"] += 1
    else:
        n = int(np.exp(rng.uniform(np.log(300), np.log(300000))))
        parts = synth.particles(3, n)
        parts[:, :2] *= world / 23.0
        if what < 0.7:
            wa, ma, ca = pf_measure(multi, parts, pts)
            wb, mb, cb = pf_measure(one, parts, pts)
            assert np.allclose(wa, wb, rtol=1e-12, atol=1e-300), (counts, n)
            assert np.allclose(ma, mb, rtol=1e-10, atol=1e-13) and np.allclose(ca, cb, rtol=1e-8, atol=1e-13), (counts, n)
            counts["measure"] += 1
        else:
            sa, sb = multi.scorePoses(pts, parts), one.scorePoses(pts, parts)
            assert np.array_equal(sa, sb), (counts, n)
            counts["score_poses"] += 1
    counts["dealt"] += multi.matcher_variant().startswith("multi[8]/host/")
    if rng.random() < 0.5:
        time.sleep(rng.uniform(0.0, 0.002))
counts["seconds"] = round(time.time() - t0, 1)
counts["seed"] = seed
multi.close()
one.close()
print(counts)
