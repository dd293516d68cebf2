"""One-off: the mapper's call sequences on ONE matcher that lives through many maps.

A matcher instance of the node sees reset / addScans / scoreScan / matchScan / scorePoints in
whatever order the mapper and the particle filter produce (reference src/ndt_mapper.cpp:299-312,
508-515, 634-643; src/particle_filter.cpp:81-87), on grids whose extent changes with every scan.
Per cycle here: a random map (tests/test_gpu_fuzz._random_case, occasionally a 30 m range_max so
that the grid is tens of thousands of cells), installed by addScans on the SAME matcher as the
cycle before, then a random subset of the calls in a random order -- so that an install's pending
map-bytes job is consumed by a scoreScan, by a small measure, by the search itself, or never --
each checked against the oracle (grid cells and skipping bitwise, scores to 1e-9).

    python experiments/fuzz_cycle.py FIRST_SEED CYCLES
"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib as O  # noqa: E402
import test_gpu_fuzz as F  # noqa: E402
from ndt_2d_amd import ScanMatcherNDT  # noqa: E402

first, cycles = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(77000 + first)
gpu = None
bad = []
counts = {}
for c in range(cycles):
    params, scans, scan_pose, query, poses = F._random_case(rng)
    if rng.random() < 0.15:
        params["range_max"] = 30.0
        params["ndt_resolution"] = float(rng.choice([0.25, 0.5]))
    if gpu is None or rng.random() < 0.05:
        gpu = ScanMatcherNDT(0)                   # (now and then a new instance)
    gpu.initialize("fuzz-cycle", **params)
    ref = O.ScanMatcherNDT()
    ref.initialize(**params)
    ref.addScans(scans)
    gpu.reset()
    gpu.addScans(scans)
    calls = [k for k in ("grid", "scoreScan", "scorePoints", "measure", "matchScan", "matchScan2", "pair", "pair")
             if rng.random() < 0.6]
    rng.shuffle(calls)
    try:
        for call in calls:
            counts[call] = counts.get(call, 0) + 1
            if call == "grid":
                assert np.array_equal(gpu.grid()[0], ref.ndt.cells6(), equal_nan=True), "grid"
            elif call == "scoreScan":
                g, e = gpu.scoreScan(scan_pose, query), ref.scoreScan(scan_pose, query)
                assert (np.isnan(g) and np.isnan(e)) or abs(g - e) < 1e-9, ("scoreScan", g, e)
            elif call == "scorePoints":
                g, e = gpu.scorePoints(query, scan_pose), ref.scorePoints(query, scan_pose)
                assert (np.isnan(g) and np.isnan(e)) or abs(g - e) < 1e-9, ("scorePoints", g, e)
            elif call == "pair":
                # the mapper's pair: scoreScan(scan), matchScan(scan, ...) -- after the first one a
                # matcher launches the search ahead; no per-candidate scores here, so it is collected
                g, e = gpu.scoreScan(scan_pose, query), ref.scoreScan(scan_pose, query)
                assert (np.isnan(g) and np.isnan(e)) or abs(g - e) < 1e-9, ("pair scoreScan", g, e)
                got = gpu.matchScan(scan_pose, query)
                exp = ref.matchScan(scan_pose, query, want_scores=True)
                assert got["n_candidates"] == exp["n_candidates"]
                assert (np.isnan(got["score"]) and np.isnan(exp["score"])) or abs(got["score"] - exp["score"]) < 1e-9, "pair score"
                finite = exp["scores"][~np.isnan(exp["scores"])]
                if finite.size > 1 and np.sort(finite)[1] - finite.min() > 1e-9:
                    assert got["best_index"] == exp["best_index"], "pair index"
                    assert np.array_equal(got["pose"], exp["pose"])
                if abs(np.nansum(exp["scores"])) > 1e-6:
                    assert np.allclose(got["covariance"], exp["covariance"], rtol=1e-7, atol=1e-12, equal_nan=True), "pair cov"
            elif call == "measure":
                w = gpu.scorePoses(query, poses)
                w_exp = O.pf_measure(ref, poses, query)
                assert np.allclose(w, w_exp, rtol=0, atol=1e-9, equal_nan=True), "measure"
            else:
                got = gpu.matchScan(scan_pose, query, want_scores=True)
                exp = ref.matchScan(scan_pose, query, want_scores=True)
                assert got["n_candidates"] == exp["n_candidates"]
                assert np.allclose(got["scores"], exp["scores"], rtol=0, atol=1e-9, equal_nan=True), \
                    ("scores", gpu.last_variant())
                counts[gpu.last_variant()] = counts.get(gpu.last_variant(), 0) + 1
                if call == "matchScan2":
                    # the unskipped control of whichever mapping served it
                    v = gpu.last_variant()
                    ctl = "small-noskip" if "small-lattice" in v else "lane-noskip" if "lane-per" in v else None
                    if ctl is not None:
                        gpu.set_variant(ctl)
                        try:
                            again = gpu.matchScan(scan_pose, query, want_scores=True)
                            assert np.array_equal(again["scores"], got["scores"], equal_nan=True), ("noskip", v)
                        finally:
                            gpu.set_variant("auto")
    except AssertionError as e:
        bad.append((c, calls, str(e)[:200]))
        gpu = None
print("cycles %d (seed %d): %d failures" % (cycles, first, len(bad)))
for b in bad[:10]:
    print(b)
print(sorted(counts.items(), key=lambda kv: -kv[1]))
if gpu is not None:
    print("search ahead (launched, collected) of the last matcher:", gpu.search_ahead_stats())
