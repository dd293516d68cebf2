"""Parity tests proper: the HIP path, called through the C-ABI, against the oracle
on the same seeded inputs and against the committed golden fixtures.

Tolerances: BASELINE.json asks for scores within 1e-5 and the same best pose.
The kernels keep the reference's operation order, so what is actually observed
is ~1e-13 (wave-tree vs sequential summation, exp() ulps); the tests assert
both the contractual 1e-5 and a tighter 1e-9 regression bound."""
import ctypes as C
import json
import math
import os
import threading

import numpy as np
import pytest

import oracle_lib as O
from ndt_2d_amd import ScanMatcherNDT, _capi, pf_measure, synth
from ndt_2d_amd import dist as shard

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL_CONTRACT = 1e-5   # BASELINE.json north_star
TOL_TIGHT = 1e-9      # regression bound on raw (un-normalised) scores


def _pair(cfg, **override):
    """(gpu matcher, oracle matcher, scans, guess, points) for a synthetic config."""
    scans = synth.map_scans(cfg)
    params = synth.matcher_params(cfg, **override)
    gpu = ScanMatcherNDT(0)
    gpu.initialize("test", **params)
    gpu.addScans(scans)
    ref = O.ScanMatcherNDT()
    ref.initialize(**params)
    ref.addScans(scans)
    guess, pts, _ = synth.query_scan(cfg)
    return gpu, ref, scans, guess, pts


def _check_match(got, exp, n_beams):
    assert got["n_candidates"] == exp["n_candidates"]
    assert got["best_index"] == exp["best_index"]
    assert np.array_equal(got["pose"], exp["pose"])
    assert abs(got["score"] - exp["score"]) < TOL_CONTRACT
    assert abs(got["score"] - exp["score"]) < TOL_TIGHT
    assert np.allclose(got["covariance"], exp["covariance"], rtol=1e-9, atol=0, equal_nan=True)
    if got.get("scores") is not None and exp.get("scores") is not None:
        err = np.max(np.abs(got["scores"] - exp["scores"]))
        assert err < TOL_CONTRACT and err < TOL_TIGHT
        assert np.max(np.abs(got["scores"] - exp["scores"]) / n_beams) < TOL_CONTRACT


@pytest.fixture(scope="module")
def cfg1():
    return _pair(1)


def test_extension_is_loaded_in_tree():
    # the native library this process uses is the in-tree build
    assert os.path.samefile(os.path.dirname(_capi.LIB_PATH),
                            os.path.join(os.path.dirname(GOLDEN), "..", "ndt_2d_amd"))
    maps = open("/proc/self/maps").read()
    _capi.lib()
    maps = open("/proc/self/maps").read()
    assert "libndt2d_hip.so" in maps


def test_cfg1_match_vs_oracle_and_golden(cfg1):
    gpu, ref, _, guess, pts = cfg1
    got = gpu.matchScan(guess, pts, want_scores=True)
    exp = ref.matchScan(guess, pts, want_scores=True)
    _check_match(got, exp, 720)
    g = np.load(os.path.join(GOLDEN, "cfg1_match.npz"))
    assert np.array_equal(gpu.grid()[0], g["cells6"])          # host NDT build, bit-exact
    assert np.max(np.abs(got["scores"] - g["scores"])) < TOL_TIGHT
    assert got["best_index"] == int(g["best_index"])
    assert np.array_equal(got["pose"], g["pose"])
    assert abs(got["score"] - float(g["score"])) < TOL_TIGHT
    assert np.allclose(got["covariance"], g["covariance"], rtol=1e-9, atol=0)
    assert "small-lattice" in gpu.last_variant()


def test_score_scan_and_score_points(cfg1):
    gpu, ref, _, guess, pts = cfg1
    assert abs(gpu.scoreScan(guess, pts) - ref.scoreScan(guess, pts)) < TOL_TIGHT
    for pose in [(0.13, -0.07, 0.031), (-1.2, 0.8, 2.5), (3.9, 3.9, -3.1), (50.0, 50.0, 0.0)]:
        assert abs(gpu.scorePoints(pts, pose) - ref.scorePoints(pts, pose)) < TOL_TIGHT
    # a matching pose scores lower (better) than a displaced one
    assert gpu.scorePoints(pts, (0.13, -0.07, 0.031)) < gpu.scorePoints(pts, (0.5, 0.5, 0.3))


def test_plugin_default_parameters_subsample(cfg1):
    # laser_max_beams = 100 of 720 beams, 21 x 21 x 80 lattice (reference defaults)
    _, _, scans, _, pts = cfg1
    g = np.load(os.path.join(GOLDEN, "cfg1_match.npz"))
    p = json.loads(str(g["default_params_json"]))
    gpu = ScanMatcherNDT(0)
    gpu.initialize("local_scan_matcher", **p)
    gpu.addScans(scans)
    got = gpu.matchScan(g["default_scan_pose"], pts, want_scores=True)
    assert got["n_candidates"] == 21 * 21 * 80
    assert np.max(np.abs(got["scores"] - g["default_scores"])) < TOL_TIGHT
    assert got["best_index"] == int(g["default_best_index"])
    assert np.array_equal(got["pose"], g["default_pose"])
    assert abs(got["score"] - float(g["default_score"])) < TOL_TIGHT
    assert np.allclose(got["covariance"], g["default_covariance"], rtol=1e-9, atol=0)


def test_device_layer_direct_and_theta_sharding(cfg1):
    """ndt2d_set_grid / set_beams / set_search / match on the raw device layer with
    the golden grid; the slab records combine to the unsharded result."""
    g = np.load(os.path.join(GOLDEN, "cfg1_match.npz"))
    p = json.loads(str(g["params_json"]))
    L = _capi.lib()
    h = C.c_void_p()
    assert L.ndt2d_create(C.byref(h), 0) == 0
    try:
        cells = np.ascontiguousarray(g["cells6"])
        assert L.ndt2d_set_grid(h, _capi.dptr(cells), int(g["size_x"]), int(g["size_y"]),
                                float(g["cell_size"]), float(g["origin"][0]),
                                float(g["origin"][1])) == 0
        pts = np.ascontiguousarray(g["points"])
        assert L.ndt2d_set_beams(h, _capi.dptr(pts), len(pts)) == 0
        dth = O.search_offsets(p["search_angular_size"], p["search_angular_resolution"])
        dlin = O.search_offsets(p["search_linear_size"], p["search_linear_resolution"])
        sp = g["scan_pose"]
        cos_t = np.array([math.cos(sp[2] + d) for d in dth])
        sin_t = np.array([math.sin(sp[2] + d) for d in dth])
        assert L.ndt2d_set_search(h, sp[0], sp[1], _capi.dptr(dth), _capi.dptr(cos_t),
                                  _capi.dptr(sin_t), len(dth), _capi.dptr(dlin), len(dlin)) == 0
        per = len(dlin) ** 2

        def run(b, e):
            res = _capi.MatchResult()
            sc = np.zeros((e - b) * per)
            assert L.ndt2d_match(h, b, e, _capi.dptr(sc), C.byref(res)) == 0, L.ndt2d_last_error(h)
            rec = np.zeros(12)
            rec[0] = res.best_score
            rec[1] = -1.0 if res.best_index == _capi.NO_INDEX else float(res.best_index)
            rec[2:] = res.acc[:]
            assert res.n_candidates == (e - b) * per
            return rec, sc

        full, sc_full = run(0, len(dth))
        assert np.max(np.abs(sc_full - g["scores"])) < TOL_TIGHT
        assert int(full[1]) == int(g["best_index"])
        for world in (2, 3, 8):
            recs, scs = zip(*[run(*shard.shard_range(len(dth), r, world)) for r in range(world)])
            assert np.array_equal(np.concatenate(scs), sc_full)  # per-candidate scores: bitwise
            s, i, acc = shard.combine_match_records(recs)
            assert (s, i) == (full[0], int(full[1]))
            assert np.allclose(acc, full[2:], rtol=1e-12, atol=0)
            assert np.allclose(shard.covariance_from_acc(acc), g["covariance"], rtol=1e-9, atol=0)
        # interleaved shares (rank r takes theta steps r, r + world, ...): the same
        # per-candidate scores, the same winner by flat index in the whole lattice
        def run_strided(first, stride, count):
            n = count * per
            assert L.ndt2d_match_launch_strided(h, first, stride, count, None, None) == _capi.OK
            res = _capi.MatchResult()
            assert L.ndt2d_match_fetch(h, C.byref(res)) == _capi.OK
            rec = np.zeros(12)
            rec[0] = res.best_score
            rec[1] = -1.0 if res.best_index == _capi.NO_INDEX else float(res.best_index)
            rec[2:] = res.acc[:]
            assert res.n_candidates == n
            return rec

        for world in (2, 3, 8):
            recs = [run_strided(*shard.shard_strided(len(dth), r, world)) for r in range(world)]
            s, i, acc = shard.combine_match_records(recs)
            assert (s, i) == (full[0], int(full[1]))
            assert np.allclose(acc, full[2:], rtol=1e-12, atol=0)
        # ... and through the lane-per-candidate kernel (this lattice is small enough to
        # get the wave mapping by default)
        assert L.ndt2d_set_variant(h, b"lane") == _capi.OK
        try:
            whole = run_strided(0, 1, len(dth))
            assert int(whole[1]) == int(full[1]) and abs(whole[0] - full[0]) < TOL_TIGHT
            recs = [run_strided(*shard.shard_strided(len(dth), r, 3)) for r in range(3)]
            s, i, acc = shard.combine_match_records(recs)
            assert (s, i) == (whole[0], int(whole[1]))
            assert np.allclose(acc, whole[2:], rtol=1e-12, atol=0)
        finally:
            assert L.ndt2d_set_variant(h, b"auto") == _capi.OK
        assert L.ndt2d_match_launch_strided(h, 0, 0, 1, None, None) == _capi.ERR_INVALID
        assert L.ndt2d_match_launch_strided(h, 1, 2, len(dth), None, None) == _capi.ERR_INVALID
        assert L.ndt2d_match_launch_strided(h, len(dth), 1, 1, None, None) == _capi.ERR_INVALID
        # bad ranges are rejected
        res = _capi.MatchResult()
        assert L.ndt2d_match(h, 5, 5, None, C.byref(res)) == _capi.ERR_INVALID
        assert L.ndt2d_match(h, 0, len(dth) + 1, None, C.byref(res)) == _capi.ERR_INVALID
    finally:
        L.ndt2d_destroy(h)


def test_kernel_variants_agree(cfg1):
    """Every candidate mapping / grid placement against the oracle; the two grid
    placements of one mapping agree bitwise (same arithmetic, different memory)."""
    gpu, ref, _, guess, pts = cfg1
    exp = ref.matchScan(guess, pts, want_scores=True)
    got = {}
    try:
        for name, tag in [("lane", "lane-per-candidate/lds-grid"),
                          ("small", "lane-per-candidate/small-lattice"),
                          ("wave-lds", "wave-per-candidate/lds-grid"),
                          ("wave-global", "wave-per-candidate/global-grid")]:
            gpu.set_variant(name)
            got[name] = gpu.matchScan(guess, pts, want_scores=True)
            assert tag in gpu.last_variant(), gpu.last_variant()
            _check_match(got[name], exp, 720)
    finally:
        gpu.set_variant("auto")
    assert np.array_equal(got["wave-lds"]["scores"], got["wave-global"]["scores"])
    # the lane mapping sums the beams in the reference's order: only exp() ulps remain
    d_lane = np.abs(got["lane"]["scores"] - exp["scores"])
    d_wave = np.abs(got["wave-lds"]["scores"] - exp["scores"])
    assert d_lane.max() <= d_wave.max() + 1e-13
    assert d_lane.max() < 1e-12
    # the small-lattice form adds in-order chunk sums in chunk order
    assert np.abs(got["small"]["scores"] - exp["scores"]).max() < 1e-12
    assert got["small"]["best_index"] == got["lane"]["best_index"] == exp["best_index"]


@pytest.mark.parametrize("cfg,override", [
    (1, {}),
    (1, dict(ndt_resolution=0.1, search_linear_size=0.3, search_angular_size=0.1)),
    (1, dict(ndt_resolution=1.0, search_linear_size=0.6, search_linear_resolution=0.03)),
    (2, dict(search_angular_size=0.05)),                                    # 4x4 sub-cell map
    (3, dict(search_angular_size=0.05, search_linear_size=0.5)),            # records gathered from HBM
])
def test_lane_skipping_is_bit_exact(cfg, override):
    """Every term the lane mapping skips -- empty sub-cells, sub-cells whose exponent
    bound is below the lane's threshold, exponents below the threshold -- is a term
    that cannot change the sum: with all skipping disabled ("lane-noskip": every beam
    of every candidate evaluated exactly) every candidate score is bit-identical."""
    gpu, ref, _, guess, pts = _pair(cfg, **override)
    try:
        gpu.set_variant("lane")
        fast = gpu.matchScan(guess, pts, want_scores=True)
        gpu.set_variant("lane-noskip")
        full = gpu.matchScan(guess, pts, want_scores=True)
        assert "lane-per-candidate" in gpu.last_variant()
    finally:
        gpu.set_variant("auto")
    assert np.array_equal(fast["scores"], full["scores"])
    assert fast["best_index"] == full["best_index"] and fast["score"] == full["score"]
    assert np.array_equal(fast["covariance"], full["covariance"], equal_nan=True)
    assert (fast["scores"] < 0).sum() > 100          # the search does hit the map
    exp = ref.matchScan(guess, pts, want_scores=True)
    _check_match(full, exp, min(720, len(pts)))


@pytest.mark.parametrize("cfg,override", [
    (1, {}),
    (1, dict(laser_max_beams=100, search_linear_size=0.05, search_linear_resolution=0.005,
             search_angular_size=0.1, search_angular_resolution=0.0025)),      # plugin defaults
    (1, dict(laser_max_beams=7, search_linear_size=0.1, search_angular_size=0.02)),   # one short group
    (1, dict(ndt_resolution=0.1, search_linear_size=0.3, search_angular_size=0.1)),
    (1, dict(ndt_resolution=0.3, search_linear_size=0.2, search_angular_size=0.05)),  # true division
    (3, dict(search_angular_size=0.05, search_linear_size=0.1)),            # windowed 201 x 201 map
])
def test_small_lattice_skipping_is_bit_exact(cfg, override):
    """The small-lattice search (beams split across the waves of a block) with all
    skipping disabled gives bit-identical candidate scores, and both give the oracle's
    result; a second run gives the same bits (fixed-order combination of the chunks)."""
    gpu, ref, _, guess, pts = _pair(cfg, **override)
    try:
        gpu.set_variant("small")
        fast = gpu.matchScan(guess, pts, want_scores=True)
        assert "small-lattice" in gpu.last_variant(), gpu.last_variant()
        again = gpu.matchScan(guess, pts, want_scores=True)
        gpu.set_variant("small-noskip")
        full = gpu.matchScan(guess, pts, want_scores=True)
    finally:
        gpu.set_variant("auto")
    assert np.array_equal(fast["scores"], full["scores"])
    assert np.array_equal(fast["scores"], again["scores"])
    assert fast["best_index"] == full["best_index"] and fast["score"] == full["score"]
    assert np.array_equal(fast["covariance"], full["covariance"], equal_nan=True)
    assert np.array_equal(fast["covariance"], again["covariance"], equal_nan=True)
    exp = ref.matchScan(guess, pts, want_scores=True)
    _check_match(fast, exp, min(gpu.params["laser_max_beams"], len(pts)))


def test_timing_events_can_be_switched_off(cfg1):
    """ndt2d_set_timing(0) (what the pluginlib shim does): same results, no event pair."""
    gpu, _, _, guess, pts = cfg1
    want = gpu.matchScan(guess, pts)
    L = _capi.lib()
    assert L.ndt2d_set_timing(gpu.device_handle, 0) == _capi.OK
    try:
        got = gpu.matchScan(guess, pts)
        with pytest.raises(_capi.Ndt2dError):
            gpu.last_launch_ms()
    finally:
        assert L.ndt2d_set_timing(gpu.device_handle, 1) == _capi.OK
    assert got["best_index"] == want["best_index"] and got["score"] == want["score"]
    assert np.array_equal(got["covariance"], want["covariance"])
    gpu.matchScan(guess, pts)
    assert gpu.last_launch_ms()[0] > 0.0


def test_replaced_beams_void_a_prepared_search(cfg1):
    """prepare_search(); a scoring call with ANOTHER scan; match_launch() must fail loudly
    instead of searching with the wrong beams -- and with the SAME scan it must work (the
    matcher recognises the beams it already holds and uploads nothing)."""
    gpu, _, _, guess, pts = cfg1
    n_th, _, _ = gpu.prepare_search(guess, pts)
    gpu.match_launch(0, n_th)
    want = gpu.match_fetch()
    gpu.scorePoses(pts, np.array([[0.1, 0.0, 0.0], [0.0, 0.1, 0.2]]))   # same scan
    gpu.match_launch(0, n_th)
    assert np.array_equal(gpu.match_fetch(), want)
    gpu.scorePoints(pts[::2], (0.0, 0.0, 0.0))                           # another scan
    with pytest.raises(_capi.Ndt2dError):
        gpu.match_launch(0, n_th)
    gpu.prepare_search(guess, pts)
    gpu.match_launch(0, n_th)
    assert np.array_equal(gpu.match_fetch(), want)


@pytest.mark.parametrize("max_beams", [100, 208, 209, 720])
def test_score_scan_then_match_scan_share_the_beams(max_beams):
    """The mapper's sequence scoreScan(scan) -> matchScan(scan) (src/ndt_mapper.cpp:514-515):
    up to 208 beams travel to the device as arguments of the scoring kernel, which leaves
    them in the context for the search; more are uploaded once.  Either way the search must
    see exactly this scan's beams -- and a different scan in between must replace them."""
    gpu, ref, _, guess, pts = _pair(1, laser_max_beams=max_beams, search_linear_size=0.1,
                                    search_angular_size=0.05)
    fresh, _, _, _, _ = _pair(1, laser_max_beams=max_beams, search_linear_size=0.1,
                              search_angular_size=0.05)
    # (this test is about the DEVICE's single-pose path; scans of up to 256 beams would
    # otherwise be scored on the host, tests/test_gpu_single_pose_host.py)
    gpu.set_single_pose_path("device")
    other = pts[::-1].copy() * 0.9
    s_other = gpu.scoreScan(guess, other)
    assert abs(s_other - ref.scoreScan(guess, other)) < TOL_TIGHT
    s = gpu.scoreScan(guess, pts)
    assert s == fresh.scorePoses(pts, [guess])[0]
    assert abs(s - ref.scoreScan(guess, pts)) < TOL_TIGHT
    got = gpu.matchScan(guess, pts, want_scores=True)
    want = fresh.matchScan(guess, pts, want_scores=True)
    assert np.array_equal(got["scores"], want["scores"])
    assert got["best_index"] == want["best_index"] and got["score"] == want["score"]
    _check_match(got, ref.matchScan(guess, pts, want_scores=True), min(max_beams, 720))
    # and the other way round: the search's upload serves the scoring call that follows
    assert gpu.scoreScan(guess, pts) == s


def test_few_poses_take_the_block_per_pose_kernel_bit_identically():
    """scorePoints / scoreScan (ONE pose) and up to 8 poses run a block-per-pose kernel
    whose sums are built in the batched kernel's order: bit-identical scores."""
    gpu, ref, _, guess, pts = _pair(3)
    gpu.set_single_pose_path("device")     # (the device's single-pose kernel is what is compared here)
    parts = synth.particles(3, 4096)
    parts[:128, :2] = guess[:2] + parts[:128, :2] / 23.0 * 0.4
    parts[:128, 2] = guess[2] + parts[:128, 2] / np.pi * 0.1
    batch = gpu.scorePoses(pts, parts)     # more than kFewPosesMax: the batched kernel
    assert "compact" in gpu.last_variant()
    assert (batch < 0).sum() > 100
    for i in range(0, 256, 7):
        assert gpu.scorePoints(pts, parts[i]) == batch[i]
        assert "block-per-pose" in gpu.last_variant()
    for n in (2, 5, 8):
        assert np.array_equal(gpu.scorePoses(pts, parts[:n]), batch[:n])
        assert "block-per-pose" in gpu.last_variant()
    assert np.array_equal(gpu.scorePoses(pts, parts[:9]), batch[:9])
    assert gpu.scoreScan(guess, pts) == gpu.scorePoses(pts, [guess])[0]
    w_ref = O.pf_measure(ref, parts[:32], pts)
    assert np.max(np.abs(batch[:32] - w_ref)) < TOL_TIGHT
    # fewer beams than chunks, one beam
    for sub in (pts[:5], pts[100:101]):
        assert gpu.scorePoints(sub, parts[3]) == gpu.scorePoses(sub, parts)[3]
        assert abs(gpu.scorePoints(sub, parts[3]) - ref.scorePoints(sub, parts[3])) < TOL_TIGHT


@pytest.mark.parametrize("n", [1, 8, 9, 63, 500, 2048, 2049])
def test_small_particle_sets_are_measured_in_one_launch(n):
    """A filter of the node's size (<= 500 particles by default, reference
    src/ndt_mapper.cpp:81-82): scoring and updateStatistics in one launch of the
    block-per-pose kernel; scores bit-identical to the batched kernel's, statistics equal to
    the reference's (src/particle_filter.cpp:163-218).  2049: the batched kernels again."""
    gpu, ref, _, guess, pts = _pair(3)
    parts = synth.particles(3, 4096)
    parts[:2500, :2] = guess[:2] + parts[:2500, :2] / 23.0 * 0.4
    parts[:2500, 2] = guess[2] + parts[:2500, 2] / np.pi * 0.1
    batch = gpu.scorePoses(pts, parts)
    assert "compact" in gpu.last_variant()
    got = gpu.scorePoses(pts, parts[:n])
    assert ("block-per-pose" in gpu.last_variant()) == (n <= 2048)
    assert np.array_equal(got, batch[:n])
    w_ref = O.pf_measure(ref, parts[:n], pts)
    assert np.max(np.abs(got - w_ref)) < TOL_TIGHT
    cov_prev = np.zeros((3, 3))
    cov_prev[2, 2] = 0.125
    w, mean, cov = pf_measure(gpu, parts[:n], pts, cov_prev=cov_prev)
    assert ("block-per-pose" in gpu.last_variant()) == (n <= 2048)
    w_n, mean_ref, cov_ref = O.pf_update_statistics(parts[:n], w_ref, cov_prev=cov_prev)
    assert np.allclose(w, w_n, rtol=1e-11, atol=1e-18)
    assert abs(w.sum() - 1.0) < 1e-12
    assert np.allclose(mean, mean_ref, rtol=1e-10, atol=1e-12)
    assert np.allclose(cov, cov_ref, rtol=1e-8, atol=1e-12)
    # a second call on the same beams (the filter's next measure): same bits
    w2, mean2, cov2 = pf_measure(gpu, parts[:n], pts, cov_prev=cov_prev)
    assert np.array_equal(w, w2) and np.array_equal(mean, mean2) and np.array_equal(cov, cov2)


def test_launch_timing_history(cfg1):
    """ndt2d_launch_history_ms: durations of back-to-back launches, read afterwards."""
    gpu, _, _, guess, pts = cfg1
    n_th, _, _ = gpu.prepare_search(guess, pts)
    for _ in range(5):
        gpu.match_launch(0, n_th)
    hist = gpu.launch_history_ms(5)
    last, n_kernels = gpu.last_launch_ms()
    assert len(hist) == 5 and all(0.0 < t < 50.0 for t in hist)
    assert hist[-1] == last and n_kernels >= 1
    assert len(gpu.launch_history_ms(3)) == 3
    gpu.match_fetch()
    # the ring keeps the last 256 launches
    for _ in range(300):
        gpu.match_launch(0, 1)
    hist = gpu.launch_history_ms(1000)
    assert len(hist) == 256 and all(0.0 < t < 50.0 for t in hist)
    assert hist[-1] == gpu.last_launch_ms()[0]
    gpu.match_fetch()


def test_runs_are_deterministic(cfg1):
    gpu, _, _, guess, pts = cfg1
    a = gpu.matchScan(guess, pts, want_scores=True)
    b = gpu.matchScan(guess, pts, want_scores=True)
    assert np.array_equal(a["scores"], b["scores"])
    assert np.array_equal(a["covariance"], b["covariance"]) and a["score"] == b["score"]


def test_non_power_of_two_cell_size_uses_true_division():
    # ndt_resolution 0.1: t * (1/0.1) != t / 0.1 in general (reference ndt_model.cpp:210-211)
    gpu, ref, _, guess, pts = _pair(1, ndt_resolution=0.1, search_linear_size=0.3,
                                    search_angular_size=0.1)
    got = gpu.matchScan(guess, pts, want_scores=True)
    exp = ref.matchScan(guess, pts, want_scores=True)
    assert "/div" in gpu.last_variant()
    _check_match(got, exp, 720)
    parts = synth.particles(3, 2048)
    parts[:, :2] *= 4.5 / 23.0
    assert np.max(np.abs(gpu.scorePoses(pts, parts) - O.pf_measure(ref, parts, pts))) < TOL_TIGHT


@pytest.mark.parametrize("n_beams", [1, 63, 64, 65, 100, 129, 719, 1500])
def test_ragged_beam_counts(cfg1, n_beams):
    # tail lanes / every beams-per-lane specialisation incl. the > 1024-beam variant
    _, _, scans, guess, _ = cfg1
    w = synth.world_of(1)
    pts = synth.scan(w, (0.13, -0.07, 0.031), 900 + n_beams, n_beams=n_beams)
    params = synth.matcher_params(1, laser_max_beams=4000, search_linear_size=0.2,
                                  search_angular_size=0.05)
    gpu = ScanMatcherNDT(0)
    gpu.initialize("t", **params)
    gpu.addScans(scans)
    ref = O.ScanMatcherNDT()
    ref.initialize(**params)
    ref.addScans(scans)
    _check_match(gpu.matchScan(guess, pts, want_scores=True),
                 ref.matchScan(guess, pts, want_scores=True), n_beams)
    poses = synth.particles(3, 300)
    poses[:, :2] *= 4.0 / 23.0
    assert np.max(np.abs(gpu.scorePoses(pts, poses) - O.pf_measure(ref, poses, pts))) < TOL_TIGHT


@pytest.mark.parametrize("lin,ang", [((0.004, 0.005), (0.001, 0.0025)),    # 2 x 2 x 1 lattice
                                     ((0.0, 0.005), (0.0, 0.0025)),         # empty lattice
                                     ((0.011, 0.005), (0.006, 0.0025)),     # 5 x 5 x 5
                                     ((0.3, 0.07), (0.02, 0.013))])          # 9 x 9 x 4, ragged patches
def test_tiny_and_ragged_lattices(cfg1, lin, ang):
    _, _, scans, guess, pts = cfg1
    params = synth.matcher_params(1, search_linear_size=lin[0], search_linear_resolution=lin[1],
                                  search_angular_size=ang[0], search_angular_resolution=ang[1])
    ref = O.ScanMatcherNDT()
    ref.initialize(**params)
    ref.addScans(scans)
    gpu = ScanMatcherNDT(0)
    gpu.initialize("t", **params)
    gpu.addScans(scans)
    exp = ref.matchScan(guess, pts, pose=(9.0, 9.0, 9.0), want_scores=True)
    for variant in ("lane", "wave"):
        gpu.set_variant(variant)
        got = gpu.matchScan(guess, pts, pose=(9.0, 9.0, 9.0), want_scores=True)
        assert got["n_candidates"] == exp["n_candidates"]
        if exp["n_candidates"] == 0:
            # the loops never run: pose untouched, 0/0 covariance, 0.0 returned
            assert got["score"] == exp["score"] == 0.0
            assert np.array_equal(got["pose"], [9.0, 9.0, 9.0])
            assert np.isnan(got["covariance"]).all() and np.isnan(exp["covariance"]).all()
        else:
            _check_match(got, exp, 720)


def test_single_beam_scan(cfg1):
    gpu, ref, _, guess, pts = cfg1
    one = pts[100:101]
    _check_match(gpu.matchScan(guess, one, want_scores=True), ref.matchScan(guess, one, want_scores=True), 1)


def test_cfg2_full_size():
    """BASELINE.json configs[1]: 2,000,000 candidates x 720 beams on one GPU."""
    gpu, ref, _, guess, pts = _pair(2)
    got = gpu.matchScan(guess, pts, want_scores=True)
    assert got["n_candidates"] == 2000000
    # the headline configuration: compacted records in LDS, two blocks per CU
    assert "lane-per-candidate/lds-grid/compact-records" in gpu.last_variant(), gpu.last_variant()
    exp = ref.matchScan(guess, pts, omp_threads=os.cpu_count(), want_scores=True)
    want = _big_winner(2)
    assert got["best_index"] == exp["best_index"] == want["best_index"]
    # EVERY score against the CPU path (src/scan_matcher_ndt.cpp:127),
    # raw sums and the returned /N scale, all 2,000,000 candidates
    d_all = np.abs(got["scores"] - exp["scores"])
    assert float(d_all.max()) < TOL_TIGHT, float(d_all.max())
    assert float(d_all.max()) / 720 < 1e-5
    assert int(np.argmin(exp["scores"])) == got["best_index"]
    assert [float(v).hex() for v in got["pose"]] == want["pose_hex"]
    assert abs(got["score"] - want["score"]) < TOL_TIGHT
    assert np.array_equal(got["pose"], exp["pose"])
    assert abs(got["score"] - exp["score"]) < TOL_TIGHT
    assert np.allclose(got["covariance"], exp["covariance"], rtol=1e-9, atol=0)
    # sampled candidates against the oracle's NDT::likelihood on the reference's
    # points_outer / points_inner construction (scan_matcher_ndt.cpp:106-125)
    p = gpu.params
    dth = O.search_offsets(p["search_angular_size"], p["search_angular_resolution"])
    dlin = O.search_offsets(p["search_linear_size"], p["search_linear_resolution"])
    rng = np.random.default_rng(7)
    idx = np.unique(np.concatenate([rng.integers(0, 2000000, 1500), [0, 1999999, got["best_index"]]]))
    for flat in idx:
        ith, ix, iy = shard.decode_index(int(flat), len(dlin))
        c, s = math.cos(guess[2] + dth[ith]), math.sin(guess[2] + dth[ith])
        ox = pts[:, 0] * c - pts[:, 1] * s + guess[0]
        oy = pts[:, 0] * s + pts[:, 1] * c + guess[1]
        inner = np.stack([ox + dlin[ix], oy + dlin[iy]], axis=1)
        assert abs(got["scores"][flat] + ref.ndt.likelihood(inner)) < TOL_TIGHT
    # the winner is the global minimum, ties to the lowest flat index
    assert got["best_index"] == int(np.argmin(got["scores"]))
    assert got["score"] * 720 == got["scores"].min()
    # accumulators against a float64 recomputation from the GPU's own scores
    sc = got["scores"]
    assert got["covariance"][2, 2] > 0
    s_sum = math.fsum(sc)
    k22 = math.fsum(np.repeat(dth, len(dlin) ** 2) ** 2 * sc)
    u2 = math.fsum(np.repeat(dth, len(dlin) ** 2) * sc)
    assert got["covariance"][2, 2] == pytest.approx(k22 / s_sum + (u2 / s_sum) ** 2, rel=1e-9)


def test_theta_slabs_of_one_search_give_the_single_launch_result(monkeypatch):
    """A lattice beyond kLaneSlabItems work items is searched slab after slab inside ONE
    ndt2d_match_launch.  Forced here on the cfg-2 lattice (169 items per
    theta step): every candidate score bitwise as from the single launch, same winner."""
    gpu, ref, _, guess, pts = _pair(2)
    one = gpu.matchScan(guess, pts, want_scores=True)
    assert "lane-per-candidate" in gpu.last_variant()
    for slab_items, n_slabs in ((5000, 7), (169, 29)):   # 29 / 1 -> clamped to 7 theta steps per slab
        monkeypatch.setenv("NDT2D_LANE_SLAB_ITEMS", str(slab_items))
        many = gpu.matchScan(guess, pts, want_scores=True)
        assert "lane-per-candidate" in gpu.last_variant()
        assert gpu.last_launch_ms()[1] == 3 * n_slabs + 1
        assert np.array_equal(many["scores"], one["scores"])
        assert many["best_index"] == one["best_index"] and many["score"] == one["score"]
        assert np.array_equal(many["pose"], one["pose"])
        assert np.allclose(many["covariance"], one["covariance"], rtol=1e-11, atol=0)
        # ... and a rank's interleaved share of it
        monkeypatch.delenv("NDT2D_LANE_SLAB_ITEMS")
        n_th, n_lin, _ = gpu.prepare_search(guess, pts)
        gpu.match_launch_strided(1, 3, (n_th - 1 + 2) // 3)
        want = gpu.match_fetch()
        monkeypatch.setenv("NDT2D_LANE_SLAB_ITEMS", str(slab_items))
        gpu.match_launch_strided(1, 3, (n_th - 1 + 2) // 3)
        got = gpu.match_fetch()
        assert (got[0], got[1]) == (want[0], want[1])
        assert np.allclose(got[2:], want[2:], rtol=1e-11, atol=0)
    monkeypatch.delenv("NDT2D_LANE_SLAB_ITEMS")
    # slabs of a lattice whose beams are cut into parts (6,760 items)
    mid, ref_mid, _, guess, pts = _pair(2, search_angular_size=0.1)
    one = mid.matchScan(guess, pts, want_scores=True)
    assert "beam-parts" in mid.last_variant()
    monkeypatch.setenv("NDT2D_LANE_SLAB_ITEMS", "2000")
    many = mid.matchScan(guess, pts, want_scores=True)
    monkeypatch.delenv("NDT2D_LANE_SLAB_ITEMS")
    assert "beam-parts" in mid.last_variant() and mid.last_launch_ms()[1] > 4
    assert np.array_equal(many["scores"], one["scores"])
    assert many["best_index"] == one["best_index"] and many["score"] == one["score"]
    assert np.allclose(many["covariance"], one["covariance"], rtol=1e-11, atol=0)
    _check_match(many, ref_mid.matchScan(guess, pts, want_scores=True), 720)


def test_lattice_beyond_two_to_the_24_work_items_keeps_the_lane_mapping():
    """513 x 513 x 4189 = 1.1e9 candidates = 17.7 M (theta, patch) work items x 256 beams:
    beyond the 2^24 items one launch can hold (it took the 4x slower wave mapping in
    round 2), now nine slabs of the lane-per-candidate search.  The combined winner equals
    the best of separate launches over theta ranges, its score the oracle's likelihood."""
    gpu, ref, _, guess, pts = _pair(1, search_linear_size=2.56, search_linear_resolution=0.01,
                                    search_angular_size=math.pi, search_angular_resolution=0.0015,
                                    laser_max_beams=256)
    n_th, n_lin, n_beams = gpu.prepare_search(guess, pts)
    p1 = (n_lin + 7) // 8
    assert n_th * p1 * p1 > 2 ** 24 and n_beams == 256
    gpu.match_launch(0, n_th)
    full = gpu.match_fetch()
    assert "lane-per-candidate" in gpu.last_variant(), gpu.last_variant()
    assert gpu.last_launch_ms()[1] > 4
    recs = []
    bounds = np.linspace(0, n_th, 12).astype(int)
    for a, b in zip(bounds[:-1], bounds[1:]):
        gpu.match_launch(int(a), int(b))
        recs.append(gpu.match_fetch())
        assert gpu.last_launch_ms()[1] == 4          # each a single slab: table, search, two reduction stages
    s, i, acc = shard.combine_match_records(recs)
    assert (s, i) == (full[0], int(full[1]))
    assert np.allclose(acc, full[2:], rtol=1e-10, atol=0)
    p = gpu.params
    dth = O.search_offsets(p["search_angular_size"], p["search_angular_resolution"])
    dlin = O.search_offsets(p["search_linear_size"], p["search_linear_resolution"])
    assert (len(dth), len(dlin)) == (n_th, n_lin)
    ith, ix, iy = shard.decode_index(i, n_lin)
    use = pts[(np.arange(256) * (len(pts) / 256.0)).astype(int)]
    c, sn = math.cos(guess[2] + dth[ith]), math.sin(guess[2] + dth[ith])
    inner = np.stack([use[:, 0] * c - use[:, 1] * sn + guess[0] + dlin[ix],
                      use[:, 0] * sn + use[:, 1] * c + guess[1] + dlin[iy]], axis=1)
    assert abs(full[0] + ref.ndt.likelihood(inner)) < TOL_TIGHT
    # one theta step of the last slab in full against the oracle
    import torch
    t = n_th - 3
    d_scores = torch.zeros(n_lin * n_lin, dtype=torch.float64, device="cuda:0")
    gpu.match_launch(t, t + 1, scores_ptr=d_scores.data_ptr())
    gpu.synchronize()
    slab = d_scores.cpu().numpy()
    assert slab.min() >= full[0]
    c, sn = math.cos(guess[2] + dth[t]), math.sin(guess[2] + dth[t])
    ox = use[:, 0] * c - use[:, 1] * sn + guess[0]
    oy = use[:, 0] * sn + use[:, 1] * c + guess[1]
    rng = np.random.default_rng(12)
    for f in rng.integers(0, n_lin * n_lin, 300):
        inner = np.stack([ox + dlin[f // n_lin], oy + dlin[f % n_lin]], axis=1)
        assert abs(slab[f] + ref.ndt.likelihood(inner)) < TOL_TIGHT


def test_mid_size_lattices_cut_a_candidates_beams_into_parts(monkeypatch):
    """Lattices of a few thousand (theta, patch) work items with many beams: one item of the
    large search is a chain of up to 720 exact evaluations, 0.2 ms for a wave however few
    items there are.  The beams are therefore cut into 4 (or 2) parts, each a work item of
    its own, and a second kernel adds a candidate's part sums in order.  Skipping stays
    bit-exact, the cut follows the WHOLE lattice (so score bits do not depend on the
    sharding), and the scores are within a few ulps of the uncut search's."""
    gpu, ref, _, guess, pts = _pair(2, search_linear_size=1.0, search_linear_resolution=0.02,
                                    search_angular_size=0.1, search_angular_resolution=0.005)
    exp = ref.matchScan(guess, pts, want_scores=True)
    got = gpu.matchScan(guess, pts, want_scores=True)           # 40 x 169 = 6,760 items
    assert "compact-records/beam-parts" in gpu.last_variant(), gpu.last_variant()
    assert gpu.last_launch_ms()[1] == 5   # table, search, combine, two reduction stages
    _check_match(got, exp, 720)
    gpu.set_variant("lane-noskip")
    full = gpu.matchScan(guess, pts, want_scores=True)
    gpu.set_variant("auto")
    assert "beam-parts" in gpu.last_variant()
    assert np.array_equal(got["scores"], full["scores"])
    assert got["best_index"] == full["best_index"] and np.array_equal(got["covariance"], full["covariance"])
    # theta shards of the same lattice: the same bits for every candidate
    import torch
    n_th, n_lin, _ = gpu.prepare_search(guess, pts)
    per = n_lin * n_lin
    for a, b in ((0, 7), (7, 29), (29, n_th)):
        d = torch.zeros((b - a) * per, dtype=torch.float64, device="cuda:0")
        gpu.match_launch(a, b, scores_ptr=d.data_ptr())
        gpu.synchronize()
        assert "beam-parts" in gpu.last_variant()
        assert np.array_equal(d.cpu().numpy(), got["scores"][a * per:b * per])
    # ... also for shares of fewer than a CU-load of items each (8 interleaved shares of 5 theta
    # steps = 845 items: a launch this small on its own would take the uncut 256-thread form)
    for r in range(8):
        first, stride, count = shard.shard_strided(n_th, r, 8)
        d = torch.zeros(count * per, dtype=torch.float64, device="cuda:0")
        gpu.match_launch_strided(first, stride, count, scores_ptr=d.data_ptr())
        gpu.synchronize()
        assert "beam-parts" in gpu.last_variant(), gpu.last_variant()
        mine = np.concatenate([got["scores"][t * per:(t + 1) * per] for t in range(first, n_th, stride)])
        assert np.array_equal(d.cpu().numpy(), mine)
    # the uncut search (one running sum per candidate, the reference's order)
    monkeypatch.setenv("NDT2D_LANE_PARTS", "1")
    uncut = gpu.matchScan(guess, pts, want_scores=True)
    monkeypatch.delenv("NDT2D_LANE_PARTS")
    assert "beam-parts" not in gpu.last_variant()
    assert uncut["best_index"] == got["best_index"]
    assert np.max(np.abs(uncut["scores"] - got["scores"])) < 1e-12
    # two parts, and a map whose records are gathered from HBM
    gpu2, ref2, _, guess2, pts2 = _pair(2, search_linear_size=1.0, search_linear_resolution=0.02,
                                        search_angular_size=0.2, search_angular_resolution=0.005)
    got2 = gpu2.matchScan(guess2, pts2, want_scores=True)        # 80 x 169 = 13,520 items
    assert "beam-parts" in gpu2.last_variant()
    _check_match(got2, ref2.matchScan(guess2, pts2, want_scores=True), 720)
    gpu3, ref3, _, guess3, pts3 = _pair(3, search_linear_size=0.5, search_linear_resolution=0.02,
                                        search_angular_size=0.1, search_angular_resolution=0.005)
    got3 = gpu3.matchScan(guess3, pts3, want_scores=True)        # 40 x 49 = 1,960 ... small; force the lane form
    gpu3.set_variant("lane")
    lane3 = gpu3.matchScan(guess3, pts3, want_scores=True)
    assert "lds-map+global-records/beam-parts" in gpu3.last_variant(), gpu3.last_variant()
    gpu3.set_variant("auto")
    exp3 = ref3.matchScan(guess3, pts3, want_scores=True)
    _check_match(got3, exp3, 720)
    _check_match(lane3, exp3, 720)


@pytest.mark.parametrize("cfg,search", [
    (1, dict(search_linear_size=0.5, search_linear_resolution=0.05, search_angular_size=0.2,
             search_angular_resolution=0.01)),
    (3, dict(search_linear_size=0.3, search_linear_resolution=0.05, search_angular_size=0.05,
             search_angular_resolution=0.01)),
])
def test_grid_installed_as_a_list_of_cells_equals_the_dense_install(cfg, search):
    """ndt2d_set_grid_sparse (the cells that hold points, in any order; what addScans sends
    since round 3) against ndt2d_set_grid (every cell): the same device state -- every
    candidate score of a search bitwise, with and without skipping, and the same records
    back through ndt2d_get_grid."""
    from ndt_2d_amd import host_build_grid
    scans = synth.map_scans(cfg)
    params = synth.matcher_params(cfg)
    cells, sx, sy, ox, oy = host_build_grid(params["ndt_resolution"], params["range_max"], scans)
    listed = np.flatnonzero(cells[:, 5] > 0).astype(np.uint32)
    assert 0 < len(listed) < len(cells) and np.any(cells[listed, 5] < 5)    # also cells that cannot score
    rng = np.random.default_rng(3)
    rng.shuffle(listed)
    sparse6 = np.ascontiguousarray(cells[listed])
    guess, pts, _ = synth.query_scan(cfg)
    pose = guess + np.array([0.03, -0.02, 0.01])
    dth = O.search_offsets(search["search_angular_size"], search["search_angular_resolution"])
    dlin = O.search_offsets(search["search_linear_size"], search["search_linear_resolution"])
    cos_t = np.array([math.cos(pose[2] + d) for d in dth])
    sin_t = np.array([math.sin(pose[2] + d) for d in dth])
    L = _capi.lib()
    out = {}
    for how in ("dense", "sparse"):
        h = C.c_void_p()
        assert L.ndt2d_create(C.byref(h), 0) == 0
        try:
            if how == "dense":
                assert L.ndt2d_set_grid(h, _capi.dptr(cells), sx, sy, 0.25, ox, oy) == 0
            else:
                assert L.ndt2d_set_grid_sparse(h, listed.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               _capi.dptr(sparse6), len(listed), sx, sy, 0.25, ox, oy) == 0
            assert L.ndt2d_set_beams(h, _capi.dptr(pts), len(pts)) == 0
            assert L.ndt2d_set_search(h, pose[0], pose[1], _capi.dptr(dth), _capi.dptr(cos_t),
                                      _capi.dptr(sin_t), len(dth), _capi.dptr(dlin), len(dlin)) == 0
            got = {}
            for variant in (b"auto", b"lane", b"lane-noskip", b"small", b"wave"):
                if L.ndt2d_set_variant(h, variant) != 0:
                    continue
                res = _capi.MatchResult()
                sc = np.zeros(len(dth) * len(dlin) ** 2)
                rc = L.ndt2d_match(h, 0, len(dth), _capi.dptr(sc), C.byref(res))
                if rc != 0:
                    continue          # (a variant this lattice / map does not support)
                got[variant] = (sc, res.best_index, res.best_score, L.ndt2d_last_variant(h))
            back = np.zeros_like(cells)
            assert L.ndt2d_get_grid(h, _capi.dptr(back), len(back), None, None, None, None, None) == 0
            out[how] = (got, back)
        finally:
            L.ndt2d_destroy(h)
    assert set(out["dense"][0]) == set(out["sparse"][0]) and b"auto" in out["dense"][0]
    for variant, (sc, bi, bs, name) in out["dense"][0].items():
        sc2, bi2, bs2, name2 = out["sparse"][0][variant]
        assert np.array_equal(sc, sc2), variant
        assert (bi, bs) == (bi2, bs2)
    assert np.array_equal(out["dense"][1], cells) and np.array_equal(out["sparse"][1], cells)
    assert np.array_equal(out["sparse"][0][b"lane"][0], out["sparse"][0][b"lane-noskip"][0])
    assert (out["dense"][0][b"auto"][0] < 0).sum() > 100


@pytest.mark.parametrize("resolution", [0.125, 0.1])
def test_particles_on_a_grid_whose_bitmap_does_not_fit_lds(resolution):
    """cfg-5's 200 x 200 m map at 0.125 m (1601 x 1601 cells, a 320 KB occupancy bitmap) and at
    0.1 m (2001 x 2001, true division): the batched scorePoints kernel screens with one bit per
    block of 2 x 2 (4 x 4) cells instead of falling back to the dense kernel.  Weights against
    the oracle, the screen against its exact control bit for bit, and against the dense kernel."""
    gpu, ref, _, guess, pts = _pair(5, ndt_resolution=resolution)
    parts = synth.particles(5, 20000)
    parts[:4000, :2] = guess[:2] + (parts[:4000, :2] / 95.0) * 3.0      # a cluster around the true pose
    got = gpu.scorePoses(pts, parts)
    assert "compact/coarse-bitmap" in gpu.last_variant(), gpu.last_variant()
    exp = O.pf_measure(ref, parts, pts, omp_threads=os.cpu_count())
    assert np.max(np.abs(got - exp)) < TOL_TIGHT
    assert (exp < -1e-3).sum() > 500
    gpu.set_variant("compact-exact")
    exact = gpu.scorePoses(pts, parts)
    assert "compact/coarse-bitmap" in gpu.last_variant(), gpu.last_variant()
    gpu.set_variant("dense")
    dense = gpu.scorePoses(pts, parts)
    assert "compact" not in gpu.last_variant()
    gpu.set_variant("auto")
    assert np.array_equal(got, exact)
    assert np.max(np.abs(got - dense)) < 1e-12
    # a new grid invalidates the block bitmap: the same matcher on the cfg-3 map
    gpu.reset()
    gpu.addScans(synth.map_scans(3))
    p3 = synth.particles(3, 5000)
    w3 = gpu.scorePoses(pts, p3)
    ref3 = O.ScanMatcherNDT()
    ref3.initialize(**dict(synth.matcher_params(5, ndt_resolution=resolution)))
    ref3.addScans(synth.map_scans(3))
    assert np.max(np.abs(w3 - O.pf_measure(ref3, p3, pts))) < TOL_TIGHT


def _big_winner(cfg):
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_winners.json")) as f:
        return json.load(f)["cfg%d" % cfg]


def test_cfg4_sized_lattice_properties():
    """BASELINE.json configs[3] lattice (501 x 501 x 1257 = 315.5M candidates) on one GPU:
    the oracle's winner over the WHOLE lattice (tests/golden/big_winners.json, 3.5 min on
    8 cores) from the full launch, from 8 contiguous slabs and from 8 round-robin shares;
    in-slab sampled parity; consistency with the cfg-2 search it contains."""
    gpu, ref, _, guess, pts = _pair(4)
    want = _big_winner(4)
    n_th, n_lin, n_beams = gpu.prepare_search(guess, pts)
    assert (n_th, n_lin, n_beams) == (1257, 501, 720) == (want["n_theta"], want["n_linear"], 720)
    gpu.match_launch(0, n_th)
    full = gpu.match_fetch()
    # "select the same best pose" (north_star; src/scan_matcher_ndt.cpp:128-134): the room
    # is 4-fold symmetric but the noisy scan makes ONE basin the oracle's winner
    assert int(full[1]) == want["best_index"] == 80443810
    recs = []
    for r in range(8):
        gpu.match_launch(*shard.shard_range(n_th, r, 8))
        recs.append(gpu.match_fetch())
    s, i, acc = shard.combine_match_records(recs)
    assert (s, i) == (full[0], int(full[1]))
    assert np.allclose(acc, full[2:], rtol=1e-11, atol=0)
    recs = []
    for r in range(8):
        gpu.match_launch_strided(*shard.shard_strided(n_th, r, 8))
        recs.append(gpu.match_fetch())
    s8, i8, acc8 = shard.combine_match_records(recs)
    assert (s8, i8) == (full[0], int(full[1]))
    assert np.allclose(acc8, full[2:], rtol=1e-11, atol=0)
    out = gpu.finish_match(full)
    assert [float(v).hex() for v in out["pose"]] == want["pose_hex"]
    assert abs(out["score"] - want["score"]) < TOL_TIGHT
    assert np.allclose(out["covariance"], want["covariance"], rtol=1e-9, atol=0)
    k = round((out["pose"][2] - 0.031) / (math.pi / 2))
    assert k == -1 and abs(out["pose"][2] - (0.031 + k * math.pi / 2)) <= 0.005 + 1e-9
    assert np.hypot(out["pose"][0], out["pose"][1]) < 0.3
    # the winner's score against the oracle's likelihood of that candidate
    p = gpu.params
    dth = O.search_offsets(p["search_angular_size"], p["search_angular_resolution"])
    dlin = O.search_offsets(p["search_linear_size"], p["search_linear_resolution"])
    ith, ix, iy = shard.decode_index(i, n_lin)
    assert (dlin[ix], dlin[iy], dth[ith]) == tuple(out["pose"])
    c, sn = math.cos(guess[2] + dth[ith]), math.sin(guess[2] + dth[ith])
    inner = np.stack([pts[:, 0] * c - pts[:, 1] * sn + guess[0] + dlin[ix],
                      pts[:, 0] * sn + pts[:, 1] * c + guess[1] + dlin[iy]], axis=1)
    assert abs(full[0] + ref.ndt.likelihood(inner)) < TOL_TIGHT
    # one whole theta slab against the oracle restricted to that theta (angular loop of 1)
    t = 700
    import torch
    d_scores = torch.zeros(n_lin * n_lin, dtype=torch.float64, device="cuda:0")
    gpu.match_launch(t, t + 1, scores_ptr=d_scores.data_ptr())
    gpu.synchronize()
    slab = d_scores.cpu().numpy()
    c, sn = math.cos(guess[2] + dth[t]), math.sin(guess[2] + dth[t])
    ox = pts[:, 0] * c - pts[:, 1] * sn + guess[0]
    oy = pts[:, 0] * sn + pts[:, 1] * c + guess[1]
    rng = np.random.default_rng(11)
    for f in rng.integers(0, n_lin * n_lin, 400):
        inner = np.stack([ox + dlin[f // n_lin], oy + dlin[f % n_lin]], axis=1)
        assert abs(slab[f] + ref.ndt.likelihood(inner)) < TOL_TIGHT


def test_match_scan_on_a_large_map_uses_the_windowed_lane_kernel():
    """matchScan against the 201 x 201 global map (scan-match localisation,
    reference src/ndt_mapper.cpp:547-566): the grid does not fit in LDS, the
    search window does."""
    gpu, ref, _, guess, pts = _pair(3, search_linear_size=0.5, search_linear_resolution=0.05,
                                    search_angular_size=0.1, search_angular_resolution=0.01)
    wrong = guess + np.array([0.21, -0.13, 0.04])          # start off the true pose
    exp = ref.matchScan(wrong, pts, want_scores=True)
    # (8,820 candidates: left to itself the library takes the wave mapping here, below)
    gpu.set_variant("lane")
    try:
        got = gpu.matchScan(wrong, pts, want_scores=True)
        assert "lane-per-candidate/lds-map+global-records" in gpu.last_variant(), gpu.last_variant()
        _check_match(got, exp, 720)
        assert np.all(np.abs(wrong + got["pose"] - guess) <= [0.05, 0.05, 0.01])
        # a scan pose near the map edge and one outside the map
        for pose in [(-24.0, 20.0, 0.5), (60.0, 60.0, 0.0)]:
            a = gpu.matchScan(pose, pts, want_scores=True)
            b = ref.matchScan(pose, pts, want_scores=True)
            _check_match(a, b, 720)
    finally:
        gpu.set_variant("auto")
    # left to itself the library gives a lattice this small the small-lattice form of the
    # lane mapping (window copied from the per-cell map bytes, records gathered from HBM)
    alt = gpu.matchScan(wrong, pts, want_scores=True)
    assert "small-lattice" in gpu.last_variant(), gpu.last_variant()
    _check_match(alt, exp, 720)
    for pose in [(-24.0, 20.0, 0.5), (60.0, 60.0, 0.0)]:
        _check_match(gpu.matchScan(pose, pts, want_scores=True),
                     ref.matchScan(pose, pts, want_scores=True), 720)
    # and the wave-per-candidate kernel (records gathered from HBM through L2)
    gpu.set_variant("wave")
    alt = gpu.matchScan(wrong, pts, want_scores=True)
    gpu.set_variant("auto")
    assert "wave-per-candidate/global-grid" in gpu.last_variant()
    _check_match(alt, exp, 720)
    for pose in [(-24.0, 20.0, 0.5), (60.0, 60.0, 0.0)]:
        _check_match(gpu.matchScan(pose, pts, want_scores=True),
                     ref.matchScan(pose, pts, want_scores=True), 720)


def test_wide_windows_keep_the_lane_mapping():
    """cfg-5's 801 x 801 map and its scan, whose longest beams reach past 30 m: the search
    window is wider than the 256 cells a one-byte map coordinate holds, so the map goes to
    one byte per 2 x 2 (or 4 x 4) block of grid cells and every live lane takes the
    reference's own index arithmetic.  Same results, skipping still bit-exact."""
    gpu, ref, _, guess, pts = _pair(5, search_linear_size=0.5, search_linear_resolution=0.05,
                                    search_angular_size=0.1, search_angular_resolution=0.01)
    reach = np.max(np.hypot(pts[:, 0], pts[:, 1])) + 0.5
    assert 2 * reach / 0.25 > 256
    wrong = guess + np.array([0.16, -0.12, 0.03])
    exp = ref.matchScan(wrong, pts, want_scores=True)
    # (a lattice this small takes the small-lattice search, its map copied from the grid's bytes
    # per 2 x 2 block of cells -- round 2 left it to the wave mapping; the others on request)
    small = gpu.matchScan(wrong, pts, want_scores=True)
    assert "small-lattice" in gpu.last_variant(), gpu.last_variant()
    _check_match(small, exp, 720)
    gpu.set_variant("small-noskip")
    control_small = gpu.matchScan(wrong, pts, want_scores=True)
    assert "small-lattice" in gpu.last_variant(), gpu.last_variant()
    assert np.array_equal(small["scores"], control_small["scores"])
    gpu.set_variant("wave")
    wave = gpu.matchScan(wrong, pts, want_scores=True)
    assert "wave-per-candidate" in gpu.last_variant(), gpu.last_variant()
    _check_match(wave, exp, 720)
    gpu.set_variant("lane")
    got = gpu.matchScan(wrong, pts, want_scores=True)
    assert "lane-per-candidate/lds-map+global-records/block-map" in gpu.last_variant(), gpu.last_variant()
    _check_match(got, exp, 720)
    gpu.set_variant("lane-noskip")
    control = gpu.matchScan(wrong, pts, want_scores=True)
    gpu.set_variant("auto")
    assert np.array_equal(got["scores"], control["scores"])
    # a larger lattice (dynamic work items, the patch pre-test on the coarse map), near the
    # map's edge and outside it
    big, ref_big, _, _, _ = _pair(5, search_linear_size=1.0, search_linear_resolution=0.02,
                                  search_angular_size=0.075, search_angular_resolution=0.005)
    for pose in (wrong, (-95.0, 90.0, 0.4), (140.0, 0.0, 0.0)):
        a = big.matchScan(pose, pts, want_scores=True)
        assert "lane-per-candidate" in big.last_variant() and "block-map" in big.last_variant()
        _check_match(a, ref_big.matchScan(pose, pts, want_scores=True), 720)


def test_mapping_follows_the_lattice_size(cfg1):
    """Small lattices (the plugin's defaults, cfg-1) take the small-lattice form of the
    lane-per-candidate mapping (beams split across the waves of a block), large ones the
    persistent form (a window wider than 1,024 cells would take the wave-per-candidate
    mapping).  All give the oracle's result."""
    gpu, ref, _, guess, pts = _pair(1)
    got = gpu.matchScan(guess, pts, want_scores=True)
    assert "small-lattice" in gpu.last_variant(), gpu.last_variant()
    exp = ref.matchScan(guess, pts, want_scores=True)
    _check_match(got, exp, 720)
    big, ref_big, _, _, _ = _pair(1, search_linear_size=0.25, search_linear_resolution=0.01,
                                  search_angular_size=0.1, search_angular_resolution=0.002)
    got = big.matchScan(guess, pts, want_scores=True)      # 50 x 50 x 100 candidates
    assert "lane-per-candidate" in big.last_variant(), big.last_variant()
    _check_match(got, ref_big.matchScan(guess, pts, want_scores=True), 720)


@pytest.mark.parametrize("cfg", [1, 3, 5])
def test_device_ndt_build_is_bit_identical(cfg):
    """N1: addScans on the GPU (stable sort by cell + in-order per-cell recurrence)
    against the oracle's NDT, bit for bit, and against the host build."""
    scans = synth.map_scans(cfg)
    params = synth.matcher_params(cfg)
    ref = O.ScanMatcherNDT()
    ref.initialize(**params)
    ref.addScans(scans)
    exp = ref.ndt.cells6()
    got = {}
    for mode in ("device", "host"):
        gpu = ScanMatcherNDT(0)
        gpu.initialize("t", **params)
        gpu.set_build_mode(mode)
        gpu.addScans(scans)
        cells, sx, sy, cs, ox, oy = gpu.grid()
        assert (sx, sy) == synth.CONFIGS[cfg]["grid"] == (ref.ndt.size_x, ref.ndt.size_y)
        assert (ox, oy) == ref.ndt.origin and cs == 0.25
        assert np.array_equal(cells, exp), mode
        got[mode] = gpu
    # scoring against the device-built grid equals scoring against the host-built one
    _, pts, _ = synth.query_scan(cfg)
    poses = synth.particles(3, 2000)
    poses[:, :2] *= synth.CONFIGS[cfg]["world"][0] / 23.0
    assert np.array_equal(got["device"].scorePoses(pts, poses), got["host"].scorePoses(pts, poses))


def test_device_ndt_build_edge_cases():
    rng = np.random.default_rng(5)
    w = synth.world_of(1)
    # rotated scan poses, an empty scan, a scan entirely outside the extent, 0.1 m cells
    scans = []
    for k in range(12):
        pose = (rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-3.1, 3.1))
        scans.append((pose, synth.scan(w, pose, 7000 + k, n_beams=300 + 17 * k)))
    scans.insert(3, ((0.2, 0.1, 1.0), np.zeros((0, 2))))
    scans.append(((0.0, 0.0, 0.0), np.full((40, 2), 500.0)))
    for res in (0.25, 0.1, 1.0):
        ref = O.ScanMatcherNDT()
        ref.initialize(ndt_resolution=res, range_max=4.75)
        ref.addScans(scans)
        gpu = ScanMatcherNDT(0)
        gpu.initialize("t", ndt_resolution=res, range_max=4.75)
        gpu.set_build_mode("device")
        gpu.addScans(scans)
        cells, sx, sy, cs, ox, oy = gpu.grid()
        assert (sx, sy, ox, oy) == (ref.ndt.size_x, ref.ndt.size_y) + ref.ndt.origin
        assert np.array_equal(cells, ref.ndt.cells6()), res
    # re-building with a different map replaces the grid; reset clears it
    gpu.addScans(scans[:2])
    ref.addScans(scans[:2])
    assert np.array_equal(gpu.grid()[0], ref.ndt.cells6())
    gpu.reset()
    assert not gpu.has_ndt()


def test_cfg3_particles_golden_through_device_layer():
    g = np.load(os.path.join(GOLDEN, "cfg3_poses256.npz"))
    ncell = int(g["size_x"]) * int(g["size_y"])
    cells = np.zeros((ncell, 6))
    cells[g["occupied_index"]] = g["occupied_cells6"]
    L = _capi.lib()
    h = C.c_void_p()
    assert L.ndt2d_create(C.byref(h), 0) == 0
    try:
        assert L.ndt2d_set_grid(h, _capi.dptr(cells), int(g["size_x"]), int(g["size_y"]),
                                float(g["cell_size"]), float(g["origin"][0]),
                                float(g["origin"][1])) == 0
        pts = np.ascontiguousarray(g["points"])
        assert L.ndt2d_set_beams(h, _capi.dptr(pts), len(pts)) == 0
        parts = np.ascontiguousarray(g["particles"])
        w = np.zeros(len(parts))
        st = np.zeros(8)
        assert L.ndt2d_score_poses(h, _capi.dptr(parts), len(parts), _capi.dptr(w),
                                   _capi.dptr(st)) == 0, L.ndt2d_last_error(h)
        assert np.max(np.abs(w - g["weights_raw"])) < TOL_TIGHT
        assert st[0] == pytest.approx(g["weights_raw"].sum(), rel=1e-12)
        assert st[1] == pytest.approx((g["weights_raw"] * parts[:, 0]).sum(), rel=1e-10)
        assert b"compact" in L.ndt2d_last_variant(h)
        # the dense kernels (records gathered per lane; 201 x 201 grid does not fit in LDS)
        assert L.ndt2d_set_variant(h, b"dense") == 0
        w2 = np.zeros(len(parts))
        assert L.ndt2d_score_poses(h, _capi.dptr(parts), len(parts), _capi.dptr(w2), None) == 0
        assert b"global-grid" in L.ndt2d_last_variant(h)
        assert np.max(np.abs(w2 - g["weights_raw"])) < TOL_TIGHT
        assert np.max(np.abs(w2 - w)) < 1e-13
    finally:
        L.ndt2d_destroy(h)


def test_cfg3_full_particle_filter_measure():
    """BASELINE.json configs[2]: 100,000 particles x 720 beams, 201 x 201 NDT."""
    gpu, ref, _, _, pts = _pair(3)
    parts = synth.particles(3)
    parts[:5000, 0] = 1.0 + (parts[:5000, 0] / 23.0) * 0.2   # a cluster near the true pose
    parts[:5000, 1] = 0.5 + (parts[:5000, 1] / 23.0) * 0.2
    parts[:5000, 2] = 0.3 + (parts[:5000, 2] / math.pi) * 0.05
    w_raw = gpu.scorePoses(pts, parts)
    w_ref = O.pf_measure(ref, parts, pts, omp_threads=os.cpu_count())
    assert np.max(np.abs(w_raw - w_ref)) < TOL_CONTRACT
    assert np.max(np.abs(w_raw - w_ref)) < TOL_TIGHT
    cov_prev = np.zeros((3, 3))
    cov_prev[2, 2] = 0.125                                   # cov_(2,2) accumulates (:216)
    w, mean, cov = pf_measure(gpu, parts, pts, cov_prev=cov_prev)
    w_n, mean_ref, cov_ref = O.pf_update_statistics(parts, w_ref, cov_prev=cov_prev)
    assert np.allclose(w, w_n, rtol=1e-9, atol=1e-18)
    assert abs(w.sum() - 1.0) < 1e-12
    assert np.allclose(mean, mean_ref, rtol=1e-9, atol=1e-12)
    assert np.allclose(cov, cov_ref, rtol=1e-8, atol=1e-12)
    assert np.argmax(w) < 5000


def test_sharded_particle_statistics_on_device():
    """The multi-GPU particle flow on one GPU: two shards score their particles, the
    [2, 8] moment table is summed (the all-reduce), each shard finalises with the
    global sums; weights, mean and covariance equal the unsharded result."""
    import torch
    gpu, ref, _, _, pts = _pair(3)
    parts = synth.particles(3, 50000)
    parts[:3000, 0] = 1.0 + (parts[:3000, 0] / 23.0) * 0.2
    parts[:3000, 1] = 0.5 + (parts[:3000, 1] / 23.0) * 0.2
    parts[:3000, 2] = 0.3 + (parts[:3000, 2] / math.pi) * 0.05
    w_full, mean_full, cov_full = pf_measure(gpu, parts, pts)
    w_ref = O.pf_measure(ref, parts, pts, omp_threads=os.cpu_count())
    w_n, mean_ref, cov_ref = O.pf_update_statistics(parts, w_ref)
    assert np.allclose(w_full, w_n, rtol=1e-9, atol=1e-18)
    assert np.allclose(mean_full, mean_ref, rtol=1e-9, atol=1e-12)
    assert np.allclose(cov_full, cov_ref, rtol=1e-8, atol=1e-12)

    gpu.prepare_beams(pts)
    dev = torch.device("cuda:0")
    shards = [parts[:20001], parts[20001:]]
    d_parts = [torch.from_numpy(np.ascontiguousarray(s)).to(dev) for s in shards]
    d_w = [torch.zeros(len(s), dtype=torch.float64, device=dev) for s in shards]
    table = torch.zeros((2, 8), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for r in range(2):
        gpu.score_poses_launch(d_parts[r].data_ptr(), len(shards[r]), d_w[r].data_ptr(),
                               table[r].data_ptr())
    gpu.synchronize()
    total = table.sum(dim=0).contiguous()          # what the all-reduce leaves on every rank
    outs = torch.zeros((2, 8), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for r in range(2):
        gpu.pf_finalize_launch(d_parts[r].data_ptr(), len(shards[r]), d_w[r].data_ptr(),
                               total.data_ptr(), outs[r].data_ptr())
    gpu.synchronize()
    w = np.concatenate([t.cpu().numpy() for t in d_w])
    o = outs.cpu().numpy()
    assert np.allclose(w, w_full, rtol=1e-12, atol=0)
    assert np.allclose(o[0, :7], o[1, :7], rtol=0, atol=0)               # same global sums
    assert np.allclose(o[0, 1:4], mean_full, rtol=1e-12, atol=1e-15)
    assert np.allclose([o[0, 4], o[0, 5], o[0, 6]],
                       [cov_full[0, 0], cov_full[0, 1], cov_full[1, 1]], rtol=1e-9, atol=1e-15)
    assert o[:, 7].sum() == pytest.approx(cov_full[2, 2], rel=1e-10)


def test_cfg5_sized_particle_set():
    """BASELINE.json configs[4] on one GPU: 1,000,000 particles, 801 x 801 NDT."""
    gpu, ref, _, _, pts = _pair(5)
    assert gpu.grid()[1:3] == (801, 801)
    parts = synth.particles(5)
    w = gpu.scorePoses(pts, parts)
    w_ref = O.pf_measure(ref, parts, pts, omp_threads=os.cpu_count())
    assert np.max(np.abs(w - w_ref)) < TOL_TIGHT
    # sharding the particle set leaves every weight bit-identical, whatever the
    # launch geometry each shard size selects
    halves = np.concatenate([gpu.scorePoses(pts, parts[:400001]), gpu.scorePoses(pts, parts[400001:])])
    assert np.array_equal(halves, w)
    small = np.concatenate([gpu.scorePoses(pts, parts[i:i + 7777]) for i in range(0, 70000, 7777)])
    assert np.array_equal(small, w[:len(small)])
    # the FP32 screening of phase A leaves every weight bit-identical
    try:
        gpu.set_variant("compact-exact")
        assert np.array_equal(gpu.scorePoses(pts, parts), w)
    finally:
        gpu.set_variant("auto")


@pytest.mark.parametrize("cfg", [1, 3])
def test_particle_screening_is_bit_exact(cfg):
    """Poses inside, on the rim of and far outside the grid, beams ending on cell
    boundaries: the screened kernel ("auto") and the exact phase A ("compact-exact")
    give the same bits, and both match the oracle."""
    gpu, ref, _, _, pts = _pair(cfg, laser_max_beams=2000)
    _, sx, sy, cell, ox, oy = gpu.grid()
    rng = np.random.default_rng(cfg)
    n = 20000
    parts = np.stack([rng.uniform(ox, ox + sx * cell, n), rng.uniform(oy, oy + sy * cell, n),
                      rng.uniform(-np.pi, np.pi, n)], axis=1)
    # a quarter of the poses around / beyond the edge of the grid
    edge = rng.integers(0, n, n // 4)
    parts[edge, 0] = ox + rng.choice([-1.0, 0.0, 1.0], len(edge)) * sx * cell * rng.uniform(0.9, 1.6, len(edge))
    parts[edge[: len(edge) // 2], 1] = oy + sy * cell * rng.uniform(-0.5, 1.5, len(edge) // 2)
    parts[rng.integers(0, n, 50)] = [1e9, -1e9, 0.3]
    # beams that end exactly on cell boundaries for axis-aligned poses on the lattice
    parts[:64, 0] = ox + cell * rng.integers(1, sx - 1, 64)
    parts[:64, 1] = oy + cell * rng.integers(1, sy - 1, 64)
    parts[:64, 2] = rng.choice([0.0, np.pi / 2, np.pi, -np.pi / 2], 64)
    beams = np.concatenate([pts, cell * rng.integers(-12, 12, size=(80, 2)).astype(np.float64)])
    want = O.pf_measure(ref, parts, beams, omp_threads=os.cpu_count())
    try:
        got = gpu.scorePoses(beams, parts)
        assert "compact" in gpu.last_variant()
        gpu.set_variant("compact-exact")
        exact = gpu.scorePoses(beams, parts)
    finally:
        gpu.set_variant("auto")
    assert np.array_equal(got, exact)
    assert np.max(np.abs(got - want)) < TOL_TIGHT
    assert (got < 0).sum() > n // 2


def test_no_ndt_returns_zero_and_leaves_outputs():
    # reference src/scan_matcher_ndt.cpp:80,159
    gpu = ScanMatcherNDT(0)
    gpu.initialize("t", range_max=5.0)
    r = gpu.matchScan((0, 0, 0), [(1.0, 1.0)], pose=(7.0, 8.0, 9.0))
    assert r["score"] == 0.0 and np.array_equal(r["pose"], [7.0, 8.0, 9.0])
    assert r["covariance"] is None
    assert gpu.scorePoints([(1.0, 1.0)], (0, 0, 0)) == 0.0
    assert gpu.scoreScan((0, 0, 0), [(1.0, 1.0)]) == 0.0
    assert np.array_equal(gpu.scorePoses([(1.0, 1.0)], [(0, 0, 0), (1, 1, 1)]), [0.0, 0.0])
    gpu.addScans([((0, 0, 0), [(1.1, 1.1), (1.11, 1.1), (1.1, 1.11), (1.09, 1.1), (1.1, 1.09)])])
    assert gpu.has_ndt() and gpu.scorePoints([(1.1, 1.1)], (0, 0, 0)) < 0.0
    gpu.reset()
    assert not gpu.has_ndt() and gpu.scorePoints([(1.1, 1.1)], (0, 0, 0)) == 0.0
    # device layer: compute before set_grid is an error, not a silent zero
    L = _capi.lib()
    h = C.c_void_p()
    assert L.ndt2d_create(C.byref(h), 0) == 0
    assert L.ndt2d_match_launch(h, 0, 1, None, None) == _capi.ERR_NO_GRID
    L.ndt2d_destroy(h)


def test_all_out_of_grid_scan(cfg1):
    gpu, ref, _, guess, pts = cfg1
    far = pts + 100.0
    got = gpu.matchScan(guess, far, pose=(1.0, 2.0, 3.0), want_scores=True)
    exp = ref.matchScan(guess, far, pose=(1.0, 2.0, 3.0))
    assert got["score"] == exp["score"] == 0.0
    assert got["best_index"] == _capi.NO_INDEX
    assert np.array_equal(got["pose"], [1.0, 2.0, 3.0])
    assert np.isnan(got["covariance"]).all() and np.isnan(exp["covariance"]).all()
    assert not got["scores"].any()


def test_cell_population_and_grid_edges():
    # n < 5 cells score exactly 0 (ndt_model.cpp:107); a point exactly on origin_x is
    # inside, one ulp below is outside; the last column [size-1] is inside (:203-218)
    pts4 = [(0.5, 0.5), (0.6, 0.5), (0.5, 0.6), (0.4, 0.45)]
    edge = [(-5.0, -5.0), (-4.99, -4.98), (-4.97, -4.99), (-4.98, -4.96), (-4.96, -4.97)]
    top = [(5.9, 5.9), (5.95, 5.9), (5.9, 5.95), (5.85, 5.92), (5.93, 5.86)]
    for cells_pts in (pts4, pts4 + [(0.55, 0.52)], edge, top):
        gpu = ScanMatcherNDT(0)
        gpu.initialize("t", ndt_resolution=1.0, range_max=5.0, laser_max_beams=10)
        gpu.addScans([((0.0, 0.0, 0.0), cells_pts)])
        ref = O.ScanMatcherNDT()
        ref.initialize(ndt_resolution=1.0, range_max=5.0, laser_max_beams=10)
        ref.addScans([((0.0, 0.0, 0.0), cells_pts)])
        assert gpu.grid()[1:3] == (11, 11)
        probes = [cells_pts[0], (-5.0, -5.0), (np.nextafter(-5.0, -6.0), -5.0),
                  (-5.0, np.nextafter(-5.0, -6.0)), (5.999, 5.999), (6.0, 5.5), (5.5, 6.0),
                  (np.nextafter(6.0, 0.0), 5.9)]
        for q in probes:
            got = gpu.scorePoints([q], (0.0, 0.0, 0.0))
            exp = ref.scorePoints([q], (0.0, 0.0, 0.0))
            assert abs(got - exp) < 1e-12, (cells_pts[0], q, got, exp)
    ref4 = O.ScanMatcherNDT()
    ref4.initialize(ndt_resolution=1.0, range_max=5.0)
    ref4.addScans([((0.0, 0.0, 0.0), pts4)])
    assert ref4.scorePoints([(0.5, 0.5)], (0, 0, 0)) == 0.0


def test_degenerate_cell_propagates_nan_like_the_reference(cfg1):
    """Five identical points give a zero covariance; Matrix2d::inverse() of it is
    inf/NaN (reference src/ndt_model.cpp:99) and every point scored against that cell
    is NaN.  The NaN must reach the same candidates, leave the argmin to the others
    and poison the covariance exactly as in the reference."""
    _, _, scans, guess, pts = cfg1
    bad = ((0.0, 0.0, 0.0), np.tile([[3.625, 0.625]], (6, 1)))  # binary-exact, in an otherwise empty cell inside the east wall
    params = synth.matcher_params(1, search_linear_size=0.3, search_angular_size=0.06)
    ref = O.ScanMatcherNDT()
    ref.initialize(**params)
    ref.addScans([bad])
    assert np.isnan(ref.ndt.cells6()).any()
    gpu = ScanMatcherNDT(0)
    gpu.initialize("t", **params)
    for mode in ("host", "device"):
        gpu.set_build_mode(mode)
        gpu.addScans([bad])
        assert np.array_equal(gpu.grid()[0], ref.ndt.cells6(), equal_nan=True)
    poses = np.array([(0.0, 0.0, 0.0), (0.05, 0.02, 0.0), (2.0, 2.0, 1.0)])
    probe = np.array([(3.6, 0.6), (1.0, 1.0), (3.7, 0.7)])
    a, b = gpu.scorePoses(probe, poses), O.pf_measure(ref, poses, probe)
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.isnan(a).any() and not np.isnan(a).all()
    assert np.allclose(a, b, rtol=0, atol=1e-12, equal_nan=True)
    # the same cell inside a healthy map
    ref.addScans(scans + [bad])
    exp = ref.matchScan(guess, pts, want_scores=True)
    assert np.isnan(exp["scores"]).any() and not np.isnan(exp["scores"]).all()
    for variant in ("lane", "wave"):
        gpu.set_variant(variant)
        gpu.addScans(scans + [bad])
        got = gpu.matchScan(guess, pts, want_scores=True)
        assert np.array_equal(np.isnan(got["scores"]), np.isnan(exp["scores"])), variant
        assert np.allclose(got["scores"], exp["scores"], rtol=0, atol=1e-9, equal_nan=True)
        assert got["best_index"] == exp["best_index"] and got["score"] == pytest.approx(exp["score"], abs=1e-12)
        assert np.array_equal(got["pose"], exp["pose"])
        assert np.isnan(got["covariance"]).all() and np.isnan(exp["covariance"]).all()
    gpu.set_variant("auto")


def test_reference_known_answer_through_the_gpu():
    # reference test/ndt_model_tests.cpp:191-230: likelihood((3.5, 3.5)) = 0.7659 +- 1e-3
    v = json.load(open(os.path.join(GOLDEN, "reference_ndt_model_tests.json")))["test_ndt"]
    gpu = ScanMatcherNDT(0)
    gpu.initialize("t", ndt_resolution=1.0, range_max=5.0)
    gpu.addScans([(v["scan_pose"], v["scan_points"])])
    got = -gpu.scorePoints(v["query"], (0.0, 0.0, 0.0))
    assert got == pytest.approx(v["likelihood"], abs=v["tol"])
    assert got == pytest.approx(0.76592833836492369, rel=1e-12)   # SURVEY.md section 4 digits


def test_concurrent_instances_keep_their_results_apart():
    """Three matcher instances busy at once on three threads -- the node's own small search
    and a 500-particle measure (results through host-coherent memory behind sequence
    flags, reductions finished by the last block of a launch) next to a large-lattice
    search and a batched particle set: every call returns the bits its instance returns on
    its own."""
    jobs = {}

    def setup(name, cfg, **override):
        m = ScanMatcherNDT(0)
        m.initialize(name, **synth.matcher_params(cfg, **override))
        m.addScans(synth.map_scans(cfg))
        guess, pts, _ = synth.query_scan(cfg)
        return m, guess, pts

    m_small, g_small, p_small = setup("local", 1, laser_max_beams=100, search_linear_size=0.05,
                                      search_linear_resolution=0.005, search_angular_size=0.1,
                                      search_angular_resolution=0.0025)
    m_large, g_large, p_large = setup("global", 2)
    m_pf, g_pf, p_pf = setup("pf", 3)
    parts = synth.particles(3, 3000)
    parts[:1500, :2] = g_pf[:2] + parts[:1500, :2] / 23.0 * 0.4

    def small():
        r = m_small.matchScan(g_small, p_small)
        return r["pose"].tobytes() + np.float64(r["score"]).tobytes() + r["covariance"].tobytes()

    def large():
        r = m_large.matchScan(g_large, p_large)
        return r["pose"].tobytes() + np.float64(r["score"]).tobytes() + r["covariance"].tobytes()

    def filt():
        w, mean, cov = pf_measure(m_pf, parts[:500], p_pf)
        big = m_pf.scorePoses(p_pf, parts)
        return w.tobytes() + mean.tobytes() + cov.tobytes() + big.tobytes()

    jobs = {"small": (small, 300), "large": (large, 40), "filter": (filt, 150)}
    alone = {k: f() for k, (f, _) in jobs.items()}
    assert "small-lattice" in m_small.last_variant() and "compact-records" in m_large.last_variant()
    errors = []

    def work(name):
        f, n = jobs[name]
        try:
            for i in range(n):
                if f() != alone[name]:
                    errors.append((name, i))
                    return
        except Exception as e:  # pragma: no cover
            errors.append((name, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in jobs]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors


def test_empty_scan_gives_nan(cfg1):
    gpu, ref, _, guess, _ = cfg1
    empty = np.zeros((0, 2))
    assert math.isnan(gpu.scorePoints(empty, (0, 0, 0))) and math.isnan(ref.scorePoints(empty, (0, 0, 0)))
    r = gpu.matchScan(guess, empty, pose=(1.0, 1.0, 1.0))
    assert math.isnan(r["score"]) and np.isnan(r["covariance"]).all()
    assert np.array_equal(r["pose"], [1.0, 1.0, 1.0])


def test_two_instances_on_two_threads(cfg1):
    """The node runs a local and a global matcher concurrently on two threads
    (reference src/ndt_mapper.cpp:141-142,508-515,634-643)."""
    _, ref, scans, guess, pts = cfg1
    exp = ref.matchScan(guess, pts)
    results, errors = {}, []

    def work(name):
        try:
            m = ScanMatcherNDT(0)
            m.initialize(name, **synth.matcher_params(1))
            for _ in range(3):
                m.reset()
                m.addScans(scans)
                results[name] = m.matchScan(guess, pts)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(n,)) for n in ("local", "global")]
    [t.start() for t in threads]
    [t.join() for t in threads]
    assert not errors, errors
    for r in results.values():
        assert np.array_equal(r["pose"], exp["pose"]) and abs(r["score"] - exp["score"]) < TOL_TIGHT


@pytest.mark.parametrize("sx,sy,n_listed", [
    (1, 1, 0), (1, 1, 1), (33, 31, 0), (33, 31, 256), (33, 31, 257), (257, 129, 5000),
    (1023, 1025, 3000), (3000, 2000, 70000),
])
def test_list_install_shapes(sx, sy, n_listed):
    """The install kernel of ndt2d_set_grid_sparse (ndt2d_build.hip: the listed cells 256 per
    step, all others 1024 per step behind the host's occupancy words, grid-stride beyond 2048
    blocks): an empty list, a single cell, lists that end on and just past a step, grids whose
    cell count is not a multiple of anything, cells that cannot score (n < 5) in the list.
    Against the dense install of the same cells: the records back (ndt2d_get_grid), a search's
    candidate scores and the particle weights, all bitwise."""
    rng = np.random.default_rng(sx * 7919 + sy * 31 + n_listed)
    ncell = sx * sy
    listed = rng.choice(ncell, size=n_listed, replace=False).astype(np.uint32)
    if n_listed > 2:
        listed[0], listed[1] = 0, ncell - 1                       # the corners of the grid
        listed = np.unique(listed).astype(np.uint32)
        rng.shuffle(listed)
    n = len(listed)
    res = 0.25
    cx = (listed % sx + 0.5) * res
    cy = (listed // sx + 0.5) * res
    rec = np.zeros((n, 6))
    rec[:, 0] = cx + rng.uniform(-0.1, 0.1, n)
    rec[:, 1] = cy + rng.uniform(-0.1, 0.1, n)
    a, b, t = rng.uniform(20, 400, n), rng.uniform(20, 400, n), rng.uniform(0, math.pi, n)
    rec[:, 2] = a * np.cos(t) ** 2 + b * np.sin(t) ** 2
    rec[:, 3] = (a - b) * np.cos(t) * np.sin(t)
    rec[:, 4] = a * np.sin(t) ** 2 + b * np.cos(t) ** 2
    rec[:, 5] = rng.integers(1, 40, n)                            # some below 5: listed, cannot score
    dense = np.zeros((ncell, 6))
    dense[listed] = rec
    # beams around a listed cell that can score, if there is one
    scoring = np.flatnonzero(rec[:, 5] >= 5)
    centre = rec[scoring[0], :2] if len(scoring) else np.array([0.1, 0.1])
    pts = rng.uniform(-1.0, 1.0, (64, 2))
    pose = np.array([centre[0], centre[1], 0.3])
    dth = np.linspace(-0.05, 0.05, 5)
    dlin = np.linspace(-0.25, 0.25, 11)
    cos_t, sin_t = np.cos(pose[2] + dth), np.sin(pose[2] + dth)
    poses = np.column_stack([rng.uniform(0, sx * res, 3000), rng.uniform(0, sy * res, 3000),
                             rng.uniform(-3, 3, 3000)])
    if len(scoring):
        poses[:1000, :2] = rec[rng.choice(scoring, 1000), :2] + rng.uniform(-0.5, 0.5, (1000, 2))
    L = _capi.lib()
    out = {}
    for how in ("dense", "sparse"):
        h = C.c_void_p()
        assert L.ndt2d_create(C.byref(h), 0) == 0
        try:
            for _ in range(2):        # (twice: the second install finds the buffers of the first)
                if how == "dense":
                    assert L.ndt2d_set_grid(h, _capi.dptr(dense), sx, sy, res, 0.0, 0.0) == 0
                else:
                    idx = listed if n else np.zeros(1, np.uint32)
                    assert L.ndt2d_set_grid_sparse(h, idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                   _capi.dptr(rec if n else np.zeros((1, 6))), n,
                                                   sx, sy, res, 0.0, 0.0) == 0
            assert L.ndt2d_set_beams(h, _capi.dptr(pts), len(pts)) == 0
            assert L.ndt2d_set_search(h, pose[0], pose[1], _capi.dptr(dth), _capi.dptr(cos_t),
                                      _capi.dptr(sin_t), len(dth), _capi.dptr(dlin), len(dlin)) == 0
            got = {}
            for variant in (b"auto", b"lane", b"lane-noskip"):
                assert L.ndt2d_set_variant(h, variant) == 0
                r = _capi.MatchResult()
                sc = np.zeros(len(dth) * len(dlin) ** 2)
                assert L.ndt2d_match(h, 0, len(dth), _capi.dptr(sc), C.byref(r)) == 0, variant
                got[variant] = (sc, r.best_index, r.best_score)
            assert L.ndt2d_set_variant(h, b"auto") == 0
            w = np.zeros(len(poses))
            assert L.ndt2d_score_poses(h, _capi.dptr(poses), len(poses), _capi.dptr(w), None) == 0, (how, L.ndt2d_last_error(h))
            back = np.full_like(dense, -1.0)
            assert L.ndt2d_get_grid(h, _capi.dptr(back), len(back), None, None, None, None, None) == 0
            out[how] = (got, w, back)
        finally:
            L.ndt2d_destroy(h)
    assert np.array_equal(out["sparse"][2], dense) and np.array_equal(out["dense"][2], dense)
    for variant, (sc, bi, bs) in out["dense"][0].items():
        sc2, bi2, bs2 = out["sparse"][0][variant]
        assert np.array_equal(sc, sc2), variant
        assert (bi, bs) == (bi2, bs2), variant
    assert np.array_equal(out["sparse"][0][b"lane"][0], out["sparse"][0][b"lane-noskip"][0])
    assert np.array_equal(out["dense"][1], out["sparse"][1])
    if len(scoring) > 10:
        assert (out["sparse"][0][b"auto"][0] < 0).any() and (out["sparse"][1] < 0).any()


@pytest.mark.parametrize("cfg,override", [
    (1, dict(search_angular_size=0.1, search_angular_resolution=0.0025, search_linear_size=0.05,
             search_linear_resolution=0.005, laser_max_beams=100)),
    (5, dict(search_angular_size=0.1, search_angular_resolution=0.0025, search_linear_size=0.05,
             search_linear_resolution=0.005, laser_max_beams=100)),
])
def test_map_bytes_of_an_install_arrive_whichever_call_comes_first(cfg, override):
    """After addScans the map bytes around the listed cells are still to do (only the
    small-lattice search reads them): the next few-pose launch -- the mapper's scoreScan,
    reference src/ndt_mapper.cpp:514 -- carries the job in spare blocks, else the search runs it
    first.  Every order of calls gives the same search, bit for bit, and the oracle's."""
    scans = synth.map_scans(cfg)
    params = synth.matcher_params(cfg, **override)
    ref = O.ScanMatcherNDT()
    ref.initialize(**params)
    ref.addScans(scans)
    guess, pts, _ = synth.query_scan(cfg)
    exp = ref.matchScan(guess, pts, want_scores=True)
    poses = np.tile(guess, (300, 1)) + np.random.default_rng(5).uniform(-0.3, 0.3, (300, 3))
    results = {}
    for order in ("match", "scoreScan", "scorePoints", "measure", "twice"):
        gpu = ScanMatcherNDT(0)
        gpu.initialize("test", **params)
        gpu.addScans(scans)
        if order == "scoreScan":
            assert abs(gpu.scoreScan(guess, pts) - ref.scoreScan(guess, pts)) < TOL_TIGHT
        elif order == "scorePoints":
            gpu.scorePoints(pts, guess)
        elif order == "measure":
            w = gpu.scorePoses(pts, poses)           # 300 poses: the block-per-pose launch
            assert "block-per-pose" in gpu.last_variant()
            assert np.all(np.isfinite(w))
        elif order == "twice":
            gpu.reset()
            gpu.addScans(scans)                      # an install whose job was never run, then another
            gpu.scoreScan(guess, pts)
        got = gpu.matchScan(guess, pts, want_scores=True)
        assert "small-lattice" in gpu.last_variant(), gpu.last_variant()
        _check_match(got, exp, params["laser_max_beams"])
        results[order] = got
    first = results["match"]
    for order, got in results.items():
        assert np.array_equal(got["scores"], first["scores"]), order
        assert got["best_index"] == first["best_index"] and got["score"] == first["score"], order
        assert np.array_equal(got["covariance"], first["covariance"], equal_nan=True), order


def test_list_install_rejects_a_cell_listed_twice():
    """ndt2d_set_grid_sparse: two records for one cell that can score would race in the
    install kernel -- refused (NDT2D_ERR_INVALID), and the context is left without a grid."""
    L = _capi.lib()
    h = C.c_void_p()
    assert L.ndt2d_create(C.byref(h), 0) == 0
    try:
        idx = np.array([5, 9, 5], np.uint32)
        rec = np.tile(np.array([0.3, 0.3, 100.0, 0.0, 100.0, 9.0]), (3, 1))
        rc = L.ndt2d_set_grid_sparse(h, idx.ctypes.data_as(C.POINTER(C.c_uint32)), _capi.dptr(rec), 3,
                                     8, 8, 0.25, 0.0, 0.0)
        assert rc == 1 and b"listed twice" in L.ndt2d_last_error(h)
        assert L.ndt2d_has_grid(h) == 0
        idx[2] = 6
        assert L.ndt2d_set_grid_sparse(h, idx.ctypes.data_as(C.POINTER(C.c_uint32)), _capi.dptr(rec), 3,
                                       8, 8, 0.25, 0.0, 0.0) == 0
        assert L.ndt2d_has_grid(h) == 1
    finally:
        L.ndt2d_destroy(h)


@pytest.mark.parametrize("max_beams", [100, 720])
def test_score_scan_launches_the_search_of_its_scan_ahead(max_beams):
    """The mapper's pair scoreScan(scan), matchScan(scan, ...) (reference src/ndt_mapper.cpp:
    514-515): once a matcher has seen it, scoreScan queues the search behind its own kernel and
    matchScan collects it.  Same bits as a matcher that never launches ahead, and the oracle's;
    calls that break the pattern (another pose, another scan, per-candidate scores wanted, a
    scorePoints or a reset in between) drop the search and still get their own results."""
    over = dict(search_angular_size=0.1, search_angular_resolution=0.0025, search_linear_size=0.05,
                search_linear_resolution=0.005, laser_max_beams=max_beams)
    # (720 beams do not travel as kernel arguments: one staged upload, then the same two launches)
    scans = synth.map_scans(1)
    params = synth.matcher_params(1, **over)
    ref = O.ScanMatcherNDT()
    ref.initialize(**params)
    ref.addScans(scans)
    guess, pts, _ = synth.query_scan(1)
    other = guess + np.array([0.02, 0.01, -0.004])
    pts2 = pts[::-1].copy()
    exp = {k: ref.matchScan(g, p, want_scores=True) for k, (g, p) in
           dict(a=(guess, pts), b=(other, pts), c=(guess, pts2)).items()}
    exp_score = ref.scoreScan(guess, pts)

    def blob(r):
        return (r["score"], r["best_index"], r["pose"].tobytes(), r["covariance"].tobytes())

    plain = ScanMatcherNDT(0)
    plain.initialize("plain", **params)
    plain.set_search_ahead(False)
    plain.addScans(scans)
    want = {k: plain.matchScan(g, p) for k, (g, p) in dict(a=(guess, pts), b=(other, pts), c=(guess, pts2)).items()}
    for k in want:
        _check_match(want[k], exp[k], max_beams)
    assert plain.search_ahead_stats() == (0, 0)

    gpu = ScanMatcherNDT(0)
    gpu.initialize("ahead", **params)
    # the single poses on the device, so that scoreScan's kernel carries the search behind it and a
    # scorePoints in between breaks the pattern (scored on the host it would leave the pending
    # search alone: tests/test_gpu_single_pose_host.py covers that form of the cycle)
    gpu.set_single_pose_path("device")
    for cycle in range(4):
        gpu.reset()
        gpu.addScans(scans)
        assert abs(gpu.scoreScan(guess, pts) - exp_score) < TOL_TIGHT
        assert blob(gpu.matchScan(guess, pts)) == blob(want["a"]), cycle
    assert gpu.search_ahead_stats() == (3, 3)          # the first cycle shows the pair, the others use it
    # the pattern breaks: every call still gets its own result, the search launched ahead is dropped
    launched = 3
    for breaker in ("pose", "scan", "scores", "scorePoints", "reset", "measure", "grid"):
        gpu.scoreScan(guess, pts)                       # pair seen: launches ahead
        launched += 1
        assert gpu.search_ahead_stats() == (launched, 3), breaker
        if breaker == "pose":
            assert blob(gpu.matchScan(other, pts)) == blob(want["b"])
        elif breaker == "scan":
            assert blob(gpu.matchScan(guess, pts2)) == blob(want["c"])
        elif breaker == "scores":
            got = gpu.matchScan(guess, pts, want_scores=True)
            _check_match(got, exp["a"], max_beams)
            assert blob(got) == blob(want["a"])
        elif breaker == "scorePoints":
            assert abs(gpu.scorePoints(pts, guess) - exp_score) < TOL_TIGHT
            assert blob(gpu.matchScan(guess, pts)) == blob(want["a"])
        elif breaker == "reset":
            gpu.reset()
            gpu.addScans(scans)
            assert blob(gpu.matchScan(guess, pts)) == blob(want["a"])
        elif breaker == "measure":
            w = gpu.scorePoses(pts, np.tile(guess, (40, 1)))
            assert np.all(np.abs(w - exp_score) < TOL_TIGHT)
            assert blob(gpu.matchScan(guess, pts)) == blob(want["a"])
        else:
            assert np.array_equal(gpu.grid()[0], ref.ndt.cells6(), equal_nan=True)
            assert blob(gpu.matchScan(guess, pts)) == blob(want["a"])
        # dropped: the next scoreScan does not launch ahead until the pair has been seen again
        gpu.scoreScan(guess, pts)
        assert gpu.search_ahead_stats() == (launched, 3), breaker
        assert blob(gpu.matchScan(guess, pts)) == blob(want["a"])     # ... which this is
    gpu.scoreScan(guess, pts)
    assert blob(gpu.matchScan(guess, pts)) == blob(want["a"])
    assert gpu.search_ahead_stats() == (launched + 1, 4)


def test_score_launch_and_fetch_in_two_steps(cfg1):
    """ndt2d_score_poses_beams_launch / ndt2d_score_fetch (the kernel-argument launch of
    ndt2d_score_poses_beams without the wait): the scores of the one-call form bit for bit;
    what is not a kernel-argument launch is refused, a fetch without a launch is a state
    error, results nobody fetched are dropped by the next launch."""
    gpu, ref, scans, guess, pts = cfg1
    cells, sx, sy, _, ox, oy = gpu.grid()
    beams = np.ascontiguousarray(pts[:100])
    poses = np.stack([guess + np.array([0.01 * k, -0.02 * k, 0.003 * k]) for k in range(8)])
    L = _capi.lib()
    h = C.c_void_p()
    assert L.ndt2d_create(C.byref(h), 0) == 0
    try:
        assert L.ndt2d_set_grid(h, _capi.dptr(cells), sx, sy, 0.25, ox, oy) == 0
        want = np.zeros(8)
        assert L.ndt2d_score_poses_beams(h, _capi.dptr(beams), 100, _capi.dptr(poses), 8, _capi.dptr(want)) == 0
        got = np.zeros(8)
        assert L.ndt2d_score_fetch(h, _capi.dptr(got)) == 5                      # NDT2D_ERR_STATE: nothing launched
        assert L.ndt2d_score_poses_beams_launch(h, _capi.dptr(beams), 100, _capi.dptr(poses), 8) == 0
        assert L.ndt2d_score_fetch(h, _capi.dptr(got)) == 0
        assert np.array_equal(got, want)
        # the beams the context now holds (NULL), fewer poses; a launch nobody fetches, then another
        assert L.ndt2d_score_poses_beams_launch(h, None, 0, _capi.dptr(poses), 8) == 0
        assert L.ndt2d_score_poses_beams_launch(h, None, 0, _capi.dptr(poses[3:]), 2) == 0
        got2 = np.zeros(2)
        assert L.ndt2d_score_fetch(h, _capi.dptr(got2)) == 0
        assert np.array_equal(got2, want[3:5])
        assert L.ndt2d_score_fetch(h, _capi.dptr(got2)) == 5
        # not kernel-argument launches: more than 8 poses, more than 208 beams
        many = np.tile(poses, (2, 1))
        assert L.ndt2d_score_poses_beams_launch(h, _capi.dptr(beams), 100, _capi.dptr(many), 16) == 5
        wide = np.ascontiguousarray(pts[:300])
        assert L.ndt2d_score_poses_beams_launch(h, _capi.dptr(wide), 300, _capi.dptr(poses), 8) == 5
        # ... and the context still scores with the beams it held
        again = np.zeros(8)
        assert L.ndt2d_score_poses(h, _capi.dptr(poses), 8, _capi.dptr(again), None) == 0
        assert np.array_equal(again, want)
        # (ref was initialised with laser_max_beams = 720: 100 points are taken whole)
        for k in range(8):
            assert abs(want[k] - ref.scorePoints(beams, poses[k])) < TOL_TIGHT
    finally:
        L.ndt2d_destroy(h)
