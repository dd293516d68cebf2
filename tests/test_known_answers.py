"""The outer loops of the hot path against known answers derived WITHOUT the oracle.

tests/golden/known_answers.json is written by tests/golden/make_known_answers.py from
the mathematics of the reference's lines in 60-digit arithmetic (no code shared with
oracle/ or the kernels): a 2-cell (+ one 4-point cell) map built by addScans from two
scans, a six-beam scan at a rotated pose, the 3 x 3 x 3 lattice.  It pins what the
reference's own tests do not: addScans' extent (src/scan_matcher_ndt.cpp:52-66), the
candidate order / strict-< argmin / returned best/N of matchScan (:103-134,148), its
accumulators and covariance formula (:137-146) and scorePoints' transform (:156-178).
Tolerance 1e-12: the two sides differ by IEEE rounding only.
"""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-12


@pytest.fixture(scope="module")
def ka():
    with open(os.path.join(HERE, "golden", "known_answers.json")) as f:
        return json.load(f)


def _scans(ka):
    return [(np.array(s["pose"]), np.array(s["points"])) for s in ka["map_scans"]]


def _check_grid(ka, cells6, size_x, size_y, origin):
    g = ka["grid"]
    assert (size_x, size_y) == (g["size_x"], g["size_y"])
    assert tuple(origin) == tuple(g["origin"])
    occupied = {c["index"]: c for c in g["cells"]}
    for idx in range(size_x * size_y):
        rec = cells6[idx]
        if idx in occupied:
            c = occupied[idx]
            assert rec[5] == c["n"]
            assert np.allclose(rec[0:2], c["mean"], rtol=0, atol=TOL)
            assert np.allclose(rec[2:5], c["information"], rtol=1e-12, atol=0)
        else:
            assert rec[5] == 0


def _check_match(ka, got, n_beams):
    m = ka["match"]
    assert got["n_candidates"] == m["n_candidates"] == 27
    assert np.max(np.abs(got["scores"] - np.array(m["scores"]))) < TOL
    assert got["best_index"] == m["best_index"]
    assert tuple(got["pose"]) == tuple(m["pose"])          # the accumulated offsets, exact
    assert abs(got["score"] - m["score"]) < TOL
    assert got["score"] * n_beams == pytest.approx(min(m["scores"]), abs=TOL)
    assert np.allclose(got["covariance"], m["covariance"], rtol=1e-11, atol=0)


def test_the_lattice_is_the_binary_exact_one(ka):
    p = ka["params"]
    assert list(O.search_offsets(p["search_linear_size"], p["search_linear_resolution"])) == ka["offsets"]
    assert list(O.search_offsets(p["search_angular_size"], p["search_angular_resolution"])) == ka["offsets"]


def test_oracle_reproduces_the_known_answers(ka):
    ref = O.ScanMatcherNDT()
    ref.initialize(**ka["params"])
    ref.addScans(_scans(ka))
    ndt = ref.ndt
    _check_grid(ka, ndt.cells6(), ndt.size_x, ndt.size_y, ndt.origin)
    beams = np.array(ka["beams"])
    got = ref.matchScan(ka["scan_pose"], beams, want_scores=True)
    _check_match(ka, got, len(beams))
    # the all-core variant used for baselines and full-lattice winners: same winner
    par = ref.matchScan(ka["scan_pose"], beams, omp_threads=2)
    assert par["best_index"] == ka["match"]["best_index"]
    assert tuple(par["pose"]) == tuple(ka["match"]["pose"])
    assert np.allclose(par["covariance"], ka["match"]["covariance"], rtol=1e-11, atol=0)
    for sp in ka["score_points"]:
        assert abs(ref.scorePoints(beams, sp["pose"]) - sp["score"]) < TOL
    assert abs(ref.scoreScan(ka["score_points"][1]["pose"], beams) - ka["score_points"][1]["score"]) < TOL


def _check_particles(ka, raw, w, mean, cov):
    pk = ka["particles"]
    if raw is not None:
        assert np.max(np.abs(np.asarray(raw) - pk["raw_weights"])) < TOL
    assert np.allclose(w, pk["weights"], rtol=1e-11, atol=1e-15)
    assert np.allclose(mean, pk["mean"], rtol=1e-11, atol=1e-15)
    cov, want = np.asarray(cov), np.array(pk["cov"])
    for r, c in ((0, 0), (0, 1), (1, 0), (1, 1), (2, 2)):
        assert cov[r, c] == pytest.approx(want[r, c], rel=1e-10)


def test_oracle_reproduces_the_known_particle_statistics(ka):
    """ParticleFilter::measure + updateStatistics (src/particle_filter.cpp:78-89,163-218)."""
    ref = O.ScanMatcherNDT()
    ref.initialize(**ka["params"])
    ref.addScans(_scans(ka))
    pk = ka["particles"]
    raw = O.pf_measure(ref, np.array(pk["poses"]), np.array(ka["beams"]))
    w, mean, cov = O.pf_update_statistics(pk["poses"], raw, cov_prev=pk["cov_before"])
    _check_particles(ka, raw, w, mean, cov)


def test_host_ndt_build_reproduces_the_known_cells(ka):
    """addScans' host build of the product library (no GPU needed)."""
    from ndt_2d_amd.scan_matcher import host_build_grid
    p = ka["params"]
    cells, sx, sy, ox, oy = host_build_grid(p["ndt_resolution"], p["range_max"], _scans(ka))
    _check_grid(ka, cells, sx, sy, (ox, oy))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["auto", "wave", "lane", "lane-noskip", "small", "small-noskip"])
def test_gpu_reproduces_the_known_answers(ka, variant):
    from ndt_2d_amd import ScanMatcherNDT
    gpu = ScanMatcherNDT(0)
    gpu.initialize("known_answers", **ka["params"])
    gpu.addScans(_scans(ka))
    cells, sx, sy, _, ox, oy = gpu.grid()
    _check_grid(ka, cells, sx, sy, (ox, oy))
    beams = np.array(ka["beams"])
    gpu.set_variant(variant)
    got = gpu.matchScan(ka["scan_pose"], beams, want_scores=True)
    _check_match(ka, got, len(beams))
    # the raw accumulators k, u, s of src/scan_matcher_ndt.cpp:137-140
    gpu.prepare_search(ka["scan_pose"], beams)
    gpu.match_launch(0, 3)
    rec = gpu.match_fetch()
    m = ka["match"]
    k = np.array(m["k"])
    assert np.allclose(rec[2:8], [k[0, 0], k[0, 1], k[0, 2], k[1, 1], k[1, 2], k[2, 2]], rtol=1e-11, atol=1e-15)
    assert np.allclose(rec[8:11], m["u"], rtol=1e-11, atol=1e-15)
    assert rec[11] == pytest.approx(m["s"], rel=1e-12)
    gpu.set_variant("auto")
    for sp in ka["score_points"]:
        assert abs(gpu.scorePoints(beams, sp["pose"]) - sp["score"]) < TOL
    assert abs(gpu.scoreScan(ka["score_points"][1]["pose"], beams) - ka["score_points"][1]["score"]) < TOL
    poses = np.array([sp["pose"] for sp in ka["score_points"]])
    assert np.max(np.abs(gpu.scorePoses(beams, poses) - [sp["score"] for sp in ka["score_points"]])) < TOL
    gpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["auto", "batched"])
def test_gpu_reproduces_the_known_particle_statistics(ka, variant):
    """measure + updateStatistics on the device: in one launch for a set of this size
    ("auto": block per particle, statistics by the last block) and through the batched
    kernels ("batched")."""
    from ndt_2d_amd import ScanMatcherNDT, pf_measure
    gpu = ScanMatcherNDT(0)
    gpu.initialize("known_particles", **ka["params"])
    gpu.addScans(_scans(ka))
    pk = ka["particles"]
    beams = np.array(ka["beams"])
    gpu.set_variant(variant)
    raw = gpu.scorePoses(beams, np.array(pk["poses"]))
    w, mean, cov = pf_measure(gpu, np.array(pk["poses"]), beams, cov_prev=pk["cov_before"])
    assert ("block-per-pose" in gpu.last_variant()) == (variant == "auto")
    _check_particles(ka, raw, w, mean, cov)
    gpu.close()


# ---- rows next to the hot path: tests/golden/make_known_answers_next.py

@pytest.fixture(scope="module")
def ka_next():
    with open(os.path.join(os.path.dirname(__file__), "golden", "known_answers_next.json")) as f:
        return json.load(f)


def _conversion_args(ka_next):
    sc = ka_next["scan_conversion"]
    return sc, dict(ranges=np.array(sc["ranges"], dtype=np.float32), angle_min=sc["angle_min"],
                    angle_increment=sc["angle_increment"], range_max=sc["range_max"],
                    laser=tuple(sc["laser"]), motion=tuple(sc["motion"]))


def test_oracle_reproduces_the_known_scan_conversion(ka_next):
    """LaserScan -> Scan with de-skew, both beam orders (src/ndt_mapper.cpp:385-453)."""
    sc, args = _conversion_args(ka_next)
    for inverted, key in ((False, "points"), (True, "points_inverted")):
        got = O.convert_scan(inverted=inverted, **args)
        assert got.shape == (len(sc[key]), 2)
        assert np.max(np.abs(got - np.array(sc[key]))) < TOL


def test_oracle_reproduces_the_known_motion_samples(ka_next):
    """MotionModel::sample with given normals, incl. normal_distribution<float>'s float
    arithmetic, reverse motion and the trans <= 0.01 branch (src/motion_model.cpp:45-83)."""
    mm = ka_next["motion_model"]
    z = np.array(mm["z"], dtype=np.float32)
    for case in mm["cases"]:
        poses, params = O.motion_sample(*case["motion"], mm["alphas"], mm["poses"], z)
        assert np.max(np.abs(params - np.array(case["params"]))) < TOL
        assert np.max(np.abs(poses - np.array(case["poses_after"]))) < TOL


@pytest.mark.gpu
def test_gpu_reproduces_the_known_scan_conversion(ka_next):
    from ndt_2d_amd import ScanMatcherNDT
    gpu = ScanMatcherNDT(0)
    gpu.initialize("known_conversion", ndt_resolution=1.0, range_max=10.0)
    sc, args = _conversion_args(ka_next)
    for inverted, key in ((False, "points"), (True, "points_inverted")):
        got = gpu.convertScan(inverted=inverted, **args)
        assert got.shape == (len(sc[key]), 2)
        assert np.max(np.abs(got - np.array(sc[key]))) < TOL
    gpu.close()


@pytest.mark.gpu
def test_gpu_reproduces_the_known_motion_samples(ka_next):
    from ndt_2d_amd import ScanMatcherNDT, pf_update
    gpu = ScanMatcherNDT(0)
    gpu.initialize("known_motion", ndt_resolution=1.0, range_max=10.0)
    mm = ka_next["motion_model"]
    z = np.array(mm["z"], dtype=np.float32)
    poses = np.array(mm["poses"])
    for case in mm["cases"]:
        after, _, _, _ = pf_update(gpu, poses, np.full(len(poses), 1.0 / len(poses)),
                                   *case["motion"], mm["alphas"], noise=z)
        assert np.max(np.abs(after - np.array(case["poses_after"]))) < TOL
    gpu.close()
