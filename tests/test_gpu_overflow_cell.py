"""exp() above its overflow threshold.  Cell::score returns std::exp(exponent)
(reference src/ndt_model.cpp:115).  A cell whose >= 5 points are identical -- a robot standing
still in front of a distant wall -- has a covariance of pure rounding noise; for one such cell in
eight it comes out NEGATIVE on both axes, Cell::compute's clamp branch (:88-96) divides by
det = 0.001 * large^2 > 0 and the information matrix is negative definite with entries of 1e19:
every point of that cell but the mean itself has an exponent of +1e15 and more, the reference's
likelihood is +inf and the candidate's score -inf.  The kernels' lean exp (no range fix-ups) must
return +inf there too, in every variant, and the -inf scores must force the same argmin."""
import numpy as np
import pytest

import oracle_lib as O
from ndt_2d_amd import ScanMatcherNDT, synth

pytestmark = pytest.mark.gpu


def overflow_cell_point():
    """Six identical points inside the (otherwise empty) cell [3.5, 3.75) x [0.5, 0.75) of the
    cfg-1 map whose covariance rounds negative on both axes (found on the CPU, seeded)."""
    rng = np.random.default_rng(7)
    for _ in range(4000):
        x, y = rng.uniform(3.52, 3.73), rng.uniform(0.52, 0.73)
        c = O.Cell()
        for _ in range(6):
            c.addPoint(x, y)
        c.compute()
        if c.covariance[0, 0] < 0 and c.covariance[1, 1] < 0 and c.score(x + 0.01, y - 0.02) == np.inf:
            return x, y
    raise AssertionError("no such cell in 4000 draws")


def test_the_oracle_has_such_cells():
    x, y = overflow_cell_point()
    c = O.Cell()
    for _ in range(6):
        c.addPoint(x, y)
    c.compute()
    info = c.information
    assert info[0, 0] < -1e15 and info[1, 1] < -1e15      # negative definite by rounding
    assert c.score(x + 0.01, y - 0.02) == np.inf and c.score(x, y + 1e-6) == np.inf


@pytest.mark.parametrize("search", [dict(search_linear_size=0.3, search_angular_size=0.06), dict()])
def test_overflowing_exponent_gives_inf_like_the_reference(search):
    x, y = overflow_cell_point()
    bad = ((0.0, 0.0, 0.0), np.tile([[x, y]], (6, 1)))
    scans = synth.map_scans(1)
    guess, pts, _ = synth.query_scan(1)
    params = synth.matcher_params(1, **search)
    ref = O.ScanMatcherNDT()
    ref.initialize(**params)
    ref.addScans([bad])
    gpu = ScanMatcherNDT(0)
    gpu.initialize("t", **params)
    # the cell itself: host build and device build give the oracle's records bit for bit
    for mode in ("host", "device"):
        gpu.set_build_mode(mode)
        gpu.addScans([bad])
        assert np.array_equal(gpu.grid()[0], ref.ndt.cells6()), mode
    # a batch of poses through the few-pose kernel and the batched (compact) particle kernel
    poses = np.array([(0.0, 0.0, 0.0), (0.05, 0.02, 0.0), (2.0, 2.0, 1.0), (-0.01, 0.03, 0.002)])
    probe = np.array([(x + 0.01, y - 0.02), (1.0, 1.0), (x, y), (3.7, 0.7)])
    exp = O.pf_measure(ref, poses, probe)
    assert np.isneginf(exp).any() and not np.isneginf(exp).all()
    got = gpu.scorePoses(probe, poses)
    assert np.array_equal(np.isneginf(got), np.isneginf(exp))
    finite = np.isfinite(exp)
    assert np.allclose(got[finite], exp[finite], rtol=0, atol=1e-12)
    many = np.tile(poses, (2048, 1))
    for variant in ("auto", "compact-exact", "dense", "lds", "global"):
        gpu.set_variant(variant)
        got = gpu.scorePoses(probe, many)
        assert "lane-per-pose" in gpu.last_variant(), gpu.last_variant()
        assert np.array_equal(np.isneginf(got), np.tile(np.isneginf(exp), 2048)), variant
        assert np.allclose(got[np.tile(finite, 2048)], np.tile(exp, 2048)[np.tile(finite, 2048)], rtol=0, atol=1e-12)
    gpu.set_variant("auto")
    # single poses on the host path (libm exp) and on the device
    for where in ("host", "device"):
        gpu.set_single_pose_path(where, 256)
        for q in range(len(poses)):
            a, b = gpu.scorePoints(probe, poses[q]), ref.scorePoints(probe, poses[q])
            assert (a == b) if np.isinf(b) else abs(a - b) < 1e-12, (where, q, a, b)
    gpu.set_single_pose_path("host", 256)

    # the cell inside a healthy map: matchScan in every mapping
    ref.addScans(scans + [bad])
    exp = ref.matchScan(guess, pts, want_scores=True)
    n_inf = int(np.isneginf(exp["scores"]).sum())
    assert 0 < n_inf < len(exp["scores"]) and not np.isnan(exp["scores"]).any()
    assert exp["score"] == -np.inf
    assert exp["best_index"] == int(np.argmax(np.isneginf(exp["scores"])))   # the first -inf in visiting order
    for variant in ("auto", "small", "lane", "lane-noskip", "wave"):
        gpu.set_variant(variant)
        for mode in ("host", "device"):
            gpu.set_build_mode(mode)
            gpu.addScans(scans + [bad])
            got = gpu.matchScan(guess, pts, want_scores=True)
            tag = (variant, mode, gpu.last_variant())
            assert np.array_equal(np.isneginf(got["scores"]), np.isneginf(exp["scores"])), tag
            finite = np.isfinite(exp["scores"])
            assert np.allclose(got["scores"][finite], exp["scores"][finite], rtol=0, atol=1e-9), tag
            assert got["best_index"] == exp["best_index"] and got["score"] == -np.inf, tag
            assert np.array_equal(got["pose"], exp["pose"]), tag
            # s = -inf: covariance = (1 / s) k + (1 / s^2) u u^T is NaN throughout, as in the reference
            assert np.isnan(got["covariance"]).all() and np.isnan(exp["covariance"]).all(), tag
    gpu.set_variant("auto")
