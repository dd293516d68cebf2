"""Cell::compute's eigenvalues (reference src/ndt_model.cpp:84-85,
Eigen::EigenSolver<Eigen::Matrix2d>): the transcription of Eigen 3.4.0's RealSchur / EigenSolver
for a 2 x 2 input (oracle/ndt2d_oracle.c eigenvalues_2x2, ndt_2d_amd/csrc/ndt2d_eigen2.h) against
the closed form d + p +- z that rounds 1-4 used, against exact arithmetic, and -- on the GPU --
the host build and the device build against the oracle in BOTH forms."""
import fractions
import math

import numpy as np
import pytest

import oracle_lib as O
from ndt_2d_amd import synth


def oracle_cells(cfg, form):
    O.set_eigen_form(form)
    try:
        m = O.ScanMatcherNDT()
        m.initialize(**synth.matcher_params(cfg))
        m.addScans(synth.map_scans(cfg))
        return np.array(m.ndt.cells6()).copy()
    finally:
        O.set_eigen_form("eigen")


def eigenvalues_of(points, form):
    O.set_eigen_form(form)
    try:
        c = O.Cell()
        for x, y in points:
            c.addPoint(x, y)
        c.compute()
        return c
    finally:
        O.set_eigen_form("eigen")


@pytest.mark.parametrize("cfg", [1, 3])
def test_the_two_forms_differ_in_a_few_cells_by_ulps(cfg):
    """How much of a map the choice touches: the information matrix of a cell changes only
    through the clamp branch (:88-96: determinant = 0.001 * large^2, or which branch is taken)."""
    a, b = oracle_cells(cfg, "eigen"), oracle_cells(cfg, "closed")
    computed = a[:, 5] >= 3
    differs = np.any(a != b, axis=1)
    assert np.array_equal(a[:, [0, 1, 5]], b[:, [0, 1, 5]])           # means and counts do not depend on it
    assert not differs[~computed].any()
    n = int(differs.sum())
    assert 0 < n <= 0.05 * computed.sum(), (n, int(computed.sum()))   # cfg-1: 8 of 249, cfg-3: 19 of 3218
    rel = np.abs(a[differs][:, 2:5] - b[differs][:, 2:5]) / np.maximum(np.abs(b[differs][:, 2:5]), 1e-300)
    assert rel.max() < 4e-15                                          # a few ulps
    # no cell changes BRANCH on these maps: a branch flip would change the matrix by far more than ulps
    assert rel.max() < 1e-12


def test_transcription_against_exact_arithmetic():
    """Both forms are within a few ulps of the exact eigenvalues of the covariance they are given
    (exact rational arithmetic for the covariance entries, 60-digit square root)."""
    mpmath = pytest.importorskip("mpmath")
    mpmath.mp.dps = 60
    rng = np.random.default_rng(11)
    worst = {"eigen": 0.0, "closed": 0.0}
    for _ in range(300):
        n = int(rng.integers(3, 12))
        ang = rng.uniform(0, math.pi)
        pts = [(3.0 + rng.normal(0, 0.05) * math.cos(ang) - rng.normal(0, 0.003) * math.sin(ang),
                1.0 + rng.normal(0, 0.05) * math.sin(ang) + rng.normal(0, 0.003) * math.cos(ang)) for _ in range(n)]
        for form in ("eigen", "closed"):
            c = eigenvalues_of(pts, form)
            cov = c.covariance
            a, b, d = (mpmath.mpf(float(cov[0, 0])), mpmath.mpf(float(cov[0, 1])), mpmath.mpf(float(cov[1, 1])))
            p = (a - d) / 2
            z = mpmath.sqrt(p * p + b * b)
            large, small = (a + d) / 2 + z, (a + d) / 2 - z
            info = c.information
            if small < mpmath.mpf("0.001") * large:
                det = mpmath.mpf("0.001") * large * large
                want = d / det
                # (0.001 is not the double 0.001 and the product rounds twice: a few ulps)
                got = mpmath.mpf(float(info[0, 0]))
                worst[form] = max(worst[form], float(abs(got - want) / abs(want)))
    assert 0 < worst["eigen"] < 2e-15 and 0 < worst["closed"] < 2e-15, worst


def test_deflated_and_degenerate_inputs():
    # an exactly diagonal covariance: both forms return the diagonal
    pts = [(1.0, 2.0), (1.5, 2.0), (0.5, 2.0), (1.0, 2.25), (1.0, 1.75)]
    a, b = eigenvalues_of(pts, "eigen"), eigenvalues_of(pts, "closed")
    assert np.array_equal(a.information, b.information)
    assert a.covariance[0, 1] == 0.0
    # identical points at binary-exact coordinates: zero covariance, Matrix2d::inverse() of it
    same = [(3.625, 0.625)] * 6
    a, b = eigenvalues_of(same, "eigen"), eigenvalues_of(same, "closed")
    assert np.isnan(a.information).any() and np.array_equal(np.isnan(a.information), np.isnan(b.information))
    # a covariance of 1e-310 (denormal): Eigen's RealSchur calls a matrix below DBL_MIN zero
    tiny = [(0.0, 0.0), (1e-155, 0.0), (0.0, 1e-155), (-1e-155, 0.0), (0.0, -1e-155)]
    a = eigenvalues_of(tiny, "eigen")
    assert a.n == 5 and a.valid


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [1, 3, 5])
@pytest.mark.parametrize("form", ["eigen", "closed"])
def test_host_and_device_builds_follow_the_oracle_in_both_forms(cfg, form):
    from ndt_2d_amd import ScanMatcherNDT
    exp = oracle_cells(cfg, form)
    scans = synth.map_scans(cfg)
    for mode in ("host", "device"):
        gpu = ScanMatcherNDT(0)
        gpu.initialize("t", **synth.matcher_params(cfg))
        gpu.set_eigenvalue_form(form)
        gpu.set_build_mode(mode)
        gpu.addScans(scans)
        assert np.array_equal(gpu.grid()[0], exp), (cfg, form, mode)
        gpu.close()


def test_clamp_screen_never_skips_a_cell_that_needs_its_eigenvalues(tmp_path):
    """Round 6: Cell::compute on the host and on the device asks ndt2d::clamp_test_surely_false
    first and computes the eigenvalues only when the answer is no (csrc/ndt2d_eigen2.h).  Compiled
    from the product's header: on 4 million covariances -- half of them within a decade of the
    threshold -- a `yes` never meets a cell that either eigenvalue form sends to the clamp branch,
    and the smallest eigenvalue ratio it lets through is four times the threshold."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "eigen_screen_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(root, "ndt_2d_amd", "csrc"),
                           os.path.join(root, "tests", "cpp", "eigen_screen_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    out = json.loads(r.stdout)
    assert r.returncode == 0 and out["violations"] == 0, out
    assert out["screened"] > 0.2 * out["cases"] and out["clamp_branch"] > 0.2 * out["cases"]
    assert out["near_threshold"] > 0.1 * out["cases"]
    assert out["least_ratio_screened"] > 0.00110     # (the screen's limit q = 0.0011 is r = 0.0011024)
