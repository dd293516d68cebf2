"""A second, independent statement of the hot path -- numpy, written from the reference
sources, sharing no code with oracle/ndt2d_oracle.c -- against the oracle on cfg-1.

The reference has no test for matchScan / scorePoints / the NDT build beyond single
cells (oracle header: "parity unpinned"), so two independent restatements agreeing
is the strongest check available here against a shared misreading.  Differences
that remain are numerical only: numpy's exp / LAPACK's eigenvalues vs glibc / the
closed form (tolerances below)."""
import math

import numpy as np
import pytest

import oracle_lib as O
from ndt_2d_amd import synth


class NumpyNDT:
    """reference src/ndt_model.cpp, straight: per-cell incremental moments, covariance,
    eigenvalue clamp, inverse; likelihood of points."""

    def __init__(self, cell_size, size_x, size_y, origin_x, origin_y):
        self.cell = cell_size
        self.sx = int(size_x / cell_size + 1)            # :121-122
        self.sy = int(size_y / cell_size + 1)
        self.ox, self.oy = origin_x, origin_y
        n = self.sx * self.sy
        self.n = np.zeros(n)
        self.mean = np.zeros((n, 2))
        self.corr = np.zeros((n, 2, 2))
        self.info = np.zeros((n, 2, 2))

    def index(self, x, y):                                  # :203-218
        if x < self.ox or y < self.oy:
            return -1
        gx, gy = int((x - self.ox) / self.cell), int((y - self.oy) / self.cell)
        if gx >= self.sx or gy >= self.sy:
            return -1
        return gy * self.sx + gx

    def add_scan(self, pose, pts):                          # :132-152
        c, s = math.cos(pose[2]), math.sin(pose[2])
        for px, py in pts:
            x = pose[0] + (px * c - py * s)
            y = pose[1] + (px * s + py * c)
            i = self.index(x, y)
            if i < 0:
                continue
            n = self.n[i]                                    # :50-63
            p = np.array([x, y])
            self.mean[i] = (self.mean[i] * n + p) / (n + 1)
            for a in range(2):
                for b in range(a, 2):
                    self.corr[i, a, b] = (self.corr[i, a, b] * n + p[a] * p[b]) / (n + 1)
            self.n[i] = n + 1

    def compute(self):                                      # :65-103,154-160
        for i in np.nonzero(self.n >= 3)[0]:
            n = self.n[i]
            cov = np.zeros((2, 2))
            for a in range(2):
                for b in range(a, 2):
                    cov[a, b] = cov[b, a] = (self.corr[i, a, b] - self.mean[i, a] * self.mean[i, b]) * (n / (n - 1))
            ev = np.linalg.eigvals(cov).real               # Eigen::EigenSolver (:84-85)
            small, large = min(ev), max(ev)
            if small < 0.001 * large:
                det = (0.001 * large) * large
                self.info[i] = np.array([[cov[1, 1], -cov[0, 1]], [-cov[1, 0], cov[0, 0]]]) / det
            else:
                self.info[i] = np.linalg.inv(cov)

    def likelihood(self, X, Y):
        """sum over the last axis of per-point likelihoods; X, Y: [..., n_points]"""
        inside = (X >= self.ox) & (Y >= self.oy)
        gx = ((X - self.ox) / self.cell).astype(np.int64)
        gy = ((Y - self.oy) / self.cell).astype(np.int64)
        inside &= (gx >= 0) & (gx < self.sx) & (gy >= 0) & (gy < self.sy)
        idx = np.where(inside, gy * self.sx + gx, 0)
        ok = inside & (self.n[idx] >= 5)                    # :107 (n < 5 -> 0.0)
        q0 = X - self.mean[idx, 0]
        q1 = Y - self.mean[idx, 1]
        I = self.info[idx]
        e = (-0.5 * q0 * I[..., 0, 0] + -0.5 * q1 * I[..., 1, 0]) * q0 + \
            (-0.5 * q0 * I[..., 0, 1] + -0.5 * q1 * I[..., 1, 1]) * q1      # :113-114
        with np.errstate(over="ignore", invalid="ignore"):
            lik = np.where(ok, np.exp(np.where(ok, e, 0.0)), 0.0)
        return lik.sum(axis=-1)


def _offsets(size, res):
    out, v = [], -size
    while v < size:                                          # scan_matcher_ndt.cpp:103,117,119
        out.append(v)
        v += res
    return np.array(out)


@pytest.fixture(scope="module")
def cfg1():
    scans = synth.map_scans(1)
    p = synth.matcher_params(1)
    # addScans (:49-74), with the numeric_limits<double>::min() start of max_x_ / max_y_
    tiny = np.finfo(np.float64).tiny
    min_x = min(min(s[0][0] - p["range_max"] for s in scans), np.finfo(np.float64).max)
    max_x = max(max(s[0][0] + p["range_max"] for s in scans), tiny)
    min_y = min(min(s[0][1] - p["range_max"] for s in scans), np.finfo(np.float64).max)
    max_y = max(max(s[0][1] + p["range_max"] for s in scans), tiny)
    ndt = NumpyNDT(p["ndt_resolution"], max_x - min_x, max_y - min_y, min_x, min_y)
    for pose, pts in scans:
        ndt.add_scan(pose, pts)
    ndt.compute()
    ref = O.ScanMatcherNDT()
    ref.initialize(**p)
    ref.addScans(scans)
    return ndt, ref, p


def test_ndt_build_agrees(cfg1):
    ndt, ref, _ = cfg1
    cells = ref.ndt.cells6()
    assert (ndt.sx, ndt.sy) == (ref.ndt.size_x, ref.ndt.size_y) == (41, 41)
    assert np.array_equal(ndt.n, cells[:, 5])
    built = ndt.n >= 3
    assert np.allclose(ndt.mean[built], cells[built, :2], rtol=1e-14, atol=0)
    got = np.stack([ndt.info[:, 0, 0], ndt.info[:, 0, 1], ndt.info[:, 1, 1]], axis=1)
    # LAPACK eigenvalues / inverse vs the closed forms: a few ulps, amplified by the
    # conditioning of thin-wall covariances
    assert np.allclose(got[built], cells[built, 2:5], rtol=1e-7, atol=1e-6)
    assert (ndt.n >= 5).sum() > 100


def test_match_scan_agrees(cfg1):
    ndt, ref, p = cfg1
    guess, pts, _ = synth.query_scan(1)
    exp = ref.matchScan(guess, pts, want_scores=True)
    dth = _offsets(p["search_angular_size"], p["search_angular_resolution"])
    dlin = _offsets(p["search_linear_size"], p["search_linear_resolution"])
    assert len(dth) * len(dlin) ** 2 == exp["n_candidates"] == 17640
    use = min(p["laser_max_beams"], len(pts))
    step = float(len(pts)) / use
    sub = pts[[int(i * step) for i in range(use)]]
    scores = np.zeros((len(dth), len(dlin), len(dlin)))
    for t, d in enumerate(dth):
        c, s = math.cos(guess[2] + d), math.sin(guess[2] + d)
        ox = sub[:, 0] * c - sub[:, 1] * s + guess[0]
        oy = sub[:, 0] * s + sub[:, 1] * c + guess[1]
        X = ox[None, None, :] + dlin[:, None, None]
        Y = oy[None, None, :] + dlin[None, :, None]
        scores[t] = -ndt.likelihood(X + 0 * Y, Y + 0 * X)
    flat = scores.reshape(-1)
    # the information matrices differ in the last digits (see above): 1e-6 on sums of 720
    assert np.max(np.abs(flat - exp["scores"])) < 1e-6
    best = int(np.argmin(flat))
    assert best == exp["best_index"]
    assert flat[best] / use == pytest.approx(exp["score"], abs=1e-9)
    t, rem = divmod(best, len(dlin) ** 2)
    assert np.allclose([dlin[rem // len(dlin)], dlin[rem % len(dlin)], dth[t]], exp["pose"], rtol=0, atol=0)
    # covariance (:137-146)
    DX, DY = np.meshgrid(dlin, dlin, indexing="ij")
    k = np.zeros((3, 3))
    u = np.zeros(3)
    ssum = 0.0
    for t, d in enumerate(dth):
        x = np.stack([DX.ravel(), DY.ravel(), np.full(DX.size, d)], axis=1)
        sc = scores[t].ravel()
        k += (x[:, :, None] * x[:, None, :] * sc[:, None, None]).sum(axis=0)
        u += (x * sc[:, None]).sum(axis=0)
        ssum += sc.sum()
    cov = (1 / ssum) * k + (1 / (ssum * ssum)) * np.outer(u, u)
    assert np.allclose(cov, exp["covariance"], rtol=1e-6, atol=1e-12)


def test_update_statistics_agrees():
    """ParticleFilter::updateStatistics (src/particle_filter.cpp:163-218), vectorised."""
    rng = np.random.default_rng(5)
    n = 5000
    parts = np.stack([rng.normal(2.0, 0.4, n), rng.normal(-1.0, 0.7, n), rng.normal(3.0, 0.5, n)], axis=1)
    parts[:, 2] = (parts[:, 2] + np.pi) % (2 * np.pi) - np.pi
    w_raw = -rng.uniform(0.01, 1.0, n)              # scores are negative (:86)
    cov_prev = np.zeros((3, 3))
    cov_prev[2, 2] = 0.0625
    w, mean, cov = O.pf_update_statistics(parts, w_raw, cov_prev)
    wn = w_raw / w_raw.sum()                        # :166-174
    assert np.allclose(w, wn, rtol=1e-13, atol=0) and np.all(wn > 0)
    mx, my = (wn * parts[:, 0]).sum(), (wn * parts[:, 1]).sum()
    mth = math.atan2((wn * np.sin(parts[:, 2])).sum(), (wn * np.cos(parts[:, 2])).sum())   # :205
    assert np.allclose(mean, [mx, my, mth], rtol=1e-11, atol=1e-13)
    want = np.zeros((3, 3))
    want[0, 0] = (wn * parts[:, 0] ** 2).sum() - mx * mx                                   # :208-215
    want[0, 1] = want[1, 0] = (wn * parts[:, 0] * parts[:, 1]).sum() - mx * my
    want[1, 1] = (wn * parts[:, 1] ** 2).sum() - my * my
    d = (mth - parts[:, 2] + np.pi) % (2 * np.pi) - np.pi                                   # shortest_angular_distance
    want[2, 2] = cov_prev[2, 2] + (wn * d * d).sum()                                        # :216, never zeroed
    assert np.allclose(cov, want, rtol=1e-9, atol=1e-12)


def test_score_points_agrees(cfg1):
    ndt, ref, p = cfg1
    _, pts, _ = synth.query_scan(1)
    rng = np.random.default_rng(0)
    for pose in np.stack([rng.uniform(-2, 2, 20), rng.uniform(-2, 2, 20), rng.uniform(-3.1, 3.1, 20)], axis=1):
        c, s = math.cos(pose[2]), math.sin(pose[2])                   # conversions.hpp:64-68
        X = pose[0] + (c * pts[:, 0] - s * pts[:, 1])
        Y = pose[1] + (s * pts[:, 0] + c * pts[:, 1])
        want = -ndt.likelihood(X, Y) / len(pts)                         # :156-178
        assert ref.scorePoints(pts, pose) == pytest.approx(want, abs=1e-8)
