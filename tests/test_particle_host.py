"""CPU tests for the particle-filter rows around the hot path:
the oracle's MotionModel restatement against the reference's own scenario
(reference test/particle_tests.cpp:74-140), and the host-side KLD resampling rule
against a plain sequential restatement of the reference loop."""
import numpy as np
import pytest

import oracle_lib as O
from ndt_2d_amd.particle_filter import (kld_leaf_count, kld_leaf_keys, kld_resample_indices)

ALPHAS = [0.1, 0.1, 0.1, 0.1, 0.0]   # particle_tests.cpp:76-77


def _sample(rng, dx, dy, dth, poses):
    z = rng.standard_normal((len(poses), 3)).astype(np.float32)
    return O.motion_sample(dx, dy, dth, ALPHAS, poses, z)[0]


@pytest.mark.parametrize("seed", range(6))
def test_oracle_motion_model_reference_scenario(seed):
    """The reference's test_particle_filter, same 50 poses, same tolerances.  The
    scenario is one random draw in the reference too, and its last expectation
    (mean y = 2.0 +- 0.5 after two forward steps) sits ~3 sigma from the model's
    true mean of ~1.6 (headings have spread by then, see the analytic test below):
    about three quarters of all seeds satisfy every tolerance, these six do."""
    rng = np.random.default_rng(seed)
    zero = np.zeros((50, 3))

    def near(poses, expect, tol):
        m = poses.mean(axis=0)
        assert np.all(np.abs(m - np.array(expect)) < tol), (m, expect)

    # forward motion (:80-86)
    near(_sample(rng, 1.0, 0.0, 0.0, zero), (1.0, 0.0, 0.0), 0.3)
    for sign in (1.0, -1.0):
        # in-place rotation, then forward twice (:88-126)
        p = _sample(rng, 0.0, 0.0, sign * 1.57, zero)
        near(p, (0.0, 0.0, sign * 1.57), 0.3)
        p = _sample(rng, 1.0, 0.0, 0.0, p)
        near(p, (0.0, sign * 1.0, sign * 1.57), 0.3)
        p = _sample(rng, 1.0, 0.0, 0.0, p)
        near(p, (0.0, sign * 2.0, sign * 1.57), 0.5)
        # in-place rotation with a small disturbance (:128-142)
        near(_sample(rng, 0.01, -0.01, sign * 1.57, zero), (0.0, 0.0, sign * 1.57), 0.3)


def test_oracle_motion_model_matches_its_analytic_mean():
    """Rotate in place by 1.57, then drive 1.0 forward: theta ~ N(1.57, a1 1.57^2),
    heading noise r1 ~ N(0, a2), so E[y] = sin(1.57) exp(-(a1 1.57^2 + a2) / 2) and
    E[x] = cos(1.57) exp(...)."""
    rng = np.random.default_rng(42)
    n = 400000
    p = _sample(rng, 0.0, 0.0, 1.57, np.zeros((n, 3)))
    # circular statistics: theta is wrapped into (-pi, pi] (:82)
    r = np.exp(1j * p[:, 2]).mean()
    assert abs(np.angle(r) - 1.57) < 5e-3
    assert abs(np.sqrt(-2.0 * np.log(abs(r))) - np.sqrt(0.1) * 1.57) < 5e-3
    # a4 rot2_^2 spreads the particles along their (old) heading while turning (:63-65)
    assert abs(p[:, 0].std() - np.sqrt(0.1) * 1.57) < 5e-3 and abs(p[:, 1].std()) < 1e-12
    q = _sample(rng, 1.0, 0.0, 0.0, p)
    shrink = np.exp(-0.5 * (0.1 * 1.57 ** 2 + 0.1))
    assert abs((q[:, 1] - p[:, 1]).mean() - np.sin(1.57) * shrink) < 5e-3
    assert abs((q[:, 0] - p[:, 0]).mean() - np.cos(1.57) * shrink) < 5e-3


def test_oracle_motion_model_noise_free_is_the_odometry():
    """z = 0: every pose moves by exactly (trans, rot1, rot2) in its own frame."""
    poses = np.array([[0.0, 0.0, 0.0], [1.0, -2.0, 0.5], [3.0, 4.0, -3.0]])
    out, params = O.motion_sample(0.3, 0.4, 0.25, ALPHAS, poses, np.zeros((3, 3), np.float32))
    rot1, trans, rot2 = params[:3]
    assert trans == pytest.approx(0.5) and rot1 == pytest.approx(np.arctan2(0.4, 0.3))
    f = np.float32
    for a, b in zip(poses, out):
        assert b[0] == a[0] + float(f(trans)) * np.cos(a[2] + float(f(rot1)))
        assert b[1] == a[1] + float(f(trans)) * np.sin(a[2] + float(f(rot1)))
        assert b[2] == O.lib().orc_normalize_angle(a[2] + float(f(rot1)) + float(f(rot2)))


def test_oracle_motion_model_small_translation_has_no_heading():
    """trans <= 0.01 takes rot1 = 0 (motion_model.cpp:50)."""
    _, params = O.motion_sample(0.005, 0.005, 0.3, ALPHAS, np.zeros((1, 3)),
                                np.zeros((1, 3), np.float32))
    assert params[0] == 0.0 and params[2] == pytest.approx(0.3)
    # reverse motion: rot1 = pi, but sigma uses the distance to the nearer of 0 / pi (:55-58)
    _, params = O.motion_sample(-1.0, 0.0, 0.0, ALPHAS, np.zeros((1, 3)),
                                np.zeros((1, 3), np.float32))
    assert abs(params[0]) == pytest.approx(np.pi)
    assert params[3] == pytest.approx(np.sqrt(0.1))     # a2 * trans^2 only


def test_kd_tree_leaf_count_reference_scenario():
    """reference test/particle_tests.cpp:47-72 (leaf 0.5 x 0.5 x 0.25)."""
    leaf = (0.5, 0.5, 0.25)
    poses = []
    expect = []
    for pose, count in (((0.0, 0.0, 0.0), 1), ((0.0, 0.0, 0.0), 1), ((0.75, 0.0, 0.0), 2),
                        ((-0.75, 0.0, 0.0), 3), ((0.75, 0.75, 0.0), 4)):
        poses.append(pose)
        expect.append(count)
        assert kld_leaf_count(kld_leaf_keys(np.array(poses), leaf)) == count


def _resample_loop(draws, keys, min_particles, max_particles, kld_err, kld_z):
    """ParticleFilter::resample's loop, statement by statement
    (reference src/particle_filter.cpp:106-134)."""
    leaves = set()
    out = []
    mx = max_particles
    it = iter(draws)
    while len(out) < max(min_particles, mx):
        p = next(it)
        leaves.add(tuple(keys[p]))
        out.append(p)
        k = len(leaves)
        if k > 1:
            a = (k - 1) / (2.0 * kld_err)
            b = 2.0 / (9.0 * (k - 1))
            c = 1.0 - b + np.sqrt(b) * kld_z
            mx = int(a * c * c * c)
        if len(out) >= max_particles:
            break
    return np.array(out, dtype=np.int64)


@pytest.mark.parametrize("seed", range(12))
def test_kld_resample_rule_matches_sequential_loop(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(10, 400))
    spread = rng.choice([0.05, 0.5, 3.0])
    particles = rng.normal(0.0, spread, size=(n, 3))
    keys = kld_leaf_keys(particles)
    min_p = int(rng.integers(1, 60))
    max_p = int(rng.integers(min_p, 3000))
    draws = rng.integers(0, n, size=max_p)
    kld_err, kld_z = float(rng.choice([0.01, 0.05])), float(rng.choice([0.99, 2.33]))
    got = kld_resample_indices(draws, keys, min_p, max_p, kld_err, kld_z)
    want = _resample_loop(draws, keys, min_p, max_p, kld_err, kld_z)
    assert np.array_equal(got, want)
    assert min(min_p, max_p) <= len(got) <= max_p


@pytest.mark.parametrize("seed", range(12))
def test_native_kld_resample_matches_sequential_loop(seed):
    """ndt2d_kld_resample (host code of the library, no GPU needed) against the loop above.
    Its draws invert the cumulative weights: draw i = first particle whose running
    weight sum exceeds u_i * total (sequential sums, as the C code forms them)."""
    from ndt_2d_amd.particle_filter import kld_resample_native
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(10, 400))
    spread = rng.choice([0.05, 0.5, 3.0])
    particles = rng.normal(0.0, spread, size=(n, 3))
    weights = rng.random(n) ** 3
    if seed % 3 == 0:
        weights[rng.integers(0, n, size=n // 2)] = 0.0      # particles that can never be drawn
    min_p = int(rng.integers(1, 60))
    max_p = int(rng.integers(min_p, 3000))
    kld_err, kld_z = float(rng.choice([0.01, 0.05])), float(rng.choice([0.99, 2.33]))
    u = rng.random(max_p)
    got = kld_resample_native(particles, weights, min_p, max_p, kld_err, kld_z, u)
    cdf = np.zeros(n)
    total = 0.0
    for i in range(n):
        total += weights[i]
        cdf[i] = total
    draws = np.minimum(np.searchsorted(cdf, u * total, side="right"), n - 1)
    want = _resample_loop(draws, kld_leaf_keys(particles), min_p, max_p, kld_err, kld_z)
    assert np.array_equal(got.astype(np.int64), want)
    assert np.all(weights[got] > 0.0)


def test_native_kld_resample_edge_cases():
    from ndt_2d_amd import _capi
    from ndt_2d_amd.particle_filter import kld_resample_native
    p = np.zeros((5, 3))
    w = np.full(5, 0.2)
    # one leaf only: Mx stays max_particles, so max_particles draws are kept (:105-107)
    assert len(kld_resample_native(p, w, 3, 40, 0.01, 0.99, np.linspace(0, 0.999, 40))) == 40
    assert len(kld_resample_native(p, w, 3, 0, 0.01, 0.99, np.zeros(0))) == 0
    # two far-apart leaves: the bound drops to a handful and min_particles takes over
    p2 = np.array([[0.0, 0.0, 0.0], [10.0, 10.0, 1.0]])
    got = kld_resample_native(p2, [0.5, 0.5], 7, 1000, 0.05, 0.99, np.tile([0.1, 0.9], 500))
    assert list(got[:4]) == [0, 1, 0, 1] and 7 <= len(got) < 100
    # u == 1 - ulp and a cumulative sum that rounds: never past the last particle
    got = kld_resample_native(p2, [0.1, 0.2], 1, 4, 0.05, 0.99, [np.nextafter(1.0, 0.0)] * 4)
    assert set(got) == {1}
    with pytest.raises(_capi.Ndt2dError):
        kld_resample_native(p2, [0.5, 0.5], 1, 10, 0.05, 0.99, np.zeros(3))   # too few uniforms
