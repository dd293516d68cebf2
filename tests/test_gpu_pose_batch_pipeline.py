"""A large pose batch handed over in ordinary host memory is cut into pieces whose uploads,
scoring and downloads overlap (ndt2d_set_pipeline_pieces; ParticleFilter::measure of a big filter
through the drop-in boundary, reference src/particle_filter.cpp:78-89).  The cut must not show:
raw scores bit for bit, normalised weights and statistics to rounding, and the oracle's weights."""
import os

import numpy as np
import pytest

from ndt_2d_amd import ScanMatcherNDT, pf_measure, synth

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big_filter():
    cfg = 3
    gpu = ScanMatcherNDT(0)
    gpu.initialize("pf", **synth.matcher_params(cfg))
    scans = synth.map_scans(cfg)
    gpu.addScans(scans)
    ref = O.ScanMatcherNDT()
    ref.initialize(**synth.matcher_params(cfg))
    ref.addScans(scans)
    _, pts, _ = synth.query_scan(cfg)
    rng = np.random.default_rng(7)
    pa = synth.particles(cfg)
    # 300,017 particles: above the pipelining threshold, not a multiple of anything
    parts = np.concatenate([pa, pa[:100000] + rng.normal(0, 0.05, (100000, 3)), np.tile(pa, (2, 1))[:100017] + rng.normal(0, 0.3, (100017, 3))])
    return gpu, ref, pts, parts


@pytest.mark.parametrize("pieces", [0, 2, 3, 4, 7, 16])
def test_pipelined_batch_leaves_no_trace(big_filter, pieces):
    gpu, ref, pts, parts = big_filter
    try:
        gpu.set_pipeline_pieces(1)
        s1 = gpu.scorePoses(pts, parts)
        assert gpu.last_pipeline_pieces() == 1
        w1, mean1, cov1 = pf_measure(gpu, parts, pts)
        gpu.set_pipeline_pieces(pieces)
        s = gpu.scorePoses(pts, parts)
        used = gpu.last_pipeline_pieces()
        # (no piece below 32,768 poses: 300,017 poses are cut into nine at most)
        assert used == min(4 if pieces == 0 else pieces, len(parts) // 32768), (pieces, used)
        w, mean, cov = pf_measure(gpu, parts, pts)
        assert gpu.last_pipeline_pieces() == used
    finally:
        gpu.set_pipeline_pieces(0)
    assert np.array_equal(s, s1)
    # the contract of include/ndt2d_hip.h (ndt2d_set_pipeline_pieces): the moment sums are added in
    # piece order, so the normalised weights agree to within 64 ulps of the sums' magnitude -- every
    # weight is a raw score over the total, and only the total's last bits move
    total = float(np.sum(s1))
    assert np.max(np.abs(w * total - w1 * total)) <= 64 * np.spacing(np.max(np.abs(s1)))
    assert np.max(np.abs(w - w1)) <= 1e-15 * np.max(np.abs(w1)) * 64
    assert np.allclose(mean, mean1, rtol=0, atol=1e-12)
    assert np.allclose(cov, cov1, rtol=1e-11, atol=1e-13)


def test_pipelined_measure_matches_the_oracle(big_filter):
    gpu, ref, pts, parts = big_filter
    gpu.set_pipeline_pieces(0)
    w, mean, cov = pf_measure(gpu, parts, pts)
    assert gpu.last_pipeline_pieces() == 4
    raw = O.pf_measure(ref, parts, pts, omp_threads=os.cpu_count())
    want = raw / raw.sum()
    assert np.max(np.abs(w - want)) < 1e-9 * np.max(np.abs(want))
    # a small batch is not cut
    gpu.scorePoses(pts, parts[:5000])
    assert gpu.last_pipeline_pieces() == 1


def test_pipeline_pieces_argument_is_checked(big_filter):
    gpu = big_filter[0]
    from ndt_2d_amd import Ndt2dError
    for bad in (-1, 17):
        with pytest.raises(Ndt2dError):
            gpu.set_pipeline_pieces(bad)
