"""The particle kernel's per-wave queue (ndt2d_poses_compact.hip): a screening block's candidates are
appended lane by lane when the whole block fits the ring, in rounds when it does not.  Particle
sets that make EVERY pair a candidate (all particles at the scan's own pose: 2,048 candidates per
block of 32 beams), sets that straddle the ring's capacity, and ordinary ones must give the bits
of the unscreened control ("compact-exact") and the oracle's weights."""
import os

import numpy as np
import pytest

import oracle_lib as O
from ndt_2d_amd import ScanMatcherNDT, synth

pytestmark = pytest.mark.gpu


def _pair(cfg):
    scans = synth.map_scans(cfg)
    params = synth.matcher_params(cfg)
    gpu = ScanMatcherNDT(0)
    gpu.initialize("q", **params)
    gpu.addScans(scans)
    ref = O.ScanMatcherNDT()
    ref.initialize(**params)
    ref.addScans(scans)
    _, pts, true = synth.query_scan(cfg)
    return gpu, ref, pts, np.asarray(true, dtype=np.float64)


@pytest.mark.parametrize("cfg", [1, 3])
@pytest.mark.parametrize("dense_share", [1.0, 0.5, 0.1, 0.03])
def test_dense_candidate_blocks(cfg, dense_share):
    gpu, ref, pts, true = _pair(cfg)
    rng = np.random.default_rng(int(dense_share * 1000) + cfg)
    n = 4096 + 37
    parts = synth.particles(cfg, n) if cfg == 3 else np.stack(
        [rng.uniform(-3.5, 3.5, n), rng.uniform(-3.5, 3.5, n), rng.uniform(-np.pi, np.pi, n)], axis=1)
    dense = rng.random(n) < dense_share
    # at (and within millimetres of) the pose the scan was taken from: every beam ends in a cell with a distribution
    parts[dense] = true + rng.normal(0.0, 0.002, (int(dense.sum()), 3)) * [1.0, 1.0, 0.1]
    want = O.pf_measure(ref, parts, pts, omp_threads=os.cpu_count())
    try:
        gpu.set_variant("batched")
        got = gpu.scorePoses(pts, parts)
        assert "compact" in gpu.last_variant()
        gpu.set_variant("compact-exact")
        exact = gpu.scorePoses(pts, parts)
    finally:
        gpu.set_variant("auto")
    assert np.array_equal(got, exact)
    assert np.max(np.abs(got - want)) < 1e-9
    if dense_share == 1.0:
        assert (got < -0.2).all()      # every particle sees the map the scan was taken in
