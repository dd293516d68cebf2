"""GPU parity for the OccupancyGrid renderer (reference
src/occupancy_grid.cpp:47-185).  Counts are integers and the threshold test is the
reference's own double arithmetic on them, so the published map must be
bit-identical to the oracle's sequential loop: np.array_equal throughout."""
import numpy as np
import pytest

import oracle_lib as O
from ndt_2d_amd import ScanMatcherNDT, synth
from ndt_2d_amd.occupancy_grid import OccupancyGrid

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    m = ScanMatcherNDT(0)
    m.initialize("occ", **synth.matcher_params(1))
    return m


def _same(got, want):
    for k in ("resolution", "width", "height", "origin_x", "origin_y"):
        assert got[k] == want[k], (k, got[k], want[k])
    assert got["data"].shape == want["data"].shape
    assert np.array_equal(got["data"], want["data"])


@pytest.mark.parametrize("cfg,resolution,occ_thresh", [(1, 0.05, 0.25), (1, 0.1, 0.5),
                                                       (1, 0.03, 0.1), (3, 0.05, 0.25)])
def test_synthetic_maps_match_oracle(device, cfg, resolution, occ_thresh):
    scans = synth.map_scans(cfg)
    want = O.OccupancyGrid(resolution, occ_thresh).getMsg(scans)
    got = OccupancyGrid(resolution, occ_thresh, device).getMsg(scans)
    _same(got, want)
    values, counts = np.unique(got["data"], return_counts=True)
    assert set(values) == {-1, 0, 100} and counts.min() > 100


def test_bounds_persist_and_grow_like_the_reference(device):
    """num_scans_ / min_x_ ... are state of the generator (:51-54,156-185): bounds are
    extended by the new scans only, and re-rounded at every update."""
    scans = synth.map_scans(1)
    ref = O.OccupancyGrid(0.05, 0.25)
    gpu = OccupancyGrid(0.05, 0.25, device)
    for upto in (1, 1, 4, 9):
        want = ref.getMsg(scans[:upto])
        got = gpu.getMsg(scans[:upto])
        _same(got, want)
        assert np.array_equal(gpu.bounds, ref.bounds) and gpu.num_scans == ref.num_scans
    # a far-away scan enlarges the map
    far = [((6.0, -5.0, 0.7), np.array([[1.0, 0.0], [0.0, 2.0], [-1.5, 0.5]]))]
    _same(gpu.getMsg(scans + far), ref.getMsg(scans + far))
    assert gpu.bounds[1] > 6.0 and gpu.bounds[2] < -5.0


def test_random_scans_and_edge_cases(device):
    rng = np.random.default_rng(4)
    for trial in range(12):
        scans = []
        for _ in range(int(rng.integers(1, 6))):
            pose = (rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-np.pi, np.pi))
            n = int(rng.integers(0, 300))
            scans.append((pose, rng.uniform(-6, 6, size=(n, 2))))
        res = float(rng.choice([0.05, 0.1, 0.25, 0.3]))
        thresh = float(rng.choice([0.1, 0.25, 0.5]))
        _same(OccupancyGrid(res, thresh, device).getMsg(scans), O.OccupancyGrid(res, thresh).getMsg(scans))
    # a scan whose pose lies outside the bounding box of all points: the ray starts
    # outside the grid (the reference would write out of bounds there; both skip)
    outside = [((20.0, 20.0, 0.0), np.array([[-18.0, -19.0], [-17.0, -19.5]]))]
    _same(OccupancyGrid(0.1, 0.25, device).getMsg(outside), O.OccupancyGrid(0.1, 0.25).getMsg(outside))
    # zero-length rays (point at the pose) hit their own cell; empty scans draw nothing
    tiny = [((0.5, 0.5, 0.3), np.zeros((3, 2))), ((1.0, 1.0, 0.0), np.zeros((0, 2)))]
    got = OccupancyGrid(0.1, 0.25, device).getMsg(tiny)
    _same(got, O.OccupancyGrid(0.1, 0.25).getMsg(tiny))
    assert (got["data"] == 100).sum() == 1 and (got["data"] == 0).sum() == 0
    # no scans at all: the 10 x 10 padding map, all unknown
    got = OccupancyGrid(0.1, 0.25, device).getMsg([])
    _same(got, O.OccupancyGrid(0.1, 0.25).getMsg([]))
    assert got["width"] == 10 and got["height"] == 10 and np.all(got["data"] == -1)
