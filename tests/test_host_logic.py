"""Host-side logic of the product (no GPU): the NDT build of addScans, the
search lattice, the synthetic generator and the sharding helpers, checked
against the oracle."""
import json
import math
import os

import numpy as np
import pytest

import oracle_lib as O
from ndt_2d_amd import dist as shard
from ndt_2d_amd import host_build_grid, search_offsets, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("size,res,count", [
    (0.05, 0.005, 21), (0.1, 0.0025, 80), (0.5, 0.05, 21), (0.2, 0.01, 40),
    (1.0, 0.02, 100), (0.5, 0.005, 200), (5.0, 0.02, 501), (math.pi, 0.005, 1257)])
def test_search_offsets_match_the_reference_loops(size, res, count):
    # SURVEY.md table T1; reference src/scan_matcher_ndt.cpp:103,117,119
    got = search_offsets(size, res)
    assert len(got) == count
    assert np.array_equal(got, O.search_offsets(size, res))


@pytest.mark.parametrize("cfg", [1, 3])
def test_host_ndt_build_is_bit_identical_to_the_oracle(cfg):
    scans = synth.map_scans(cfg)
    p = synth.matcher_params(cfg)
    cells, sx, sy, ox, oy = host_build_grid(p["ndt_resolution"], p["range_max"], scans)
    m = O.ScanMatcherNDT()
    m.initialize(**p)
    m.addScans(scans)
    assert (sx, sy) == synth.CONFIGS[cfg]["grid"] == (m.ndt.size_x, m.ndt.size_y)
    assert (ox, oy) == m.ndt.origin
    assert np.array_equal(cells, m.ndt.cells6())


def test_cfg5_grid_extent():
    # 801 x 801 cells; only the extent is checked here
    c = synth.CONFIGS[5]
    k, pitch = c["map_lattice"]
    span = (k - 1) / 2.0 * pitch
    extent = 2 * (span + c["range_max"])
    assert int(extent / 0.25 + 1) == 801


def test_host_ndt_build_on_the_reference_test_scan():
    # reference test/ndt_model_tests.cpp:191-230 through the product's host NDT
    ref = json.load(open(os.path.join(GOLDEN, "reference_ndt_model_tests.json")))["test_ndt"]
    # NDT(1.0, 10, 10, -5, -5): a pose at the origin with range_max 5 gives that extent
    cells, sx, sy, ox, oy = host_build_grid(1.0, 5.0, [(ref["scan_pose"], ref["scan_points"])])
    assert (sx, sy, ox, oy) == (11, 11, -5.0, -5.0)
    c = cells[96]
    assert c[5] == 5
    q0, q1 = 3.5 - c[0], 3.5 - c[1]
    lik = math.exp(-0.5 * (q0 * (c[2] * q0 + c[3] * q1) + q1 * (c[3] * q0 + c[4] * q1)))
    assert lik == pytest.approx(ref["likelihood"], abs=ref["tol"])


def test_min_quirk_of_add_scans_extent():
    # max_x_ starts at numeric_limits<double>::min() (reference scan_matcher_ndt.cpp:54,56):
    # a map entirely at negative coordinates still extends to ~0
    pts = np.array([[0.5, 0.0], [0.0, 0.5]])
    cells, sx, sy, ox, oy = host_build_grid(0.25, 1.0, [((-10.0, -10.0, 0.0), pts)])
    m = O.ScanMatcherNDT()
    m.initialize(ndt_resolution=0.25, range_max=1.0)
    m.addScans([((-10.0, -10.0, 0.0), pts)])
    assert (sx, sy) == (m.ndt.size_x, m.ndt.size_y) == (45, 45)
    assert (ox, oy) == (-11.0, -11.0)


def test_synth_scan_is_deterministic_and_in_room():
    w = synth.world_of(1)
    a = synth.scan(w, (0.13, -0.07, 0.031), 101)
    b = synth.scan(w, (0.13, -0.07, 0.031), 101)
    assert np.array_equal(a, b)
    r = np.hypot(a[:, 0], a[:, 1])
    assert r.min() > 1.0 and r.max() < 4.2 * math.sqrt(2)
    assert synth.pose_blocked(w, 2.0, 2.0) and not synth.pose_blocked(w, 0.0, 0.0)


def test_shard_range_partitions():
    for n in (1, 7, 200, 1257):
        for world in (1, 2, 3, 8):
            ranges = [shard.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_combine_match_records_first_wins_on_ties():
    # two slabs with the same best score: the earlier slab (lower flat index) wins,
    # as the reference's strict `<` does (scan_matcher_ndt.cpp:128)
    a = np.zeros(12); a[0] = -3.0; a[1] = 10; a[2:] = 1.0
    b = np.zeros(12); b[0] = -3.0; b[1] = 500; b[2:] = 2.0
    none = np.zeros(12); none[1] = -1.0
    s, i, acc = shard.combine_match_records([a, b, none])
    assert (s, i) == (-3.0, 10) and np.all(acc == 3.0)
    s, i, acc = shard.combine_match_records([none, none])
    assert s == 0.0 and i is None
    # interleaved shares: the tie goes to the lower flat index whatever the rank
    s, i, acc = shard.combine_match_records([b, none, a])
    assert (s, i) == (-3.0, 10)
    better = b.copy(); better[0] = -3.5
    assert shard.combine_match_records([a, better])[:2] == (-3.5, 500)


def test_shard_strided_partitions():
    for n in (1, 7, 40, 201, 1257):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                first, stride, count = shard.shard_strided(n, r, world)
                seen += [first + stride * k for k in range(count)]
            assert sorted(seen) == list(range(n))
    assert shard.decode_index(1065647, 100) == (106, 56, 47)
