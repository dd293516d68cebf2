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


# ---- round 6: addScans' host build adds a scan's four quarters side by side ----

def _random_scans(rng, n_scans, n_beams, spread, reach):
    scans = []
    for _ in range(n_scans):
        pose = (float(rng.uniform(-spread, spread)), float(rng.uniform(-spread, spread)),
                float(rng.uniform(-math.pi, math.pi)))
        ang = np.linspace(-math.pi, math.pi, n_beams, endpoint=False)
        r = rng.uniform(0.0, reach, n_beams)
        scans.append((pose, np.stack([r * np.cos(ang), r * np.sin(ang)], axis=1)))
    return scans


@pytest.mark.parametrize("cfg", [1, 3])
def test_quarters_side_by_side_give_the_sequential_bits(cfg):
    """HostNdt::add_scan (csrc/ndt2d_host.cpp): four chains of Cell::addPoint in flight, the points
    of a cell still in the reference's order -- every cell bit for bit the sequential loop's."""
    from ndt_2d_amd.scan_matcher import BUILD_SEQUENTIAL
    scans = synth.map_scans(cfg)
    p = synth.matcher_params(cfg)
    a = host_build_grid(p["ndt_resolution"], p["range_max"], scans)
    b = host_build_grid(p["ndt_resolution"], p["range_max"], scans, BUILD_SEQUENTIAL)
    assert a[1:] == b[1:] and np.array_equal(a[0], b[0])


@pytest.mark.parametrize("seed", range(12))
def test_quarters_that_share_cells_keep_the_order(seed):
    """Adversarial scans: ranges shorter than a cell (EVERY quarter lands in the robot's own cell
    and its neighbours), beams beyond the grid, beam counts that are no multiple of four, a
    non-power-of-two cell size (true divide in getIndex), scans too short to be cut."""
    from ndt_2d_amd.scan_matcher import BUILD_SEQUENTIAL
    rng = np.random.default_rng(600 + seed)
    n_beams = [720, 719, 37, 33, 31, 1001, 64, 5, 360, 722, 90, 128][seed]
    reach = [0.3, 6.0, 0.1, 2.0, 1.0, 12.0, 0.6, 3.0, 0.25, 40.0, 0.05, 1.5][seed]
    res = [0.25, 0.25, 0.25, 0.1, 0.25, 0.3, 0.25, 0.25, 0.5, 0.25, 0.25, 0.07][seed]
    scans = _random_scans(rng, 7, n_beams, 0.6, reach)
    if seed == 9:
        scans[3] = (scans[3][0], scans[3][1][:0])           # an empty scan among the others
    a = host_build_grid(res, 3.0, scans)
    b = host_build_grid(res, 3.0, scans, BUILD_SEQUENTIAL)
    assert a[1:] == b[1:] and np.array_equal(a[0], b[0])
    assert (a[0][:, 5] > 0).sum() > 0
    # ... and both are the oracle's
    m = O.ScanMatcherNDT()
    m.initialize(ndt_resolution=res, range_max=3.0)
    m.addScans(scans)
    assert np.array_equal(a[0], m.ndt.cells6())


def test_builds_reuse_their_storage_across_geometries():
    """The matcher rebuilds its NDT in the storage of the one before (the cell stamps of
    add_scan are never cleared: ids from a running counter): a sequence of different maps
    through ONE library instance gives what each gives alone."""
    from ndt_2d_amd.scan_matcher import BUILD_SEQUENTIAL
    rng = np.random.default_rng(77)
    for i in range(6):
        scans = _random_scans(rng, 3 + i, 200 + 40 * i, 1.0 + i, 2.0 + i)
        a = host_build_grid(0.25, 4.0 + i, scans)
        b = host_build_grid(0.25, 4.0 + i, scans, BUILD_SEQUENTIAL)
        assert np.array_equal(a[0], b[0])


# ---- round 6: nothing unwinds through the C boundary (csrc/ndt2d_guard.h) ----

@pytest.mark.parametrize("pose,range_max", [
    ((1e15, 0.0, 0.0), 4.75), ((0.0, -1e15, 0.0), 4.75), ((float("nan"), 0.0, 0.0), 4.75),
    ((0.0, float("inf"), 0.0), 4.75), ((0.0, 0.0, 0.0), float("inf")), ((0.0, 0.0, 0.0), float("nan")),
    ((0.0, 0.0, 0.0), 1e12), ((3e5, 3e5, 0.0), 3e5)])
def test_a_grid_no_memory_could_hold_is_an_error_code(pose, range_max):
    """addScans sizes its NDT by the scan poses +- range_max (reference src/scan_matcher_ndt.cpp:52-66):
    a pose of 1e15, an infinite range_max or a NaN must come back as a status -- not as a
    std::bad_alloc / std::length_error unwinding into the caller, not as a size_t cast from NaN."""
    import ctypes as C
    from ndt_2d_amd import _capi
    from ndt_2d_amd._capi import dptr
    L = _capi.lib()
    poses = np.array([[0.0, 0.0, 0.0], list(pose)], dtype=np.float64)
    pts = np.zeros((4, 2))
    off = np.array([0, 2, 4], dtype=np.uint64)
    sx, sy = C.c_uint32(0), C.c_uint32(0)
    ox, oy = C.c_double(0), C.c_double(0)
    for flags in (0, 1):
        rc = L.ndt2d_host_build_grid_ex(0.25, range_max, dptr(poses), dptr(pts), off.ctypes.data_as(C.POINTER(C.c_size_t)),
                                        2, flags, None, 0, C.byref(sx), C.byref(sy), C.byref(ox), C.byref(oy))
        assert rc == _capi.ERR_INVALID
    # ... and the library goes on working
    cells, gx, gy, _, _ = host_build_grid(0.25, 4.75, synth.map_scans(1))
    assert (gx, gy) == (41, 41) and (cells[:, 5] > 0).any()


@pytest.mark.parametrize("size,res", [(1e9, 1e-9), (float("inf"), 0.01), (1.0, 1e-300), (float("nan"), 0.1),
                                      (1.0, float("nan")), (1e308, 1.0)])
def test_a_lattice_that_would_not_end_is_an_error_code(size, res):
    import ctypes as C
    from ndt_2d_amd import _capi
    n = C.c_size_t(0)
    assert _capi.lib().ndt2d_search_offsets(size, res, None, 0, C.byref(n)) == _capi.ERR_INVALID
    assert len(search_offsets(0.05, 0.005)) == 21


def test_random_scan_sets_build_bit_for_bit_like_the_sequential_loop_and_the_oracle():
    """A slice of experiments/fuzz_host_build.py (142,850 cases without a difference in round 6): random
    scan sets, ranges that make degenerate cells (NaN information) included -- compared as bits."""
    from ndt_2d_amd.scan_matcher import BUILD_SEQUENTIAL
    rng = np.random.default_rng(66)
    for _ in range(250):
        n_scans = int(rng.integers(1, 8))
        nb = int(rng.choice([1, 7, 31, 32, 33, 100, 360, 719, 720]))
        res = float(rng.choice([0.05, 0.1, 0.25, 0.3, 1.0]))
        rmax = float(rng.choice([0.5, 2.0, 4.75]))
        spread = float(rng.choice([0.0, 0.01, 0.3, 2.0]))
        reach = float(rng.choice([0.02, 0.2, 1.0, 5.0, 30.0]))
        scans = []
        for _ in range(n_scans):
            pose = (float(rng.uniform(-spread, spread)), float(rng.uniform(-spread, spread)),
                    float(rng.uniform(-math.pi, math.pi)))
            ang = np.linspace(-math.pi, math.pi, nb, endpoint=False)
            r = [rng.uniform(0, reach, nb), np.full(nb, reach) * rng.uniform(0.99, 1.01, nb),
                 rng.choice([0.0, reach, reach * 0.5], nb)][int(rng.integers(0, 3))]
            scans.append((pose, np.stack([r * np.cos(ang), r * np.sin(ang)], axis=1)))
        a = host_build_grid(res, rmax, scans)
        b = host_build_grid(res, rmax, scans, BUILD_SEQUENTIAL)
        m = O.ScanMatcherNDT()
        m.initialize(ndt_resolution=res, range_max=rmax)
        m.addScans(scans)
        c = np.ascontiguousarray(m.ndt.cells6())
        assert a[1:] == b[1:]
        assert np.array_equal(a[0].view(np.uint64), b[0].view(np.uint64))
        assert np.array_equal(a[0].view(np.uint64), c.view(np.uint64))
