"""GPU parity for the LaserScan -> Scan conversion (SURVEY.md 8(f) row N2,
reference src/ndt_mapper.cpp:385-453) and its fusion with matchScan.

Tolerance: the kept set and the order are exact; coordinates differ by the
device sincos's last ulps (1e-12 absolute bound at these ranges)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from ndt_2d_amd import ScanMatcherNDT, _capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher():
    m = ScanMatcherNDT(0)
    m.initialize("scan", **synth.matcher_params(1))
    return m


def _case(seed, n, dirty=True):
    rng = np.random.default_rng(seed)
    ranges = rng.uniform(0.3, 12.0, size=n).astype(np.float32)
    if dirty:
        ranges[rng.random(n) < 0.05] = np.nan
        ranges[rng.random(n) < 0.05] = np.inf
        ranges[rng.random(n) < 0.05] = 40.0
    return dict(ranges=ranges, angle_min=float(rng.uniform(-3.2, -1.0)),
                angle_increment=float(rng.uniform(0.002, 0.01)), range_max=10.0,
                laser=tuple(rng.uniform(-0.3, 0.3, size=3)),
                motion=tuple(rng.uniform(-0.1, 0.1, size=3)))


@pytest.mark.parametrize("inverted", [False, True])
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 720, 1023, 1024, 1025, 1081, 5000, 70000])
def test_conversion_matches_oracle(matcher, n, inverted):
    c = _case(1000 + n, n)
    want = O.convert_scan(inverted=inverted, **c)
    got = matcher.convertScan(inverted=inverted, **c)
    assert got.shape == want.shape
    if len(want):
        assert np.max(np.abs(got - want)) < 1e-12


def test_conversion_edge_cases(matcher):
    r = np.array([1.0, np.nan, 2.0, 30.0, 3.0, np.inf, 10.0], dtype=np.float32)
    assert np.array_equal(matcher.convertScan(r, 0.0, 0.0, 10.0),
                          [[1, 0], [2, 0], [3, 0], [10, 0]])
    assert np.array_equal(matcher.convertScan(r, 0.0, 0.0, 10.0, inverted=True),
                          [[10, 0], [3, 0], [2, 0]])
    assert len(matcher.convertScan(r[:1], 0.0, 0.0, 10.0, inverted=True)) == 0
    assert len(matcher.convertScan(np.zeros(0, np.float32), 0.0, 0.1, 10.0)) == 0
    # everything filtered
    assert len(matcher.convertScan(np.full(100, np.nan, np.float32), 0.0, 0.1, 10.0)) == 0
    # nothing filtered, no motion: plain polar -> cartesian
    n = 720
    rr = np.linspace(1.0, 9.0, n).astype(np.float32)
    pts = matcher.convertScan(rr, -np.pi, 2 * np.pi / n, 10.0)
    want = O.convert_scan(rr, -np.pi, 2 * np.pi / n, 10.0)
    assert len(pts) == n and np.max(np.abs(pts - want)) < 1e-12
    assert np.allclose(np.hypot(pts[:, 0], pts[:, 1]), rr, rtol=1e-12)


def test_conversion_device_pointer_entry(matcher):
    """ndt2d_convert_scan_launch on caller-owned device buffers."""
    import torch
    c = _case(7, 2000)
    want = O.convert_scan(**c)
    s = torch.cuda.Stream()
    matcher.set_stream(s.cuda_stream)
    with torch.cuda.stream(s):
        d_r = torch.from_numpy(c["ranges"]).cuda()
        d_p = torch.zeros((2000, 2), dtype=torch.float64, device="cuda")
        d_i = torch.zeros(2, dtype=torch.float64, device="cuda")
    s.synchronize()
    L = _capi.lib()
    desc = matcher._laser_scan(c["angle_min"], c["angle_increment"], c["range_max"], False,
                               c["laser"], c["motion"])
    rc = L.ndt2d_convert_scan_launch(matcher.device_handle, d_r.data_ptr(), 2000, C.byref(desc),
                                     d_p.data_ptr(), d_i.data_ptr())
    assert rc == _capi.OK
    matcher.synchronize()
    info = d_i.cpu().numpy()
    assert int(info[0]) == len(want)
    got = d_p.cpu().numpy()[:len(want)]
    assert np.max(np.abs(got - want)) < 1e-12
    rmax = np.hypot(want[:, 0], want[:, 1]).max()
    assert rmax <= info[1] <= rmax * (1 + 1e-9)
    # bad arguments are rejected
    assert L.ndt2d_convert_scan_launch(matcher.device_handle, None, 10, C.byref(desc),
                                       d_p.data_ptr(), d_i.data_ptr()) == _capi.ERR_INVALID
    assert L.ndt2d_convert_scan_launch(matcher.device_handle, d_r.data_ptr(), 0, C.byref(desc),
                                       d_p.data_ptr(), d_i.data_ptr()) == _capi.ERR_INVALID


@pytest.mark.parametrize("max_beams", [100, 720, 5000])
def test_fused_beams_are_the_subsampled_points(matcher, max_beams):
    """ndt2d_set_beams_from_ranges leaves in the beam buffer exactly what
    subsampling the converted points on the host would upload
    (reference src/scan_matcher_ndt.cpp:95-96,110)."""
    import torch
    c = _case(11, 1081)
    want_pts = O.convert_scan(**c)
    use = min(max_beams, len(want_pts))
    step = float(len(want_pts)) / use
    want_beams = want_pts[[int(i * step) for i in range(use)]]
    L = _capi.lib()
    desc = matcher._laser_scan(c["angle_min"], c["angle_increment"], c["range_max"], False,
                               c["laser"], c["motion"])
    npts, nb = C.c_size_t(0), C.c_size_t(0)
    rc = L.ndt2d_set_beams_from_ranges(matcher.device_handle,
                                       c["ranges"].ctypes.data_as(C.POINTER(C.c_float)), 1081,
                                       C.byref(desc), max_beams, C.byref(npts), C.byref(nb))
    assert rc == _capi.OK
    assert npts.value == len(want_pts) and nb.value == use
    n_out = C.c_size_t(0)
    ptr = L.ndt2d_scan_points(matcher.device_handle, C.byref(n_out))
    assert n_out.value == len(want_pts) and ptr
    # score the fused beams against host-uploaded beams at a few poses: same sums
    scans = synth.map_scans(1)
    matcher.addScans(scans)
    poses = np.array([[0.0, 0.0, 0.0], [0.3, -0.2, 0.1], [-1.0, 0.5, -0.4]])
    s = torch.cuda.Stream()
    matcher.set_stream(s.cuda_stream)
    with torch.cuda.stream(s):
        d_p = torch.from_numpy(poses).cuda()
        d_s = torch.zeros(3, dtype=torch.float64, device="cuda")
    s.synchronize()
    matcher.score_poses_launch(d_p.data_ptr(), 3, d_s.data_ptr())
    matcher.synchronize()
    fused = d_s.cpu().numpy()
    ref = O.ScanMatcherNDT()
    params = synth.matcher_params(1)
    params["laser_max_beams"] = use
    ref.initialize(**params)
    ref.addScans(scans)
    want = np.array([ref.scorePoints(want_beams, p) for p in poses])
    assert np.max(np.abs(fused - want)) < 1e-9


@pytest.mark.parametrize("cfg,inverted", [(1, False), (1, True), (3, False)])
def test_match_laser_scan_equals_match_scan_on_converted_points(cfg, inverted):
    """The fused entry gives matchScan's result on the oracle-converted points."""
    scans = synth.map_scans(cfg)
    params = synth.matcher_params(cfg)
    gpu = ScanMatcherNDT(0)
    gpu.initialize("scan", **params)
    gpu.addScans(scans)
    ref = O.ScanMatcherNDT()
    ref.initialize(**params)
    ref.addScans(scans)
    guess, pts, _ = synth.query_scan(cfg)
    n = len(pts)
    ranges = np.hypot(pts[:, 0], pts[:, 1]).astype(np.float32)
    rng = np.random.default_rng(5)
    ranges[rng.random(n) < 0.03] = np.nan
    ranges[rng.random(n) < 0.03] = 1e6
    a_min, a_inc = -np.pi, 2.0 * np.pi / n
    if inverted:
        # an upside-down laser sweeps the other way: index i looks along -(a_min + i inc)
        # = the direction of original beam n-1-i when a_min = -pi + inc
        ranges = ranges[::-1].copy()
        a_min = -np.pi + a_inc
    motion = (0.02, -0.01, 0.015)
    conv = dict(angle_min=a_min, angle_increment=a_inc, range_max=params["range_max"],
                inverted=inverted, laser=(0.05, 0.0, 0.0), motion=motion)
    points = O.convert_scan(ranges, **conv)
    assert 100 < len(points) < n     # NaN, 1e6 and beams beyond range_max are dropped
    exp = ref.matchScan(guess, points, pose=[0.0, 0.0, 0.0])
    got = gpu.matchLaserScan(guess, ranges, pose=[0.0, 0.0, 0.0], **conv)
    assert got["n_points"] == len(points)
    # device-converted points differ from the oracle's by ulps: same winner, scores to 1e-9
    assert np.array_equal(got["pose"], exp["pose"])
    assert abs(got["score"] - exp["score"]) < 1e-9
    assert np.allclose(got["covariance"], exp["covariance"], rtol=1e-7, atol=0, equal_nan=True)
    # and the same as the unfused path through the C-ABI
    unf = gpu.matchScan(guess, gpu.convertScan(ranges, **conv), pose=[0.0, 0.0, 0.0])
    assert np.array_equal(unf["pose"], got["pose"]) and unf["score"] == got["score"]


def test_match_laser_scan_without_points_or_map(matcher):
    m = ScanMatcherNDT(0)
    m.initialize("scan", **synth.matcher_params(1))
    r = np.full(100, 3.0, np.float32)
    out = m.matchLaserScan([0, 0, 0], r, -1.0, 0.02, 10.0, pose=[9.0, 9.0, 9.0])
    assert out["score"] == 0.0 and np.array_equal(out["pose"], [9.0, 9.0, 9.0])   # no NDT (:80)
    m.addScans(synth.map_scans(1))
    out = m.matchLaserScan([0, 0, 0], np.full(100, np.nan, np.float32), -1.0, 0.02, 10.0,
                           pose=[9.0, 9.0, 9.0])
    assert out["n_points"] == 0 and np.isnan(out["score"])      # 0.0 / 0 (:148)
    assert np.array_equal(out["pose"], [9.0, 9.0, 9.0])
