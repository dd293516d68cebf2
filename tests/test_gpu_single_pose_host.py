"""scorePoints / scoreScan of ONE pose are served on the host from the host NDT
(ndt2d_matcher_set_single_pose_path, include/ndt2d_hip.h) -- the unchanged
ParticleFilter::measure calls scorePoints once per particle (reference
src/particle_filter.cpp:81-87).  That path restates the reference's arithmetic with
libm's exp, so it must give the oracle's bits; the device path (score_few_kernel) gives
the same within rounding of the device exp."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(cfg, **override):
    import oracle_lib as O
    from ndt_2d_amd import ScanMatcherNDT, synth
    p = synth.matcher_params(cfg, **override)
    scans = synth.map_scans(cfg)
    gpu = ScanMatcherNDT(0)
    gpu.initialize("m", **p)
    gpu.addScans(scans)
    ref = O.ScanMatcherNDT()
    ref.initialize(**p)
    ref.addScans(scans)
    return gpu, ref, scans


@pytest.mark.parametrize("build", ["host", "device"])
def test_host_single_pose_path_gives_the_oracles_bits(build):
    from ndt_2d_amd import ScanMatcherNDT, synth
    import oracle_lib as O
    p = synth.matcher_params(1, laser_max_beams=100)
    scans = synth.map_scans(1)
    gpu = ScanMatcherNDT(0)
    gpu.initialize("m", **p)
    gpu.set_build_mode(build)
    gpu.addScans(scans)
    ref = O.ScanMatcherNDT()
    ref.initialize(**p)
    ref.addScans(scans)
    _, pts, _ = synth.query_scan(1)
    poses = synth.particles(3, 400)
    poses[:, :2] *= 4.0 / 23.0
    host = np.array([gpu.scorePoints(pts, q) for q in poses])
    want = np.array([ref.scorePoints(pts, q) for q in poses])
    assert np.array_equal(host, want)          # same operations, same libm: bit for bit
    assert np.count_nonzero(want) > 100
    gpu.set_single_pose_path("device")
    dev = np.array([gpu.scorePoints(pts, q) for q in poses])
    assert gpu.last_variant().startswith("poses/block-per-pose")
    assert float(np.max(np.abs(dev - want))) < 1e-12
    # scoreScan = scorePoints(scan points, scan pose)
    gpu.set_single_pose_path("host")
    for q in poses[:20]:
        assert gpu.scoreScan(q, pts) == ref.scoreScan(q, pts)


def test_long_scans_and_batches_stay_on_the_device():
    from ndt_2d_amd import synth
    gpu, ref, _ = _pair(1)                      # laser_max_beams = 720 > 256
    _, pts, _ = synth.query_scan(1)
    q = (0.1, -0.05, 0.02)
    got = gpu.scorePoints(pts, q)
    assert gpu.last_variant().startswith("poses/block-per-pose")
    assert abs(got - ref.scorePoints(pts, q)) < 1e-12
    gpu.set_single_pose_path("host", 1024)      # the threshold is a parameter
    assert gpu.scorePoints(pts, q) == ref.scorePoints(pts, q)
    poses = np.tile(np.array(q), (64, 1))
    s = gpu.scorePoses(pts, poses)              # a batch: never the host
    assert gpu.last_variant().startswith("poses/")
    assert float(np.max(np.abs(s - ref.scorePoints(pts, q)))) < 1e-12


def test_edge_cases_follow_the_reference():
    from ndt_2d_amd import ScanMatcherNDT, synth
    gpu, ref, _ = _pair(1, laser_max_beams=100)
    _, pts, _ = synth.query_scan(1)
    # a pose that throws every beam off the grid: -0.0 summed 100 times over 100
    far = (100.0, 100.0, 0.3)
    assert gpu.scorePoints(pts, far) == ref.scorePoints(pts, far)
    # points on the lower edge / just below it, upper edge (getIndex, src/ndt_model.cpp:203-218)
    edge = np.array([[-5.0, -5.0], [np.nextafter(-5.0, -10.0), 0.0], [5.0, 5.0], [5.25, 0.0], [0.0, 5.2499]])
    ident = (0.0, 0.0, 0.0)
    assert gpu.scorePoints(edge, ident) == ref.scorePoints(edge, ident)
    # NaN beam: propagates as in the reference
    nan_pts = pts.copy()
    nan_pts[3, 0] = np.nan
    a, b = gpu.scorePoints(nan_pts, ident), ref.scorePoints(nan_pts, ident)
    assert (np.isnan(a) and np.isnan(b)) or a == b
    # no NDT: 0.0
    empty = ScanMatcherNDT(0)
    empty.initialize("e", **synth.matcher_params(1, laser_max_beams=100))
    assert empty.scorePoints(pts, ident) == 0.0
    # reset() takes the host copy away too
    gpu.reset()
    assert gpu.scorePoints(pts, ident) == 0.0


def test_mapper_cycle_with_host_scoring_still_launches_the_search_ahead():
    """reset + addScans + scoreScan + matchScan (reference src/ndt_mapper.cpp:508-515): once the
    pair has been seen, scoreScan launches the scan's search and scores on the host meanwhile."""
    from ndt_2d_amd import synth
    gpu, ref, scans = _pair(1, laser_max_beams=100, search_linear_size=0.05, search_linear_resolution=0.005,
                            search_angular_size=0.1, search_angular_resolution=0.0025)
    w = synth.world_of(1)
    for k in range(6):
        pose = (0.1 + 0.01 * k, -0.05, 0.02)
        pts = synth.scan(w, (0.13 + 0.01 * k, -0.07, 0.031), 900 + k)
        gpu.reset()
        gpu.addScans(scans)
        s = gpu.scoreScan(pose, pts)
        assert s == ref.scoreScan(pose, pts)
        got = gpu.matchScan(pose, pts)
        exp = ref.matchScan(pose, pts)
        assert got["best_index"] == exp["best_index"]
        assert np.array_equal(got["pose"], exp["pose"])
        assert abs(got["score"] - exp["score"]) < 1e-12
    launched, collected = gpu.search_ahead_stats()
    assert launched >= 4 and collected == launched


def test_a_large_device_built_grid_is_not_fetched_for_single_poses():
    """A grid built on the DEVICE (maps of 73,728 points and more) has no host copy; fetching one
    for a single pose would move the whole dense grid over PCIe (cfg-5: 801 x 801 cells, 31 MB) to
    save a 30 us launch.  Above 65,536 cells the single-pose calls take the device path instead --
    same score within the device exp's rounding -- while the cfg-3 grid (40,401 cells) is fetched
    once and scored on the host, bit for bit the oracle's."""
    import oracle_lib as O
    from ndt_2d_amd import ScanMatcherNDT, synth
    for cfg, on_host in ((3, True), (5, False)):
        p = synth.matcher_params(cfg, laser_max_beams=100)
        scans = synth.map_scans(cfg)
        gpu = ScanMatcherNDT(0)
        gpu.initialize("m", **p)
        gpu.set_build_mode("device")
        gpu.addScans(scans)
        ref = O.ScanMatcherNDT()
        ref.initialize(**p)
        ref.addScans(scans)
        guess, pts, _ = synth.query_scan(cfg)
        poses = synth.particles(cfg, 64)
        got = np.array([gpu.scorePoints(pts, q) for q in poses] + [gpu.scoreScan(guess, pts)])
        want = np.array([ref.scorePoints(pts, q) for q in poses] + [ref.scoreScan(guess, pts)])
        assert np.count_nonzero(want) > 3
        if on_host:
            assert np.array_equal(got, want), cfg
        else:
            assert gpu.last_variant().startswith("poses/block-per-pose"), gpu.last_variant()
            assert float(np.max(np.abs(got - want))) < 1e-12, cfg
        gpu.close()
