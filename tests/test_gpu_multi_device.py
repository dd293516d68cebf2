"""The multi-device matcher behind the C-ABI (ndt2d_matcher_create_multi): one handle,
several device contexts in one process, matchScan's theta steps dealt round-robin and
particle batches in contiguous ranges, the per-device records exchanged once (RCCL
all-reduce, or the host-coherent result blocks).  The GPU box has one GPU: the devices
are several contexts on it (host exchange; RCCL refuses a device twice), and the RCCL
path runs with one rank."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "ndt_2d_amd")


def _build(tmp_path):
    exe = os.path.join(str(tmp_path), "multi_device")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-pedantic",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "multi_device.c"),
           "-L", LIBDIR, "-lndt2d_hip", "-lm", "-Wl,-rpath," + LIBDIR, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("ids,exchange,label", [("0,0,0", "host", "multi[3]/host/"),
                                                ("0,0", "auto", "multi[2]/host/"),
                                                ("0", "rccl", "multi[1]/rccl/")])
def test_c_host_multi_device_equals_single_device(tmp_path, ids, exchange, label):
    """cfg-1, cfg-2 (winner 1065647) and cfg-4 (winner 80443810) through a plain-C host:
    score, index and pose bit for bit the single-device ones, covariance within 1e-9."""
    exe = _build(tmp_path)
    r = subprocess.run([exe, ids, exchange], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] is True
    assert out["cfg2"]["best_index"] == 1065647 and out["cfg2"]["n_candidates"] == 2000000
    assert out["cfg4"]["best_index"] == 80443810 and out["cfg4"]["n_candidates"] == 315508257
    for c in ("cfg1", "cfg2", "cfg4"):
        assert out[c]["same_winner_score_pose"] is True
        assert out[c]["variant"].startswith(label)
    assert out["pf_measure"]["variant"].startswith(label)


def test_c_host_cfg5_is_sharded_by_the_default_thresholds(tmp_path):
    """BASELINE.json configs[4] -- 1,000,000 particles x 720 beams, 801 x 801 NDT -- through
    eight contexts and the host exchange WITHOUT touching the thresholds (the reference's call
    site: src/ndt_mapper.cpp:474 -> src/particle_filter.cpp:78-89): the call is dealt out, raw
    scores are bit for bit the single-device ones, weights and statistics agree to rounding."""
    exe = _build(tmp_path)
    r = subprocess.run([exe, "0,0,0,0,0,0,0,0", "host", "cfg5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] is True and out["devices"] == 8
    c = out["cfg5"]
    assert c["grid"] == [801, 801] and c["particles"] == 1000000
    assert c["units"] >= c["default_min_pose_units"]
    assert c["raw_scores_bit_identical"] is True
    assert c["max_rel_weight_diff"] < 1e-12
    assert c["variant"].startswith("multi[8]/host/") and c["score_poses_variant"].startswith("multi[8]/host/")


def test_multi_device_all_scores_against_the_oracle():
    """Every one of cfg-1's 17,640 scores from three contexts (theta steps interleaved)
    against the CPU oracle, and the golden result."""
    import oracle_lib as O
    from ndt_2d_amd import ScanMatcherNDT, synth
    m = ScanMatcherNDT(device_ids=[0, 0, 0])
    assert m.device_count() == 3
    m.set_multi_min_units(0)
    m.initialize("global_scan_matcher", **synth.matcher_params(1))
    scans = synth.map_scans(1)
    m.addScans(scans)
    guess, pts, _ = synth.query_scan(1)
    got = m.matchScan(guess, pts, want_scores=True)
    assert m.matcher_variant().startswith("multi[3]/host/")
    ref = O.ScanMatcherNDT()
    ref.initialize(**synth.matcher_params(1))
    ref.addScans(scans)
    exp = ref.matchScan(guess, pts, want_scores=True)
    assert got["n_candidates"] == exp["n_candidates"] == 17640
    assert float(np.max(np.abs(got["scores"] - exp["scores"]))) < 1e-9
    assert got["best_index"] == exp["best_index"]
    assert np.array_equal(got["pose"], exp["pose"])
    assert abs(got["score"] - exp["score"]) < 1e-12
    assert np.allclose(got["covariance"], exp["covariance"], rtol=1e-9, atol=0)
    # ... and a single-device matcher gives the same scores bit for bit (same kernels, same steps)
    one = ScanMatcherNDT(0)
    one.initialize("one", **synth.matcher_params(1))
    one.addScans(scans)
    single = one.matchScan(guess, pts, want_scores=True)
    assert np.array_equal(single["scores"], got["scores"])


def test_more_devices_than_theta_steps_and_small_calls_stay_on_the_first_device():
    from ndt_2d_amd import ScanMatcherNDT, synth
    scans = synth.map_scans(1)
    guess, pts, _ = synth.query_scan(1)
    # 3 theta steps on 5 contexts: two devices have nothing to search
    p = synth.matcher_params(1, search_angular_size=0.015, search_angular_resolution=0.01)
    m = ScanMatcherNDT(device_ids=[0] * 5)
    m.set_multi_min_units(0)
    m.initialize("few_steps", **p)
    m.addScans(scans)
    got = m.matchScan(guess, pts, want_scores=True)
    one = ScanMatcherNDT(0)
    one.initialize("one", **p)
    one.addScans(scans)
    want = one.matchScan(guess, pts, want_scores=True)
    assert got["n_candidates"] == want["n_candidates"] == 3 * 21 * 21
    assert np.array_equal(got["scores"], want["scores"])
    assert got["best_index"] == want["best_index"] and got["score"] == want["score"]
    # the default threshold: a node-sized search is not dealt out
    d = ScanMatcherNDT(device_ids=[0, 0])
    d.initialize("default", **synth.matcher_params(1, laser_max_beams=100))
    d.addScans(scans)
    res = d.matchScan(guess, pts)
    assert not d.matcher_variant().startswith("multi[")
    assert res["best_index"] == one_default(scans).matchScan(guess, pts)["best_index"]
    assert d.scorePoints(pts, guess) == one_default(scans).scorePoints(pts, guess)


def one_default(scans):
    from ndt_2d_amd import ScanMatcherNDT, synth
    m = ScanMatcherNDT(0)
    m.initialize("default1", **synth.matcher_params(1, laser_max_beams=100))
    m.addScans(scans)
    return m


def test_multi_device_particle_measure_against_the_oracle():
    """ParticleFilter::measure on three contexts: contiguous ranges, one exchange of the
    moment sums, the theta variance with a second -- against the CPU oracle."""
    import oracle_lib as O
    from ndt_2d_amd import ScanMatcherNDT, pf_measure, synth
    scans = synth.map_scans(1)
    _, pts, _ = synth.query_scan(1)
    parts = synth.particles(3, 30001)
    parts[:, :2] *= 4.0 / 23.0
    for ids, exchange in (([0, 0, 0], "host"), ([0], "rccl")):
        m = ScanMatcherNDT(device_ids=ids)
        m.set_exchange(exchange)
        m.set_multi_min_units(0)
        m.initialize("pf", **synth.matcher_params(1))
        m.addScans(scans)
        w, mean, cov = pf_measure(m, parts, pts, cov_prev=np.diag([0.0, 0.0, 0.25]))
        assert m.matcher_variant().startswith("multi[%d]/%s/" % (len(ids), exchange))
        ref = O.ScanMatcherNDT()
        ref.initialize(**synth.matcher_params(1))
        ref.addScans(scans)
        w_ref = O.pf_measure(ref, parts, pts)
        w_ref_n, mean_ref, cov_ref = O.pf_update_statistics(parts, w_ref, cov_prev=np.diag([0.0, 0.0, 0.25]))
        assert float(np.max(np.abs(w * w_ref.sum() - w_ref))) < 1e-5
        assert np.allclose(w, w_ref_n, rtol=1e-9, atol=1e-15)
        assert np.allclose(mean, mean_ref, rtol=1e-9, atol=1e-12)
        assert np.allclose(cov, cov_ref, rtol=1e-8, atol=1e-12)
        # scorePoses dealt out: the un-normalised scores
        s = m.scorePoses(pts, parts)
        assert float(np.max(np.abs(s - w_ref))) < 1e-9


def test_bad_device_lists_are_refused_cleanly():
    import ctypes as C
    from ndt_2d_amd import _capi
    L = _capi.lib()
    m = C.c_void_p()
    ids = (C.c_int * 2)(0, 99)                 # the second device does not exist: the first context is released
    assert L.ndt2d_matcher_create_multi(C.byref(m), ids, 2) == _capi.ERR_INVALID and not m.value
    assert L.ndt2d_matcher_create_multi(C.byref(m), ids, 0) == _capi.ERR_INVALID
    assert L.ndt2d_matcher_create_multi(C.byref(m), None, 2) == _capi.ERR_INVALID
    one = (C.c_int * 1)(0)
    assert L.ndt2d_matcher_create_multi(C.byref(m), one, 1) == _capi.OK
    assert L.ndt2d_matcher_device_count(m) == 1 and L.ndt2d_matcher_device_at(m, 1) is None
    assert L.ndt2d_matcher_set_exchange(m, b"bogus") == _capi.ERR_INVALID
    assert L.ndt2d_matcher_destroy(m) == _capi.OK
