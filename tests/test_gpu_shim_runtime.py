"""The pluginlib shim RUN as the node runs it: tests/cpp/shim_runtime.cpp holds the plugin
object through ndt_2d::ScanMatcherPtr, with the reference's own Scan / Pose2d / Point classes,
and calls initialize / addScans / scoreScan / matchScan / scorePoints / reset in the node's
order.  The binary is built where /root/reference exists (oracle/Makefile, _ref/shim_runtime:
the reference's src/scan.cpp compiled from where it lies; Eigen3 / rclcpp / pluginlib replaced
by tests/stubs/) and travels to the GPU box."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from ndt_2d_amd import ScanMatcherNDT, synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "shim_runtime")


def _run(tmp_path, scans, qpose, qpts, poses, params):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/shim_runtime not built (needs /root/reference at build time)")
    inp, outp = os.path.join(str(tmp_path), "in.bin"), os.path.join(str(tmp_path), "out.bin")
    sp, allpts, offsets = O._pack_scans(scans)
    with open(inp, "wb") as f:
        f.write(struct.pack("<Q", len(scans)))
        f.write(sp.astype("<f8").tobytes())
        f.write(offsets.astype("<u8").tobytes())
        f.write(allpts.astype("<f8").tobytes())
        f.write(np.asarray(qpose, "<f8").tobytes())
        f.write(struct.pack("<Q", len(qpts)))
        f.write(np.asarray(qpts, "<f8").tobytes())
        f.write(struct.pack("<Q", len(poses)))
        f.write(np.asarray(poses, "<f8").tobytes())
    args = ["%s=%s" % kv for kv in params.items()]
    r = subprocess.run([EXE, inp, outp] + args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    v = np.fromfile(outp, dtype="<f8")
    n = len(poses)
    return dict(score_scan=v[0], match=v[1], pose=v[2:5], cov=v[5:14].reshape(3, 3),
                score_points=v[14:14 + n], batch=v[14 + n:14 + 2 * n], after_reset=v[14 + 2 * n])


def _yaml(name, p):
    """The node's parameter overrides for matcher `name` (reference src/scan_matcher_ndt.cpp:37-44)."""
    out = {"range_max": repr(p["range_max"])}
    for k in ("ndt_resolution", "search_angular_resolution", "search_angular_size",
              "search_linear_resolution", "search_linear_size", "laser_max_beams"):
        out["%s.%s" % (name, k)] = repr(p[k])
    return out


def test_plugin_object_with_the_reference_defaults(tmp_path):
    """The six declared parameters left at the reference's defaults (100 beams, 21 x 21 x 80
    candidates): every return value against the Python mirror of the same library (bit for
    bit) and the CPU oracle."""
    scans = synth.map_scans(1)
    guess, pts, _ = synth.query_scan(1)
    qpose = (0.11, -0.05, 0.02)
    poses = synth.particles(3, 64)
    poses[:, :2] *= 4.0 / 23.0
    got = _run(tmp_path, scans, qpose, pts, poses, {"range_max": "4.75"})
    p = synth.matcher_params(1, search_linear_size=0.05, search_linear_resolution=0.005,
                             search_angular_size=0.1, search_angular_resolution=0.0025, laser_max_beams=100)
    m = ScanMatcherNDT(0)
    m.initialize("global_scan_matcher", **p)
    m.addScans(scans)
    ref = O.ScanMatcherNDT()
    ref.initialize(**p)
    ref.addScans(scans)
    want = m.matchScan(qpose, pts)
    exp = ref.matchScan(qpose, pts)
    assert got["match"] == want["score"] and np.array_equal(got["pose"], want["pose"])
    assert np.array_equal(got["cov"], want["covariance"])
    assert np.array_equal(got["pose"], exp["pose"]) and abs(got["match"] - exp["score"]) < 1e-12
    assert np.allclose(got["cov"], exp["covariance"], rtol=1e-9, atol=0)
    # single poses: scored on the host, the oracle's bits
    assert got["score_scan"] == ref.scoreScan(qpose, pts)
    assert np.array_equal(got["score_points"], np.array([ref.scorePoints(pts, q) for q in poses]))
    # the batched interface (one launch) agrees with them to rounding
    assert float(np.max(np.abs(got["batch"] - got["score_points"]))) < 1e-12
    assert got["after_reset"] == 0.0


def test_plugin_object_over_three_device_contexts(tmp_path):
    """<name>.device_ids = [0, 0, 0], <name>.exchange = host and a loop-closure-sized search
    (cfg-2's lattice): the unchanged matchScan call is dealt to three contexts."""
    scans = synth.map_scans(2)
    guess, pts, _ = synth.query_scan(2)
    p = synth.matcher_params(2)
    poses = synth.particles(3, 16)
    poses[:, :2] *= 4.0 / 23.0
    over = _yaml("global_scan_matcher", p)
    over["global_scan_matcher.device_ids"] = "0,0,0"
    over["global_scan_matcher.exchange"] = "host"
    got = _run(tmp_path, scans, guess, pts, poses, over)
    m = ScanMatcherNDT(0)
    m.initialize("global_scan_matcher", **p)
    m.addScans(scans)
    want = m.matchScan(guess, pts)
    assert want["best_index"] == 1065647
    assert got["match"] == want["score"] and np.array_equal(got["pose"], want["pose"])
    assert np.allclose(got["cov"], want["covariance"], rtol=1e-9, atol=0)
    assert float(np.max(np.abs(got["score_points"] - m.scorePoses(pts, poses)))) < 1e-12
