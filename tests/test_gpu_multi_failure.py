"""One device of a multi-device call FAILS while the others are in flight (a hooks build of the
library makes the k-th launch return NDT2D_ERR_HIP): the call must return that error with every
device waited out -- no search left pending, no copy into the caller's buffers in flight, the
device threads released from their barrier -- and the next call must give the right answer on all
devices.  ndt2d_host.cpp drain_devices / first_failure, DeviceWorkers::barrier."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent("""
    import ctypes, json, sys, time
    import numpy as np
    sys.path.insert(0, %(root)r)
    from ndt_2d_amd import Ndt2dError, ScanMatcherNDT, _capi, pf_measure, synth
    hooks = ctypes.CDLL(_capi.LIB_PATH)
    out = {}
    m = ScanMatcherNDT(device_ids=[0, 0, 0, 0])
    m.set_multi_min_units(0)
    m.initialize("t", **synth.matcher_params(1))
    m.addScans(synth.map_scans(1))
    guess, pts, _ = synth.query_scan(1)
    good = m.matchScan(guess, pts, want_scores=True)
    out["variant"] = m.matcher_variant()
    faults = []
    for kth in (1, 2, 3, 4):            # the launch of each of the four devices in turn
        hooks.ndt2d_test_fail_launch(kth)
        t0 = time.perf_counter()
        try:
            m.matchScan(guess, pts)
            faults.append("no error")
        except Ndt2dError as e:
            faults.append({"code": e.code, "seconds": time.perf_counter() - t0, "message": str(e)})
        hooks.ndt2d_test_fail_launch(0)
        again = m.matchScan(guess, pts, want_scores=True)
        faults[-1] = dict(faults[-1], recovers=bool(np.array_equal(again["scores"], good["scores"])
                                                    and again["best_index"] == good["best_index"])) \\
            if isinstance(faults[-1], dict) else faults[-1]
    out["search_faults"] = faults

    parts = synth.particles(3, 20000)
    parts[:, :2] *= 4.0 / 23.0
    w_good, mean_good, cov_good = pf_measure(m, parts, pts)
    out["pf_variant"] = m.matcher_variant()
    pf_faults = []
    for kth in (1, 3):
        hooks.ndt2d_test_fail_launch(kth)
        t0 = time.perf_counter()
        try:
            pf_measure(m, parts, pts)
            pf_faults.append("no error")
        except Ndt2dError as e:
            pf_faults.append({"code": e.code, "seconds": time.perf_counter() - t0, "message": str(e)})
        hooks.ndt2d_test_fail_launch(0)
        w, mean, cov = pf_measure(m, parts, pts)
        if isinstance(pf_faults[-1], dict):
            pf_faults[-1]["recovers"] = bool(np.array_equal(w, w_good) and np.array_equal(mean, mean_good))
    out["pf_faults"] = pf_faults
    print(json.dumps(out))
""")


def test_a_failing_device_is_waited_out_and_the_matcher_recovers(tmp_path):
    from ndt_2d_amd import _capi
    from ndt_2d_amd import build as _build
    hooks_lib = _build.build_test_hooks()
    script = os.path.join(str(tmp_path), "child.py")
    with open(script, "w") as f:
        f.write(CHILD % {"root": ROOT})
    env = dict(os.environ, NDT2D_HIP_LIB=hooks_lib)
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["variant"].startswith("multi[4]/host/") and out["pf_variant"].startswith("multi[4]/host/")
    for fault in out["search_faults"] + out["pf_faults"]:
        assert isinstance(fault, dict), out
        assert fault["code"] == _capi.ERR_HIP and "injected" in fault["message"], fault
        assert fault["seconds"] < 5.0 and fault["recovers"] is True, fault
