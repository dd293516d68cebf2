"""The bounded polls of the kernels that wait for their own blocks -- the reducing block of the
small-lattice search (ndt2d_match_small.hip) and the last block of score_few_kernel
(ndt2d_poses_compact.hip) -- are FORCED to trip: a build of the library with -DNDT2D_TEST_HOOKS
(python -m ndt_2d_amd.build --test-hooks; test infrastructure, never loaded by the package)
lets a test tell one producer to withhold its `done` word.  The call must then return
NDT2D_ERR_HIP within the bound -- no trap, no hang -- and the context must stay usable: the next
call gives the result the call before the fault gave."""
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
    from ndt_2d_amd import Ndt2dError, ScanMatcherNDT, _capi, synth
    hooks = ctypes.CDLL(_capi.LIB_PATH)
    out = {"lib": _capi.LIB_PATH}
    m = ScanMatcherNDT(0)
    m.initialize("t", **synth.matcher_params(1, laser_max_beams=100, search_linear_size=0.05,
                                             search_linear_resolution=0.005, search_angular_size=0.1,
                                             search_angular_resolution=0.0025))
    m.addScans(synth.map_scans(1))
    guess, pts, _ = synth.query_scan(1)

    # -- the small-lattice search: one record's `done` word withheld
    good = m.matchScan(guess, pts)
    out["search_variant"] = m.last_variant()
    assert hooks.ndt2d_test_drop_done_small(3) == 0
    t0 = time.perf_counter()
    try:
        m.matchScan(guess, pts)
        out["search_fault"] = "no error"
    except Ndt2dError as e:
        out["search_fault"] = {"code": e.code, "seconds": time.perf_counter() - t0, "message": str(e)}
    assert hooks.ndt2d_test_drop_done_small(0) == 0
    again = m.matchScan(guess, pts)
    out["search_recovers"] = bool(again["best_index"] == good["best_index"] and again["score"] == good["score"]
                                  and np.array_equal(again["pose"], good["pose"]))

    # -- the few-pose kernel: one pose's `done` word withheld (scores only, then with statistics)
    poses = synth.particles(3, 64)
    poses[:, :2] *= 4.0 / 23.0
    good_s = m.scorePoses(pts, poses)
    out["few_variant"] = m.last_variant()
    assert hooks.ndt2d_test_drop_done_few(5) == 0
    for label, call in (("few_fault", lambda: m.scorePoses(pts, poses)),
                        ("few_stats_fault", lambda: __import__("ndt_2d_amd").pf_measure(m, poses, pts))):
        t0 = time.perf_counter()
        try:
            call()
            out[label] = "no error"
        except Ndt2dError as e:
            out[label] = {"code": e.code, "seconds": time.perf_counter() - t0, "message": str(e)}
    assert hooks.ndt2d_test_drop_done_few(0) == 0
    out["few_recovers"] = bool(np.array_equal(m.scorePoses(pts, poses), good_s))
    out["search_after_few"] = bool(m.matchScan(guess, pts)["best_index"] == good["best_index"])
    print(json.dumps(out))
""")


def test_bounded_polls_trip_and_the_context_survives(tmp_path):
    import json
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
    assert out["lib"] == hooks_lib
    assert "small-lattice" in out["search_variant"], out["search_variant"]
    for key in ("search_fault", "few_fault", "few_stats_fault"):
        fault = out[key]
        assert isinstance(fault, dict), (key, fault)
        assert fault["code"] == _capi.ERR_HIP, (key, fault)
        assert "gave up" in fault["message"], (key, fault)
        assert fault["seconds"] < 30.0, (key, fault)          # the bound: half a second of the chip clock in the test build (ten in the product)
    assert out["search_recovers"] and out["few_recovers"] and out["search_after_few"], out
