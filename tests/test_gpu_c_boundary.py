"""Nothing unwinds through the C boundary (csrc/ndt2d_guard.h; SURVEY.md 8b "Errors": the
plugin's callers -- rclcpp's executor threads -- expect no exception): inputs that would size a
container beyond memory come back as a status, the matcher stays usable, and the next valid
call gives the oracle's answer."""
import numpy as np
import pytest

import oracle_lib as O
from ndt_2d_amd import Ndt2dError, ScanMatcherNDT, _capi, synth




@pytest.fixture()
def matcher():
    m = ScanMatcherNDT(0)
    yield m
    m.close()


def _expect(code, fn):
    with pytest.raises(Ndt2dError) as e:
        fn()
    assert e.value.code == code, e.value
    return e.value


@pytest.mark.gpu
@pytest.mark.parametrize("bad_pose", [(1e15, 0.0, 0.0), (0.0, -1e15, 0.0), (float("nan"), 0.0, 0.0),
                                      (0.0, float("inf"), 0.0), (0.0, 0.0, float("nan"))])
def test_add_scans_with_an_absurd_pose_is_a_status_and_the_matcher_lives(matcher, bad_pose):
    p = synth.matcher_params(1)
    matcher.initialize("guard", **p)
    scans = synth.map_scans(1)
    bad = list(scans) + [(bad_pose, scans[0][1])]
    _expect(_capi.ERR_INVALID, lambda: matcher.addScans(bad))
    # no NDT after the failed call: the reference's "no map yet" answers (src/scan_matcher_ndt.cpp:80,159)
    guess, pts, _ = synth.query_scan(1)
    assert matcher.scoreScan(guess, pts) == 0.0
    # ... and a valid map afterwards searches to the oracle's result
    matcher.addScans(scans)
    got = matcher.matchScan(guess, pts)
    ref = O.ScanMatcherNDT()
    ref.initialize(**p)
    ref.addScans(scans)
    exp = ref.matchScan(guess, pts)
    assert got["best_index"] == exp["best_index"] and abs(got["score"] - exp["score"]) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("range_max", [float("inf"), float("nan"), -float("inf")])
def test_initialize_refuses_a_non_finite_range_max(matcher, range_max):
    p = dict(synth.matcher_params(1), range_max=range_max)
    _expect(_capi.ERR_INVALID, lambda: matcher.initialize("guard", **p))


@pytest.mark.gpu
def test_a_range_max_that_needs_2_to_the_31_cells_is_refused_before_allocating(matcher):
    p = dict(synth.matcher_params(1), range_max=1.0e7)       # 8e7 x 8e7 cells
    matcher.initialize("guard", **p)
    _expect(_capi.ERR_INVALID, lambda: matcher.addScans(synth.map_scans(1)))
    for mode in ("device", "host"):
        matcher.set_build_mode(mode)
        _expect(_capi.ERR_INVALID, lambda: matcher.addScans(synth.map_scans(1)))


@pytest.mark.gpu
@pytest.mark.parametrize("override", [
    dict(search_linear_size=5.0, search_linear_resolution=1e-5, search_angular_size=0.5,
         search_angular_resolution=0.005),                    # 10^6 x 10^6 x 200 = 2 x 10^14 candidates
    dict(search_linear_size=float("inf")), dict(search_angular_resolution=float("nan")),
    dict(search_angular_size=1e6, search_angular_resolution=1e-6),
    dict(search_linear_size=1.0, search_linear_resolution=1e-300)])
def test_a_lattice_beyond_any_search_is_refused_at_initialize(matcher, override):
    """2^40 candidates and more: the offsets alone would fill the host's memory (the reference's
    loops, src/scan_matcher_ndt.cpp:103,117,119, would simply never return)."""
    p = dict(synth.matcher_params(1), **override)
    _expect(_capi.ERR_INVALID, lambda: matcher.initialize("guard", **p))
    # the matcher keeps the parameters it had and still works
    good = synth.matcher_params(1)
    matcher.initialize("guard", **good)
    matcher.addScans(synth.map_scans(1))
    guess, pts, _ = synth.query_scan(1)
    assert matcher.matchScan(guess, pts)["best_index"] is not None


def test_every_multi_line_entry_point_is_guarded():
    """The two macros bracket every extern "C" body of the host and device layers (a new entry
    point without them fails here, not in production)."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ndt_2d_amd", "csrc")
    missing = []
    for name in ("ndt2d_host.cpp", "ndt2d_device.hip"):
        lines = open(os.path.join(root, name)).read().split("\n")
        for i, ln in enumerate(lines):
            m = re.match(r'^(?:extern "C" )?int (ndt2d_\w+)\(', ln)
            if not m:
                continue
            j = i
            while lines[j] != "{" and "{" not in lines[j]:
                j += 1
            if lines[j] != "{":
                continue                                  # a one-line accessor
            k = j + 1
            while lines[k] != "}":
                k += 1
            body = "\n".join(lines[j:k])
            if "NDT2D_C_TRY" not in body and "catch (...)" not in body:
                missing.append(m.group(1))
    assert not missing, missing
