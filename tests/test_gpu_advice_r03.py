"""Regression tests for the advisor's round-3 findings: contracts that were
documented but not enforced."""
import ctypes as C

import numpy as np
import pytest

from ndt_2d_amd import ScanMatcherNDT, _capi, synth

pytestmark = pytest.mark.gpu


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def test_a_cell_listed_twice_is_refused_whether_it_scores_or_not():
    """The duplicate check of the list install covered scoring cells only; a non-scoring cell
    listed twice raced in ndt2d_get_grid's scatter."""
    L = _capi.lib()
    h = C.c_void_p()
    assert L.ndt2d_create(C.byref(h), 0) == 0
    try:
        idx = np.array([5, 9, 5], np.uint32)
        rec = np.tile(np.array([0.3, 0.3, 100.0, 0.0, 100.0, 9.0]), (3, 1))
        rec[0, 5] = 2.0      # first listing of cell 5 cannot score (n < 5), the second can
        rc = L.ndt2d_set_grid_sparse(h, _u32p(idx), _capi.dptr(rec), 3, 8, 8, 0.25, 0.0, 0.0)
        assert rc == _capi.ERR_INVALID and b"listed twice" in L.ndt2d_last_error(h)
        rec[2, 5] = 3.0      # neither can
        rc = L.ndt2d_set_grid_sparse(h, _u32p(idx), _capi.dptr(rec), 3, 8, 8, 0.25, 0.0, 0.0)
        assert rc == _capi.ERR_INVALID and L.ndt2d_has_grid(h) == 0
    finally:
        L.ndt2d_destroy(h)


def test_an_open_stage_is_void_after_another_grid_call():
    """ndt2d_grid_stage_begin ... a dense ndt2d_set_grid in between ... _commit: the commit must
    fail (NDT2D_ERR_STATE) instead of installing whatever the staging buffer holds now."""
    L = _capi.lib()
    h = C.c_void_p()
    assert L.ndt2d_create(C.byref(h), 0) == 0
    try:
        pi, pc = C.POINTER(C.c_uint32)(), C.POINTER(C.c_double)()
        assert L.ndt2d_grid_stage_begin(h, 8, 8, 4, C.byref(pi), C.byref(pc)) == 0
        dense = np.zeros((64, 6))
        assert L.ndt2d_set_grid(h, _capi.dptr(dense), 8, 8, 0.25, 0.0, 0.0) == 0
        assert L.ndt2d_grid_stage_commit(h, 0, 0.25, 0.0, 0.0) == _capi.ERR_STATE
        assert L.ndt2d_has_grid(h) == 1       # the dense grid stands
        assert L.ndt2d_grid_stage_begin(h, 8, 8, 4, C.byref(pi), C.byref(pc)) == 0
        assert L.ndt2d_clear_grid(h) == 0
        assert L.ndt2d_grid_stage_commit(h, 0, 0.25, 0.0, 0.0) == _capi.ERR_STATE
    finally:
        L.ndt2d_destroy(h)


def test_a_raw_launch_on_the_device_handle_cannot_steal_the_search_launched_ahead():
    """scoreScan launches the next matchScan's search; a caller that launches or fetches on
    ndt2d_matcher_device(m) itself in between used to make that matchScan return the wrong
    search's record.  The matcher now checks the context's launch / fetch counters."""
    params = synth.matcher_params(1, search_linear_size=0.05, search_linear_resolution=0.005,
                                  search_angular_size=0.1, search_angular_resolution=0.0025, laser_max_beams=100)
    scans = synth.map_scans(1)
    guess, pts, _ = synth.query_scan(1)
    plain = ScanMatcherNDT(0)
    plain.initialize("plain", **params)
    plain.set_search_ahead(False)
    plain.addScans(scans)
    want = plain.matchScan(guess, pts)
    for path in ("host", "device"):
        m = ScanMatcherNDT(0)
        m.initialize("ahead", **params)
        m.set_single_pose_path(path)
        m.addScans(scans)
        for _ in range(3):                      # the pair is seen, searches go out ahead
            m.scoreScan(guess, pts)
            got = m.matchScan(guess, pts)
            assert got["best_index"] == want["best_index"] and got["score"] == want["score"]
        launched0, collected0 = m.search_ahead_stats()
        assert launched0 >= 1 and collected0 == launched0
        # now interfere between the two calls: another search on the raw device handle
        m.scoreScan(guess, pts)
        m.match_launch(0, 3)                    # theta steps 0..2 only: a different record
        other = m.match_fetch()
        got = m.matchScan(guess, pts)
        assert got["best_index"] == want["best_index"] and got["score"] == want["score"], path
        assert np.array_equal(got["pose"], want["pose"])
        assert other[0] <= 0.0
        launched1, collected1 = m.search_ahead_stats()
        assert launched1 == launched0 + 1 and collected1 == collected0     # dropped, not collected
