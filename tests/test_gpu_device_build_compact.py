"""A grid built on the device (addScans of many scans: a loop closure's map, csrc/ndt2d_build.hip)
carries the compacted records + cell -> record table the searches keep in LDS, as a host-installed
grid does: the same kernel forms run on both and every candidate score is the same bits."""
import numpy as np
import pytest

from ndt_2d_amd import ScanMatcherNDT, synth

pytestmark = pytest.mark.gpu

SEARCHES = {
    "default": dict(search_linear_size=0.05, search_linear_resolution=0.005, search_angular_size=0.1,
                    search_angular_resolution=0.0025, laser_max_beams=100),
    "mid_1352": dict(search_linear_size=1.0, search_linear_resolution=0.02, search_angular_size=0.02,
                     search_angular_resolution=0.005),
    "mid_6760": dict(search_linear_size=1.0, search_linear_resolution=0.02, search_angular_size=0.1,
                     search_angular_resolution=0.005),
    "large": dict(search_linear_size=1.0, search_linear_resolution=0.02, search_angular_size=0.2,
                  search_angular_resolution=0.005),
}


@pytest.mark.parametrize("name", sorted(SEARCHES))
def test_search_on_a_device_built_grid_is_the_search_on_a_host_built_one(name):
    scans = synth.map_scans(2)
    guess, pts, _ = synth.query_scan(2)
    got = {}
    for mode in ("host", "device"):
        m = ScanMatcherNDT(0)
        m.initialize(name, **synth.matcher_params(2, **SEARCHES[name]))
        m.set_build_mode(mode)
        m.addScans(scans)
        r = m.matchScan(guess, pts, want_scores=True)
        got[mode] = (r, m.last_variant(), m.grid()[0])
        m.close()
    (rh, vh, gh), (rd, vd, gd) = got["host"], got["device"]
    assert np.array_equal(gh, gd, equal_nan=True)
    assert vh == vd, (vh, vd)                       # the same kernel form (compacted records where the host's has them)
    assert rd["best_index"] == rh["best_index"] and rd["score"] == rh["score"]
    assert np.array_equal(rd["pose"], rh["pose"])
    assert np.array_equal(rd["scores"], rh["scores"])
    assert np.array_equal(rd["covariance"], rh["covariance"])


def test_large_grids_are_not_compacted():
    """65,535 cells and more: no uint16 rank table; the search still runs (records gathered)."""
    m = ScanMatcherNDT(0)
    m.initialize("big", **synth.matcher_params(5, search_linear_size=0.1, search_linear_resolution=0.02,
                                               search_angular_size=0.02, search_angular_resolution=0.005))
    m.set_build_mode("device")
    m.addScans(synth.map_scans(5))
    guess, pts, _ = synth.query_scan(5)
    r = m.matchScan(guess, pts)
    assert r["score"] < 0.0 and "compact-records" not in m.last_variant()
    m.close()
