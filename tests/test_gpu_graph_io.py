"""A map that went through the reference's graph file format drives the GPU path:
bag -> Graph -> findNearest -> addScans -> matchScan / OccupancyGrid, compared with
the oracle fed from the same loaded scans (alternative)."""
import numpy as np
import pytest

import oracle_lib as O
from ndt_2d_amd import ScanMatcherNDT, synth
from ndt_2d_amd.graph_io import Constraint, Graph, Scan
from ndt_2d_amd.occupancy_grid import OccupancyGrid

pytestmark = pytest.mark.gpu


def test_loaded_graph_drives_loop_closure_matching_and_map_rendering(tmp_path):
    scans = synth.map_scans(1)
    graph = Graph(True)
    for i, (pose, pts) in enumerate(scans):
        graph.scans.append(Scan(i, pose, pts))
        if i:
            graph.constraints.append(Constraint(i - 1, i, (0.25, 0.0, 0.0), np.eye(3), False))
    bag = str(tmp_path / "map")
    graph.save(bag)
    loaded = Graph(True, bag)

    # the loop-closure step's selection (reference src/ndt_mapper.cpp:615-635)
    guess, pts, _ = synth.query_scan(1)
    query = Scan(len(scans), guess, pts)
    near = loaded.findNearest(query, dist=100.0)
    assert len(near) == len(scans)
    chosen = loaded.scan_tuples(sorted(near)[:6])

    params = synth.matcher_params(1)
    m = ScanMatcherNDT(0)
    m.initialize("global_scan_matcher", **params)
    m.addScans(chosen)
    got = m.matchScan(guess, pts)
    om = O.ScanMatcherNDT()
    om.initialize(**params)
    om.addScans(chosen)
    want = om.matchScan(guess, pts)
    assert np.array_equal(got["pose"], want["pose"])
    assert abs(got["score"] - want["score"]) < 1e-5
    assert np.allclose(got["covariance"], want["covariance"], rtol=1e-9, atol=1e-12)

    # the same through the in-memory scans: the file format loses nothing
    m2 = ScanMatcherNDT(0)
    m2.initialize("direct", **params)
    m2.addScans([scans[i] for i in sorted(near)[:6]])
    direct = m2.matchScan(guess, pts)
    assert direct["score"] == got["score"] and np.array_equal(direct["pose"], got["pose"])

    grid = OccupancyGrid(0.05, 0.25, m).getMsg(loaded.scan_tuples())
    ref = O.OccupancyGrid(0.05, 0.25).getMsg(loaded.scan_tuples())
    assert np.array_equal(grid["data"], ref["data"])
