"""ndt_2d::Graph's bag format (alternative): the reference's own
read_write_test scenario (reference test/graph_tests.cpp:32-139) through
ndt_2d_amd.graph_io, plus the byte layout of the two CDR messages and the SQLite
schema the reference's rosbag2 writer produces.

PARITY UNPINNED against files written by the reference itself: rosbag2 / rclcpp are
absent here, so no reference-written bag exists to read; the layout follows the
published formats (see graph_io's header)."""
import math
import os
import sqlite3
import struct

import numpy as np
import pytest

from ndt_2d_amd.graph_io import (Constraint, Graph, Scan, deserialize_constraint,
                                 deserialize_scan, serialize_constraint, serialize_scan)


def _reference_graph():
    """test/graph_tests.cpp:40-96"""
    graph = Graph(True)
    scan0 = Scan(0, (0.0, 1.0, 0.0), [(2.0, 3.0), (3.0, 3.0), (4.0, 4.0)])
    graph.scans.append(scan0)
    scan1 = Scan(1, (1.0, 2.5, 0.05), [(1.0, 1.5), (2.0, 1.5), (3.0, 2.5)])
    graph.scans.append(scan1)
    info = np.zeros((3, 3))
    info[0, 0] = info[1, 1] = 100.0
    info[2, 2] = 20.0
    graph.constraints.append(Constraint(0, 1, (1.0, 1.5, 0.0), info, True))
    return graph, scan0, scan1


def test_barycenter_poses_of_the_reference_scenario():
    """test/graph_tests.cpp:55-58, 75-79, 112-116"""
    _, scan0, scan1 = _reference_graph()
    b0 = scan0.getBarycenterPose()
    assert b0[0] == 3.0 and b0[1] == (10 / 3.0) + 1 and b0[2] == 0.0
    c_x = math.cos(0.05) * 2.0 - math.sin(0.05) * (5.5 / 3.0) + 1.0
    c_y = math.sin(0.05) * 2.0 + math.cos(0.05) * (5.5 / 3.0) + 2.5
    b1 = scan1.getBarycenterPose()
    assert b1[0] == pytest.approx(c_x, rel=4e-16) and b1[1] == pytest.approx(c_y, rel=4e-16)
    assert b1[2] == 0.05


def test_find_nearest_of_the_reference_scenario():
    """test/graph_tests.cpp:98-121"""
    graph, _, _ = _reference_graph()
    scan2 = Scan(2, (1.0, 2.3, 0.05), [(1.0, 1.5), (2.0, 1.5), (3.0, 2.5)])
    near = graph.findNearest(scan2)
    assert len(near) == 2 and near[0] == 0 and near[1] == 1
    # nearest first (nanoflann sorts): squared distances 0.0193 (scan0) and 0.2^2 = 0.04
    # (scan1) -- and the radius is compared with the SQUARED distance (L2_Simple_Adaptor)
    assert graph.findNearest(scan2, dist=0.05) == [0, 1]
    assert graph.findNearest(scan2, dist=0.03) == [0]
    assert graph.findNearest(scan2, dist=0.01) == []
    assert graph.findNearest(scan2, limit_scan_index=1) == [0]
    assert Graph(False).findNearest(scan2) == []


def test_read_write_round_trip(tmp_path):
    """test/graph_tests.cpp:98-139"""
    graph, _, _ = _reference_graph()
    bag = str(tmp_path / "test_graph")
    assert graph.save(bag) is True
    new_graph = Graph(True, bag)
    assert len(new_graph.scans) == 2
    assert len(new_graph.scans[0].points) == 3 and len(new_graph.scans[1].points) == 3
    assert len(new_graph.constraints) == 1
    c = new_graph.constraints[0]
    assert (c.begin, c.end) == (0, 1)
    assert tuple(c.transform) == (1.0, 1.5, 0.0)
    assert c.information[0, 0] == 100.0 and c.information[1, 1] == 100.0
    assert c.information[2, 2] == 20.0 and c.switchable is True
    # beyond the reference's checks: everything survives bit for bit
    for a, b in zip(graph.scans, new_graph.scans):
        assert a.id == b.id and np.array_equal(a.pose, b.pose) and np.array_equal(a.points, b.points)
    assert np.array_equal(c.information, graph.constraints[0].information)
    with pytest.raises(FileExistsError):
        graph.save(bag)      # the rosbag2 writer refuses an existing directory too


def test_bag_layout_is_rosbag2_sqlite3(tmp_path):
    graph, _, _ = _reference_graph()
    bag = str(tmp_path / "g")
    graph.save(bag)
    assert sorted(os.listdir(bag)) == ["g_0.db3", "metadata.yaml"]
    meta = open(os.path.join(bag, "metadata.yaml")).read()
    assert "storage_identifier: sqlite3" in meta and "message_count: 3" in meta
    assert "type: ndt_2d/msg/Scan" in meta and "type: ndt_2d/msg/Constraint" in meta
    import yaml
    info = yaml.safe_load(meta)["rosbag2_bagfile_information"]
    assert info["relative_file_paths"] == ["g_0.db3"]
    assert [t["message_count"] for t in info["topics_with_message_count"]] == [2, 1]
    con = sqlite3.connect(os.path.join(bag, "g_0.db3"))
    assert con.execute("SELECT name, type, serialization_format FROM topics ORDER BY id").fetchall() \
        == [("scans", "ndt_2d/msg/Scan", "cdr"), ("constraints", "ndt_2d/msg/Constraint", "cdr")]
    rows = con.execute("SELECT topic_id, timestamp FROM messages ORDER BY id").fetchall()
    assert rows == [(1, 0), (1, 0), (2, 0)]
    assert con.execute("SELECT name FROM sqlite_master WHERE type='index'").fetchall() == [("timestamp_idx",)]


def test_cdr_byte_layout():
    """Offsets counted from the end of the 4-byte encapsulation header."""
    scan = Scan(7, (1.5, -2.5, 0.25), [(1.0, 2.0), (3.0, 4.0)])
    b = serialize_scan(scan)
    assert b[:4] == b"\x00\x01\x00\x00"
    body = b[4:]
    assert struct.unpack_from("<Q", body, 0) == (7,)
    assert struct.unpack_from("<7d", body, 8) == (1.5, -2.5, 0.0, 0.0, 0.0, 0.0, 0.25)
    assert struct.unpack_from("<I", body, 64) == (2,)
    assert struct.unpack_from("<6d", body, 72) == (1.0, 2.0, 0.0, 3.0, 4.0, 0.0)   # 4 pad bytes at 68
    assert len(body) == 72 + 48
    assert len(serialize_scan(Scan(1))) == 4 + 68       # empty sequence: just the length

    c = Constraint(3, 9, (0.5, 0.25, -0.125), np.arange(9.0).reshape(3, 3), True)
    b = serialize_constraint(c)
    body = b[4:]
    assert struct.unpack_from("<2q", body, 0) == (3, 9)
    assert struct.unpack_from("<7d", body, 16) == (0.5, 0.25, -0.125, 0.0, 0.0, 0.0, 1.0)
    assert struct.unpack_from("<9d", body, 72) == tuple(np.arange(9.0))
    assert body[144] == 1 and len(body) == 145


def test_reader_accepts_big_endian_and_empty_scans():
    scan = Scan(2 ** 40 + 5, (0.1, 0.2, -3.0), [(1.25, -1.0)])
    le = serialize_scan(scan)
    body = le[4:]
    be = b"\x00\x00\x00\x00" + struct.pack(">Q7dI", *struct.unpack_from("<Q7dI", body, 0)) + b"\0" * 4 \
        + struct.pack(">3d", *struct.unpack_from("<3d", body, 72))
    for data in (le, be):
        got = deserialize_scan(data)
        assert got.id == scan.id and np.array_equal(got.pose, scan.pose)
        assert np.array_equal(got.points, scan.points)
    empty = deserialize_scan(serialize_scan(Scan(4, (1.0, 2.0, 3.0))))
    assert empty.points.shape == (0, 2) and empty.getBarycenterPose().tolist() == [1.0, 2.0, 3.0]
    with pytest.raises(ValueError):
        deserialize_scan(b"\x01\x02\x03\x04rest")
    c = deserialize_constraint(serialize_constraint(Constraint(-5, 6, (1, 2, 3), np.eye(3), False)))
    assert (c.begin, c.end, c.switchable) == (-5, 6, False) and np.array_equal(c.information, np.eye(3))


def test_large_graph_round_trip_feeds_the_matcher_input(tmp_path):
    """A synthetic map saved and loaded: the loaded scans are what addScans is given."""
    from ndt_2d_amd import synth
    scans = synth.map_scans(1)
    graph = Graph(True)
    for i, (pose, pts) in enumerate(scans):
        graph.scans.append(Scan(i, pose, pts))
    for i in range(len(scans) - 1):
        graph.constraints.append(Constraint(i, i + 1, (0.25, 0.0, 0.0), np.eye(3) * 50.0, i % 2 == 0))
    bag = str(tmp_path / "map")
    graph.save(bag)
    loaded = Graph(True, bag)
    assert len(loaded.scans) == len(scans) and len(loaded.constraints) == len(scans) - 1
    for (pose, pts), (lpose, lpts) in zip(scans, loaded.scan_tuples()):
        assert np.array_equal(np.asarray(pose, dtype=np.float64), lpose) and np.array_equal(pts, lpts)
    assert [c.switchable for c in loaded.constraints] == [i % 2 == 0 for i in range(len(scans) - 1)]
    near = loaded.findNearest(loaded.scans[4], dist=100.0)
    assert near[0] == 4 and sorted(near) == list(range(len(scans)))
