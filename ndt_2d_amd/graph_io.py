"""The pose graph's file format: ndt_2d::Graph save / load (SURVEY.md 8(f) row N4's
alternative), so that maps written by the reference can be fed to the GPU path
(`ScanMatcherNDT.addScans`, `OccupancyGrid.getMsg`) instead of synthetic ones.

Mirror of the reference's Graph for the parts around the file (reference
include/ndt_2d/graph.hpp:43-131, src/graph.cpp:49-189): `Graph(use_barycenter,
filename)` reads a bag, `save(filename)` writes one, `findNearest` is the radius
search the loop-closure step uses to pick the scans it hands to addScans
(src/ndt_mapper.cpp:615-635).

The reference writes with rosbag2_cpp::Writer and rclcpp::Serialization
(src/graph.cpp:107-165); neither library is in /root/reference (ROS 2 packages
rosbag2_cpp / rosbag2_storage_default_plugins and rmw's CDR serialisation,
versions unpinned by package.xml:17-24), so this module restates their published
formats and its parity with files written by the reference itself is UNPINNED:

 * storage: a directory `<name>/` holding `metadata.yaml` and `<name>_0.db3`, an
   SQLite 3 file with the tables `topics(id, name, type, serialization_format,
   offered_qos_profiles)` and `messages(id, topic_id, timestamp, data)` plus the
   index `timestamp_idx` (rosbag2_storage_sqlite3, bag format version 4/5 as of
   ROS 2 Galactic / Humble);
 * messages: OMG CDR (XCDR1), 4-byte encapsulation header `00 01 00 00` =
   little-endian plain CDR, every primitive aligned to its size counted from the end
   of that header, sequences as uint32 length + elements.
     ndt_2d/msg/Scan       = uint64 id, geometry_msgs/Pose pose (7 float64),
                             geometry_msgs/Point[] points (3 float64 each)
     ndt_2d/msg/Constraint = int64 begin, int64 end, geometry_msgs/Transform
                             (7 float64), float64[9] information, bool switchable
   (reference msg/Scan.msg, msg/Constraint.msg).

The reader accepts both byte orders of the encapsulation header and any topic
order; like the reference it treats every topic that is not "scans" as constraints
(src/graph.cpp:82).  Only the standard library is used (sqlite3, struct).
"""
import math
import os
import sqlite3
import struct

import numpy as np

SCAN_TOPIC = ("scans", "ndt_2d/msg/Scan")
CONSTRAINT_TOPIC = ("constraints", "ndt_2d/msg/Constraint")


class Scan:
    """ndt_2d::Scan (reference include/ndt_2d/scan.hpp, src/scan.cpp:35-92): id, pose
    (x, y, theta), robot-frame points [n, 2]."""

    def __init__(self, scan_id, pose=(0.0, 0.0, 0.0), points=None):
        self.id = int(scan_id)
        self.pose = np.array(pose, dtype=np.float64)
        self.points = np.zeros((0, 2)) if points is None else \
            np.array(points, dtype=np.float64).reshape(-1, 2)

    def getBarycenterPose(self):
        """src/scan.cpp:72-90: pose + mean of the rotated points (sequential sums)."""
        x, y, th = self.pose
        if len(self.points):
            c, s = math.cos(th), math.sin(th)
            cx = cy = 0.0
            for px, py in self.points:
                cx += c * px - s * py
                cy += s * px + c * py
            x += cx / len(self.points)
            y += cy / len(self.points)
        return np.array([x, y, th])

    def as_tuple(self):
        """(pose_xyt, points[n, 2]) as ScanMatcherNDT.addScans / OccupancyGrid.getMsg take it."""
        return self.pose, self.points


class Constraint:
    """ndt_2d::Constraint (reference include/ndt_2d/constraint.hpp)."""

    def __init__(self, begin=0, end=0, transform=(0.0, 0.0, 0.0), information=None,
                 switchable=False):
        self.begin = int(begin)
        self.end = int(end)
        self.transform = np.array(transform, dtype=np.float64)
        self.information = np.zeros((3, 3)) if information is None else \
            np.array(information, dtype=np.float64).reshape(3, 3)
        self.switchable = bool(switchable)


# ---- CDR --------------------------------------------------------------------------------

class _Writer:
    def __init__(self):
        self.body = bytearray()

    def align(self, n):
        self.body.extend(b"\0" * (-len(self.body) % n))

    def put(self, fmt, *values):
        self.align(struct.calcsize(fmt[-1]))
        self.body.extend(struct.pack("<" + fmt, *values))

    def bytes(self):
        return b"\x00\x01\x00\x00" + bytes(self.body)


class _Reader:
    def __init__(self, data):
        data = bytes(data)
        if len(data) < 4 or data[0] != 0 or data[1] not in (0, 1):
            raise ValueError("not a plain-CDR message (encapsulation header %r)" % data[:4])
        self.endian = "<" if data[1] == 1 else ">"
        self.body = data[4:]
        self.at = 0

    def get(self, fmt):
        size = struct.calcsize(fmt[-1])
        self.at += -self.at % size
        values = struct.unpack_from(self.endian + fmt, self.body, self.at)
        self.at += struct.calcsize(self.endian + fmt)
        return values


def serialize_scan(scan):
    """ndt_2d/msg/Scan as Graph::save fills it (src/graph.cpp:121-135): theta travels
    in pose.orientation.w, the rest of the quaternion and z are zero."""
    w = _Writer()
    w.put("Q", scan.id)
    w.put("7d", scan.pose[0], scan.pose[1], 0.0, 0.0, 0.0, 0.0, scan.pose[2])
    w.put("I", len(scan.points))
    if len(scan.points):
        w.align(8)
        pts = np.zeros((len(scan.points), 3), dtype="<f8")
        pts[:, :2] = scan.points
        w.body.extend(pts.tobytes())
    return w.bytes()


def deserialize_scan(data):
    """src/graph.cpp:66-80"""
    r = _Reader(data)
    (scan_id,) = r.get("Q")
    pose = r.get("7d")
    (n,) = r.get("I")
    points = np.zeros((n, 2))
    if n:
        r.at += -r.at % 8
        raw = np.frombuffer(r.body, dtype=r.endian + "f8", count=3 * n, offset=r.at).reshape(n, 3)
        points = raw[:, :2].astype(np.float64)
    return Scan(scan_id, (pose[0], pose[1], pose[6]), points)


def serialize_constraint(c):
    """src/graph.cpp:144-158; geometry_msgs/Transform's rotation keeps its default
    (0, 0, 0, 1), theta travels in translation.z."""
    w = _Writer()
    w.put("2q", c.begin, c.end)
    w.put("7d", c.transform[0], c.transform[1], c.transform[2], 0.0, 0.0, 0.0, 1.0)
    w.put("9d", *c.information.reshape(9))
    w.put("?", c.switchable)
    return w.bytes()


def deserialize_constraint(data):
    """src/graph.cpp:84-101"""
    r = _Reader(data)
    begin, end = r.get("2q")
    tf = r.get("7d")
    info = r.get("9d")
    (switchable,) = r.get("?")
    return Constraint(begin, end, tf[:3], info, switchable)


# ---- bag --------------------------------------------------------------------------------

_METADATA = """rosbag2_bagfile_information:
  version: 5
  storage_identifier: sqlite3
  duration:
    nanoseconds: 0
  starting_time:
    nanoseconds_since_epoch: 0
  message_count: {count}
  topics_with_message_count:
{topics}  compression_format: ""
  compression_mode: ""
  relative_file_paths:
    - {db}
  files:
    - path: {db}
      starting_time:
        nanoseconds_since_epoch: 0
      duration:
        nanoseconds: 0
      message_count: {count}
"""

_TOPIC_ENTRY = """    - topic_metadata:
        name: {name}
        type: {type}
        serialization_format: cdr
        offered_qos_profiles: ""
      message_count: {count}
"""


def _db_files(path):
    """The .db3 files of a bag directory (or the file itself), in name order."""
    if os.path.isfile(path):
        return [path]
    files = sorted(f for f in os.listdir(path) if f.endswith(".db3"))
    if not files:
        raise FileNotFoundError("no .db3 storage file in %s" % path)
    return [os.path.join(path, f) for f in files]


class Graph:
    """ndt_2d::Graph: `scans` and `constraints`, loaded from `filename` when given."""

    def __init__(self, use_barycenter, filename=None):
        self.use_barycenter = bool(use_barycenter)
        self.scans = []
        self.constraints = []
        if filename is not None:
            self._load(filename)

    # src/graph.cpp:49-104
    def _load(self, filename):
        for db in _db_files(filename):
            con = sqlite3.connect("file:%s?mode=ro" % db, uri=True)
            try:
                names = dict(con.execute("SELECT id, name FROM topics"))
                rows = con.execute("SELECT topic_id, data FROM messages ORDER BY timestamp, id")
                for topic_id, data in rows:
                    if names.get(topic_id) == SCAN_TOPIC[0]:
                        self.scans.append(deserialize_scan(data))
                    else:
                        self.constraints.append(deserialize_constraint(data))
            finally:
                con.close()

    # src/graph.cpp:107-165
    def save(self, filename):
        os.makedirs(filename, exist_ok=False)
        base = os.path.basename(os.path.normpath(filename))
        db = base + "_0.db3"
        con = sqlite3.connect(os.path.join(filename, db))
        try:
            con.execute("CREATE TABLE topics(id INTEGER PRIMARY KEY, name TEXT NOT NULL, "
                        "type TEXT NOT NULL, serialization_format TEXT NOT NULL, "
                        "offered_qos_profiles TEXT NOT NULL)")
            con.execute("CREATE TABLE messages(id INTEGER PRIMARY KEY, topic_id INTEGER NOT NULL, "
                        "timestamp INTEGER NOT NULL, data BLOB NOT NULL)")
            con.execute("CREATE INDEX timestamp_idx ON messages (timestamp ASC)")
            groups = ((SCAN_TOPIC, [serialize_scan(s) for s in self.scans]),
                      (CONSTRAINT_TOPIC, [serialize_constraint(c) for c in self.constraints]))
            topic_id = 0
            for (name, type_name), blobs in groups:
                if not blobs:
                    continue   # the writer creates a topic at its first message
                topic_id += 1
                con.execute("INSERT INTO topics VALUES (?, ?, ?, 'cdr', '')",
                            (topic_id, name, type_name))
                # every message carries the default-constructed rclcpp::Time: 0 (:115)
                con.executemany("INSERT INTO messages(topic_id, timestamp, data) VALUES (?, 0, ?)",
                                [(topic_id, b) for b in blobs])
            con.commit()
        finally:
            con.close()
        topics = "".join(_TOPIC_ENTRY.format(name=n, type=t, count=len(b))
                         for (n, t), b in groups if b)
        with open(os.path.join(filename, "metadata.yaml"), "w") as f:
            f.write(_METADATA.format(count=len(self.scans) + len(self.constraints),
                                     topics=topics, db=db))
        return True

    # src/graph.cpp:167-189
    def findNearest(self, scan, dist=10.0, limit_scan_index=-1):
        """Indices of the scans within the search radius of `scan`, nearest first.
        nanoflann's L2_Simple_Adaptor works on SQUARED distances, so `dist` is compared
        with the squared distance, as in the reference (graph.hpp:114-115, :183)."""
        limit = limit_scan_index if limit_scan_index > 0 else len(self.scans)
        q = scan.getBarycenterPose() if self.use_barycenter else scan.pose
        found = []
        for i, s in enumerate(self.scans[:limit]):
            p = s.getBarycenterPose() if self.use_barycenter else s.pose
            d2 = (p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2
            if d2 < dist:   # RadiusResultSet::addPoint keeps dist < radius
                found.append((d2, i))
        found.sort()
        return [i for _, i in found]

    def scan_tuples(self, indices=None):
        """The scans as ScanMatcherNDT.addScans / OccupancyGrid.getMsg take them."""
        scans = self.scans if indices is None else [self.scans[i] for i in indices]
        return [s.as_tuple() for s in scans]
