"""ctypes loader for the parity oracle (oracle/libndt2d_oracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Never imported by ndt_2d_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ORACLE_DIR, "libndt2d_oracle.so")

_dp = C.POINTER(C.c_double)
UINT64_MAX = (1 << 64) - 1


def build(force=False):
    src = [os.path.join(_ORACLE_DIR, f) for f in ("ndt2d_oracle.c", "ndt2d_oracle.h")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-B", "libndt2d_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


class OrcCell(C.Structure):
    _fields_ = [
        ("valid", C.c_int),
        ("n", C.c_double),
        ("mean", C.c_double * 2),
        ("covariance", C.c_double * 4),
        ("correlation", C.c_double * 4),
        ("information", C.c_double * 4),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    vp = C.c_void_p
    d = C.c_double
    sz = C.c_size_t

    def sig(name, res, args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args

    sig("orc_cell_init", None, [C.POINTER(OrcCell)])
    sig("orc_cell_add_point", None, [C.POINTER(OrcCell), d, d])
    sig("orc_cell_compute", None, [C.POINTER(OrcCell)])
    sig("orc_set_eigen_form", None, [C.c_int])
    sig("orc_get_eigen_form", C.c_int, [])
    sig("orc_cell_score", d, [C.POINTER(OrcCell), d, d])
    sig("orc_ndt_create", vp, [d, d, d, d, d])
    sig("orc_ndt_destroy", None, [vp])
    sig("orc_ndt_add_scan", None, [vp, d, d, d, _dp, sz])
    sig("orc_ndt_compute", None, [vp])
    sig("orc_ndt_likelihood_point", d, [vp, d, d])
    sig("orc_ndt_likelihood_points", d, [vp, _dp, sz])
    sig("orc_ndt_likelihood_scan", d, [vp, d, d, d, _dp, sz])
    sig("orc_ndt_get_index", C.c_int, [vp, d, d])
    sig("orc_ndt_size_x", sz, [vp])
    sig("orc_ndt_size_y", sz, [vp])
    sig("orc_ndt_cell_size", d, [vp])
    sig("orc_ndt_origin_x", d, [vp])
    sig("orc_ndt_origin_y", d, [vp])
    sig("orc_ndt_cells", C.POINTER(OrcCell), [vp])
    sig("orc_ndt_export_cells6", None, [vp, _dp])
    sig("orc_search_offsets", sz, [d, d, _dp, sz])
    sig("orc_matcher_create", vp, [])
    sig("orc_matcher_destroy", None, [vp])
    sig("orc_matcher_initialize", None, [vp, d, d, d, d, d, sz, d])
    sig("orc_matcher_add_scans", None, [vp, _dp, _dp, C.POINTER(sz), sz])
    sig("orc_matcher_match_scan", d,
        [vp, _dp, _dp, sz, _dp, _dp, _dp, sz, C.POINTER(sz), C.POINTER(C.c_uint64)])
    sig("orc_matcher_match_scan_omp", d, [vp, _dp, _dp, sz, _dp, _dp, C.c_int])
    sig("orc_matcher_match_scan_omp_ex", d,
        [vp, _dp, _dp, sz, _dp, _dp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int)])
    sig("orc_matcher_match_scan_omp_scores", d,
        [vp, _dp, _dp, sz, _dp, _dp, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int), _dp, sz])
    sig("orc_matcher_score_points", d, [vp, _dp, sz, _dp])
    sig("orc_matcher_score_scan", d, [vp, _dp, _dp, sz])
    sig("orc_matcher_reset", None, [vp])
    sig("orc_matcher_has_ndt", C.c_int, [vp])
    sig("orc_matcher_ndt", vp, [vp])
    sig("orc_pf_measure", None, [vp, _dp, sz, _dp, sz, _dp, C.c_int])
    sig("orc_pf_measure_omp", None, [vp, _dp, sz, _dp, sz, _dp, C.c_int])
    sig("orc_pf_update_statistics", None, [_dp, _dp, sz, _dp, _dp])
    sig("orc_motion_sample", None, [d, d, d, _dp, _dp, sz, C.POINTER(C.c_float), _dp])
    sig("orc_pf_init", None, [d, d, d, d, d, d, _dp, sz, C.POINTER(C.c_float)])
    sig("orc_normalize_angle", d, [d])
    sig("orc_shortest_angular_distance", d, [d, d])
    _lib = L
    return L


def set_eigen_form(form):
    """"eigen" (default): Eigen 3.4.0's EigenSolver transcribed; "closed": the closed form."""
    lib().orc_set_eigen_form({"eigen": 0, "closed": 1}[form])


def _arr(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


class Cell:
    """Mirror of ndt_2d::Cell (reference include/ndt_2d/ndt_model.hpp:43-65)."""

    def __init__(self):
        self.c = OrcCell()
        lib().orc_cell_init(C.byref(self.c))

    def addPoint(self, x, y):
        lib().orc_cell_add_point(C.byref(self.c), x, y)

    def compute(self):
        lib().orc_cell_compute(C.byref(self.c))

    def score(self, x, y):
        return lib().orc_cell_score(C.byref(self.c), x, y)

    @property
    def valid(self):
        return bool(self.c.valid)

    @property
    def n(self):
        return self.c.n

    @property
    def mean(self):
        return np.array(self.c.mean[:])

    @property
    def covariance(self):
        return np.array(self.c.covariance[:]).reshape(2, 2)

    @property
    def correlation(self):
        return np.array(self.c.correlation[:]).reshape(2, 2)

    @property
    def information(self):
        return np.array(self.c.information[:]).reshape(2, 2)


class _NDTView:
    def __init__(self, ptr):
        self.p = ptr

    @property
    def size_x(self):
        return lib().orc_ndt_size_x(self.p)

    @property
    def size_y(self):
        return lib().orc_ndt_size_y(self.p)

    @property
    def cell_size(self):
        return lib().orc_ndt_cell_size(self.p)

    @property
    def origin(self):
        return lib().orc_ndt_origin_x(self.p), lib().orc_ndt_origin_y(self.p)

    def getIndex(self, x, y):
        return lib().orc_ndt_get_index(self.p, x, y)

    def likelihood(self, points=None, pose=None):
        pts, pp = _arr(np.asarray(points, dtype=np.float64).reshape(-1, 2))
        if pose is None:
            return lib().orc_ndt_likelihood_points(self.p, pp, len(pts))
        return lib().orc_ndt_likelihood_scan(self.p, pose[0], pose[1], pose[2], pp, len(pts))

    def likelihood_point(self, x, y):
        return lib().orc_ndt_likelihood_point(self.p, x, y)

    def cells6(self):
        out = np.zeros((self.size_x * self.size_y, 6), dtype=np.float64)
        lib().orc_ndt_export_cells6(self.p, out.ctypes.data_as(_dp))
        return out

    def cell(self, index):
        return lib().orc_ndt_cells(self.p)[index]


class NDT(_NDTView):
    """Mirror of ndt_2d::NDT (reference include/ndt_2d/ndt_model.hpp:67-134)."""

    def __init__(self, cell_size, size_x, size_y, origin_x, origin_y):
        super().__init__(lib().orc_ndt_create(cell_size, size_x, size_y, origin_x, origin_y))

    def __del__(self):
        if getattr(self, "p", None):
            lib().orc_ndt_destroy(self.p)
            self.p = None

    def addScan(self, pose, points):
        pts, pp = _arr(np.asarray(points, dtype=np.float64).reshape(-1, 2))
        lib().orc_ndt_add_scan(self.p, pose[0], pose[1], pose[2], pp, len(pts))

    def compute(self):
        lib().orc_ndt_compute(self.p)


def search_offsets(size, res):
    n = lib().orc_search_offsets(size, res, None, 0)
    out = np.zeros(n, dtype=np.float64)
    lib().orc_search_offsets(size, res, out.ctypes.data_as(_dp), n)
    return out


class ScanMatcherNDT:
    """Mirror of ndt_2d::ScanMatcherNDT (reference scan_matcher_ndt.hpp:42-105)."""

    def __init__(self):
        self.m = lib().orc_matcher_create()
        self.params = None

    def __del__(self):
        if getattr(self, "m", None):
            lib().orc_matcher_destroy(self.m)
            self.m = None

    def initialize(self, ndt_resolution=0.25, search_angular_resolution=0.0025,
                   search_angular_size=0.1, search_linear_resolution=0.005,
                   search_linear_size=0.05, laser_max_beams=100, range_max=0.0):
        self.params = dict(ndt_resolution=ndt_resolution,
                           search_angular_resolution=search_angular_resolution,
                           search_angular_size=search_angular_size,
                           search_linear_resolution=search_linear_resolution,
                           search_linear_size=search_linear_size,
                           laser_max_beams=laser_max_beams, range_max=range_max)
        lib().orc_matcher_initialize(self.m, ndt_resolution, search_angular_resolution,
                                     search_angular_size, search_linear_resolution,
                                     search_linear_size, laser_max_beams, range_max)

    def addScans(self, scans):
        """scans: list of (pose_xyt, points[n,2])."""
        poses = np.ascontiguousarray([s[0] for s in scans], dtype=np.float64).reshape(-1, 3)
        pts = [np.asarray(s[1], dtype=np.float64).reshape(-1, 2) for s in scans]
        offsets = np.zeros(len(scans) + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum([len(p) for p in pts])
        allpts = np.ascontiguousarray(np.concatenate(pts) if pts else np.zeros((0, 2)))
        lib().orc_matcher_add_scans(
            self.m, poses.ctypes.data_as(_dp), allpts.ctypes.data_as(_dp),
            offsets.ctypes.data_as(C.POINTER(C.c_size_t)), len(scans))

    def reset(self):
        lib().orc_matcher_reset(self.m)

    @property
    def ndt(self):
        p = lib().orc_matcher_ndt(self.m)
        return _NDTView(p) if p else None

    def matchScan(self, scan_pose, points, pose=None, want_scores=False, omp_threads=None):
        """Returns dict(score, pose, covariance, [scores], n_candidates, best_index)."""
        sp, spp = _arr(scan_pose)
        pts, pp = _arr(np.asarray(points, dtype=np.float64).reshape(-1, 2))
        pose_io = np.array([0.0, 0.0, 0.0] if pose is None else pose, dtype=np.float64)
        cov = np.zeros(9, dtype=np.float64)
        if omp_threads is not None:
            best = C.c_uint64(0)
            used = C.c_int(0)
            scores, sp_ptr, cap = None, None, 0
            if want_scores:
                p = self.params
                n_th = len(search_offsets(p["search_angular_size"], p["search_angular_resolution"]))
                n_lin = len(search_offsets(p["search_linear_size"], p["search_linear_resolution"]))
                cap = n_th * n_lin * n_lin
                scores = np.zeros(cap, dtype=np.float64)
                sp_ptr = scores.ctypes.data_as(_dp)
            score = lib().orc_matcher_match_scan_omp_scores(
                self.m, spp, pp, len(pts), pose_io.ctypes.data_as(_dp),
                cov.ctypes.data_as(_dp), int(omp_threads), C.byref(best), C.byref(used), sp_ptr, cap)
            return dict(score=score, pose=pose_io, covariance=cov.reshape(3, 3), scores=scores,
                        best_index=best.value, threads_used=used.value)
        ncand = C.c_size_t(0)
        best = C.c_uint64(0)
        scores = None
        sp_ptr, cap = None, 0
        if want_scores:
            p = self.params
            n_th = len(search_offsets(p["search_angular_size"], p["search_angular_resolution"]))
            n_lin = len(search_offsets(p["search_linear_size"], p["search_linear_resolution"]))
            cap = n_th * n_lin * n_lin
            scores = np.zeros(cap, dtype=np.float64)
            sp_ptr = scores.ctypes.data_as(_dp)
        score = lib().orc_matcher_match_scan(
            self.m, spp, pp, len(pts), pose_io.ctypes.data_as(_dp), cov.ctypes.data_as(_dp),
            sp_ptr, cap, C.byref(ncand), C.byref(best))
        return dict(score=score, pose=pose_io, covariance=cov.reshape(3, 3), scores=scores,
                    n_candidates=ncand.value, best_index=best.value)

    def scorePoints(self, points, pose):
        pts, pp = _arr(np.asarray(points, dtype=np.float64).reshape(-1, 2))
        ps, psp = _arr(pose)
        return lib().orc_matcher_score_points(self.m, pp, len(pts), psp)

    def scoreScan(self, scan_pose, points):
        pts, pp = _arr(np.asarray(points, dtype=np.float64).reshape(-1, 2))
        ps, psp = _arr(scan_pose)
        return lib().orc_matcher_score_scan(self.m, psp, pp, len(pts))


def pf_measure(matcher, particles, points, copy_points=False, omp_threads=None):
    """ParticleFilter::measure loop (reference particle_filter.cpp:81-87): raw weights."""
    pa, pap = _arr(np.asarray(particles, dtype=np.float64).reshape(-1, 3))
    pts, pp = _arr(np.asarray(points, dtype=np.float64).reshape(-1, 2))
    w = np.zeros(len(pa), dtype=np.float64)
    if omp_threads is not None:
        lib().orc_pf_measure_omp(matcher.m, pap, len(pa), pp, len(pts),
                                 w.ctypes.data_as(_dp), int(omp_threads))
    else:
        lib().orc_pf_measure(matcher.m, pap, len(pa), pp, len(pts), w.ctypes.data_as(_dp),
                             1 if copy_points else 0)
    return w


def pf_update_statistics(particles, weights, cov_prev=None):
    """ParticleFilter::updateStatistics (reference particle_filter.cpp:163-218).
    Returns (normalised weights, mean[3], cov[3,3])."""
    pa, pap = _arr(np.asarray(particles, dtype=np.float64).reshape(-1, 3))
    w = np.array(weights, dtype=np.float64)
    mean = np.zeros(3)
    cov = np.zeros(9) if cov_prev is None else np.array(cov_prev, dtype=np.float64).reshape(9)
    lib().orc_pf_update_statistics(pap, w.ctypes.data_as(_dp), len(pa),
                                   mean.ctypes.data_as(_dp), cov.ctypes.data_as(_dp))
    return w, mean, cov.reshape(3, 3)


def motion_sample(dx, dy, dth, alphas, poses, z):
    """MotionModel::sample (reference motion_model.cpp:45-83) with given standard
    normals z[n, 3] (float32).  Returns (new poses[n, 3], params[6])."""
    ps = np.array(poses, dtype=np.float64).reshape(-1, 3).copy()
    zz = np.ascontiguousarray(z, dtype=np.float32).reshape(-1, 3)
    a, ap = _arr(alphas)
    params = np.zeros(6)
    lib().orc_motion_sample(dx, dy, dth, ap, ps.ctypes.data_as(_dp), len(ps),
                            zz.ctypes.data_as(C.POINTER(C.c_float)), params.ctypes.data_as(_dp))
    return ps, params


def pf_init(x, y, theta, sigma_x, sigma_y, sigma_theta, z):
    """ParticleFilter::init sampling loop (reference particle_filter.cpp:53-69)
    with given standard normals z[n, 3] (float32).  Returns poses[n, 3]."""
    zz = np.ascontiguousarray(z, dtype=np.float32).reshape(-1, 3)
    ps = np.zeros((len(zz), 3))
    lib().orc_pf_init(x, y, theta, sigma_x, sigma_y, sigma_theta, ps.ctypes.data_as(_dp),
                      len(zz), zz.ctypes.data_as(C.POINTER(C.c_float)))
    return ps


class OrcLaserScan(C.Structure):
    _fields_ = [("angle_min", C.c_float), ("angle_increment", C.c_float),
                ("range_max", C.c_double), ("inverted", C.c_int),
                ("laser_x", C.c_double), ("laser_y", C.c_double), ("laser_theta", C.c_double),
                ("motion_x", C.c_double), ("motion_y", C.c_double), ("motion_theta", C.c_double)]


def convert_scan(ranges, angle_min, angle_increment, range_max, inverted=False,
                 laser=(0.0, 0.0, 0.0), motion=(0.0, 0.0, 0.0)):
    """LaserScan -> Scan points (reference ndt_mapper.cpp:385-453).  Returns points[m, 2]."""
    r = np.ascontiguousarray(ranges, dtype=np.float32)
    d = OrcLaserScan(angle_min, angle_increment, range_max, 1 if inverted else 0,
                     laser[0], laser[1], laser[2], motion[0], motion[1], motion[2])
    out = np.zeros((max(len(r), 1), 2))
    f = lib().orc_convert_scan
    f.restype = C.c_size_t
    f.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.POINTER(OrcLaserScan), _dp]
    n = f(r.ctypes.data_as(C.POINTER(C.c_float)), len(r), C.byref(d), out.ctypes.data_as(_dp))
    return out[:n].copy()


def _pack_scans(scans):
    poses = np.ascontiguousarray([s[0] for s in scans], dtype=np.float64).reshape(-1, 3)
    pts = [np.ascontiguousarray(s[1], dtype=np.float64).reshape(-1, 2) for s in scans]
    offsets = np.zeros(len(scans) + 1, dtype=np.uint64)
    if scans:
        offsets[1:] = np.cumsum([len(p) for p in pts])
    allpts = np.ascontiguousarray(np.concatenate(pts) if pts else np.zeros((0, 2)))
    return poses, allpts, offsets


class OccupancyGrid:
    """Mirror of ndt_2d::OccupancyGrid (reference include/ndt_2d/occupancy_grid.hpp:44-74)."""

    def __init__(self, resolution, occ_thresh):
        self.resolution = resolution
        self.occ_thresh = occ_thresh
        self.bounds = np.zeros(4)        # min_x_, max_x_, min_y_, max_y_ (:37-40)
        self.num_scans = 0

    def getMsg(self, scans):
        """Returns dict(resolution, width, height, origin_x, origin_y, data[height, width])."""
        L = lib()
        scans = list(scans)
        poses, allpts, offsets = _pack_scans(scans)
        szp = C.POINTER(C.c_size_t)
        L.orc_occupancy_update_bounds.restype = None
        L.orc_occupancy_update_bounds.argtypes = [_dp, C.c_double, _dp, _dp, szp, C.c_size_t,
                                                  C.c_size_t]
        L.orc_occupancy_render.restype = None
        L.orc_occupancy_render.argtypes = [_dp, C.c_double, C.c_double, _dp, _dp, szp, C.c_size_t,
                                           C.POINTER(C.c_uint32), _dp, C.c_void_p]
        off_p = offsets.ctypes.data_as(szp)
        if len(scans) != self.num_scans:                       # :51-54
            L.orc_occupancy_update_bounds(self.bounds.ctypes.data_as(_dp), self.resolution,
                                          poses.ctypes.data_as(_dp), allpts.ctypes.data_as(_dp),
                                          off_p, self.num_scans, len(scans))
            self.num_scans = len(scans)
        wh = (C.c_uint32 * 2)()
        origin = np.zeros(2)
        args = (self.bounds.ctypes.data_as(_dp), self.resolution, self.occ_thresh,
                poses.ctypes.data_as(_dp), allpts.ctypes.data_as(_dp), off_p, len(scans), wh,
                origin.ctypes.data_as(_dp))
        L.orc_occupancy_render(*args, None)
        data = np.zeros((wh[1], wh[0]), dtype=np.int8)
        L.orc_occupancy_render(*args, data.ctypes.data_as(C.c_void_p))
        return dict(resolution=self.resolution, width=int(wh[0]), height=int(wh[1]),
                    origin_x=origin[0], origin_y=origin[1], data=data)
