"""Python-side handle on the matcher layer of libndt2d_hip.so.

`ScanMatcherNDT` mirrors the reference's ndt_2d::ScanMatcherNDT plugin object
(reference include/ndt_2d/scan_matcher_ndt.hpp:42-105): the same six methods
with the same argument meaning and error behaviour, so the parity tests read
like tests of the reference.  Every score is computed by the HIP kernels
through the C-ABI; nothing here evaluates a likelihood.
"""
import ctypes as C
import weakref

import numpy as np

from . import _capi
from ._capi import Ndt2dError, dptr

# defaults of the declared parameters, reference src/scan_matcher_ndt.cpp:37-44
DEFAULT_PARAMS = dict(ndt_resolution=0.25, search_angular_resolution=0.0025,
                      search_angular_size=0.1, search_linear_resolution=0.005,
                      search_linear_size=0.05, laser_max_beams=100)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a.reshape(shape) if shape is not None else a


def search_offsets(size, res):
    """Values visited by the reference's `for (v = -size; v < size; v += res)`."""
    L = _capi.lib()
    n = C.c_size_t(0)
    L.ndt2d_search_offsets(size, res, None, 0, C.byref(n))
    out = np.zeros(n.value, dtype=np.float64)
    if n.value:
        L.ndt2d_search_offsets(size, res, dptr(out), n.value, C.byref(n))
    return out


def _pack_scans(scans):
    poses = _f64([s[0] for s in scans], (-1, 3)) if scans else np.zeros((0, 3))
    pts = [_f64(s[1], (-1, 2)) for s in scans]
    offsets = np.zeros(len(scans) + 1, dtype=np.uint64)
    if scans:
        offsets[1:] = np.cumsum([len(p) for p in pts])
    allpts = _f64(np.concatenate(pts) if pts else np.zeros((0, 2)))
    return poses, allpts, offsets


BUILD_SEQUENTIAL = 1      # include/ndt2d_hip.h NDT2D_BUILD_SEQUENTIAL
BUILD_CLOSED_FORM = 2     # NDT2D_BUILD_CLOSED_FORM


def host_build_grid(ndt_resolution, range_max, scans, flags=0):
    """addScans' NDT build on the host only (no GPU): (cells6, size_x, size_y, ox, oy).
    flags: BUILD_SEQUENTIAL (the reference's loop as it stands instead of a scan's four
    quarters side by side: same bits), BUILD_CLOSED_FORM."""
    L = _capi.lib()
    poses, allpts, offsets = _pack_scans(scans)
    sx, sy = C.c_uint32(0), C.c_uint32(0)
    ox, oy = C.c_double(0), C.c_double(0)
    off_p = offsets.ctypes.data_as(C.POINTER(C.c_size_t))
    rc = L.ndt2d_host_build_grid_ex(ndt_resolution, range_max, dptr(poses), dptr(allpts), off_p,
                                    len(scans), flags, None, 0, C.byref(sx), C.byref(sy), C.byref(ox),
                                    C.byref(oy))
    if rc != _capi.OK:
        raise Ndt2dError(rc, "ndt2d_host_build_grid")
    cells = np.zeros((sx.value * sy.value, 6), dtype=np.float64)
    rc = L.ndt2d_host_build_grid_ex(ndt_resolution, range_max, dptr(poses), dptr(allpts), off_p,
                                    len(scans), flags, dptr(cells), len(cells), C.byref(sx), C.byref(sy),
                                    C.byref(ox), C.byref(oy))
    if rc != _capi.OK:
        raise Ndt2dError(rc, "ndt2d_host_build_grid")
    return cells, sx.value, sy.value, ox.value, oy.value


def _free_pinned(lib, state, address):
    lib.ndt2d_host_free(state["handle"], C.c_void_p(address))


class ScanMatcherNDT:
    """ndt_2d::ScanMatcherNDT over the MI355X kernels."""

    def __init__(self, device_id=0, device_ids=None):
        """device_ids (a list): one matcher over several GPUs of this process
        (ndt2d_matcher_create_multi) -- matchScan's theta steps and particle batches are
        dealt to them; a device may be named twice (several contexts on one GPU, host
        exchange)."""
        self._L = _capi.lib()
        self._m = C.c_void_p()
        if device_ids is None:
            rc = self._L.ndt2d_matcher_create(C.byref(self._m), int(device_id))
        else:
            ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
            rc = self._L.ndt2d_matcher_create_multi(C.byref(self._m), ids, len(device_ids))
        if rc != _capi.OK:
            self._m = None
            raise Ndt2dError(rc, "ndt2d_matcher_create",
                             "no usable GPU; this library has no CPU fallback")
        self.params = dict(DEFAULT_PARAMS, range_max=0.0)

    def device_count(self):
        return self._L.ndt2d_matcher_device_count(self._m)

    def set_exchange(self, mode):
        """How a multi-device matcher exchanges its per-device records: "auto", "host"
        (host-coherent result blocks, no collective) or "rccl" (one all-reduce)."""
        self._check(self._L.ndt2d_matcher_set_exchange(self._m, mode.encode()), "set_exchange")

    def set_multi_min_units(self, units):
        """Work (candidates x beams, particles x beams) below which a multi-device matcher
        stays on its first device."""
        self._check(self._L.ndt2d_matcher_set_multi_min_units(self._m, float(units)),
                    "set_multi_min_units")

    def set_multi_thresholds(self, min_search_units, min_pose_units):
        """The two thresholds apart: candidates x beams of a search (default 1e9), particles x
        beams of a pose batch (default 2e8)."""
        self._check(self._L.ndt2d_matcher_set_multi_thresholds(self._m, float(min_search_units),
                                                               float(min_pose_units)), "set_multi_thresholds")

    def multi_thresholds(self):
        a, b = _capi.C.c_double(0.0), _capi.C.c_double(0.0)
        self._check(self._L.ndt2d_matcher_get_multi_thresholds(self._m, _capi.C.byref(a), _capi.C.byref(b)),
                    "get_multi_thresholds")
        return a.value, b.value

    def last_fanout_us(self):
        """Of the last dealt call: when each device's launch had been queued (us from the call's start)."""
        out = np.zeros(64, dtype=np.float64)
        n = _capi.C.c_size_t(0)
        self._check(self._L.ndt2d_matcher_last_fanout_us(self._m, _capi.dptr(out), 64, _capi.C.byref(n)),
                    "last_fanout_us")
        return out[:min(n.value, 64)].copy()

    def matcher_variant(self):
        """ndt2d_matcher_last_variant: "multi[n]/rccl/..." when the last call was dealt out."""
        v = self._L.ndt2d_matcher_last_variant(self._m)
        return v.decode() if v else ""

    def close(self):
        if getattr(self, "_m", None):
            # host_alloc() buffers stay valid while any numpy view of them is alive: each is
            # freed by its own finalizer (with a NULL handle once this context is gone)
            state = getattr(self, "_pinned_state", None)
            if state is not None:
                state["handle"] = None
            self._L.ndt2d_matcher_destroy(self._m)
            self._m = None

    def host_alloc(self, shape):
        """float64 array in pinned, GPU-mapped host memory (ndt2d_host_alloc): the
        host-pointer entry points read / write such buffers in place over PCIe instead
        of staging and copying them.  The memory is released when the last numpy view of
        it is gone -- before or after close(), never under a live array."""
        shape = (shape,) if np.isscalar(shape) else tuple(shape)
        n = int(np.prod(shape))
        ptr = C.c_void_p()
        self._dev_check(self._L.ndt2d_host_alloc(self.device_handle, max(n, 1) * 8, C.byref(ptr)),
                        "ndt2d_host_alloc")
        if not hasattr(self, "_pinned_state"):
            self._pinned_state = {"handle": self.device_handle}
        buf = (C.c_double * max(n, 1)).from_address(ptr.value)
        # every view of the array keeps `buf` alive (numpy's base chain); the block goes with it
        weakref.finalize(buf, _free_pinned, self._L, self._pinned_state, ptr.value)
        return np.ctypeslib.as_array(buf)[:n].reshape(shape)

    def set_timing(self, enabled):
        """HIP events around every launch (last_launch_ms) on / off; the pluginlib shim
        runs with them off (~4.5 us per call)."""
        self._dev_check(self._L.ndt2d_set_timing(self.device_handle, 1 if enabled else 0),
                        "ndt2d_set_timing")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, where):
        if rc != _capi.OK:
            msg = self._L.ndt2d_matcher_last_error(self._m)
            raise Ndt2dError(rc, where, msg.decode() if msg else "")

    @property
    def device_handle(self):
        """The ndt2d_handle of the device layer (for sharded / device-pointer launches)."""
        return C.c_void_p(self._L.ndt2d_matcher_device(self._m))

    # -- ScanMatcher interface (reference include/ndt_2d/scan_matcher.hpp:42-91) --

    def initialize(self, name="scan_matcher", range_max=0.0, **params):
        """initialize(name, node, range_max): `params` stands for the node's parameters
        `<name>.ndt_resolution` etc. (reference src/scan_matcher_ndt.cpp:35-47)."""
        unknown = set(params) - set(DEFAULT_PARAMS)
        if unknown:
            raise KeyError("undeclared parameter(s): %s" % sorted(unknown))
        p = dict(DEFAULT_PARAMS)
        p.update(params)
        p["range_max"] = float(range_max)
        self.name = name
        self.params = p
        self._check(self._L.ndt2d_matcher_initialize(
            self._m, p["ndt_resolution"], p["search_angular_resolution"],
            p["search_angular_size"], p["search_linear_resolution"], p["search_linear_size"],
            int(p["laser_max_beams"]), p["range_max"]), "initialize")

    def addScans(self, scans):
        """scans: iterable of (pose_xyt, points[n, 2]) -- the [begin, end) range."""
        scans = list(scans)
        poses, allpts, offsets = _pack_scans(scans)
        self._check(self._L.ndt2d_matcher_add_scans(
            self._m, dptr(poses), dptr(allpts),
            offsets.ctypes.data_as(C.POINTER(C.c_size_t)), len(scans)), "addScans")

    def matchScan(self, scan_pose, points, pose=None, want_scores=False):
        """Returns dict(score, pose, covariance, n_candidates, best_index[, scores]).
        `pose` is the caller's pre-initialised out-parameter (default (0,0,0)); it is
        returned untouched when no candidate scores below 0, and covariance is
        None when there is no NDT (the reference leaves both untouched then)."""
        sp = _f64(scan_pose, (3,))
        pts = _f64(points, (-1, 2))
        pose_io = np.array([0.0, 0.0, 0.0] if pose is None else pose, dtype=np.float64)
        cov = np.full(9, np.nan)
        score = C.c_double(0.0)
        ncand = C.c_size_t(0)
        best = C.c_uint64(0)
        scores, sp_ptr, cap = None, None, 0
        if want_scores:
            p = self.params
            n_th = len(search_offsets(p["search_angular_size"], p["search_angular_resolution"]))
            n_lin = len(search_offsets(p["search_linear_size"], p["search_linear_resolution"]))
            cap = n_th * n_lin * n_lin
            scores = np.zeros(cap, dtype=np.float64)
            sp_ptr = dptr(scores)
        self._check(self._L.ndt2d_matcher_match_scan_ex(
            self._m, dptr(sp), dptr(pts), len(pts), dptr(pose_io), dptr(cov), C.byref(score),
            sp_ptr, cap, C.byref(ncand), C.byref(best)), "matchScan")
        has = bool(self._L.ndt2d_matcher_has_ndt(self._m))
        return dict(score=score.value, pose=pose_io,
                    covariance=cov.reshape(3, 3) if has else None,
                    n_candidates=ncand.value, best_index=best.value, scores=scores)

    @staticmethod
    def _laser_scan(angle_min, angle_increment, range_max, inverted, laser, motion):
        return _capi.LaserScan(angle_min, angle_increment, range_max, 1 if inverted else 0,
                               laser[0], laser[1], laser[2], motion[0], motion[1], motion[2])

    def convertScan(self, ranges, angle_min, angle_increment, range_max, inverted=False,
                    laser=(0.0, 0.0, 0.0), motion=(0.0, 0.0, 0.0)):
        """LaserScan -> Scan points on the device (reference src/ndt_mapper.cpp:385-453):
        ranges float32[n]; laser = laser_transform_; motion = odometry motion over the
        sweep (`translation`, :386-389).  Returns points[m, 2]."""
        r = np.ascontiguousarray(ranges, dtype=np.float32)
        d = self._laser_scan(angle_min, angle_increment, range_max, inverted, laser, motion)
        out = np.zeros((max(len(r), 1), 2), dtype=np.float64)
        n = C.c_size_t(0)
        self._dev_check(self._L.ndt2d_convert_scan(
            self.device_handle, r.ctypes.data_as(C.POINTER(C.c_float)), len(r), C.byref(d),
            dptr(out), C.byref(n)), "ndt2d_convert_scan")
        return out[:n.value].copy()

    def matchLaserScan(self, scan_pose, ranges, angle_min, angle_increment, range_max,
                       inverted=False, laser=(0.0, 0.0, 0.0), motion=(0.0, 0.0, 0.0), pose=None):
        """Conversion fused with matchScan: only the raw ranges cross PCIe.  Returns
        dict(score, pose, covariance, n_points)."""
        sp = _f64(scan_pose, (3,))
        r = np.ascontiguousarray(ranges, dtype=np.float32)
        d = self._laser_scan(angle_min, angle_increment, range_max, inverted, laser, motion)
        pose_io = np.array([0.0, 0.0, 0.0] if pose is None else pose, dtype=np.float64)
        cov = np.full(9, np.nan)
        score = C.c_double(0.0)
        npts = C.c_size_t(0)
        self._check(self._L.ndt2d_matcher_match_laser_scan(
            self._m, dptr(sp), r.ctypes.data_as(C.POINTER(C.c_float)), len(r), C.byref(d),
            dptr(pose_io), dptr(cov), C.byref(score), C.byref(npts)), "matchLaserScan")
        has = bool(self._L.ndt2d_matcher_has_ndt(self._m))
        return dict(score=score.value, pose=pose_io,
                    covariance=cov.reshape(3, 3) if has else None, n_points=npts.value)

    def scoreScan(self, scan_pose, points):
        sp = _f64(scan_pose, (3,))
        pts = _f64(points, (-1, 2))
        out = C.c_double(0.0)
        self._check(self._L.ndt2d_matcher_score_scan(self._m, dptr(sp), dptr(pts), len(pts),
                                                     C.byref(out)), "scoreScan")
        return out.value

    def scorePoints(self, points, pose):
        pts = _f64(points, (-1, 2))
        ps = _f64(pose, (3,))
        out = C.c_double(0.0)
        self._check(self._L.ndt2d_matcher_score_points(self._m, dptr(pts), len(pts), dptr(ps),
                                                       C.byref(out)), "scorePoints")
        return out.value

    def reset(self):
        self._check(self._L.ndt2d_matcher_reset(self._m), "reset")

    def set_search_ahead(self, enabled):
        """scoreScan launching the scan's search behind itself once the mapper's scoreScan /
        matchScan pair has been seen (include/ndt2d_hip.h, ndt2d_matcher_score_scan): on by default."""
        self._check(self._L.ndt2d_matcher_set_search_ahead(self._m, 1 if enabled else 0), "set_search_ahead")

    def search_ahead_stats(self):
        """(searches scoreScan launched ahead, how many a matchScan collected)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._check(self._L.ndt2d_matcher_search_ahead_stats(self._m, C.byref(a), C.byref(b)), "search_ahead_stats")
        return a.value, b.value

    def set_adjudication(self, enabled):
        """Near-tie adjudication of matchScan (on by default): candidates within 2^-36 (1.5e-11, relative) of the
        best are rescored on the host with the reference's arithmetic and its first-wins rule."""
        self._check(self._L.ndt2d_matcher_set_adjudication(self._m, 1 if enabled else 0), "set_adjudication")

    def settle_near_tie(self, scan_pose, record):
        """A combined record of a search sharded from outside: a marked winner (index + 0.5) is
        settled with the reference's arithmetic; returns the record with a plain index."""
        sp = _f64(scan_pose, (3,))
        rec = _f64(record, (_capi.MATCH_RECORD_DOUBLES,)).copy()
        self._check(self._L.ndt2d_matcher_settle_near_tie(self._m, dptr(sp), dptr(rec)), "settle_near_tie")
        return rec

    def adjudication_stats(self):
        """(searches whose winner came back marked near-tie, of those: winner changed, list truncated)."""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self._L.ndt2d_matcher_adjudication_stats(self._m, C.byref(a), C.byref(b), C.byref(c)),
                    "adjudication_stats")
        return a.value, b.value, c.value

    def match_near_best(self, th_begin, th_end, rel=2.0 ** -36, capacity=256):
        """ndt2d_match_near_best on the prepared search: flat indices (ascending) of the candidates
        within rel * |best| of the slab's best (the first `capacity` in visiting order), and how many
        there are."""
        idx = (C.c_uint64 * capacity)()
        n = C.c_size_t(0)
        self._dev_check(self._L.ndt2d_match_near_best(self.device_handle, th_begin, th_end, rel, idx, capacity,
                                                      C.byref(n), None), "ndt2d_match_near_best")
        return [idx[k] for k in range(min(n.value, capacity))], n.value

    def set_single_pose_path(self, where, max_beams=0):
        """Where scorePoints / scoreScan score their one pose: "host" (default; scans of up to
        max_beams subsampled beams, from the host NDT in the reference's order) or "device"."""
        self._check(self._L.ndt2d_matcher_set_single_pose_path(self._m, where.encode(), int(max_beams)),
                    "set_single_pose_path")

    def set_build_mode(self, mode):
        """Where addScans builds the NDT: "host", "device" or "auto" (bit-identical grids)."""
        self._check(self._L.ndt2d_matcher_set_build_mode(self._m, mode.encode()), "set_build_mode")

    def set_eigenvalue_form(self, form):
        """How Cell::compute's eigenvalues (src/ndt_model.cpp:84-85) are formed: "eigen" (default:
        Eigen 3.4.0's EigenSolver transcribed) or "closed" (the closed form of rounds 1-4)."""
        self._check(self._L.ndt2d_matcher_set_eigenvalue_form(self._m, form.encode()), "set_eigenvalue_form")

    # -- additive batched interface ------------------------------------------------

    def scorePoses(self, points, poses, out=None):
        """scores[i] == scorePoints(points, poses[i]), one launch.  `poses` / `out` from
        host_alloc() are used in place by the kernel (no copies)."""
        pts = _f64(points, (-1, 2))
        ps = _f64(poses, (-1, 3))
        if out is None:
            out = np.zeros(len(ps), dtype=np.float64)
        assert out.dtype == np.float64 and out.flags.c_contiguous and out.size == len(ps)
        self._check(self._L.ndt2d_matcher_score_poses(self._m, dptr(pts), len(pts), dptr(ps),
                                                      len(ps), dptr(out)), "scorePoses")
        return out

    # -- split matchScan / device-pointer launches (multi-GPU sharding, bench) ---------

    def prepare_search(self, scan_pose, points):
        """Subsample + build/upload the search tables.  Returns (n_th, n_lin, n_beams)."""
        sp = _f64(scan_pose, (3,))
        pts = _f64(points, (-1, 2))
        n_th, n_lin, n_b = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        self._check(self._L.ndt2d_matcher_prepare_search(
            self._m, dptr(sp), dptr(pts), len(pts), C.byref(n_th), C.byref(n_lin),
            C.byref(n_b)), "prepare_search")
        return n_th.value, n_lin.value, n_b.value

    def prepare_beams(self, points):
        pts = _f64(points, (-1, 2))
        n_b = C.c_size_t(0)
        self._check(self._L.ndt2d_matcher_prepare_beams(self._m, dptr(pts), len(pts),
                                                        C.byref(n_b)), "prepare_beams")
        return n_b.value

    def _dev_check(self, rc, where):
        if rc != _capi.OK:
            msg = self._L.ndt2d_last_error(self.device_handle)
            raise Ndt2dError(rc, where, msg.decode() if msg else "")

    def match_launch(self, th_begin, th_end, record_ptr=None, scores_ptr=None):
        """Asynchronous slab search; record_ptr / scores_ptr are DEVICE addresses
        (e.g. torch tensor .data_ptr()) or None."""
        self._dev_check(self._L.ndt2d_match_launch(self.device_handle, th_begin, th_end,
                                                   scores_ptr, record_ptr), "ndt2d_match_launch")

    def match_launch_strided(self, th_first, th_stride, th_count, record_ptr=None, scores_ptr=None):
        """Asynchronous search of the theta steps th_first, th_first + th_stride, ...: one
        rank's share of an interleaved sharding (ndt_2d_amd.dist.shard_strided)."""
        self._dev_check(self._L.ndt2d_match_launch_strided(
            self.device_handle, th_first, th_stride, th_count, scores_ptr, record_ptr),
            "ndt2d_match_launch_strided")

    def match_fetch(self):
        res = _capi.MatchResult()
        self._dev_check(self._L.ndt2d_match_fetch(self.device_handle, C.byref(res)),
                        "ndt2d_match_fetch")
        rec = np.zeros(_capi.MATCH_RECORD_DOUBLES)
        rec[0] = res.best_score
        rec[1] = -1.0 if res.best_index == _capi.NO_INDEX else float(res.best_index) + (0.5 if res.near_tie else 0.0)
        rec[2:] = res.acc[:]
        return rec

    def finish_match(self, record, pose=None):
        """matchScan's outputs from a (combined) 12-double record."""
        rec = _f64(record, (_capi.MATCH_RECORD_DOUBLES,))
        pose_io = np.array([0.0, 0.0, 0.0] if pose is None else pose, dtype=np.float64)
        cov = np.zeros(9)
        score = C.c_double(0.0)
        self._check(self._L.ndt2d_matcher_finish_match(self._m, dptr(rec), dptr(pose_io),
                                                       dptr(cov), C.byref(score)), "finish_match")
        return dict(score=score.value, pose=pose_io, covariance=cov.reshape(3, 3))

    def score_poses_launch(self, poses_ptr, n_poses, scores_ptr, stats_ptr=None):
        """Asynchronous batched scorePoints on DEVICE pointers."""
        self._dev_check(self._L.ndt2d_score_poses_launch(self.device_handle, poses_ptr, n_poses,
                                                         scores_ptr, stats_ptr),
                        "ndt2d_score_poses_launch")

    def pf_finalize_launch(self, poses_ptr, n_poses, weights_ptr, stats_ptr, out_ptr):
        """updateStatistics on DEVICE pointers: weights normalised in place by the
        total weight in stats (all-reduced over ranks when sharded), out[8] =
        {sum w, mean x, mean y, mean theta, cov xx, cov xy, cov yy, theta-variance part}."""
        self._dev_check(self._L.ndt2d_pf_finalize_launch(self.device_handle, poses_ptr, n_poses,
                                                         weights_ptr, stats_ptr, out_ptr),
                        "ndt2d_pf_finalize_launch")

    def pf_noise_launch(self, seed, step, first_index, n, noise_ptr):
        """Write the Philox standard-normal stream of (seed, step) for particles
        first_index .. first_index + n into DEVICE float[n][3]."""
        self._dev_check(self._L.ndt2d_pf_noise_launch(self.device_handle, seed, step, first_index,
                                                      n, noise_ptr), "ndt2d_pf_noise_launch")

    def pf_motion_launch(self, poses_ptr, n, dx, dy, dth, alphas, noise_ptr=None, seed=0, step=0,
                         first_index=0):
        """MotionModel::sample (reference src/motion_model.cpp:45-83) in place on DEVICE
        poses[n][3]; noise_ptr = DEVICE float[n][3] standard normals or None (Philox)."""
        a = _f64(alphas, (5,))
        self._dev_check(self._L.ndt2d_pf_motion_launch(self.device_handle, poses_ptr, n, dx, dy,
                                                       dth, dptr(a), noise_ptr, seed, step,
                                                       first_index), "ndt2d_pf_motion_launch")

    def pf_init_launch(self, poses_ptr, n, x, y, theta, sigma_x, sigma_y, sigma_theta,
                       noise_ptr=None, seed=0, step=0, first_index=0):
        """ParticleFilter::init sampling loop (reference src/particle_filter.cpp:53-65)."""
        self._dev_check(self._L.ndt2d_pf_init_launch(self.device_handle, poses_ptr, n, x, y, theta,
                                                     sigma_x, sigma_y, sigma_theta, noise_ptr,
                                                     seed, step, first_index),
                        "ndt2d_pf_init_launch")

    def pose_moments_launch(self, poses_ptr, n, weights_ptr, stats_ptr):
        """Moment sums of updateStatistics for DEVICE weights (None = uniform 1/n)."""
        self._dev_check(self._L.ndt2d_pose_moments_launch(self.device_handle, poses_ptr, n,
                                                          weights_ptr, stats_ptr),
                        "ndt2d_pose_moments_launch")

    def set_stream(self, stream_ptr):
        self._dev_check(self._L.ndt2d_set_stream(self.device_handle, stream_ptr),
                        "ndt2d_set_stream")
        self._bound_stream = bool(stream_ptr)

    def get_stream(self):
        """The caller-owned stream bound with set_stream, or None while the context
        launches on its own stream."""
        cur = self._L.ndt2d_get_stream(self.device_handle)
        return cur if getattr(self, "_bound_stream", None) else None

    def synchronize(self):
        self._dev_check(self._L.ndt2d_synchronize(self.device_handle), "ndt2d_synchronize")

    # -- introspection ---------------------------------------------------------------

    def has_ndt(self):
        return bool(self._L.ndt2d_matcher_has_ndt(self._m))

    def grid(self):
        """(cells6[ncell, 6], size_x, size_y, cell_size, origin_x, origin_y) of the host NDT."""
        sx, sy = C.c_uint32(0), C.c_uint32(0)
        cs, ox, oy = C.c_double(0), C.c_double(0), C.c_double(0)
        self._check(self._L.ndt2d_matcher_grid_info(self._m, C.byref(sx), C.byref(sy),
                                                    C.byref(cs), C.byref(ox), C.byref(oy)),
                    "grid_info")
        cells = np.zeros((sx.value * sy.value, 6), dtype=np.float64)
        self._check(self._L.ndt2d_matcher_grid_cells6(self._m, dptr(cells), len(cells)),
                    "grid_cells6")
        return cells, sx.value, sy.value, cs.value, ox.value, oy.value

    def last_launch_ms(self):
        ms = C.c_float(0)
        nk = C.c_int(0)
        rc = self._L.ndt2d_last_launch_ms(self.device_handle, C.byref(ms), C.byref(nk))
        if rc != _capi.OK:
            raise Ndt2dError(rc, "ndt2d_last_launch_ms")
        return ms.value, nk.value

    def launch_history_ms(self, n=256):
        """Kernel durations (ms) of the last up-to-n launches, oldest first; blocks until
        the newest has finished."""
        buf = (C.c_float * n)()
        got = C.c_size_t(0)
        self._dev_check(self._L.ndt2d_launch_history_ms(self.device_handle, buf, n, C.byref(got)),
                        "ndt2d_launch_history_ms")
        return [buf[i] for i in range(got.value)]

    def last_variant(self):
        v = self._L.ndt2d_last_variant(self.device_handle)
        return v.decode() if v else ""

    def set_pipeline_pieces(self, pieces):
        """ndt2d_set_pipeline_pieces: how a large pose batch from host memory is cut into
        overlapped upload / scoring / download pieces (0 default, 1 off)."""
        rc = self._L.ndt2d_set_pipeline_pieces(self.device_handle, int(pieces))
        if rc != 0:
            raise Ndt2dError(rc, "ndt2d_set_pipeline_pieces")

    def last_pipeline_pieces(self):
        return int(self._L.ndt2d_last_pipeline_pieces(self.device_handle))

    def set_variant(self, name):
        rc = self._L.ndt2d_set_variant(self.device_handle, name.encode())
        if rc != _capi.OK:
            raise Ndt2dError(rc, "ndt2d_set_variant")


def pf_measure(matcher, particles, points, cov_prev=None):
    """ParticleFilter::measure (reference src/particle_filter.cpp:78-89) including its
    updateStatistics (:163-218).  Returns (normalised weights, mean[3], cov[3, 3]);
    cov_prev carries cov_(2,2), which the reference accumulates across calls."""
    L = _capi.lib()
    pa = _f64(particles, (-1, 3))
    pts = _f64(points, (-1, 2))
    w = np.zeros(len(pa), dtype=np.float64)
    mean = np.zeros(3)
    cov = np.zeros(9) if cov_prev is None else np.array(cov_prev, dtype=np.float64).reshape(9)
    matcher._check(L.ndt2d_matcher_pf_measure(matcher._m, dptr(pa), len(pa), dptr(pts),
                                              len(pts), dptr(w), dptr(mean), dptr(cov)),
                   "pf_measure")
    return w, mean, cov.reshape(3, 3)


def pf_update(matcher, particles, weights, dx, dy, dth, alphas, noise=None, seed=0, step=0,
              cov_prev=None):
    """ParticleFilter::update (reference src/particle_filter.cpp:71-76): the motion model
    on every particle, then updateStatistics.  noise = float32[n, 3] standard normals, or
    None for the device's Philox stream of (seed, step).  Returns (particles, normalised
    weights, mean[3], cov[3, 3])."""
    L = _capi.lib()
    pa = _f64(particles, (-1, 3)).copy()
    w = _f64(weights, (len(pa),)).copy()
    a = _f64(alphas, (5,))
    out = np.zeros(_capi.PF_RESULT_DOUBLES)
    zp = None
    if noise is not None:
        z = np.ascontiguousarray(noise, dtype=np.float32).reshape(len(pa), 3)
        zp = z.ctypes.data_as(C.POINTER(C.c_float))
    matcher._dev_check(L.ndt2d_pf_update(matcher.device_handle, dptr(pa), len(pa), dx, dy, dth,
                                         dptr(a), zp, seed, step, dptr(w), dptr(out)),
                       "ndt2d_pf_update")
    cov = np.zeros((3, 3)) if cov_prev is None else np.array(cov_prev, dtype=np.float64).reshape(3, 3)
    return pa, w, out[1:4].copy(), statistics_covariance(out, cov)


def statistics_covariance(out, cov_prev):
    """cov_ after updateStatistics from an NDT2D_PF_RESULT_DOUBLES record: the x/y block
    is overwritten (:208-211), (2,2) accumulates (:216), the rest is kept."""
    cov = np.array(cov_prev, dtype=np.float64).reshape(3, 3).copy()
    cov[0, 0] = out[4]
    cov[0, 1] = out[5]
    cov[1, 0] = out[5]
    cov[1, 1] = out[6]
    cov[2, 2] += out[7]
    return cov
