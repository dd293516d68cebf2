"""ctypes binding of include/ndt2d_hip.h (libndt2d_hip.so).

This is the only place the package touches native code.  Importing it fails
loudly if the HIP extension has not been built -- there is no Python or CPU
fallback for any compute entry point.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# NDT2D_HIP_LIB selects another build of the same library (A/B timing of kernels)
LIB_PATH = os.environ.get("NDT2D_HIP_LIB") or os.path.join(_PKG, "libndt2d_hip.so")

OK = 0
ERR_INVALID = 1
ERR_NO_GRID = 2
ERR_HIP = 3
ERR_NO_DEVICE = 4
ERR_STATE = 5
ERR_ALLOC = 6
ERR_INTERNAL = 7
NO_INDEX = (1 << 64) - 1
MATCH_RECORD_DOUBLES = 12
POSE_STATS_DOUBLES = 8
PF_RESULT_DOUBLES = 8

_ERR_NAMES = {1: "NDT2D_ERR_INVALID", 2: "NDT2D_ERR_NO_GRID", 3: "NDT2D_ERR_HIP",
              4: "NDT2D_ERR_NO_DEVICE", 5: "NDT2D_ERR_STATE", 6: "NDT2D_ERR_ALLOC", 7: "NDT2D_ERR_INTERNAL"}


class Ndt2dError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__("%s failed: %s%s" % (where, _ERR_NAMES.get(code, str(code)),
                                               (" (" + detail + ")") if detail else ""))


class MatchResult(C.Structure):
    _fields_ = [("best_score", C.c_double), ("best_index", C.c_uint64),
                ("acc", C.c_double * 10), ("n_candidates", C.c_uint64), ("near_tie", C.c_uint64)]


class LaserScan(C.Structure):
    """ndt2d_laser_scan"""
    _fields_ = [("angle_min", C.c_float), ("angle_increment", C.c_float),
                ("range_max", C.c_double), ("inverted", C.c_int),
                ("laser_x", C.c_double), ("laser_y", C.c_double), ("laser_theta", C.c_double),
                ("motion_x", C.c_double), ("motion_y", C.c_double), ("motion_theta", C.c_double)]


class OccupancyInfo(C.Structure):
    """ndt2d_occupancy_info"""
    _fields_ = [("resolution", C.c_double), ("width", C.c_uint32), ("height", C.c_uint32),
                ("origin_x", C.c_double), ("origin_y", C.c_double)]


class World(C.Structure):
    _fields_ = [("room_half", C.c_double), ("pillar_pitch", C.c_double),
                ("pillar_half", C.c_double)]


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p
_d = C.c_double
_sz = C.c_size_t
_u64 = C.c_uint64
_u32 = C.c_uint32
_szp = C.POINTER(C.c_size_t)

# name -> (restype, argtypes); one entry per function declared in ndt2d_hip.h
SIGNATURES = {
    "ndt2d_abi_version": (C.c_int, []),
    "ndt2d_build_info": (C.c_char_p, []),
    "ndt2d_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "ndt2d_destroy": (C.c_int, [_vp]),
    "ndt2d_last_error": (C.c_char_p, [_vp]),
    "ndt2d_set_stream": (C.c_int, [_vp, _vp]),
    "ndt2d_get_stream": (_vp, [_vp]),
    "ndt2d_device_id": (C.c_int, [_vp]),
    "ndt2d_set_grid": (C.c_int, [_vp, _dp, _u32, _u32, _d, _d, _d]),
    "ndt2d_set_grid_sparse": (C.c_int, [_vp, C.POINTER(C.c_uint32), _dp, _sz, _u32, _u32, _d, _d, _d]),
    "ndt2d_build_grid": (C.c_int, [_vp, _d, _d, _dp, _dp, _szp, _sz]),
    "ndt2d_set_eigenvalue_form": (C.c_int, [_vp, C.c_char_p]),
    "ndt2d_get_grid": (C.c_int, [_vp, _dp, _sz, C.POINTER(_u32), C.POINTER(_u32), _dp, _dp, _dp]),
    "ndt2d_clear_grid": (C.c_int, [_vp]),
    "ndt2d_has_grid": (C.c_int, [_vp]),
    "ndt2d_set_beams": (C.c_int, [_vp, _dp, _sz]),
    "ndt2d_set_search": (C.c_int, [_vp, _d, _d, _dp, _dp, _dp, _sz, _dp, _sz]),
    "ndt2d_set_search_beams": (C.c_int, [_vp, _dp, _sz, _d, _d, _dp, _dp, _dp, _sz, _dp, _sz]),
    "ndt2d_match_launch": (C.c_int, [_vp, _sz, _sz, _vp, _vp]),
    "ndt2d_match_launch_strided": (C.c_int, [_vp, _sz, _sz, _sz, _vp, _vp]),
    "ndt2d_match_fetch": (C.c_int, [_vp, C.POINTER(MatchResult)]),
    "ndt2d_match": (C.c_int, [_vp, _sz, _sz, _dp, C.POINTER(MatchResult)]),
    "ndt2d_score_poses_launch": (C.c_int, [_vp, _vp, _sz, _vp, _vp]),
    "ndt2d_score_poses": (C.c_int, [_vp, _dp, _sz, _dp, _dp]),
    "ndt2d_score_poses_beams": (C.c_int, [_vp, _dp, _sz, _dp, _sz, _dp]),
    "ndt2d_grid_stage_begin": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _sz, C.POINTER(C.POINTER(C.c_uint32)),
                                         C.POINTER(C.POINTER(C.c_double))]),
    "ndt2d_grid_stage_commit": (C.c_int, [_vp, _sz, _d, _d, _d]),
    "ndt2d_score_poses_beams_launch": (C.c_int, [_vp, _dp, _sz, _dp, _sz]),
    "ndt2d_score_fetch": (C.c_int, [_vp, _dp]),
    "ndt2d_pf_finalize_launch": (C.c_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "ndt2d_pose_sums_launch": (C.c_int, [_vp, _vp, _sz, _vp]),
    "ndt2d_pose_sums_fetch": (C.c_int, [_vp, _dp]),
    "ndt2d_pf_finalize_totals_launch": (C.c_int, [_vp, _vp, _sz, _vp, _dp]),
    "ndt2d_pf_result_read": (C.c_int, [_vp, _dp]),
    "ndt2d_pf_measure": (C.c_int, [_vp, _dp, _sz, _dp, _dp]),
    "ndt2d_pf_noise_launch": (C.c_int, [_vp, _u64, _u64, _u64, _sz, _vp]),
    "ndt2d_pf_motion_launch": (C.c_int, [_vp, _vp, _sz, _d, _d, _d, _dp, _vp, _u64, _u64, _u64]),
    "ndt2d_pf_init_launch": (C.c_int, [_vp, _vp, _sz, _d, _d, _d, _d, _d, _d, _vp, _u64, _u64,
                                       _u64]),
    "ndt2d_pose_moments_launch": (C.c_int, [_vp, _vp, _sz, _vp, _vp]),
    "ndt2d_pf_update": (C.c_int, [_vp, _dp, _sz, _d, _d, _d, _dp, C.POINTER(C.c_float), _u64,
                                  _u64, _dp, _dp]),
    "ndt2d_convert_scan_launch": (C.c_int, [_vp, _vp, _sz, C.POINTER(LaserScan), _vp, _vp]),
    "ndt2d_convert_scan": (C.c_int, [_vp, C.POINTER(C.c_float), _sz, C.POINTER(LaserScan), _dp,
                                     _szp]),
    "ndt2d_set_beams_from_ranges": (C.c_int, [_vp, C.POINTER(C.c_float), _sz,
                                              C.POINTER(LaserScan), _sz, _szp, _szp]),
    "ndt2d_scan_points": (_vp, [_vp, _szp]),
    "ndt2d_occupancy_grid": (C.c_int, [_vp, _d, _d, _dp, _dp, _szp, _sz, _sz, _dp,
                                       C.POINTER(OccupancyInfo), _vp, _sz]),
    "ndt2d_device_alloc": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "ndt2d_device_free": (C.c_int, [_vp, _vp]),
    "ndt2d_copy_to_device": (C.c_int, [_vp, _vp, _vp, _sz]),
    "ndt2d_copy_to_host": (C.c_int, [_vp, _vp, _vp, _sz]),
    "ndt2d_copy_to_device_async": (C.c_int, [_vp, _vp, _vp, _sz]),
    "ndt2d_copy_to_host_async": (C.c_int, [_vp, _vp, _vp, _sz]),
    "ndt2d_match_near_best": (C.c_int, [_vp, _sz, _sz, _d, C.POINTER(C.c_uint64), _sz, _szp,
                                        C.POINTER(MatchResult)]),
    "ndt2d_match_status": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ndt2d_launch_history_ms": (C.c_int, [_vp, C.POINTER(C.c_float), _sz, _szp]),
    "ndt2d_host_alloc": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "ndt2d_host_free": (C.c_int, [_vp, _vp]),
    "ndt2d_set_timing": (C.c_int, [_vp, C.c_int]),
    "ndt2d_synchronize": (C.c_int, [_vp]),
    "ndt2d_last_launch_ms": (C.c_int, [_vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "ndt2d_last_variant": (C.c_char_p, [_vp]),
    "ndt2d_set_variant": (C.c_int, [_vp, C.c_char_p]),
    "ndt2d_set_pipeline_pieces": (C.c_int, [_vp, C.c_int]),
    "ndt2d_last_pipeline_pieces": (C.c_int, [_vp]),
    "ndt2d_matcher_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "ndt2d_matcher_create_multi": (C.c_int, [C.POINTER(_vp), C.POINTER(C.c_int), C.c_int]),
    "ndt2d_matcher_device_count": (C.c_int, [_vp]),
    "ndt2d_matcher_device_at": (_vp, [_vp, C.c_int]),
    "ndt2d_matcher_set_exchange": (C.c_int, [_vp, C.c_char_p]),
    "ndt2d_matcher_set_multi_min_units": (C.c_int, [_vp, _d]),
    "ndt2d_matcher_set_multi_thresholds": (C.c_int, [_vp, _d, _d]),
    "ndt2d_matcher_get_multi_thresholds": (C.c_int, [_vp, _dp, _dp]),
    "ndt2d_matcher_last_fanout_us": (C.c_int, [_vp, _dp, _sz, _szp]),
    "ndt2d_matcher_last_variant": (C.c_char_p, [_vp]),
    "ndt2d_matcher_set_timing": (C.c_int, [_vp, C.c_int]),
    "ndt2d_matcher_destroy": (C.c_int, [_vp]),
    "ndt2d_matcher_last_error": (C.c_char_p, [_vp]),
    "ndt2d_matcher_device": (_vp, [_vp]),
    "ndt2d_matcher_initialize": (C.c_int, [_vp, _d, _d, _d, _d, _d, _sz, _d]),
    "ndt2d_matcher_add_scans": (C.c_int, [_vp, _dp, _dp, _szp, _sz]),
    "ndt2d_matcher_set_eigenvalue_form": (C.c_int, [_vp, C.c_char_p]),
    "ndt2d_matcher_set_build_mode": (C.c_int, [_vp, C.c_char_p]),
    "ndt2d_matcher_match_scan": (C.c_int, [_vp, _dp, _dp, _sz, _dp, _dp, _dp]),
    "ndt2d_matcher_match_scan_ex": (C.c_int, [_vp, _dp, _dp, _sz, _dp, _dp, _dp, _dp, _sz,
                                             _szp, C.POINTER(C.c_uint64)]),
    "ndt2d_matcher_match_laser_scan": (C.c_int, [_vp, _dp, C.POINTER(C.c_float), _sz,
                                                 C.POINTER(LaserScan), _dp, _dp, _dp, _szp]),
    "ndt2d_matcher_prepare_search": (C.c_int, [_vp, _dp, _dp, _sz, _szp, _szp, _szp]),
    "ndt2d_matcher_finish_match": (C.c_int, [_vp, _dp, _dp, _dp, _dp]),
    "ndt2d_matcher_prepare_beams": (C.c_int, [_vp, _dp, _sz, _szp]),
    "ndt2d_matcher_score_scan": (C.c_int, [_vp, _dp, _dp, _sz, _dp]),
    "ndt2d_matcher_set_search_ahead": (C.c_int, [_vp, C.c_int]),
    "ndt2d_matcher_set_adjudication": (C.c_int, [_vp, C.c_int]),
    "ndt2d_matcher_settle_near_tie": (C.c_int, [_vp, _dp, _dp]),
    "ndt2d_matcher_adjudication_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                                   C.POINTER(C.c_uint64)]),
    "ndt2d_matcher_set_single_pose_path": (C.c_int, [_vp, C.c_char_p, _sz]),
    "ndt2d_matcher_search_ahead_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ndt2d_matcher_score_points": (C.c_int, [_vp, _dp, _sz, _dp, _dp]),
    "ndt2d_matcher_reset": (C.c_int, [_vp]),
    "ndt2d_matcher_has_ndt": (C.c_int, [_vp]),
    "ndt2d_matcher_score_poses": (C.c_int, [_vp, _dp, _sz, _dp, _sz, _dp]),
    "ndt2d_matcher_pf_measure": (C.c_int, [_vp, _dp, _sz, _dp, _sz, _dp, _dp, _dp]),
    "ndt2d_matcher_grid_info": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(_u32), _dp, _dp, _dp]),
    "ndt2d_matcher_grid_cells6": (C.c_int, [_vp, _dp, _sz]),
    "ndt2d_search_offsets": (C.c_int, [_d, _d, _dp, _sz, _szp]),
    "ndt2d_kld_resample": (C.c_int, [_dp, _dp, _sz, _sz, _sz, _d, _d, _dp, _dp, _sz,
                                     C.POINTER(_u32), _szp]),
    "ndt2d_host_build_grid": (C.c_int, [_d, _d, _dp, _dp, _szp, _sz, _dp, _sz,
                                       C.POINTER(_u32), C.POINTER(_u32), _dp, _dp]),
    "ndt2d_host_build_grid_ex": (C.c_int, [_d, _d, _dp, _dp, _szp, _sz, C.c_uint, _dp, _sz,
                                          C.POINTER(_u32), C.POINTER(_u32), _dp, _dp]),
    "ndt2d_synth_scan": (C.c_int, [C.POINTER(World), _dp, _sz, _d, C.c_uint64, _dp]),
    "ndt2d_synth_pose_blocked": (C.c_int, [C.POINTER(World), _d, _d, _d]),
    "ndt2d_synth_uniform": (C.c_int, [C.c_uint64, _sz, _dp]),
}

_lib = None


def lib():
    """The loaded library.  Raises ImportError if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "ndt_2d_amd: %s is missing -- build it with `python -m ndt_2d_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    # PyTorch-ROCm wheels bundle their own HIP runtime.  If this library pulled in
    # the system libamdhip64 first, a later `import torch` would bring a second
    # runtime into the process and find no GPUs; loading torch's first makes both
    # share one.  (Pure C/C++ hosts, e.g. the ROS plugin, never see torch.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def build_info():
    """ndt2d_build_info() of the loaded library."""
    return lib().ndt2d_build_info().decode()


def lib_source_sha256():
    """The source hash the loaded library carries (build.source_sha256() at its build time)."""
    info = build_info()
    key = "NDT2D_SOURCE_SHA256="
    at = info.find(key)
    return info[at + len(key):at + len(key) + 64] if at >= 0 else None


def dptr(a):
    return a.ctypes.data_as(_dp)
