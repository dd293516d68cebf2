"""ndt_2d_amd -- MI355X-native NDT scan-matching hot path of ndt_2d.

Host-side handles over libndt2d_hip.so (include/ndt2d_hip.h).  The package holds
only what the hot path needs: csrc/ (HIP kernels + C-ABI), the Python mirror of
the reference's ScanMatcher plugin interface, the synthetic workload generator
and the multi-GPU sharding helper.
"""
from ._capi import LIB_PATH, Ndt2dError  # noqa: F401
from .particle_filter import MotionModel, ParticleFilter  # noqa: F401
from .scan_matcher import (DEFAULT_PARAMS, ScanMatcherNDT, host_build_grid,  # noqa: F401
                           pf_measure, pf_update, search_offsets)

__all__ = ["ScanMatcherNDT", "ParticleFilter", "MotionModel", "pf_measure", "pf_update",
           "search_offsets", "host_build_grid", "DEFAULT_PARAMS", "Ndt2dError", "LIB_PATH"]
