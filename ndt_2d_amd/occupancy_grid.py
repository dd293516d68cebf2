"""ndt_2d::OccupancyGrid over the MI355X kernels.

Mirror of the reference's map renderer (reference include/ndt_2d/occupancy_grid.hpp:
44-74, src/occupancy_grid.cpp): same constructor arguments, same `getMsg(scans)`,
same persistent bounds.  The ray tracing runs on the GPU through
ndt2d_occupancy_grid; nothing here touches a map cell.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import Ndt2dError, dptr
from .scan_matcher import _pack_scans


class OccupancyGrid:
    """`device` is an ndt_2d_amd.ScanMatcherNDT (its GPU context is used) ."""

    def __init__(self, resolution, occ_thresh, device):
        self.resolution = float(resolution)
        self.occ_thresh = float(occ_thresh)
        self._device = device
        self._L = _capi.lib()
        # min_x_, max_x_, min_y_, max_y_ and num_scans_ (reference occupancy_grid.cpp:37-41)
        self.bounds = np.zeros(4, dtype=np.float64)
        self.num_scans = 0

    def getMsg(self, scans):
        """scans: iterable of (pose_xyt, points[n, 2]).  Returns dict(resolution, width,
        height, origin_x, origin_y, data[height, width] int8) -- the fields of the
        nav_msgs/OccupancyGrid the reference fills (:60-66,134-150)."""
        scans = list(scans)
        poses, allpts, offsets = _pack_scans(scans)
        off_p = offsets.ctypes.data_as(C.POINTER(C.c_size_t))
        info = _capi.OccupancyInfo()
        h = self._device.device_handle

        def call(bounded, data_ptr, cap):
            rc = self._L.ndt2d_occupancy_grid(h, self.resolution, self.occ_thresh, dptr(poses),
                                              dptr(allpts), off_p, len(scans), bounded,
                                              dptr(self.bounds), C.byref(info), data_ptr, cap)
            if rc != _capi.OK:
                msg = self._L.ndt2d_last_error(h)
                raise Ndt2dError(rc, "ndt2d_occupancy_grid", msg.decode() if msg else "")

        call(self.num_scans, None, 0)       # bounds (if the scan count changed) + meta data
        self.num_scans = len(scans)
        data = np.zeros((info.height, info.width), dtype=np.int8)
        if data.size:
            call(self.num_scans, data.ctypes.data_as(C.c_void_p), data.size)
        return dict(resolution=info.resolution, width=int(info.width), height=int(info.height),
                    origin_x=info.origin_x, origin_y=info.origin_y, data=data)
