import sys,time
sys.path.insert(0,''+__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))+'')
import numpy as np
from ndt_2d_amd import host_build_grid, synth
scans=synth.map_scans(1)
p=synth.matcher_params(1)
import ctypes as C
from ndt_2d_amd import _capi
from ndt_2d_amd.scan_matcher import _pack_scans
from ndt_2d_amd._capi import dptr
L=_capi.lib()
poses, allpts, offsets = _pack_scans(scans)
off_p = offsets.ctypes.data_as(C.POINTER(C.c_size_t))
sx, sy = C.c_uint32(0), C.c_uint32(0); ox, oy = C.c_double(0), C.c_double(0)
cells = np.zeros((41*41, 6))
ts=[]
for i in range(3000):
    t0=time.perf_counter()
    L.ndt2d_host_build_grid(0.25, p["range_max"], dptr(poses), dptr(allpts), off_p, len(scans), dptr(cells), len(cells), C.byref(sx), C.byref(sy), C.byref(ox), C.byref(oy))
    ts.append(time.perf_counter()-t0)
ts.sort(); print("host build median %.2f us  p10 %.2f"%(ts[len(ts)//2]*1e6, ts[len(ts)//10]*1e6), len(allpts))
import hashlib; print(hashlib.sha256(cells.tobytes()).hexdigest()[:16])
