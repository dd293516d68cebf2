"""Builds libndt2d_hip.so (HIP kernels + C-ABI) in-tree for gfx950.

    python -m ndt_2d_amd.build            # incremental
    python -m ndt_2d_amd.build --force

hipcc cross-compiles for gfx950 without a GPU.  The product library is built
from ndt_2d_amd/csrc only; nothing under oracle/ is compiled into it.
"""
import os
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
_CSRC = os.path.join(_PKG, "csrc")
LIB_PATH = os.path.join(_PKG, "libndt2d_hip.so")

SOURCES = ["ndt2d_kernels.hip", "ndt2d_match_lane.hip", "ndt2d_poses_compact.hip", "ndt2d_build.hip", "ndt2d_motion.hip", "ndt2d_scan.hip", "ndt2d_occupancy.hip", "ndt2d_device.hip", "ndt2d_host.cpp"]
HEADERS = [os.path.join(_CSRC, "ndt2d_kernels.h"), os.path.join(_CSRC, "ndt2d_device_fn.h"), os.path.join(_ROOT, "include", "ndt2d_hip.h")]
ARCH = "gfx950"
# -ffp-contract=off: the reference's x86-64 build has no fused multiply-add; the
# kernels keep its separate roundings (see DESIGN.md "Numerics").
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wextra",
         "-Wno-unused-parameter"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(_CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_all(force=False, verbose=False):
    if not force and not _stale():
        return LIB_PATH
    objs = []
    for src in SOURCES:
        obj = os.path.join(_CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc(), "--offload-arch=" + ARCH] + FLAGS + [
            "-I", os.path.join(_ROOT, "include"), "-I", _CSRC, "-c",
            os.path.join(_CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
