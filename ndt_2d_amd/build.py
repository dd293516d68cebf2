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

SOURCES = ["ndt2d_kernels.hip", "ndt2d_match_lane.hip", "ndt2d_match_small.hip", "ndt2d_poses_compact.hip", "ndt2d_build.hip", "ndt2d_motion.hip", "ndt2d_scan.hip", "ndt2d_occupancy.hip", "ndt2d_device.hip", "ndt2d_exchange.hip", "ndt2d_host.cpp"]
HEADERS = [os.path.join(_CSRC, "ndt2d_kernels.h"), os.path.join(_CSRC, "ndt2d_device_fn.h"), os.path.join(_CSRC, "ndt2d_lane_fn.h"), os.path.join(_CSRC, "ndt2d_exchange.h"), os.path.join(_ROOT, "include", "ndt2d_hip.h")]
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


def _obj(src):
    return os.path.join(_CSRC, os.path.splitext(src)[0] + ".o")


def _stale_sources(force):
    """Sources whose object is missing or older than the source, a header or this file."""
    common = HEADERS + [os.path.abspath(__file__)]
    t_common = max(os.path.getmtime(d) for d in common)
    out = []
    for src in SOURCES:
        obj = _obj(src)
        if force or not os.path.exists(obj) or \
                os.path.getmtime(obj) < max(t_common, os.path.getmtime(os.path.join(_CSRC, src))):
            out.append(src)
    return out


def build_all(force=False, verbose=False, jobs=4):
    stale = _stale_sources(force)
    if not stale and os.path.exists(LIB_PATH) and \
            os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(_obj(s)) for s in SOURCES):
        probe_src = os.path.join(_PKG, "tools", "latency_probe.c")
        if not os.path.exists(PROBE_PATH) or os.path.getmtime(PROBE_PATH) < max(
                os.path.getmtime(probe_src), os.path.getmtime(LIB_PATH)):
            build_tools(verbose)
        return LIB_PATH

    def compile_one(src):
        if src.endswith(".cpp"):
            # host-only translation unit (no HIP headers): the host compiler, which also
            # knows function multiversioning (the AVX2 clone of the NDT build loop)
            cmd = ["g++"] + FLAGS + ["-I", os.path.join(_ROOT, "include"), "-I", _CSRC, "-c",
                                     os.path.join(_CSRC, src), "-o", _obj(src)]
        else:
            cmd = [hipcc(), "--offload-arch=" + ARCH] + FLAGS + [
                "-I", os.path.join(_ROOT, "include"), "-I", _CSRC, "-c",
                os.path.join(_CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    if stale:
        # a few translation units at a time (each hipcc is itself two compiler passes)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(jobs, len(stale)))) as pool:
            list(pool.map(compile_one, stale))
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC"] + [_obj(s) for s in SOURCES] + \
        ["-ldl", "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    build_tools(verbose)
    return LIB_PATH


PROBE_PATH = os.path.join(_PKG, "ndt2d_latency_probe")


def build_tools(verbose=False):
    """The plain-C latency probe (ndt_2d_amd/tools/latency_probe.c): the C-ABI as a C
    host calls it, used by bench.py's default_search leg."""
    src = os.path.join(_PKG, "tools", "latency_probe.c")
    cmd = ["gcc", "-O2", "-std=c99", "-Wall", "-Wextra", "-I", os.path.join(_ROOT, "include"), src,
           "-L", _PKG, "-lndt2d_hip", "-lm", "-Wl,-rpath," + _PKG, "-o", PROBE_PATH]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.check_call(cmd)
    except (OSError, subprocess.CalledProcessError) as exc:
        # a measurement tool, not the product: the library is usable without it
        print("ndt_2d_amd.build: latency probe not built (%s)" % exc, file=sys.stderr)
        return None
    return PROBE_PATH


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
