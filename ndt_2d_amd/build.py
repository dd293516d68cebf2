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

# ndt2d_build_info.cpp is compiled apart (it receives the hash of all the others as a macro)
BUILD_INFO_SOURCE = "ndt2d_build_info.cpp"
SOURCES = ["ndt2d_kernels.hip", "ndt2d_match_lane.hip", "ndt2d_match_small.hip", "ndt2d_poses_compact.hip", "ndt2d_build.hip", "ndt2d_motion.hip", "ndt2d_scan.hip", "ndt2d_occupancy.hip", "ndt2d_device.hip", "ndt2d_exchange.hip", "ndt2d_host.cpp"]
HEADERS = [os.path.join(_CSRC, "ndt2d_kernels.h"), os.path.join(_CSRC, "ndt2d_device_fn.h"), os.path.join(_CSRC, "ndt2d_lane_fn.h"), os.path.join(_CSRC, "ndt2d_poses_fn.h"), os.path.join(_CSRC, "ndt2d_exchange.h"), os.path.join(_CSRC, "ndt2d_eigen2.h"), os.path.join(_CSRC, "ndt2d_workers.h"), os.path.join(_CSRC, "ndt2d_guard.h"), os.path.join(_ROOT, "include", "ndt2d_hip.h")]
ARCH = "gfx950"
# -ffp-contract=off: the reference's x86-64 build has no fused multiply-add; the
# kernels keep its separate roundings (see DESIGN.md "Numerics").
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wextra",
         "-Wno-unused-parameter"]


SHA_MARKER = b"NDT2D_SOURCE_SHA256="


def source_sha256():
    """sha256 over everything libndt2d_hip.so is compiled from -- csrc/*.hip, *.cpp, the headers,
    include/ndt2d_hip.h -- and the compiler flags.  build_all() bakes it into the library
    (ndt2d_build_info()); tests/conftest.py rebuilds a library whose baked hash differs, so a
    stale .so (it is git-ignored but travels to the GPU box) cannot pass for the sources."""
    import hashlib
    h = hashlib.sha256()
    paths = [os.path.join(_CSRC, s) for s in SOURCES + [BUILD_INFO_SOURCE]] + HEADERS
    for path in sorted(paths, key=os.path.basename):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update((" ".join(FLAGS) + " " + ARCH).encode())
    return h.hexdigest()


def embedded_sha256(lib_path=None):
    """The hash baked into a built library (read from the file, nothing is loaded); None if the
    file is missing or carries none."""
    lib_path = lib_path or LIB_PATH
    try:
        with open(lib_path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    at = blob.find(SHA_MARKER)
    if at < 0:
        return None
    hexd = blob[at + len(SHA_MARKER):at + len(SHA_MARKER) + 64]
    try:
        text = hexd.decode("ascii")
    except UnicodeDecodeError:
        return None
    return text if len(text) == 64 and all(ch in "0123456789abcdef" for ch in text) else None


def lib_matches_source(lib_path=None):
    return embedded_sha256(lib_path) == source_sha256()


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _obj(src):
    return os.path.join(_CSRC, os.path.splitext(src)[0] + ".o")


def _input_sha256(src):
    """What one object file is compiled from: its source, every header, the flags."""
    import hashlib
    h = hashlib.sha256()
    for path in [os.path.join(_CSRC, src)] + sorted(HEADERS, key=os.path.basename):
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update((" ".join(FLAGS) + " " + ARCH).encode())
    return h.hexdigest()


def _stale_sources(force):
    """Sources whose object is missing or was compiled from other contents (the hash of its
    inputs is kept beside it in <object>.sha; time stamps are not consulted -- a tree copied to
    another machine keeps its contents, not its times)."""
    out = []
    for src in SOURCES:
        obj = _obj(src)
        try:
            with open(obj + ".sha") as f:
                have = f.read().strip()
        except OSError:
            have = None
        if force or not os.path.exists(obj) or have != _input_sha256(src):
            out.append(src)
    return out


def build_all(force=False, verbose=False, jobs=4):
    sha = source_sha256()
    if not force and os.path.exists(LIB_PATH) and embedded_sha256() == sha:
        # the library carries the hash of the sources as they are now
        probe_src = os.path.join(_PKG, "tools", "latency_probe.c")
        if not os.path.exists(PROBE_PATH) or os.path.getmtime(PROBE_PATH) < max(
                os.path.getmtime(probe_src), os.path.getmtime(LIB_PATH)):
            build_tools(verbose)
        return LIB_PATH
    stale = _stale_sources(force)

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
        if os.path.exists(_obj(src) + ".sha"):
            os.remove(_obj(src) + ".sha")
        subprocess.check_call(cmd)
        with open(_obj(src) + ".sha", "w") as f:
            f.write(_input_sha256(src) + "\n")

    if stale:
        # a few translation units at a time (each hipcc is itself two compiler passes)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=max(1, min(jobs, len(stale)))) as pool:
            list(pool.map(compile_one, stale))
    # the hash of the sources, baked in (ndt2d_build_info)
    cmd = ["g++"] + FLAGS + ["-DNDT2D_SOURCE_SHA=\"" + sha + "\"", "-DNDT2D_BUILD_ARCH=\"" + ARCH + "\"",
                             "-I", os.path.join(_ROOT, "include"), "-c",
                             os.path.join(_CSRC, BUILD_INFO_SOURCE), "-o", _obj(BUILD_INFO_SOURCE)]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC"] + \
        [_obj(s) for s in SOURCES + [BUILD_INFO_SOURCE]] + ["-ldl", "-lpthread", "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    build_tools(verbose)
    return LIB_PATH


HOOKS_LIB_PATH = os.path.join(_PKG, "libndt2d_hip_hooks.so")
HOOKED_SOURCES = ["ndt2d_match_small.hip", "ndt2d_poses_compact.hip", "ndt2d_device.hip"]


def build_test_hooks(verbose=False):
    """libndt2d_hip_hooks.so: the library with -DNDT2D_TEST_HOOKS in the two translation units
    whose kernels wait for their own blocks (bounded polls) -- their producers can then be told
    to withhold a `done` word (ndt2d_test_drop_done_small / _few), which is how
    tests/test_gpu_bounded_poll.py makes the polls trip -- and in the device layer, where the k-th
    launch can be made to fail (ndt2d_test_fail_launch: tests/test_gpu_multi_failure.py).  Test infrastructure: never loaded by
    the package itself (NDT2D_HIP_LIB selects it in the test's subprocess)."""
    lib = build_all(verbose=verbose)
    sha = source_sha256()
    stamp = HOOKS_LIB_PATH + ".sha"
    try:
        with open(stamp) as f:
            if f.read().strip() == sha and os.path.exists(HOOKS_LIB_PATH):
                return HOOKS_LIB_PATH
    except OSError:
        pass
    # the plain objects of the non-hooked sources are linked as they are: build_all() returns early
    # when the .so already carries the tree's hash, without looking at them -- a box that received
    # the library but not the (git-ignored) objects gets them compiled here
    if any(not os.path.exists(_obj(s)) for s in SOURCES if s not in HOOKED_SOURCES):
        lib = build_all(force=True, verbose=verbose)
    objs = []
    for src in SOURCES + [BUILD_INFO_SOURCE]:
        if src == BUILD_INFO_SOURCE:
            # its own build-info object: the same source hash, marked hooks=1, so that
            # ndt2d_build_info() / lib_matches_source() tell the two libraries apart
            obj = os.path.join(_CSRC, "ndt2d_build_info.hooks.o")
            cmd = ["g++"] + FLAGS + ["-DNDT2D_SOURCE_SHA=\"" + sha + "\"", "-DNDT2D_BUILD_ARCH=\"" + ARCH + "\"",
                                     "-DNDT2D_BUILD_HOOKS=1", "-I", os.path.join(_ROOT, "include"), "-c",
                                     os.path.join(_CSRC, BUILD_INFO_SOURCE), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            objs.append(obj)
        elif src in HOOKED_SOURCES:
            obj = os.path.join(_CSRC, os.path.splitext(src)[0] + ".hooks.o")
            cmd = [hipcc(), "--offload-arch=" + ARCH] + FLAGS + ["-DNDT2D_TEST_HOOKS", "-I", os.path.join(_ROOT, "include"),
                                                                 "-I", _CSRC, "-c", os.path.join(_CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            objs.append(obj)
        else:
            objs.append(_obj(src))
    cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-ldl", "-lpthread", "-o", HOOKS_LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(sha + "\n")
    assert os.path.exists(lib)
    return HOOKS_LIB_PATH


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
    if "--test-hooks" in sys.argv:
        print(build_test_hooks(verbose=True))
