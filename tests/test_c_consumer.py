"""include/ndt2d_hip.h is a plain-C header: a gcc-built C program links against
libndt2d_hip.so and uses it (tests/c/capi_smoke.c)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "ndt_2d_amd")


def _build(tmp_path):
    from ndt_2d_amd import _capi
    assert os.path.exists(_capi.LIB_PATH)
    exe = os.path.join(str(tmp_path), "capi_smoke")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "capi_smoke.c"),
           "-L", LIBDIR, "-lndt2d_hip", "-lm", "-Wl,-rpath," + LIBDIR, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_plain_c_and_host_entry_points_work(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)


@pytest.mark.gpu
def test_c_program_reproduces_the_reference_known_answer(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "ok" in r.stdout and "likelihood((3.5,3.5)) = 0.7659" in r.stdout
