"""The device worker threads of the multi-device matcher (ndt_2d_amd/csrc/ndt2d_workers.h),
exercised on the CPU: tests/cpp/workers_check.cpp."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_workers(tmp_path):
    exe = os.path.join(str(tmp_path), "workers_check")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", "-Werror", "-pthread",
           "-I", os.path.join(ROOT, "ndt_2d_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "workers_check.cpp"),
           "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "workers ok" in r.stdout
