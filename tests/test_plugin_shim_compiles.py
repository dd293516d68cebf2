"""Syntax-checks the pluginlib shim against the reference's own interface headers
(include/ndt_2d/scan_matcher.hpp etc.).  Eigen3 / rclcpp / pluginlib are absent in
this image, so tests/stubs/ holds minimal stand-ins for THOSE (test-only).  Runs
only where /root/reference exists (not on the GPU box)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INC = "/root/reference/include"


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="reference headers not present")
@pytest.mark.parametrize("src", ["ndt_2d_amd/plugin/scan_matcher_ndt_hip.cpp",
                                 "tests/stubs/shim_instantiation.cpp"])
def test_shim_compiles_against_reference_interface(src):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror",
           "-I", os.path.join(ROOT, "tests", "stubs"), "-I", REF_INC,
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ndt_2d_amd", "plugin"),
           os.path.join(ROOT, src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_plugins_xml_registers_against_the_reference_base_class():
    xml = open(os.path.join(ROOT, "ndt_2d_amd", "plugin", "plugins.xml")).read()
    assert 'base_class_type="ndt_2d::ScanMatcher"' in xml
    assert 'type="ndt_2d_hip::ScanMatcherNDTHip"' in xml
