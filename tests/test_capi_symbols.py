"""The C-ABI library loads and exports every symbol include/ndt2d_hip.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ndt2d_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(ndt2d_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("ndt2d_create", "ndt2d_set_grid", "ndt2d_match_launch", "ndt2d_score_poses_launch",
                 "ndt2d_matcher_initialize", "ndt2d_matcher_add_scans", "ndt2d_matcher_match_scan",
                 "ndt2d_matcher_score_scan", "ndt2d_matcher_score_points", "ndt2d_matcher_reset",
                 "ndt2d_matcher_pf_measure"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from ndt_2d_amd import _capi
    lib = ctypes.CDLL(_capi.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_every_declared_symbol():
    from ndt_2d_amd import _capi
    assert sorted(_capi.SIGNATURES) == declared_functions()
    assert _capi.lib().ndt2d_abi_version() == 4


def test_library_is_built_from_these_sources():
    """The hash baked into the loaded library is the hash of the sources in this tree: what
    the tests and the bench exercise is what the repository holds, not a stale binary."""
    from ndt_2d_amd import _capi
    from ndt_2d_amd import build as _build
    if os.environ.get("NDT2D_HIP_LIB"):
        pytest.skip("an A/B library given through NDT2D_HIP_LIB")
    want = _build.source_sha256()
    assert _capi.lib_source_sha256() == want, _capi.build_info()
    assert _build.embedded_sha256() == want
    assert "arch=gfx950" in _capi.build_info()


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the compute entry points fail loudly instead of falling back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ndt_2d_amd import Ndt2dError, ScanMatcherNDT, _capi
    with pytest.raises(Ndt2dError) as ei:
        ScanMatcherNDT(0)
    assert ei.value.code in (_capi.ERR_NO_DEVICE, _capi.ERR_HIP)


def test_product_does_not_reference_the_oracle():
    """Nothing under ndt_2d_amd/ imports, links or names the oracle."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "ndt_2d_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", ".txt", ".xml")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"oracle_lib|ndt2d_oracle|libndt2d_oracle|orc_[a-z]+_", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
