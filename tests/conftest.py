import os
import sys

import pytest

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
for p in (_HERE, _ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")
    # A fresh checkout has no built library (it is git-ignored), and a library that is there
    # may have been built from other sources than the tree's (it travels to the GPU box with the
    # snapshot): build_all() compares the hash baked into the .so with the sources' and compiles
    # what differs.  hipcc cross-compiles for gfx950 without a GPU.  (NDT2D_HIP_LIB: an A/B
    # library given from outside is taken as it is.)
    from ndt_2d_amd import _capi
    from ndt_2d_amd import build as _build
    if not os.environ.get("NDT2D_HIP_LIB") and not _build.lib_matches_source():
        _build.build_all()


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
