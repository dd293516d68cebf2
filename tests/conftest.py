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
    # A fresh checkout has no built library (it is git-ignored): build it once, as
    # __graft_entry__.build() does.  hipcc cross-compiles for gfx950 without a GPU.
    from ndt_2d_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        from ndt_2d_amd import build as _build
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
