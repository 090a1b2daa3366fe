import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "texture-gs_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    """Make sure libtexgs.so exists (built in-tree; cross-compiles without a GPU)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("texgs_build", os.path.join(PKG, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build()


def pytest_collection_modifyitems(config, items):
    """`gpu`-marked tests need an MI355X: on a box without one they are SKIPPED, not errors."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="needs an MI355X (torch.cuda.is_available() is False)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
