import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "input-inference-for-control_amd")
for p in (ROOT, PKG_DIR, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _hip_library_is_built():
    """The gfx950 library is a build artefact (git-ignored): on a fresh checkout build it before the first test that
    loads it (hipcc cross-compiles without a GPU; a no-op when lib/libi2c_hip.so is newer than its sources)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("i2c_amd_build", os.path.join(PKG_DIR, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if os.path.exists(mod.HIPCC):
        mod.build_hip(verbose=False)
    yield


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no GPU is visible and -m gpu was not requested."""
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
