import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        from accumulation_amd import ffi
        return ffi.load().amsm_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """`-m gpu` on a box without a GPU must fail loudly, not pass vacuously or skip: the HIP path never degrades to the host backend."""
    expr = (config.getoption("-m") or "").strip()
    wants_gpu = "gpu" in expr and "not gpu" not in expr
    if wants_gpu and any(i.get_closest_marker("gpu") for i in items) and not _have_gpu():
        raise pytest.UsageError("-m gpu selected but libamsm.so sees no gfx950 GPU (amsm_device_count() == 0): "
                                "the GPU tests cannot run here (the host backend has its own tests: -m 'not gpu')")


@pytest.fixture(scope="session")
def have_gpu():
    return _have_gpu()


@pytest.fixture(scope="session")
def built_lib():
    """libamsm.so must exist (build it if the tree is fresh); CPU-only boxes can still load it."""
    from accumulation_amd import build, ffi
    if not os.path.exists(ffi.LIB_PATH):
        build.build_lib()
    return ffi.load()


@pytest.fixture(scope="session")
def cref():
    from oracle import cref as c
    c.build()
    c.load()
    return c
