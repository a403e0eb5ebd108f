"""Every Context of the modules in this directory is a context of the library's HOST backend (include/amsm.h: AMSM_DEVICE_HOST):
the modules re-collect the GPU suites' own test functions, which build `Context(curve)` themselves."""
import pytest

from accumulation_amd import ffi
from accumulation_amd.engine import Context


@pytest.fixture(scope="module", autouse=True)
def every_context_on_the_host_backend():
    orig = Context.__init__

    def init(self, curve=ffi.AMSM_PALLAS, device=0, stream=None):
        orig(self, curve, ffi.AMSM_DEVICE_HOST, None)

    Context.__init__ = init
    yield
    Context.__init__ = orig
