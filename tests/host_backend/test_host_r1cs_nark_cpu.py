"""test_r1cs_nark_gpu.py on the host backend (conftest.py of this directory; test_host_context_cpu.py says why)."""
from tests.test_r1cs_nark_gpu import *  # noqa: F401,F403
pytestmark = []  # (the star import brought the GPU module's `gpu` mark along: these run on the host backend, without one)
