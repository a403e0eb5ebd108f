"""test_fold_gpu.py on the host backend (conftest.py of this directory; test_host_context_cpu.py says why)."""
from tests.test_fold_gpu import (test_points_fold_full_size_scalars_vs_oracle, ctxs)  # noqa: F401,F403
