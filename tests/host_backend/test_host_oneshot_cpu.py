"""test_oneshot_gpu.py on the host backend (conftest.py of this directory; test_host_context_cpu.py says why)."""
from tests.test_oneshot_gpu import (test_golden_cases_oneshot, test_min_len_identity_bases_and_empty,  # noqa: F401
                                    test_sizes_vs_c_oracle_and_resident_key, ctxs)
