"""test_ipa_jump_gpu.py on the host backend (conftest.py of this directory; test_host_context_cpu.py says why)."""
from tests.test_ipa_jump_gpu import (test_jump_fold_vs_oracle_and_physical_folds, test_keys_that_do_not_qualify_are_refused_not_miscomputed,  # noqa: F401
                                     test_opening_with_and_without_the_jump_same_proof)
