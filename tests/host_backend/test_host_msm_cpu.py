"""test_msm_gpu.py on the host backend (conftest.py of this directory; test_host_context_cpu.py says why)."""
from tests.test_msm_gpu import (test_golden_cases, test_golden_seeded_device_key, test_scalar_out_of_range_is_an_error,
                                test_base_offset_and_min_len, test_pedersen_commit_with_hiding, test_partials_roundtrip_single_rank,
                                test_grouped_msm_vs_two_msms, test_doubling_of_a_negated_duplicate_base,
                                test_published_bls12_381_multiples_through_the_device_msm, ctxs)  # noqa: F401,F403
