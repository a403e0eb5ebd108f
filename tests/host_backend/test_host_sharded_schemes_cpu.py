"""hp_as and r1cs_nark_as over point-SHARDED committer keys on two and three gloo ranks, every rank a context of the library's host
backend: the N > 1 form of BASELINE configs 4 / 5 (dist.ShardedCommitterKey, per-rank partial records, one all-gather per commit
round) with the product library itself and no GPU -- same accumulators, proofs and witness slices as the unsharded run
(tests/test_hp_as_sharded_gpu.py / test_r1cs_nark_as_sharded_gpu.py hold the GPU form)."""
import pytest

from accumulation_amd import ffi


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_hp_as_sharded_equals_unsharded_on_the_host_backend(built_lib, make_zk, world):
    from tests.test_hp_as_sharded_gpu import run_sharded_vs_unsharded
    run_sharded_vs_unsharded(make_zk, world, device=ffi.AMSM_DEVICE_HOST)


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_r1cs_nark_as_sharded_equals_unsharded_on_the_host_backend(built_lib, make_zk):
    from tests.test_r1cs_nark_as_sharded_gpu import run_sharded_vs_unsharded
    run_sharded_vs_unsharded(make_zk, 2, device=ffi.AMSM_DEVICE_HOST)
