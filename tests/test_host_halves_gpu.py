"""A LONE host slice of [3 x 2^18, 2^20] pairs over a 20-bit key -- the reference's one blocking `commit(&ck, &[Fr], ..)` at a time
(src/hp_as/mod.rs:372-385,911-918) -- runs as two ranges over ONE bucket set so that the second half uploads while the first is
sorted and accumulated (round 6; api_pipeline.inc: host_halves_apply).  Results against the CPU restatement, for uniform vectors,
for vectors whose second (or first) half is skewed (the halves' overflow sends the call back to the one-range form), with base
offsets, through amsm_msm and amsm_pedersen_commit; AMSM_HOST_HALVES=0 gives the same points."""
import numpy as np
import pytest

from oracle import pyref as o

pytestmark = pytest.mark.gpu
N = 1 << 20


@pytest.fixture(scope="module")
def env(cref):
    from accumulation_amd import CommitterKey, Context
    c = o.PALLAS
    ctx = Context(c.curve_id)
    ck = CommitterKey.generate(ctx, 0x4A1, N + 4096)
    assert ck.window_bits == 20
    xy, _ = ck.read()
    yield c, ctx, ck, xy
    ck.free()
    ctx.close()


@pytest.mark.parametrize("n,off", [(N, 0), (N, 4096), (3 << 18, 7), (900001, 1000), (N - 63, 1)])
def test_uniform_slices_take_the_halved_form(env, cref, n, off):
    from accumulation_amd import VariableBaseMSM
    c, ctx, ck, xy = env
    sc = cref.rng_frs(c.curve_id, 0x4A2 + n, n)
    before = ctx.pipeline_stats()
    got, ginf = VariableBaseMSM.multi_scalar_mul(ck, sc, base_off=off)
    after = ctx.pipeline_stats()
    ref, rinf = cref.msm(c.curve_id, xy[off:off + n], sc)
    assert bool(ginf) == bool(rinf) and np.array_equal(got, ref)
    assert after["shared_bucket_sets"] - before["shared_bucket_sets"] == 1 and after["fallbacks"] == before["fallbacks"]
    assert after["bucket_per_lane"] - before["bucket_per_lane"] == 2  # two ranges, one reduction


def test_below_three_quarters_one_range(env, cref):
    from accumulation_amd import VariableBaseMSM
    c, ctx, ck, xy = env
    for n in (1 << 19, (3 << 18) - 1):
        sc = cref.rng_frs(c.curve_id, 0x4A3, n)
        before = ctx.pipeline_stats()["shared_bucket_sets"]
        got, ginf = VariableBaseMSM.multi_scalar_mul(ck, sc)
        assert ctx.pipeline_stats()["shared_bucket_sets"] == before
        ref, rinf = cref.msm(c.curve_id, xy[:n], sc)
        assert bool(ginf) == bool(rinf) and np.array_equal(got, ref)


@pytest.mark.parametrize("which", ["second_half_constant", "first_half_few_values", "a_quarter_equal_in_the_middle"])
def test_a_skewed_half_falls_back_and_is_exact(env, cref, which):
    from accumulation_amd import VariableBaseMSM
    c, ctx, ck, xy = env
    sc = cref.rng_frs(c.curve_id, 0x4A4, N)
    if which == "second_half_constant":
        sc[N // 2 + 5:] = sc[3]
    elif which == "first_half_few_values":
        sc[: N // 2] = sc[np.random.default_rng(1).integers(0, 3, N // 2)]
    else:
        sc[3 * N // 8: 5 * N // 8] = sc[11]
    got, ginf = VariableBaseMSM.multi_scalar_mul(ck, sc)
    ref, rinf = cref.msm(c.curve_id, xy[:N], sc)
    assert bool(ginf) == bool(rinf) and np.array_equal(got, ref), which


def test_pedersen_commit_and_the_switch(env, cref, monkeypatch):
    from accumulation_amd import Context, CommitterKey, PedersenCommitment, VariableBaseMSM, ffi
    c, ctx, ck, xy = env
    sc = cref.rng_frs(c.curve_id, 0x4A5, N)
    elems = cref.fr_to_mont(c.curve_id, sc)
    got, ginf = PedersenCommitment.commit(ck, elems, None)
    ref, rinf = cref.msm(c.curve_id, xy[:N], sc)
    assert bool(ginf) == bool(rinf) and np.array_equal(got, ref)
    monkeypatch.setenv("AMSM_HOST_HALVES", "0")
    ctx2 = Context(c.curve_id)
    try:
        ck2 = CommitterKey.load(ctx2, xy[:N], None, ffi.AMSM_BASES_PRECOMPUTE)
        got2, ginf2 = VariableBaseMSM.multi_scalar_mul(ck2, sc)
        assert ctx2.pipeline_stats()["shared_bucket_sets"] == 0
        assert bool(ginf2) == bool(rinf) and np.array_equal(got2, ref)
        ck2.free()
    finally:
        ctx2.close()
