"""The context's device memory is visible and bounded from outside (amsm_ctx_memory / amsm_ctx_trim): the grow-only MSM
workspace, the live vectors and the caching allocator behind amsm_dev_alloc / amsm_dev_free."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_workspace_allocator_and_trim(cref):
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
    ctx = Context(ffi.AMSM_PALLAS)
    try:
        lib = ctx._lib
        m0 = ctx.memory()
        assert m0["workspace_bytes"] == 0 and m0["vectors_live_bytes"] == 0
        n = 1 << 14
        ck = CommitterKey.generate(ctx, 3, n)
        v = ctx.random_vector(5, n, mont=False)
        ref, ref_inf = VariableBaseMSM.multi_scalar_mul(ck, v)
        m1 = ctx.memory()
        assert m1["workspace_bytes"] > (1 << 20) and m1["vectors_live_bytes"] >= n * 32
        # the allocator hands a freed buffer out again instead of going to the driver
        p = C.c_void_p()
        assert lib.amsm_dev_alloc(ctx._h, 12345, C.byref(p)) == 0
        first = p.value
        assert lib.amsm_dev_free(ctx._h, p) == 0
        assert ctx.memory()["vectors_pooled_bytes"] >= 12345
        q = C.c_void_p()
        assert lib.amsm_dev_alloc(ctx._h, 12300, C.byref(q)) == 0  # same 256-byte granule
        assert q.value == first
        assert lib.amsm_dev_free(ctx._h, q) == 0
        # trim: workspace and free lists go, live vectors and the key stay usable
        ctx.trim()
        m2 = ctx.memory()
        assert m2["workspace_bytes"] == 0 and m2["vectors_pooled_bytes"] == 0 and m2["vectors_live_bytes"] >= n * 32
        again, again_inf = VariableBaseMSM.multi_scalar_mul(ck, v)
        assert np.array_equal(again, ref) and again_inf == ref_inf
        xy, _ = ck.read()
        cpu, cpu_inf = cref.msm(ffi.AMSM_PALLAS, xy, v.download())
        assert np.array_equal(again, cpu) and bool(cpu_inf) == bool(again_inf)
    finally:
        ctx.close()


def test_schemes_run_unchanged_on_a_multi_device_context():
    """hp_as prove / verify / decide on a key sharded over two shards (both on GPU 0 here): the accumulator is bit-identical
    to the single-device one -- the scheme code does not know the key is sharded."""
    from accumulation_amd import Context, MultiContext, PedersenCommitment, ffi
    from accumulation_amd.hp_as import ASForHadamardProducts as AS
    from tests.test_hp_as_scheme_gpu import SchemeRng
    from tests.test_as_layers_vs_oracle_gpu import hp_inputs, hp_to_oracle
    n = 600
    accs = []
    for make in (lambda: Context(ffi.AMSM_PALLAS), lambda: MultiContext(ffi.AMSM_PALLAS, (0, 0))):
        ctx = make()
        try:
            ck = PedersenCommitment.setup(ctx, n, seed=991)
            if isinstance(ctx, MultiContext):
                assert ctx._lib.amsm_bases_num_shards(ck._h) == 2 and ctx.collective == "peer-copy"
            pk, vk, dk = AS.index(ck)
            ins = hp_inputs(ctx, ck, n, 3, True, 100)
            rng = SchemeRng(7)
            a1, p1 = AS.prove(pk, ins[:2], [], rng, None)
            a2, p2 = AS.prove(pk, ins[2:], [a1], rng, None)
            assert AS.verify(ctx, vk, [x.instance for x in ins[2:]], [a1.instance], a2.instance, p2, None)
            assert AS.decide(dk, a2, None)
            accs.append(hp_to_oracle(a2))
        finally:
            ctx.close()
    assert accs[0] == accs[1]
