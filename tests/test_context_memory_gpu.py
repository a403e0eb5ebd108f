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
        n = 1 << 16  # (keys of up to 2^15 generators are summed straight from their table: a few KiB of workspace)
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


def test_one_exchange_per_dependent_commit_round():
    """Multi-device contexts exchange partial records ONCE per sharded MSM / commit call (amsm_ctx_collectives), and the
    scheme drivers batch every group of commitments that does not depend on a Fiat-Shamir challenge in between -- so a
    prover's number of exchanges is its number of dependent commit rounds, whatever the number of vectors:
      hp_as prove, no zk: 1 (the product-polynomial commitments, src/hp_as/mod.rs:354-388); zk: 2 (+ the three hiding
      commitments, :196-214, absorbed before mu is squeezed); decide: 1 (:906-922);
      r1cs_nark prove: 1 (:216-261, all eight commitments of the zk prover in one batch);
      r1cs_nark_as prove, no zk: 1 (the nested hp_as round); zk: 3 (the prover's randomness commitments
      src/r1cs_nark_as/mod.rs:394-410, then the nested scheme's two)."""
    from accumulation_amd import MultiContext, PedersenCommitment, ffi
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.hp_as import ASForHadamardProducts as HP
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as NAS, Input, InputInstance
    from accumulation_amd.scalar_field import MODULI, Fr
    from accumulation_amd.sponge import Sha256Sponge
    from tests.test_as_layers_vs_oracle_gpu import hp_inputs
    from tests.test_hp_as_scheme_gpu import SchemeRng
    from tests.test_r1cs_nark_gpu import dummy_circuit
    ctx = MultiContext(ffi.AMSM_PALLAS, (0, 0, 0))
    try:
        def delta(f):
            before = ctx.collectives
            out = f()
            return ctx.collectives - before, out
        n = 500
        ck = PedersenCommitment.setup(ctx, n, seed=5)
        for zk, expect in ((False, 1), (True, 2)):
            ins = hp_inputs(ctx, ck, n, 3, zk, 100)
            rng = SchemeRng(7) if zk else None
            d, (acc, proof) = delta(lambda: HP.prove(ck, ins, [], rng, None))
            assert d == expect, (zk, d)
            d, ok = delta(lambda: HP.decide(ck, acc, None))
            assert ok and d == 1, d
        r = MODULI[ctx.curve]
        fr = Fr(ctx.curve)
        A, B, C_, _, _ = dummy_circuit(5, 300, 2, 3, r)
        ipk = nark.index(ctx, A, B, C_, 6, 8, key_seed=9)
        pk, vk, dk = NAS.index(ipk)
        for zk, expect_nark, expect_as in ((False, 1, 1), (True, 1, 3)):
            rng = SchemeRng(11)
            ins = []
            for _ in range(2):
                a, b = rng.field() % r, rng.field() % r
                _, _, _, inst, w = dummy_circuit(5, 300, a, b, r)
                d, proof = delta(lambda: nark.prove(ipk, inst, ctx.upload(fr.to_limbs_many(w)), zk, NAS._sponges(Sha256Sponge())[0],
                                                    rng if zk else None))
                assert d == expect_nark, (zk, d)
                ins.append(Input(InputInstance(inst, proof.first_msg), proof.second_msg))
            d, (acc, _) = delta(lambda: NAS.prove(pk, ins, [], rng if zk else None, None))
            assert d == expect_as, (zk, d)
            assert NAS.decide(dk, acc, None)
    finally:
        ctx.close()
