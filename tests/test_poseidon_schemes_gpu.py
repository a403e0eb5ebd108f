"""The Python scheme mirrors with the reference's sponge (accumulation_amd.sponge.PoseidonSponge, ark-sponge's Poseidon over
the base field) instead of the SHA-256 stand-in: the reference's template (src/lib.rs:334-395) on its heaviest scenario,
zk and no-zk, for hp_as and r1cs_nark_as; ipa_pc_as / trivial_pc_as through their `sponge_cls` hook."""
import pytest

from tests.test_hp_as_scheme_gpu import SchemeRng

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_hp_as_with_poseidon(make_zk):
    from accumulation_amd import Context, PedersenCommitment, ffi
    from accumulation_amd.hp_as import ASForHadamardProducts as AS
    from accumulation_amd.sponge import PoseidonSponge
    from tests.test_hp_as_scheme_gpu import VECTOR_LEN, generate_inputs
    ctx = Context(ffi.AMSM_PALLAS)
    try:
        ck = PedersenCommitment.setup(ctx, VECTOR_LEN, seed=4242)
        pk, vk, dk = AS.index(ck)
        inputs = generate_inputs(ctx, ck, 7, make_zk)
        rng = SchemeRng(7) if make_zk else None
        old, start = [], 0
        for k in [1, 1, 2, 3]:
            step = inputs[start:start + k]
            start += k
            acc, proof = AS.prove(pk, step, old, rng, PoseidonSponge())
            assert AS.verify(ctx, vk, [x.instance for x in step], [x.instance for x in old], acc.instance, proof, PoseidonSponge())
            # a verifier with the OTHER sponge derives different challenges and must reject
            assert not AS.verify(ctx, vk, [x.instance for x in step], [x.instance for x in old], acc.instance, proof, None) or len(step) + len(old) == 1 and not make_zk
            old.append(acc)
        assert AS.decide(dk, old[-1], None)
    finally:
        ctx.close()


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_r1cs_nark_as_with_poseidon(make_zk):
    from accumulation_amd import Context, ffi
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS, Input, InputInstance
    from accumulation_amd.scalar_field import Fr, MODULI
    from accumulation_amd.sponge import PoseidonSponge
    from tests.test_r1cs_nark_gpu import dummy_circuit
    ctx = Context(ffi.AMSM_PALLAS)
    try:
        r = MODULI[ctx.curve]
        fr = Fr(ctx.curve)
        A, B, C_, _, _ = dummy_circuit(5, 10, 2, 3, r)
        ipk = nark.index(ctx, A, B, C_, 6, 8, key_seed=31337)
        pk, vk, dk = AS.index(ipk)
        rng = SchemeRng(2024)
        inputs = []
        for _ in range(7):
            a, b = rng.field() % r, rng.field() % r
            _, _, _, inst, w = dummy_circuit(5, 10, a, b, r)
            nark_sponge, _, _ = AS._sponges(PoseidonSponge())
            proof = nark.prove(ipk, inst, ctx.upload(fr.to_limbs_many(w)), make_zk, nark_sponge, rng if make_zk else None)
            inputs.append(Input(InputInstance(inst, proof.first_msg), proof.second_msg))
        old, start = [], 0
        for k in [1, 1, 2, 3]:
            step = inputs[start:start + k]
            start += k
            acc, proof = AS.prove(pk, step, old, rng if make_zk else None, PoseidonSponge())
            assert AS.verify(ctx, vk, [x.instance for x in step], [x.instance for x in old], acc.instance, proof, PoseidonSponge())
            old.append(acc)
        assert AS.decide(dk, old[-1], None)
    finally:
        ctx.close()


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_pc_schemes_with_poseidon(make_zk):
    from accumulation_amd import Context, ffi
    from accumulation_amd import ipa_pc_as as M
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.sponge import PoseidonSponge
    from tests.test_ipa_gpu import DEGREE, generate_inputs
    ctx = Context(ffi.AMSM_PALLAS)
    AS = M.AtomicASForInnerProductArgPC
    old_as, old_pc = AS.sponge_cls, IpaPC.sponge_cls
    AS.sponge_cls = IpaPC.sponge_cls = PoseidonSponge
    try:
        pp = IpaPC.setup(ctx, DEGREE, seed=0xABCDEF)
        pk, vk, dk = AS.index(pp, DEGREE)
        rng = SchemeRng(4096)
        inputs = generate_inputs((ctx, pp), pk, 4, make_zk, rng)
        old, start = [], 0
        for k in [1, 1, 2]:
            step = inputs[start:start + k]
            start += k
            acc, proof = AS.prove(pk, step, [a.instance for a in old], rng if make_zk else None, None)
            assert AS.verify(ctx, vk, step, [a.instance for a in old], acc.instance, proof, None)
            old.append(acc)
        assert AS.decide(dk, old[-1], None)
    finally:
        AS.sponge_cls, IpaPC.sponge_cls = old_as, old_pc
        ctx.close()
