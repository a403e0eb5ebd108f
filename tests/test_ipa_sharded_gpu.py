"""ipa_pc / ipa_pc_as over a key SHARDED across the devices of a multi-device context (round 5: grouped MSMs and the IPA round shard
-- every shard sums its part of both index classes, the two sums per shard are folded on the host; the key is never folded).  On the one-GPU
test box the shards share device 0; proofs and accumulators must equal the single-device ones bit for bit."""
import numpy as np
import pytest

from accumulation_amd import Context, MultiContext, VariableBaseMSM, ffi
from tests.test_hp_as_scheme_gpu import SchemeRng

pytestmark = pytest.mark.gpu


def same_pt(a, b):
    return bool(a[1]) == bool(b[1]) and np.array_equal(np.asarray(a[0]), np.asarray(b[0]))


@pytest.mark.parametrize("devices", [(0, 0), (0, 0, 0), (0,) * 8], ids=["2_shards", "3_shards", "8_shards"])
@pytest.mark.parametrize("log_n", [6, 10, 16])
def test_grouped_msm_over_a_sharded_key(devices, log_n):
    _grouped(devices, log_n, ffi.AMSM_PALLAS)


def _grouped(devices, log_n, curve):
    from accumulation_amd import CommitterKey
    one, multi = Context(curve), MultiContext(curve, devices)
    try:
        n = (1 << log_n) + (5 if log_n == 10 else 0)
        k1, kN = CommitterKey.generate(one, 0x1DA1, n), CommitterKey.generate(multi, 0x1DA1, n)
        assert kN.num_shards == len(devices)
        v1, vN = one.random_vector(9, n, mont=True), multi.random_vector(9, n, mont=True)
        for shift in sorted({0, 1, max(0, log_n - 4), max(0, log_n - 2), log_n - 1}):
            a = VariableBaseMSM.multi_scalar_mul_grouped(k1, v1, shift, mont=True)
            before = multi.collectives
            b = VariableBaseMSM.multi_scalar_mul_grouped(kN, vN, shift, mont=True)
            assert multi.collectives - before == 0  # (the two class sums of every shard are folded on the host: no device exchange)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (devices, log_n, shift)
        k1.free()
        kN.free()
    finally:
        one.close()
        multi.close()


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
@pytest.mark.parametrize("devices,degree", [((0, 0), 63), ((0, 0, 0), 1023), ((0,) * 8, (1 << 16) - 1)], ids=["2x64", "3x1024", "8x65536"])
def test_ipa_open_and_accumulate_over_a_sharded_key(devices, degree, make_zk):
    """the whole opening (every L_j, R_j, the final key, c, the hiding terms) and one ipa_pc_as accumulation"""
    _open_and_accumulate(devices, degree, make_zk, ffi.AMSM_PALLAS)


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_sharded_key_over_bls12_381(make_zk):
    """the same over BLS12-381 (192-byte partial records, the 384-bit field's kernels; BASELINE config 3's curve): grouped MSMs over three
    shards and a whole 2^10 opening + accumulation"""
    if not make_zk:
        _grouped((0, 0, 0), 10, ffi.AMSM_BLS12_381_G1)
    _open_and_accumulate((0, 0, 0), 1023, make_zk, ffi.AMSM_BLS12_381_G1)


def _open_and_accumulate(devices, degree, make_zk, curve):
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.ipa_pc_as import AtomicASForInnerProductArgPC as AS, InputInstance
    from accumulation_amd.scalar_field import Fr
    out = []
    for ctx in (Context(curve), MultiContext(curve, devices)):
        try:
            fr = Fr(ctx.curve)
            pp = IpaPC.setup(ctx, degree, seed=0xABCDEF)
            pk, vk, dk = AS.index(pp, degree)
            assert pk.ipa_ck.comm_key.num_shards == (len(devices) if isinstance(ctx, MultiContext) else 1)
            rng = SchemeRng(4096)
            poly = ctx.random_vector(77, degree + 1, mont=True)
            comm, rand = IpaPC.commit(pk.ipa_ck, poly, make_zk, rng)
            point = rng.field() % fr.r
            before = ctx.collectives if isinstance(ctx, MultiContext) else 0
            proof = IpaPC.open(pk.ipa_ck, poly, comm, point, rand, make_zk, rng)
            if isinstance(ctx, MultiContext):  # the rounds fold their sums on the host; the final key's MSM (and the hiding
                assert 1 <= ctx.collectives - before <= 3  # polynomial's commitment) are exchanges of records between the devices
            z = ctx.vector(degree + 1)
            from accumulation_amd.engine import _ptr
            ffi.check(ctx._lib.amsm_vec_powers(ctx._h, _ptr(fr.to_limbs(point)), degree + 1, z.ptr), "powers")
            value = IpaPC._inner_product(ctx, fr, poly, z)
            assert IpaPC.check(pk.ipa_ck, comm, point, value, proof)
            acc, pr = AS.prove(pk, [InputInstance(comm, point, value, proof)], [], rng if make_zk else None, None)
            assert AS.decide(dk, acc, None)
            out.append((comm, proof, acc.instance))
        finally:
            ctx.close()
    (c1, p1, a1), (cN, pN, aN) = out
    assert same_pt(c1.comm, cN.comm)
    assert len(p1.l_vec) == len(pN.l_vec) == (degree + 1).bit_length() - 1
    for x, y in zip(p1.l_vec + p1.r_vec + [p1.final_comm_key], pN.l_vec + pN.r_vec + [pN.final_comm_key]):
        assert same_pt(x, y)
    assert p1.c == pN.c and p1.rand == pN.rand
    assert (p1.hiding_comm is None) == (pN.hiding_comm is None) and (p1.hiding_comm is None or same_pt(p1.hiding_comm, pN.hiding_comm))
    assert same_pt(a1.ipa_commitment.comm, aN.ipa_commitment.comm) and a1.point == aN.point and a1.evaluation == aN.evaluation
    assert same_pt(a1.ipa_proof.final_comm_key, aN.ipa_proof.final_comm_key) and a1.ipa_proof.c == aN.ipa_proof.c
