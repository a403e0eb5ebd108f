"""The PRODUCT (HIP MSM through the C ABI, the Pedersen commitment, a whole hp_as accumulation with the Poseidon sponge)
against vectors produced by the real arkworks stack and the reference crate itself (tools/ark_vectors).  Skips with "parity
unpinned: <file> absent" until the files exist -- see tests/ark_vectors.py."""
import numpy as np
import pytest

from oracle import pyref as o
from oracle import pyref_ser as ser
from tests import ark_vectors as av
from tests import helpers as h

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve", ["pallas", "bls12_381_g1"])
@pytest.mark.parametrize("flags", [1, 2], ids=["precomputed_key", "plain_key"])
def test_msm_vectors_vs_product(curve, flags):
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.CURVES[curve]
    cases = av.load("ark_msm.json")[curve]
    ctx = Context(c.curve_id)
    try:
        for case in cases:
            if case["kind"] == "seeded":
                n = case["n"]
                ck = CommitterKey.generate(ctx, case["seed_points"], n, flags)
                got, inf = VariableBaseMSM.multi_scalar_mul(ck, ctx.random_vector(case["seed_scalars"], n, mont=False))
            else:
                pts, sc = [av.pt(p) for p in case["points"]], av.ints(case["scalars"])
                k = min(len(pts), len(sc))
                xy, pinf = h.points_to_np(c, pts[:k])
                ck = CommitterKey.load(ctx, xy, pinf, flags)
                got, inf = VariableBaseMSM.multi_scalar_mul(ck, h.scalars_to_np(sc[:k]))
            assert h.np_to_point(c, got, inf) == av.pt(case["expected"]), case.get("name", case.get("n"))
            ck.free()
    finally:
        ctx.close()


def test_pedersen_vectors_vs_product():
    from accumulation_amd import CommitterKey, Context, PedersenCommitment
    c = o.PALLAS
    cases = av.load("ark_pedersen.json")["cases"]
    ctx = Context(c.curve_id)
    try:
        for case in cases:
            gens, H = [av.pt(p) for p in case["generators"]], av.pt(case["hiding_generator"])
            xy, _ = h.points_to_np(c, gens)
            hw, _ = h.points_to_np(c, [H])
            ck = CommitterKey.load(ctx, xy, None, 0, hiding_generator=hw[0].copy())
            elems = h.fr_mont_np(c, av.ints(case["elems"]))
            got, inf = PedersenCommitment.commit(ck, elems, None)
            assert h.np_to_point(c, got, inf) == av.pt(case["commit"])
            got, inf = PedersenCommitment.commit(ck, elems, h.fr_mont_np(c, [int(case["rand"], 16)])[0])
            assert h.np_to_point(c, got, inf) == av.pt(case["commit_hiding"])
            ck.free()
    finally:
        ctx.close()


def _read_points(c, buf, off, n):
    sz = ser.point_size(c, True)
    return [ser.point_deserialize(c, buf[off + i * sz: off + (i + 1) * sz], True) for i in range(n)], off + n * sz


def _read_vec_fr(c, buf, off):
    n = int.from_bytes(buf[off:off + 8], "little")
    off += 8
    return [ser.fr_deserialize(c, buf[off + 32 * i: off + 32 * (i + 1)]) for i in range(n)], off + 32 * n


def test_hp_as_accumulation_vs_reference_crate():
    """ASForHadamardProducts::prove without zk (src/hp_as/mod.rs:646-813) through the product's driver and Poseidon sponge:
    accumulator instance (three commitments), witness vectors and the proof's low / high commitments, as the reference
    serialised them"""
    from accumulation_amd import CommitterKey, Context, PedersenCommitment
    from accumulation_amd.hp_as import Accumulator as Input, ASForHadamardProducts, InputInstance, InputWitness, compute_hp
    from accumulation_amd.sponge import PoseidonSponge
    c = o.PALLAS
    cases = av.load("ark_hp_as.json")["cases"]
    ctx = Context(c.curve_id)
    try:
        for case in cases:
            gens, H = [av.pt(p) for p in case["generators"]], av.pt(case["hiding_generator"])
            xy, _ = h.points_to_np(c, gens)
            hw, _ = h.points_to_np(c, [H])
            ck = CommitterKey.load(ctx, xy, None, 0, hiding_generator=hw[0].copy())
            ins = []
            for inp in case["inputs"]:
                a, b = ctx.upload(h.fr_mont_np(c, av.ints(inp["a"]))), ctx.upload(h.fr_mont_np(c, av.ints(inp["b"])))
                prod = compute_hp(ctx, a, b)
                inst = InputInstance(PedersenCommitment.commit(ck, a, None), PedersenCommitment.commit(ck, b, None),
                                     PedersenCommitment.commit(ck, prod, None))
                ins.append(Input(inst, InputWitness(a, b, None)))
            AS = ASForHadamardProducts
            pk, vk, dk = AS.index(ck)
            acc, proof = AS.prove(pk, ins, [], None, PoseidonSponge(ctx.curve))
            assert AS.decide(dk, acc, PoseidonSponge(ctx.curve))
            bi, bw, bp = (bytes.fromhex(case[k]) for k in ("accumulator_instance", "accumulator_witness", "proof"))
            want_inst, _ = _read_points(c, bi, 0, 3)
            pt = lambda p: h.np_to_point(c, p[0], p[1])
            assert [pt(acc.instance.comm_1), pt(acc.instance.comm_2), pt(acc.instance.comm_3)] == want_inst
            wa, off = _read_vec_fr(c, bw, 0)
            wb, off = _read_vec_fr(c, bw, off)
            assert bw[off] == 0  # Option<randomness>::None
            assert h.fr_from_mont_np(c, acc.witness.a_vec.download()) == wa
            assert h.fr_from_mont_np(c, acc.witness.b_vec.download()) == wb
            n_low = int.from_bytes(bp[0:8], "little")
            low, off = _read_points(c, bp, 8, n_low)
            n_high = int.from_bytes(bp[off:off + 8], "little")
            high, off = _read_points(c, bp, off + 8, n_high)
            assert [pt(p) for p in proof.product_poly_comm.low] == low and [pt(p) for p in proof.product_poly_comm.high] == high
            assert bp[off] == 0  # Option<hiding_comms>::None
    finally:
        ctx.close()
