#!/usr/bin/env python3
"""Secondary measurements for the other BASELINE.json configs (bench.py stays the driver's contract):
  * MSM pairs/s at 2^16 Pallas (cfg2), 2^20 BLS12-381 G1 (cfg3), 2^18 / 2^22 Pallas (cfg4/cfg5 sizes)
  * scalar-field vector kernels at 2^22 elements against the HBM roofline (K4-K6)
  * hp_as accumulations/sec at 2^22 (cfg5; n_all = 2, no zk): t-vectors + 2 MSMs + 2 combines on the GPU
One JSON object per line.  Usage: python tools/bench_configs.py [--quick]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from accumulation_amd import CommitterKey, Context, PedersenCommitment, VariableBaseMSM, ffi  # noqa: E402
from accumulation_amd.hp_as import (ASForHadamardProducts as AS, Accumulator, InputInstance, InputWitness,  # noqa: E402
                                    combine_vectors, compute_hp, compute_t_vecs)
from accumulation_amd.scalar_field import Fr  # noqa: E402

HBM = 8000.0


def emit(**kw):
    print(json.dumps(kw), flush=True)


def bench_msm(curve, log2n, reps=40, plain=False):
    """plain: no precomputed multiples -- the true VariableBaseMSM shape (bases as handed over per call by an ark-ec patch; the
    folded keys of the IPA rounds)"""
    ctx = Context(curve)
    n = 1 << log2n
    t0 = time.time()
    ck = CommitterKey.generate(ctx, 0x5EED1001, n, ffi.AMSM_BASES_NO_PRECOMPUTE if plain else ffi.AMSM_BASES_PRECOMPUTE)
    t_key = time.time() - t0
    vecs = [ctx.random_vector(0x5EED0001 + j, n, mont=False) for j in range(4)]
    VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(3)], mont=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(reps)], mont=False)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for i in range(4):
        VariableBaseMSM.multi_scalar_mul(ck, vecs[i])
    dt_sync = (time.perf_counter() - t0) / 4
    emit(kind="msm_plain_key" if plain else "msm", curve="pallas" if curve == 0 else "bls12_381_g1", log2n=log2n, pairs_per_s=n / dt,
         ms_per_msm_pipelined=dt * 1e3, ms_per_msm_sync=dt_sync * 1e3, key_setup_s=round(t_key, 3), window_bits=ck.window_bits,
         pipeline_stats=ctx.pipeline_stats())
    ck.free()
    ctx.close()


def bench_degenerate(log2n):
    """the reference's harness commits to vec![x; len], all-zero and all-one vectors."""
    ctx = Context(ffi.AMSM_PALLAS)
    fr = Fr(ctx.curve)
    n = 1 << log2n
    ck = CommitterKey.generate(ctx, 0x5EED1001, n, ffi.AMSM_BASES_PRECOMPUTE)
    cases = {"uniform": ctx.random_vector(1, n, mont=True), "all_equal": ctx.fill(fr.to_limbs(0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF), n),
             "all_zero": ctx.fill(fr.to_limbs(0), n), "all_one": ctx.fill(fr.to_limbs(1), n)}
    for name, v in cases.items():
        VariableBaseMSM.multi_scalar_mul(ck, v, mont=True)
        t0 = time.perf_counter()
        for _ in range(3):
            VariableBaseMSM.multi_scalar_mul(ck, v, mont=True)
        emit(kind="msm_distribution", log2n=log2n, scalars=name, ms_per_msm_sync=(time.perf_counter() - t0) / 3 * 1e3)
    ck.free()
    ctx.close()


def bench_r1cs_nark_as(log2c, reps=8):
    """cfg4: r1cs_nark_as over 2^log2c constraints (DummyCircuit of examples/scaling-nark.rs:21-56), no zk:
    one accumulation = 1 input + 1 old accumulator => 2 SpMV + nested hp_as (2 MSMs) + witness combination."""
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as NAS, Input, InputInstance
    from accumulation_amd.sponge import Sha256Sponge
    ctx = Context(ffi.AMSM_PALLAS)
    fr = Fr(ctx.curve)
    nc, n_in = 1 << log2c, 5
    n_inst = n_in + 1
    A = [[(1, n_inst)] for _ in range(nc - 1)] + [[]]
    B = [[(1, n_inst + 1)] for _ in range(nc - 1)] + [[]]
    Cm = [[(1, 1)] for _ in range(nc - 1)] + [[]]
    t0 = time.time()
    ipk = nark.index(ctx, A, B, Cm, n_inst, n_inst + 2)
    t_index = time.time() - t0
    pk, vk, dk = NAS.index(ipk)

    def make_input(a, b):
        inst = [1, a * b % fr.r] + [a] * (n_in - 1)
        t1 = time.perf_counter()
        proof = nark.prove(ipk, inst, ctx.upload(fr.to_limbs_many([a, b])), False, NAS._sponges(Sha256Sponge())[0], None)
        return Input(InputInstance(inst, proof.first_msg), proof.second_msg), time.perf_counter() - t1

    i0, _ = make_input(3, 5)
    i1, t_nark = make_input(7, 11)
    acc0, _ = NAS.prove(pk, [i0], [], None, None)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        acc, proof = NAS.prove(pk, [i1], [acc0], None, None)
    dt = (time.perf_counter() - t0) / reps
    ok = NAS.verify(ctx, vk, [i1.instance], [acc0.instance], acc.instance, proof, None)
    t1 = time.perf_counter()
    dec = NAS.decide(dk, acc, None)
    emit(kind="r1cs_nark_as", log2_constraints=log2c, zk=False, accumulations_per_s=1 / dt, prove_ms=dt * 1e3,
         nark_prove_ms=t_nark * 1e3, decide_ms=(time.perf_counter() - t1) * 1e3, index_s=round(t_index, 2),
         verify_ok=bool(ok), decide_ok=bool(dec))
    ctx.close()


def bench_ipa(log2d, curve=ffi.AMSM_PALLAS, reps=3):
    """cfg1 / cfg2: ipa_pc_as with d+1 = 2^log2d: decide = one (d+1)-point MSM; prove = succinct checks + one IPA opening."""
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.ipa_pc_as import AtomicASForInnerProductArgPC as IAS, InputInstance as IpaInput
    ctx = Context(curve)
    fr = Fr(ctx.curve)
    d = (1 << log2d) - 1
    pp = IpaPC.setup(ctx, d)
    pk, vk, dk = IAS.index(pp, d)
    poly = ctx.random_vector(77, d + 1, mont=True)
    comm, rand = IpaPC.commit(pk.ipa_ck, poly, False, None)
    point = 0x123456789ABCDEF
    z = ctx.vector(d + 1)
    from accumulation_amd.engine import _ptr
    ffi.check(ctx._lib.amsm_vec_powers(ctx._h, _ptr(fr.to_limbs(point)), d + 1, z.ptr), "powers")
    value = IpaPC._inner_product(ctx, fr, poly, z)
    t0 = time.perf_counter()
    proof = IpaPC.open(pk.ipa_ck, poly, comm, point, rand, False, None)
    t_first = time.perf_counter() - t0  # includes first-use costs (code-object load, workspace growth)
    t0 = time.perf_counter()
    for _ in range(reps):
        proof = IpaPC.open(pk.ipa_ck, poly, comm, point, rand, False, None)
    t_open = (time.perf_counter() - t0) / reps
    inp = IpaInput(comm, point, value, proof)
    t0 = time.perf_counter()
    for _ in range(reps):
        acc, pr = IAS.prove(pk, [inp], [], None, None)
    t_prove = (time.perf_counter() - t0) / reps
    ok = IAS.verify(ctx, vk, [inp], [], acc.instance, pr, None)
    dec = IAS.decide(dk, acc, None)
    t0 = time.perf_counter()
    for _ in range(reps):
        dec = dec and IAS.decide(dk, acc, None)
    t_dec = (time.perf_counter() - t0) / reps
    emit(kind="ipa_pc_as", curve="pallas" if curve == ffi.AMSM_PALLAS else "bls12_381_g1", log2_degree_plus_1=log2d,
         ipa_open_first_call_ms=t_first * 1e3, ipa_open_ms=t_open * 1e3, prove_ms=t_prove * 1e3,
         accumulations_per_s=1 / t_prove, decide_ms=t_dec * 1e3, verify_ok=bool(ok), decide_ok=bool(dec))
    ctx.close()


def timed(ctx, fn, reps=20):
    fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps


def bench_vec(log2n):
    ctx = Context(ffi.AMSM_PALLAS)
    fr = Fr(ctx.curve)
    n = 1 << log2n
    a = [ctx.random_vector(10 + j, n, mont=True) for j in range(2)]
    b = [ctx.random_vector(20 + j, n, mont=True) for j in range(2)]
    ch = fr.to_limbs_many([3, 5])
    # outputs are freed by the FrVector destructor each call
    t = timed(ctx, lambda: compute_hp(ctx, a[0], b[0]))
    emit(kind="vec", op="compute_hp", log2n=log2n, ms=t * 1e3, GBps=96 * n / t / 1e9, frac_hbm=96 * n / t / 1e9 / HBM)
    t = timed(ctx, lambda: combine_vectors(ctx, a, ch))
    emit(kind="vec", op="combine_vectors(n=2)", log2n=log2n, ms=t * 1e3, GBps=96 * n / t / 1e9, frac_hbm=96 * n / t / 1e9 / HBM,
         multiplications_per_element=2)
    t = timed(ctx, lambda: compute_t_vecs(ctx, a, b, ch, n, None, skip_uncommitted=True))
    emit(kind="vec", op="compute_t_vecs(n=2, 4 in / 2 out)", log2n=log2n, ms=t * 1e3, GBps=192 * n / t / 1e9,
         frac_hbm=192 * n / t / 1e9 / HBM, multiplications_per_element=4)
    # the coefficients the PROVER passes: the first challenge of every combination is 1 (mu_0, nu^0, beta_0:
    # src/hp_as/mod.rs:241,266, src/r1cs_nark_as/mod.rs:444), and the kernels skip that multiplication
    ch1 = fr.to_limbs_many([1, 5])
    t = timed(ctx, lambda: combine_vectors(ctx, a, ch1))
    emit(kind="vec", op="combine_vectors(n=2, coefficients 1, x as in prove)", log2n=log2n, ms=t * 1e3, GBps=96 * n / t / 1e9,
         frac_hbm=96 * n / t / 1e9 / HBM, multiplications_per_element=1)
    t = timed(ctx, lambda: compute_t_vecs(ctx, a, b, ch1, n, None, skip_uncommitted=True))
    emit(kind="vec", op="compute_t_vecs(n=2, mu = 1, x as in prove)", log2n=log2n, ms=t * 1e3, GBps=192 * n / t / 1e9,
         frac_hbm=192 * n / t / 1e9 / HBM, multiplications_per_element=3)
    # the reference harness's shape (examples/scaling-as.rs:91-104): one input + two old accumulators = three (a, b) pairs
    a3 = a + [ctx.random_vector(12, n, mont=True)]
    b3 = b + [ctx.random_vector(22, n, mont=True)]
    for label, chs, mults_c, mults_t in (("arbitrary", [3, 5, 7], 3, 9), ("mu_0 = 1 as in prove", [1, 5, 7], 2, 8)):
        c3 = fr.to_limbs_many(chs)
        t = timed(ctx, lambda: combine_vectors(ctx, a3, c3))
        emit(kind="vec", op=f"combine_vectors(n=3, {label})", log2n=log2n, ms=t * 1e3, GBps=128 * n / t / 1e9,
             frac_hbm=128 * n / t / 1e9 / HBM, multiplications_per_element=mults_c)
        t = timed(ctx, lambda: compute_t_vecs(ctx, a3, b3, c3, n, None, skip_uncommitted=True))
        emit(kind="vec", op=f"compute_t_vecs(n=3, 6 in / 4 out, {label})", log2n=log2n, ms=t * 1e3, GBps=320 * n / t / 1e9,
             frac_hbm=320 * n / t / 1e9 / HBM, multiplications_per_element=mults_t)
    ctx.close()


def bench_trivial_pc_as(log2d, reps=5):
    """cfg0 (examples/scaling-as.rs:62-63,91-104): trivial_pc_as at degree 2^log2d - 1, one input accumulated into two old
    accumulators: 3 MSMs of <= 2^log2d points in prove (witness polynomials), one in decide.  The reference runs it on the
    CPU; here it goes through the same GPU path (bench.py reports the host backend's line beside it) -- far too small to fill the device."""
    from accumulation_amd.trivial_pc_as import ASForTrivialPC as TAS, Input as TInput, InputInstance as TInst, LabeledPolynomial, TrivialPC
    ctx = Context(ffi.AMSM_PALLAS)
    fr = Fr(ctx.curve)
    d = (1 << log2d) - 1
    pp = TrivialPC.setup(ctx, d)
    ck, _ = TrivialPC.trim(pp, d)
    pk, vk, dk = TAS.index(pp, d)
    rng = _Rng(5)

    def make_input():
        poly = LabeledPolynomial([rng.field() % fr.r for _ in range(d + 1)])
        point = rng.field() % fr.r
        return TInput(TInst(TrivialPC.commit(ck, poly), point, poly.evaluate(fr, point)), poly)

    a0, _ = TAS.prove(pk, [make_input()], [], None, None)
    a1, _ = TAS.prove(pk, [make_input()], [], None, None)
    inp = make_input()
    t0 = time.perf_counter()
    for _ in range(reps):
        acc, proof = TAS.prove(pk, [inp], [a0, a1], None, None)
    dt = (time.perf_counter() - t0) / reps
    ok = TAS.verify(ctx, vk, [inp.instance], [a0.instance, a1.instance], acc.instance, proof, None)
    t0 = time.perf_counter()
    dec = TAS.decide(dk, acc, None)
    t_dec = time.perf_counter() - t0
    emit(kind="trivial_pc_as", log2_degree_plus_1=log2d, accumulations_per_s=1 / dt, prove_ms=dt * 1e3, decide_ms=t_dec * 1e3,
         verify_ok=bool(ok), decide_ok=bool(dec))
    ctx.close()


class _Rng:
    """MakeZK::Enabled(rng) stand-in: .field() -> scalar < 2^254."""

    def __init__(self, seed):
        self.s = seed

    def field(self):
        import hashlib
        self.s += 1
        return int.from_bytes(hashlib.sha256(self.s.to_bytes(8, "little")).digest(), "little") >> 2


def bench_hp_as(log2n, reps=3, zk=False, constant_vectors=False):
    """cfg5.  constant_vectors: inputs generated like the reference's harness (a, b = vec![rand; len], src/hp_as/mod.rs:991-992):
    every MSM then sees all-equal scalars.  zk: MakeZK::Enabled -- 3 more commitments per accumulation, two of them to the
    prover's CONSTANT hiding vectors (:179-230)."""
    from accumulation_amd.hp_as import InputWitnessRandomness
    ctx = Context(ffi.AMSM_PALLAS)
    fr = Fr(ctx.curve)
    n = 1 << log2n
    ck = PedersenCommitment.setup(ctx, n, seed=0x5EED1001, flags=ffi.AMSM_BASES_PRECOMPUTE)
    pk, vk, dk = AS.index(ck)
    rng = _Rng(99) if zk else None

    def make_input(seed):
        if constant_vectors:
            a = ctx.fill(fr.to_limbs(_Rng(seed).field()), n)
            b = ctx.fill(fr.to_limbs(_Rng(seed + 1).field()), n)
        else:
            a = ctx.random_vector(seed, n, mont=True)
            b = ctx.random_vector(seed + 1, n, mont=True)
        if not zk:
            pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, [a, b, compute_hp(ctx, a, b)], mont=True)
            return Accumulator(InputInstance(*[(pts[i], bool(infs[i])) for i in range(3)]), InputWitness(a, b, None))
        rnd = InputWitnessRandomness(rng.field(), rng.field(), rng.field())
        c = [PedersenCommitment.commit(ck, v, fr.to_limbs(r))
             for v, r in zip((a, b, compute_hp(ctx, a, b)), (rnd.rand_1, rnd.rand_2, rnd.rand_3))]
        return Accumulator(InputInstance(*c), InputWitness(a, b, rnd))

    inp0, inp1 = make_input(100), make_input(200)
    acc0, _ = AS.prove(pk, [inp0], [], rng, None)           # warm-up accumulation (like examples/scaling-as.rs:81-88)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        acc, proof = AS.prove(pk, [inp1], [acc0], rng, None)  # timed: 1 input + 1 old accumulator => 2 MSMs (+3 zk)
    dt = (time.perf_counter() - t0) / reps
    ok = AS.verify(ctx, vk, [inp1.instance], [acc0.instance], acc.instance, proof, None)
    dec = AS.decide(dk, acc, None)
    t1 = time.perf_counter()
    dec = dec and AS.decide(dk, acc, None)
    t_dec = time.perf_counter() - t1
    emit(kind="hp_as", log2n=log2n, n_all=2, zk=zk, inputs="constant vectors (reference harness)" if constant_vectors else "uniform",
         accumulations_per_s=1 / dt, prove_ms=dt * 1e3, decide_ms=t_dec * 1e3, verify_ok=bool(ok), decide_ok=bool(dec),
         msms_per_prove=5 if zk else 2)
    ctx.close()


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    only_vec = "--vec" in sys.argv or "--vec-only" in sys.argv
    if not only_vec:
        bench_msm(ffi.AMSM_PALLAS, 16)
        bench_msm(ffi.AMSM_PALLAS, 18)
        bench_msm(ffi.AMSM_PALLAS, 20)
        bench_msm(ffi.AMSM_BLS12_381_G1, 16 if quick else 20)
        if not quick:
            bench_msm(ffi.AMSM_PALLAS, 22, reps=6)
            bench_msm(ffi.AMSM_PALLAS, 17)
            bench_msm(ffi.AMSM_PALLAS, 19)
            bench_msm(ffi.AMSM_BLS12_381_G1, 18)
            for lg in (17, 18, 19, 20, 22):  # round 4: plain keys on the bucket-per-lane pipeline
                bench_msm(ffi.AMSM_PALLAS, lg, plain=True, reps=6 if lg == 22 else 40)
            bench_msm(ffi.AMSM_BLS12_381_G1, 20, plain=True)
            for lg in (8, 12, 14):  # round 4: small keys are direct sums
                bench_msm(ffi.AMSM_PALLAS, lg)
            bench_msm(ffi.AMSM_BLS12_381_G1, 12)
    bench_vec(20 if quick else 22)
    if "--vec-only" in sys.argv:
        sys.exit(0)
    bench_hp_as(18 if quick else 22)
    bench_hp_as(18 if quick else 22, constant_vectors=True)
    bench_hp_as(18 if quick else 22, zk=True)
    bench_degenerate(16 if quick else 20)
    bench_r1cs_nark_as(12 if quick else 18)
    bench_trivial_pc_as(10)
    bench_ipa(10 if quick else 16)
    bench_ipa(10 if quick else 20, curve=ffi.AMSM_BLS12_381_G1)
