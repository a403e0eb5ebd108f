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


def bench_msm(curve, log2n, reps=12):
    ctx = Context(curve)
    n = 1 << log2n
    t0 = time.time()
    ck = CommitterKey.generate(ctx, 0x5EED1001, n, ffi.AMSM_BASES_PRECOMPUTE)
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
    emit(kind="msm", curve="pallas" if curve == 0 else "bls12_381_g1", log2n=log2n, pairs_per_s=n / dt,
         ms_per_msm_pipelined=dt * 1e3, ms_per_msm_sync=dt_sync * 1e3, key_setup_s=round(t_key, 3))
    ck.free()
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
    emit(kind="vec", op="combine_vectors(n=2)", log2n=log2n, ms=t * 1e3, GBps=96 * n / t / 1e9, frac_hbm=96 * n / t / 1e9 / HBM)
    t = timed(ctx, lambda: compute_t_vecs(ctx, a, b, ch, n, None, skip_uncommitted=True))
    emit(kind="vec", op="compute_t_vecs(n=2, 4 in / 2 out)", log2n=log2n, ms=t * 1e3, GBps=192 * n / t / 1e9,
         frac_hbm=192 * n / t / 1e9 / HBM)
    ctx.close()


def bench_hp_as(log2n, reps=3):
    ctx = Context(ffi.AMSM_PALLAS)
    n = 1 << log2n
    ck = PedersenCommitment.setup(ctx, n, seed=0x5EED1001, flags=ffi.AMSM_BASES_PRECOMPUTE)
    pk, vk, dk = AS.index(ck)

    def make_input(seed):
        a = ctx.random_vector(seed, n, mont=True)
        b = ctx.random_vector(seed + 1, n, mont=True)
        pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, [a, b, compute_hp(ctx, a, b)], mont=True)
        return Accumulator(InputInstance(*[(pts[i], bool(infs[i])) for i in range(3)]), InputWitness(a, b, None))

    inp0, inp1 = make_input(100), make_input(200)
    acc0, _ = AS.prove(pk, [inp0], [], None, None)           # warm-up accumulation (like examples/scaling-as.rs:81-88)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        acc, proof = AS.prove(pk, [inp1], [acc0], None, None)  # timed: 1 input + 1 old accumulator => 2 MSMs
    dt = (time.perf_counter() - t0) / reps
    ok = AS.verify(ctx, vk, [inp1.instance], [acc0.instance], acc.instance, proof, None)
    t1 = time.perf_counter()
    dec = AS.decide(dk, acc, None)
    t_dec = time.perf_counter() - t1
    emit(kind="hp_as", log2n=log2n, n_all=2, zk=False, accumulations_per_s=1 / dt, prove_ms=dt * 1e3,
         decide_ms=t_dec * 1e3, verify_ok=bool(ok), decide_ok=bool(dec), msms_per_prove=2)
    ctx.close()


if __name__ == "__main__":
    quick = "--quick" in sys.argv
    only_vec = "--vec" in sys.argv
    if not only_vec:
        bench_msm(ffi.AMSM_PALLAS, 16)
        bench_msm(ffi.AMSM_PALLAS, 18)
        bench_msm(ffi.AMSM_PALLAS, 20)
        bench_msm(ffi.AMSM_BLS12_381_G1, 16 if quick else 20)
        if not quick:
            bench_msm(ffi.AMSM_PALLAS, 22, reps=6)
    bench_vec(20 if quick else 22)
    bench_hp_as(18 if quick else 22)
