"""Batch / blocking MSM rates at 2^18 .. 2^20 pairs under the current environment (A/B of AMSM_BPL_MIN_LOG2 / AMSM_TOP_SPREAD):
    AMSM_BPL_MIN_LOG2=20 python tools/ab_mid_sizes.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi  # noqa: E402

for curve, name in ((ffi.AMSM_PALLAS, "pallas"), (ffi.AMSM_BLS12_381_G1, "bls12_381")):
    for lg in (18, 19, 20):
        ctx = Context(curve)
        n = 1 << lg
        t0 = time.perf_counter()
        ck = CommitterKey.generate(ctx, 1, n)
        ctx.synchronize()
        t_key = time.perf_counter() - t0
        vecs = [ctx.random_vector(10 + j, n, mont=False) for j in range(4)]
        VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(8)], mont=False)
        ctx.synchronize()
        reps = 60
        t0 = time.perf_counter()
        VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(reps)], mont=False)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for i in range(20):
            VariableBaseMSM.multi_scalar_mul(ck, vecs[i % 4], mont=False)
        ds = (time.perf_counter() - t0) / 20
        print(f"{name} 2^{lg} window {ck.window_bits} key {t_key * 1e3:.0f} ms | batch {n / dt / 1e6:.1f} M pairs/s ({dt * 1e3:.4f} ms) | "
              f"blocking {ds * 1e3:.4f} ms | {ctx.pipeline_stats()}", flush=True)
        ck.free()
        ctx.close()
