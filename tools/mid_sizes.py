"""Batch / blocking MSM rates at 2^16 .. 2^22 pairs, precomputed and plain keys, both curves (one fresh context per line):
what profiles/rNN_mid_sizes.txt holds.  Not a test."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi  # noqa: E402

only = [int(a) for a in sys.argv[1:]]  # e.g. `mid_sizes.py 18 19`: those sizes, Pallas, precomputed only (for a rocprofv3 trace)
for curve, name, sizes in ((ffi.AMSM_PALLAS, "pallas", tuple(only) or (16, 17, 18, 19, 20, 21, 22)),
                           (ffi.AMSM_BLS12_381_G1, "bls12_381", () if only else (18, 19, 20))):
    for lg in sizes:
        for flags, kind in ((ffi.AMSM_BASES_PRECOMPUTE, "precomputed"), (ffi.AMSM_BASES_NO_PRECOMPUTE, "plain")):
            if kind == "plain" and (lg > 21 or only):
                continue
            ctx = Context(curve)
            n = 1 << lg
            t0 = time.perf_counter()
            ck = CommitterKey.generate(ctx, 1, n, flags)
            ctx.synchronize()
            t_key = time.perf_counter() - t0
            vecs = [ctx.random_vector(10 + j, n, mont=False) for j in range(4)]
            reps = 40 if lg <= 20 else 12
            VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(8)], mont=False)
            VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(reps)], mont=False)
            ctx.synchronize()
            t0 = time.perf_counter()
            VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(reps)], mont=False)
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for i in range(10):
                VariableBaseMSM.multi_scalar_mul(ck, vecs[i % 4], mont=False)
            ds = (time.perf_counter() - t0) / 10
            print(f"{name} 2^{lg} {kind} window {ck.window_bits} key {t_key * 1e3:.0f} ms | batch {n / dt / 1e6:.1f} M pairs/s ({dt * 1e3:.4f} ms) | "
                  f"blocking {ds * 1e3:.4f} ms | {ctx.pipeline_stats()}", flush=True)
            ck.free()
            ctx.close()
