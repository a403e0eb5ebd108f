"""Round-4 check + rates of the generalised bucket-per-lane pipeline (not a test; tests/test_bpl_gpu.py and
tests/test_narrow_gpu.py hold the assertions): precomputed keys of 2^18 .. 2^20 generators with window widths adding up to 256
bits, plain keys with one bucket set per window, both curves -- every result compared with oracle/ark_msm.c, then batch and
blocking rates.

    python tools/r4_check.py [--no-check] [--sizes 18,19,20] [--curves pallas,bls] [--kinds precomp,plain]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--no-check", action="store_true")
ap.add_argument("--sizes", default="17,18,19,20")
ap.add_argument("--curves", default="pallas,bls")
ap.add_argument("--kinds", default="precomp,plain")
ap.add_argument("--reps", type=int, default=40)
ap.add_argument("--frac", type=int, default=0, help="MSMs of n >> frac pairs over the key of n generators")
args = ap.parse_args()

cref = None
if not args.no_check:
    from oracle import cref  # noqa: E402

    cref.build()

CURVES = {"pallas": ffi.AMSM_PALLAS, "bls": ffi.AMSM_BLS12_381_G1}
for cname in args.curves.split(","):
    curve = CURVES[cname]
    for kind in args.kinds.split(","):
        for lg in [int(x) for x in args.sizes.split(",")]:
            ctx = Context(curve)
            n = 1 << lg
            flags = ffi.AMSM_BASES_PRECOMPUTE if kind == "precomp" else ffi.AMSM_BASES_NO_PRECOMPUTE
            t0 = time.perf_counter()
            ck = CommitterKey.generate(ctx, 1, n, flags)
            ctx.synchronize()
            t_key = time.perf_counter() - t0
            vecs = [ctx.random_vector(10 + j, n, mont=False) for j in range(4)]
            if args.frac:
                vecs = [v.view(0, n >> args.frac) for v in vecs]
            ok = "unchecked"
            if cref is not None:
                xy, _ = ck.read()
                ok = "OK"
                for j, m in enumerate((n, n - 12345 if n > 20000 else n - 1, n // 2 + 1)):
                    sc = vecs[j].download()[:m]
                    off = (n - m) // 3
                    got, inf = VariableBaseMSM.multi_scalar_mul(ck, vecs[j].view(0, m), base_off=off)
                    ref, rinf = cref.msm(curve, xy[off:off + m], sc, threads=8)
                    if bool(inf) != bool(rinf) or not np.array_equal(got, ref):
                        ok = f"MISMATCH(m={m})"
                        break
            VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(8)], mont=False)
            ctx.synchronize()
            reps = args.reps
            t0 = time.perf_counter()
            VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(reps)], mont=False)
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for i in range(10):
                VariableBaseMSM.multi_scalar_mul(ck, vecs[i % 4], mont=False)
            ds = (time.perf_counter() - t0) / 10
            n = len(vecs[0])
            print(f"{cname} {kind} 2^{lg} (MSM {n}) window {ck.window_bits} key {t_key * 1e3:.0f} ms | {ok} | batch {n / dt / 1e6:.1f} M pairs/s "
                  f"({dt * 1e3:.4f} ms) | blocking {ds * 1e3:.4f} ms | {ctx.pipeline_stats()}", flush=True)
            ck.free()
            ctx.close()
