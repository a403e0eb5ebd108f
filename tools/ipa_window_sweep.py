"""Which window width should a precomputed key of 2^15 .. 2^17 generators carry when what runs over it is a CHAIN of blocking
grouped MSMs (an ipa_pc opening: one per round, each waiting for the previous round's challenge)?  For every width: the key is built
with it (amsm_ctx_set_window before the key, reset after: the MSMs then choose their pipeline from the table as usual), then blocking
plain and grouped MSMs are timed.  One fresh context per line.  Not a test."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi  # noqa: E402

curves = ((ffi.AMSM_PALLAS, "pallas"), (ffi.AMSM_BLS12_381_G1, "bls12_381"))
for curve, name in curves:
    for lg in (int(a) for a in (sys.argv[1:] or ["16", "17"])):
        for w in (0, 8, 10, 11, 12, 13, 14, 15, 16, 17):
            ctx = Context(curve)
            n = 1 << lg
            if w:
                ctx.set_window(w)
            ck = CommitterKey.generate(ctx, 1, n, ffi.AMSM_BASES_PRECOMPUTE | ffi.AMSM_BASES_NO_DIRECT_TABLE)
            ctx.set_window(0)
            vecs = [ctx.random_vector(10 + j, n, mont=True) for j in range(4)]
            res = {}
            for label, fn in (("plain", lambda v: VariableBaseMSM.multi_scalar_mul(ck, v, mont=True)),
                              ("grouped_top", lambda v: VariableBaseMSM.multi_scalar_mul_grouped(ck, v, lg - 1, mont=True)),
                              ("grouped_3", lambda v: VariableBaseMSM.multi_scalar_mul_grouped(ck, v, 3, mont=True))):
                for i in range(6):
                    fn(vecs[i % 4])
                t0 = time.perf_counter()
                for i in range(30):
                    fn(vecs[i % 4])
                res[label] = (time.perf_counter() - t0) / 30 * 1e3
            print(f"{name} 2^{lg} built with {w or 'default'} -> window {ck.window_bits} | blocking ms: " +
                  ", ".join(f"{k} {v:.4f}" for k, v in res.items()) + f" | {ctx.pipeline_stats()}", flush=True)
            ck.free()
            ctx.close()
