"""Plain keys (no precomputed multiples: the ark-ec call shape, folded IPA keys, amsm_msm_oneshot's ranges) at 2^16 .. 2^19 pairs:
the table row (msm_select.h) against the chunked pipeline at forced window widths.  M pairs/s in batches of 24 / ms blocking.
Not a test.   python tools/plain_mid_sweep.py [log2 sizes...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi  # noqa: E402


def rate(ctx, ck, vecs, n):
    VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(8)], mont=False)
    ctx.synchronize()
    t0 = time.perf_counter()
    VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(24)], mont=False)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / 24
    t0 = time.perf_counter()
    for i in range(8):
        VariableBaseMSM.multi_scalar_mul(ck, vecs[i % 4], mont=False)
    return n / dt / 1e6, (time.perf_counter() - t0) / 8 * 1e3


for lg in [int(a) for a in sys.argv[1:]] or [16, 17, 18, 19]:
    n = 1 << lg
    for curve, name in ((ffi.AMSM_PALLAS, "pallas"), (ffi.AMSM_BLS12_381_G1, "bls12_381")):
        out = []
        for label, env, window in (("table", {}, 0), ("chunked", {"AMSM_BPL_PLAIN": "0"}, 0), ("c=10", {}, 10), ("c=11", {}, 11), ("c=12", {}, 12),
                                   ("c=13", {}, 13), ("c=14", {}, 14), ("c=15", {}, 15)):
            os.environ.update(env)
            ctx = Context(curve)
            for k in env:
                del os.environ[k]
            if window:
                ctx.set_window(window)
            ck = CommitterKey.generate(ctx, 1, n, ffi.AMSM_BASES_NO_PRECOMPUTE)
            vecs = [ctx.random_vector(10 + j, n, mont=False) for j in range(4)]
            r, b = rate(ctx, ck, vecs, n)
            out.append(f"{label} {r:.0f} / {b:.3f}")
            ck.free()
            ctx.close()
        print(f"{name} plain 2^{lg}: " + " | ".join(out), flush=True)
