"""MSMs over vectors that look like an R1CS witness: a fraction of the scalars are 0 or 1 (boolean wires), the rest uniform.
ark-ec's VariableBaseMSM skips zeros and adds the points of unit scalars directly; here unit scalars all land in bucket 1 of the
lowest window -- a skewed digit distribution.  ms per MSM in batches of 12 and blocking, against the uniform vector.  Not a test."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi  # noqa: E402

for lg in (int(a) for a in (sys.argv[1:] or ["18", "20"])):
    for flags, kind in ((ffi.AMSM_BASES_PRECOMPUTE, "precomputed"), (ffi.AMSM_BASES_NO_PRECOMPUTE, "plain")):
        ctx = Context(ffi.AMSM_PALLAS)
        n = 1 << lg
        ck = CommitterKey.generate(ctx, 1, n, flags)
        base = ctx.random_vector(10, n, mont=False).download()
        rng = np.random.default_rng(3)
        for frac, label in ((0.0, "uniform"), (0.1, "10% booleans"), (0.5, "50% booleans"), (0.9, "90% booleans"), (0.5, "50% ones only")):
            h = base.copy()
            if frac:
                pick = rng.random(n) < frac
                vals = np.zeros((n, 4), dtype=np.uint64)
                vals[:, 0] = 1 if label.endswith("ones only") else rng.integers(0, 2, n)
                h[pick] = vals[pick]
            v = ctx.upload(h)
            for _ in range(2):
                VariableBaseMSM.multi_scalar_mul_batch(ck, [v] * 12, mont=False)
            ctx.synchronize()
            t0 = time.perf_counter()
            VariableBaseMSM.multi_scalar_mul_batch(ck, [v] * 12, mont=False)
            dt = (time.perf_counter() - t0) / 12
            t0 = time.perf_counter()
            for _ in range(5):
                VariableBaseMSM.multi_scalar_mul(ck, v, mont=False)
            ds = (time.perf_counter() - t0) / 5
            print(f"pallas 2^{lg} {kind} {label}: batch {dt * 1e3:.3f} ms per MSM, blocking {ds * 1e3:.3f} ms | {ctx.pipeline_stats()}", flush=True)
            v.free()
        ck.free()
        ctx.close()
