"""Batches the size a prover issues (2, 3, 6 MSMs per call) next to the 40-MSM batches of tools/mid_sizes.py: ms per CALL.
`AMSM_LIB_PATH` selects another build for an A/B.  Not a test."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi  # noqa: E402

for lg in (int(a) for a in (sys.argv[1:] or ["16", "18", "20", "22"])):
    ctx = Context(ffi.AMSM_PALLAS)
    n = 1 << lg
    ck = CommitterKey.generate(ctx, 1, n, ffi.AMSM_BASES_PRECOMPUTE)
    vecs = [ctx.random_vector(10 + j, n, mont=True) for j in range(4)]
    out = []
    for k in (1, 2, 3, 6, 40 if lg <= 20 else 12):
        reps = max(3, 60 // k)
        for _ in range(3):
            VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(k)], mont=True)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(k)], mont=True)
        out.append(f"{k} MSMs {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")
    print(f"pallas 2^{lg}: " + ", ".join(out), flush=True)
    ck.free()
    ctx.close()
