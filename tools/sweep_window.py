import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
import sys as _s
for log2n in [int(x) for x in _s.argv[1:]]:
    n = 1 << log2n
    for c in range(max(8, log2n - 6), min(21, log2n + 1)):
        ctx = Context(ffi.AMSM_PALLAS)
        ctx.set_window(c)  # (key creation and every MSM take this width: amsm_ctx_set_window)
        ck = CommitterKey.generate(ctx, 7, n, ffi.AMSM_BASES_PRECOMPUTE)
        vecs = [ctx.random_vector(100 + j, n, mont=False) for j in range(4)]
        VariableBaseMSM.multi_scalar_mul_batch(ck, vecs, mont=False)
        ctx.synchronize()
        t0 = time.perf_counter()
        VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[j % 4] for j in range(24)], mont=False)
        dt = (time.perf_counter() - t0) / 24
        t0 = time.perf_counter()
        for j in range(8):
            VariableBaseMSM.multi_scalar_mul(ck, vecs[j % 4], mont=False)
        ds = (time.perf_counter() - t0) / 8
        print(f"2^{log2n} c={c:2d}  pipelined {dt*1e3:7.3f} ms ({n/dt/1e6:6.1f} M/s)   sync {ds*1e3:7.3f} ms", flush=True)
        ck.free(); ctx.close()
