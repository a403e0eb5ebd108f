import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
ctx = Context(ffi.AMSM_PALLAS)
n = 1 << 20
ck = CommitterKey.generate(ctx, 1, n)
vecs = [ctx.random_vector(10 + j, n, mont=False) for j in range(4)]
for k in (3, 1, 2, 3, 4, 6, 8, 12, 20, 40, 20, 8, 1):
    best = 1e9
    for rep in range(3):
        ctx.synchronize()
        t = time.perf_counter()
        VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(k)], mont=False)
        best = min(best, time.perf_counter() - t)
    print(f"k={k:3d}: {best*1e3:8.3f} ms total, {best*1e3/k:.4f} per MSM", flush=True)
