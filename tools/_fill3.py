import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
n = 1 << 20
mode = sys.argv[1] if len(sys.argv) > 1 else "a"
ctx = Context(ffi.AMSM_PALLAS)
ck = CommitterKey.generate(ctx, 0x5EED1001 if mode != "a" else 1, n, ffi.AMSM_BASES_PRECOMPUTE)
seeds = [0x5EED0001 + 1000 * j for j in range(4)] if mode != "a" else [10, 11, 12, 13]
vecs = [ctx.random_vector(s, n, mont=False) for s in seeds]
ctx.synchronize()
def run(k):
    pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(k)], mont=False)
    return np.array(pts), np.array(infs)
run(3)
if mode == "c":
    ctx.set_profiling(True)
for k in (20, 20, 100, 20):
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(k)
    torch.cuda.synchronize()
    print(f"mode {mode} k={k}: {(time.perf_counter()-t)*1e3/k:.4f} ms per MSM", flush=True)
