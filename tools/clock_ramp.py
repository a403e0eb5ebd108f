"""Clock ramp after idle (profiles/r04_experiments.md section 9): 2^20-pair MSMs over the 20-bit key in batches of 20 / 5 after
idle gaps of 50 ms / 1 s, ms per MSM -- what `bench.py`'s preheat is for.  Not a test."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
n = 1 << 20
ctx = Context(ffi.AMSM_PALLAS)
ck = CommitterKey.generate(ctx, 1, n, ffi.AMSM_BASES_PRECOMPUTE)
vecs = [ctx.random_vector(10 + j, n, mont=False) for j in range(4)]
ctx.synchronize()
def run(k):
    VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(k)], mont=False)
def timed(tag, k):
    torch.cuda.synchronize()
    t = time.perf_counter()
    run(k)
    torch.cuda.synchronize()
    print(f"{tag} k={k}: {(time.perf_counter()-t)*1e3/k:.4f} ms per MSM", flush=True)
run(3)
timed("after 3-MSM warm-up", 20)
timed("back to back", 20)
time.sleep(1.0)
timed("after 1 s idle", 20)
timed("back to back", 20)
time.sleep(0.05)
timed("after 50 ms idle", 20)
time.sleep(1.0)
run(3)
timed("1 s idle, then 3 MSMs", 20)
time.sleep(1.0)
for k in (5, 5, 5, 5, 5, 5):
    timed("after idle, chunks of 5", k)
