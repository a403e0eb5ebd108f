"""Host-slice batches against device-resident vectors (profiles/r04_experiments.md section 8): pageable / page-locked slices in
batches of 12 and 48 MSMs of 2^20 pairs, and the raw copy rates.  Not a test."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
ctx = Context(ffi.AMSM_PALLAS)
n = 1 << 20
ck = CommitterKey.generate(ctx, 1, n)
vecs = [ctx.random_vector(10 + j, n, mont=False) for j in range(4)]
h = [v.download() for v in vecs]
def run(tag, reps=12):
    VariableBaseMSM.multi_scalar_mul_batch_host(ck, [h[i % 4] for i in range(3)])
    t = time.perf_counter()
    VariableBaseMSM.multi_scalar_mul_batch_host(ck, [h[i % 4] for i in range(reps)])
    dt = (time.perf_counter() - t) / reps
    print(f"{tag}: {dt*1e3:.3f} ms per MSM = {n/dt/1e6:.0f} M pairs/s", flush=True)
run("pageable 12")
run("pageable 48", 48)
for a in h: ctx.host_register(a)
run("registered 12")
run("registered 48", 48)
dv = vecs
def rund(tag, reps):
    VariableBaseMSM.multi_scalar_mul_batch(ck, [dv[i % 4] for i in range(3)], mont=False)
    t = time.perf_counter()
    VariableBaseMSM.multi_scalar_mul_batch(ck, [dv[i % 4] for i in range(reps)], mont=False)
    dt = (time.perf_counter() - t) / reps
    print(f"{tag}: {dt*1e3:.3f} ms per MSM = {n/dt/1e6:.0f} M pairs/s", flush=True)
rund("device 12", 12)
rund("device 48", 48)
# raw copy rates
d = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
for name, src in (("pageable", torch.from_numpy(h[0].view(np.uint8).reshape(-1).copy())), ("registered", torch.from_numpy(h[0].view(np.uint8).reshape(-1))),
                  ("pinned(torch)", torch.from_numpy(h[0].view(np.uint8).reshape(-1).copy()).pin_memory())):
    d.copy_(src); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): d.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"copy {name}: {dt*1e3:.3f} ms = {n*32/dt/1e9:.1f} GB/s", flush=True)
