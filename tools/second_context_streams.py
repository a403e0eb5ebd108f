"""A SECOND context on a torch stream in a process that already holds one (profiles/r04_experiments.md section 9): stream creation
order decides which hardware queues the pipeline's three streams share.  Not a test."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
n = 1 << 20
def run(tag, ctx, prof):
    ck = CommitterKey.generate(ctx, 1, n)
    vecs = [ctx.random_vector(10 + j, n, mont=False) for j in range(4)]
    VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(3)], mont=False)
    ctx.set_profiling(prof)
    for k in (20, 20, 100):
        torch.cuda.synchronize()
        t = time.perf_counter()
        VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(k)], mont=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"{tag} profiling={prof} k={k}: {dt*1e3/k:.4f} ms per MSM", flush=True)
    ctx.set_profiling(False)
ctx = Context(ffi.AMSM_PALLAS)
run("own stream", ctx, False)
run("own stream", ctx, True)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    ctx2 = Context(ffi.AMSM_PALLAS, device=0, stream=st.cuda_stream)
    run("torch stream", ctx2, False)
    run("torch stream", ctx2, True)
