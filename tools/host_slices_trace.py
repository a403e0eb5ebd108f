"""Timeline source for the host-slice question: 12 host-slice MSMs next to 12 device-vector MSMs of 2^20
pairs, phases separated by 100 ms of idle so that a rocprofv3 --kernel-trace --memory-copy-trace timeline can be cut by phase.
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/r5_hs -- python3 tools/host_slices_trace.py
Prints wall-clock per phase; tools/trace_gaps.py attributes the timeline.  Not a test."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
ctx = Context(ffi.AMSM_PALLAS)
n = 1 << 20
reps = int(os.environ.get("REPS", "12"))
ck = CommitterKey.generate(ctx, 1, n)
vecs = [ctx.random_vector(10 + j, n, mont=False) for j in range(4)]
h = [v.download() for v in vecs]
def phase(tag, fn):
    ctx.synchronize(); time.sleep(0.1)
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"PHASE {tag} t0={t0:.6f} wall_ms={dt*1e3:.3f} per_msm_ms={dt*1e3/reps:.4f}", flush=True)
for _ in range(3):  # steady clocks, grown workspaces
    VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(reps)], mont=False)
    VariableBaseMSM.multi_scalar_mul_batch_host(ck, [h[i % 4] for i in range(reps)])
phase("device", lambda: VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(reps)], mont=False))
phase("host", lambda: VariableBaseMSM.multi_scalar_mul_batch_host(ck, [h[i % 4] for i in range(reps)]))
phase("device", lambda: VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[i % 4] for i in range(reps)], mont=False))
phase("host", lambda: VariableBaseMSM.multi_scalar_mul_batch_host(ck, [h[i % 4] for i in range(reps)]))
for a in h: ctx.host_register(a)
phase("host_registered", lambda: VariableBaseMSM.multi_scalar_mul_batch_host(ck, [h[i % 4] for i in range(reps)]))
phase("host_registered", lambda: VariableBaseMSM.multi_scalar_mul_batch_host(ck, [h[i % 4] for i in range(reps)]))
