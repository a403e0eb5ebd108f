import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
for lg in (16, 17, 18):
    for env in ({"AMSM_BPL_MID": "1"}, {"AMSM_BPL_MID": "0"}):
        os.environ.update(env)
        ctx = Context(ffi.AMSM_PALLAS); n = 1 << lg
        ck = CommitterKey.generate(ctx, 7, n, ffi.AMSM_BASES_PRECOMPUTE)
        v = ctx.random_vector(100, n, mont=False)
        ctx.set_profiling(True)
        for _ in range(3): VariableBaseMSM.multi_scalar_mul(ck, v)
        st = ctx.stage_ms()
        vs = [v] * 30
        VariableBaseMSM.multi_scalar_mul_batch(ck, vs[:3], mont=False); ctx.synchronize()
        t0 = time.perf_counter(); VariableBaseMSM.multi_scalar_mul_batch(ck, vs, mont=False); dt = (time.perf_counter() - t0) / 30
        print(lg, env, "c", ck.window_bits, ctx.pipeline_stats(), {k: round(x, 3) for k, x in st.items() if x > 0.01}, "batch Mpairs/s", round(n / dt / 1e6, 1), flush=True)
        ck.free(); ctx.close()
