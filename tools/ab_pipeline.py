"""A/B of the library's documented switches (include/amsm.h "Environment") in ONE process: each configuration is a fresh Context
created under its environment overrides; configurations are measured round-robin.  (A second context of a process gets other
hardware queues than the first -- compare like positions, or use one process per configuration as tools/ab_share.py does.)

    python tools/ab_pipeline.py "" "AMSM_BPL=0" "AMSM_SHARE_BUCKETS=0" ...
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (initialises the HIP runtime the same way bench.py does)

from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi

LOG2N = int(os.environ.get("AB_LOG2N", "20"))
STEPS = int(os.environ.get("AB_STEPS", "60"))
ROUNDS = int(os.environ.get("AB_ROUNDS", "4"))
configs = sys.argv[1:] or [""]
n = 1 << LOG2N
envs = []
for c in configs:
    kv = dict(x.split("=", 1) for x in c.split(",") if x)
    envs.append(kv)
state = []
for kv in envs:
    for k, v in kv.items():
        os.environ[k] = v
    ctx = Context(ffi.AMSM_PALLAS)
    ck = CommitterKey.generate(ctx, 7, n, ffi.AMSM_BASES_PRECOMPUTE)
    vecs = [ctx.random_vector(100 + j, n, mont=False) for j in range(4)]
    for k in kv:
        del os.environ[k]
    state.append((ctx, ck, vecs, kv))
res = [[] for _ in state]
for r in range(ROUNDS + 1):
    for i, (ctx, ck, vecs, kv) in enumerate(state):
        for k, v in kv.items():
            os.environ[k] = v  # launch-time knobs are read per launch
        ctx.synchronize()
        t0 = time.perf_counter()
        VariableBaseMSM.multi_scalar_mul_batch(ck, [vecs[j % 4] for j in range(STEPS)], mont=False)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        for k in kv:
            del os.environ[k]
        if r:
            res[i].append(n * STEPS / dt / 1e6)
for c, r in zip(configs, res):
    print(f"{c or '(default)':40s} " + " ".join(f"{x:7.1f}" for x in r) + f"   median {sorted(r)[len(r) // 2]:7.1f} Mpairs/s")
