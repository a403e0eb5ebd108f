"""Per-stage device times of ONE blocking MSM under different knobs (library hipEvent pairs, amsm_ctx_set_profiling):
    python tools/l0_sweep.py "AMSM_K0=16" "AMSM_K0=24" "AMSM_L0_LDS_PAD=40000" ...
Each configuration = a fresh Context created under its environment overrides; median over AB_ROUNDS calls."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi

LOG2N = int(os.environ.get("AB_LOG2N", "20"))
ROUNDS = int(os.environ.get("AB_ROUNDS", "9"))
n = 1 << LOG2N
configs = sys.argv[1:] or [""]
for c in configs:
    kv = dict(x.split("=", 1) for x in c.split(",") if x)
    for k, v in kv.items():
        os.environ[k] = v
    ctx = Context(ffi.AMSM_PALLAS)
    ck = CommitterKey.generate(ctx, 7, n, ffi.AMSM_BASES_PRECOMPUTE)
    vec = ctx.random_vector(100, n, mont=False)
    ctx.set_profiling(True)
    rows = []
    for r in range(ROUNDS + 1):
        VariableBaseMSM.multi_scalar_mul(ck, vec, mont=False)
        if r:
            rows.append(ctx.stage_ms())
    for k in kv:
        del os.environ[k]
    names = list(rows[0].keys())
    med = {k: sorted(x[k] for x in rows)[len(rows) // 2] for k in names}
    print(f"{c or '(default)':40s} " + " ".join(f"{k}={med[k]:.3f}" for k in names) + f"  total={sum(med.values()):.3f} ms")
    ctx.close()
