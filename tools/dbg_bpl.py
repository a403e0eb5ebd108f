import sys, numpy as np
sys.path.insert(0, ".")
from oracle import cref, pyref as o
from tests import helpers as h
from accumulation_amd import CommitterKey, Context, VariableBaseMSM
C = o.PALLAS; N = 1 << 20
ctx = Context(C.curve_id); ck = CommitterKey.generate(ctx, 0x5EED1001, N); xy, _ = ck.read()
rng = np.random.default_rng(7)
def run(name, sc):
    b = ctx.pipeline_stats(); got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc); a = ctx.pipeline_stats()
    ref, rinf = cref.msm(C.curve_id, xy, sc, threads=17)
    print(name, "ok" if (np.array_equal(got, ref) and bool(inf) == bool(rinf)) else "WRONG", "bpl", a["bucket_per_lane"] - b["bucket_per_lane"], "fallback", a["fallbacks"] - b["fallbacks"], flush=True)
sc = cref.rng_scalars(31, N); idx = rng.random(N) < 0.3; sc[idx] = h.scalars_to_np([o.rng_scalar(32, 0)])[0]
run("30pct equal", sc)
for bits in (19, 18, 16, 12, 10, 6):
    small = np.zeros((N, 4), dtype=np.uint64); small[:, 0] = rng.integers(0, 1 << bits, N, dtype=np.uint64)
    run(f"small<2^{bits}", small)
run("uniform", cref.rng_scalars(5, N))
