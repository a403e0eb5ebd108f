"""`combine_vectors` at n = 2 .. 8, arbitrary / unit first coefficient (profiles/r04_experiments.md section 11); run under
AMSM_VEC_SAT=0 for round 3's 9 x 29-limb kernels.  Not a test."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accumulation_amd import Context, ffi
from accumulation_amd.hp_as import combine_vectors
from accumulation_amd.scalar_field import Fr
ctx = Context(ffi.AMSM_PALLAS)
fr = Fr(ctx.curve)
n = 1 << 22
vs = [ctx.random_vector(10 + j, n, mont=True) for j in range(8)]
for k in (2, 3, 4, 5, 6, 8):
    for first in (3, 1):
        ch = fr.to_limbs_many([first] + [5 + 2 * j for j in range(k - 1)])
        for _ in range(5):
            combine_vectors(ctx, vs[:k], ch)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            combine_vectors(ctx, vs[:k], ch)
        ctx.synchronize()
        t = (time.perf_counter() - t0) / 40
        print(f"VEC_SAT={os.environ.get('AMSM_VEC_SAT','-')} combine n={k} first={first}: {t*1e3:.4f} ms  {32*(k+1)*n/t/8e12:.3f} of HBM")
