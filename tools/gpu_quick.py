"""Quick GPU sanity run (not a test): device key generation + small MSMs vs the Python oracle."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from accumulation_amd import Context, CommitterKey, VariableBaseMSM, ffi
from oracle import pyref as o

def pts_from_np(c, xy, inf):
    return [o.point_from_mont_limbs(c, [int(v) for v in xy[i]], int(inf[i])) for i in range(xy.shape[0])]

for curve in (o.PALLAS, o.BLS12_381_G1):
    ctx = Context(curve.curve_id)
    for n, flags in ((1, 2), (5, 2), (33, 2), (300, 2), (300, 1), (1000, 1), (1000, 2)):
        t = time.time()
        ck = CommitterKey.generate(ctx, 0x5EED1001, n, flags)
        xy, inf = ck.read()
        pts = pts_from_np(curve, xy, inf)
        if n <= 33:
            ref_pts = o.rng_points(curve, 0x5EED1001, n)
            assert pts == ref_pts, f"generated bases differ n={n}"
        assert all(o.is_on_curve(curve, P) for P in pts)
        sc = o.rng_scalars(0x5EED0001, n)
        s_np = np.array([o.int_to_limbs(s, 4) for s in sc], dtype=np.uint64)
        out, is_inf = VariableBaseMSM.multi_scalar_mul(ck, s_np)
        got = o.point_from_mont_limbs(curve, [int(v) for v in out], is_inf)
        ref = o.msm_pippenger(curve, pts, sc)
        print(curve.name, n, "precomp" if ck.precomputed else "plain", "OK" if got == ref else "MISMATCH", f"{time.time()-t:.2f}s")
        assert got == ref
print("all ok")
