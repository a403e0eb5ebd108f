"""Host-side group algebra of the schemes, timed alone (no GPU): amsm_host_lincomb_batch on the shapes the drivers issue --
hp_as's instance combine (three jobs of 2 / 2 / 5 terms, powers of 128-bit challenges: full-size scalars), r1cs_nark_as's
blinded commitments (four jobs of 2 / 2 / 2 / 3 terms, 128-bit scalars) -- and one Poseidon absorb of a point.  Not a test."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from accumulation_amd import Context, CommitterKey, ffi  # noqa: E402
from accumulation_amd.engine import _ptr  # noqa: E402
from accumulation_amd.scalar_field import Fr, MODULI  # noqa: E402

lib = ffi.load()
for curve, name in ((ffi.AMSM_PALLAS, "pallas"), (ffi.AMSM_BLS12_381_G1, "bls12_381")):
    ctx = Context(curve, device=ffi.AMSM_DEVICE_HOST)
    fr = Fr(curve)
    r = MODULI[curve]
    key = CommitterKey.generate(ctx, 7, 16, ffi.AMSM_BASES_NO_PRECOMPUTE)
    xy, _ = key.read()
    w = xy.shape[1]
    rng = np.random.default_rng(5)

    def scal(bits):
        return int.from_bytes(rng.bytes(32), "little") % (1 << bits) % r

    def bench(label, shapes):
        nj = len(shapes)
        n_terms = (C.c_size_t * nj)(*[len(sh) for sh in shapes])
        xy_p, inf_p, sc_p = (C.c_void_p * nj)(), (C.c_void_p * nj)(), (C.c_void_p * nj)()
        keep, at = [], 0
        for j, sh in enumerate(shapes):
            pts = np.ascontiguousarray(xy[at:at + len(sh)])
            at = (at + len(sh)) % 8
            sc = np.ascontiguousarray(np.stack([fr.to_limbs(s) for s in sh]))
            inf = np.zeros((len(sh),), dtype=np.uint8)
            keep.append((pts, sc, inf))
            xy_p[j], inf_p[j], sc_p[j] = pts.ctypes.data, inf.ctypes.data, sc.ctypes.data
        out = np.zeros((nj, w), dtype=np.uint64)
        oinf = np.zeros((nj,), dtype=np.uint8)
        for _ in range(20):
            lib.amsm_host_lincomb_batch(curve, nj, n_terms, xy_p, inf_p, sc_p, _ptr(out), _ptr(oinf))
        t0 = time.perf_counter()
        for _ in range(200):
            lib.amsm_host_lincomb_batch(curve, nj, n_terms, xy_p, inf_p, sc_p, _ptr(out), _ptr(oinf))
        print(f"{name} {label}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call ({lib.amsm_host_threads()} helper threads)", flush=True)

    nu = scal(128)
    bench("hp_as combine (2 / 2 / 5 terms, powers of a 128-bit challenge)",
          [[1, scal(128)], [1, nu], [1, nu, nu * nu % r, pow(nu, 3, r), scal(128) * nu % r]])
    g = scal(128)
    bench("r1cs_nark_as blinded commitments (2 / 2 / 2 / 3 terms, 128-bit)", [[1, g], [1, g], [1, g], [1, g, g * g % r]])
    bench("r1cs_nark_as beta combine (3 jobs x 4 terms, 128-bit)", [[1, scal(128), scal(128), scal(128)]] * 3)
    bench("one job of 34 terms, 128-bit (an ipa_pc succinct check)", [[scal(128) for _ in range(34)]])
    chk = [1, scal(255)] + [scal(255) if i % 2 == 0 else scal(128) for i in range(32)]
    bench("three ipa_pc succinct checks in one call (3 x (34 + 2) terms, half of them full-size inverses)", [chk, [scal(255), scal(255)]] * 3)
    ctx.close()
