"""Round 4 experiment, opt-in (AMSM_RADIX=1): MIXED-RADIX digits -- W digits in radix R = m 2^k with R^W just above 2^256
(13 digits of 13 * 2^16 for keys of more than 2^19 generators, 14 digits of 5 * 2^16 for keys of 2^18 / 2^19), every digit
uniform over (-R/2, R/2], the table level w = R^w G (MsmGeom::radix_m, vec_kernels.h `digit_step`, `k_precompute_level`'s
multiply-by-m step).  It measured no faster than the power-of-two widths (profiles/r04_experiments.md section 10) and is not
the default; these tests keep it exact: MSMs, ranges, grouped MSMs and key folds over such keys against the CPU restatement
oracle/ark_msm.c, bit for bit.  Replaces ark-ec `VariableBaseMSM::multi_scalar_mul` (ext; call sites src/hp_as/mod.rs:377,
src/ipa_pc_as/mod.rs:454)."""
import os

import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h

pytestmark = pytest.mark.gpu
CASES = [(o.PALLAS, (1 << 18) + 3, 5, 163840), (o.PALLAS, 1 << 20, 13, 425984), (o.BLS12_381_G1, (1 << 19) + 1, 13, 425984)]


@pytest.fixture(scope="module", params=CASES, ids=lambda p: f"{p[0].name}-{p[1]}")
def env(request):
    from accumulation_amd import CommitterKey, Context
    c, n, m, nb = request.param
    os.environ["AMSM_RADIX"] = "1"
    os.environ["AMSM_TWO_VALUED"] = "0"
    try:
        ctx = Context(c.curve_id)
    finally:
        del os.environ["AMSM_RADIX"], os.environ["AMSM_TWO_VALUED"]
    ck = CommitterKey.generate(ctx, 0x5EED5001 + n, n)
    assert ck.precomputed
    xy, inf = ck.read()
    yield c, ctx, ck, xy, inf, n
    ck.free()
    ctx.close()


def _msm(c, ctx, ck, xy, sc, cref, off=0, mont=False):
    from accumulation_amd import VariableBaseMSM
    before = ctx.pipeline_stats()
    got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc, base_off=off, mont=mont)
    after = ctx.pipeline_stats()
    n = min(len(sc), len(xy) - off)
    ref, rinf = cref.msm(c.curve_id, xy[off:off + n], sc[:n], threads=8)
    assert bool(inf) == bool(rinf) and np.array_equal(got, ref)
    return after["bucket_per_lane"] - before["bucket_per_lane"], after["fallbacks"] - before["fallbacks"]


def test_whole_key_and_ranges(env, cref):
    c, ctx, ck, xy, inf, n = env
    assert _msm(c, ctx, ck, xy, cref.rng_scalars(0xB000, n), cref) == (1, 0)
    half = n // 2 + 11
    assert _msm(c, ctx, ck, xy, cref.rng_scalars(0xB001, half), cref, off=n - half) == (1, 0)
    for short in (4099, 1):  # far below a pair per two buckets: the other pipelines, over the same mixed-radix table
        assert _msm(c, ctx, ck, xy, cref.rng_scalars(0xB002 + short, short), cref, off=5) == (0, 0)


def test_edge_scalars(env, cref):
    """0, 1, r - 1, values whose digits sit on the fold boundary R / 2 in every position, all-ones digits"""
    c, ctx, ck, xy, inf, n = env
    m = 5 if n <= (1 << 19) and c is o.PALLAS else 13
    R = m << 16
    W = 14 if m == 5 else 13
    sc = cref.rng_scalars(0xB100, n)
    specials = [0, 1, c.r - 1, c.r - 2, (1 << 254) % c.r, sum((R // 2) * R ** w for w in range(W)) % c.r,
                sum((R // 2 + 1) * R ** w for w in range(W)) % c.r, sum((R - 1) * R ** w for w in range(W)) % c.r, R ** (W - 1) % c.r,
                (R ** (W - 1) - 1) % c.r]
    for j, v in enumerate(specials * 40):
        sc[(j * 2617) % n] = o.int_to_limbs(v, 4)
    assert _msm(c, ctx, ck, xy, sc, cref) == (1, 0)


def test_montgomery_form_scalars(env, cref):
    c, ctx, ck, xy, inf, n = env
    sc = cref.rng_scalars(0xB200, n)
    from accumulation_amd import VariableBaseMSM
    d = ctx.upload(cref.fr_to_mont(c.curve_id, sc))
    got, i0 = VariableBaseMSM.multi_scalar_mul(ck, d, mont=True)
    ref, rinf = cref.msm(c.curve_id, xy, sc, threads=8)
    assert bool(i0) == bool(rinf) and np.array_equal(got, ref)


def test_constant_vector_falls_back_and_is_exact(env, cref):
    c, ctx, ck, xy, inf, n = env
    sc = np.tile(np.array(o.int_to_limbs(0x1234567 % c.r, 4), dtype=np.uint64), (n, 1))
    took, fell = _msm(c, ctx, ck, xy, sc, cref)
    assert took == 0 or fell == 1


def test_grouped_msm(env, cref):
    from accumulation_amd import VariableBaseMSM
    c, ctx, ck, xy, inf, n = env
    sc = cref.rng_scalars(0xB300, n)
    d = ctx.upload(sc)
    for shift in (0, 9):
        pts, infs = VariableBaseMSM.multi_scalar_mul_grouped(ck, d, shift, mont=False)
        cls = (np.arange(n) >> shift) & 1
        for g in (0, 1):
            ref, rinf = cref.msm(c.curve_id, xy[cls == g], sc[cls == g], threads=8)
            assert bool(infs[g]) == bool(rinf) and np.array_equal(pts[g], ref), (shift, g)


def test_fold_through_the_mixed_radix_table(env, cref):
    """amsm_bases_fold: x cut into radix-R digits on the host, the joint ladder over the R^w multiples"""
    from accumulation_amd import CommitterKey
    from accumulation_amd.scalar_field import Fr
    c, ctx, ck, xy, inf, n = env
    fr = Fr(ctx.curve)
    n_half = n // 2
    plain = CommitterKey.load(ctx, xy, inf, 2)
    for x, nbits in ((o.rng_scalar(0xB400, 0) % (1 << 128), 128), (o.rng_scalar(0xB400, 1) % c.r, 255), (c.r - 1, 255)):
        a = ck.fold(n_half, fr.to_limbs(x), nbits)
        b = plain.fold(n_half, fr.to_limbs(x), nbits)
        ga, ia = a.read()
        gb, ib = b.read()
        assert np.array_equal(ga, gb) and np.array_equal(ia, ib), hex(x)
        i = n_half - 1
        P, Q = h.np_to_point(c, xy[i], bool(inf[i])), h.np_to_point(c, xy[n_half + i], bool(inf[n_half + i]))
        assert h.np_to_point(c, ga[i], bool(ia[i])) == o.add(c, P, o.mul(c, x % (1 << nbits), Q))
        a.free()
        b.free()
    plain.free()
