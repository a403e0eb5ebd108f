"""Group-law edge cases on points whose coordinates are EXTREME IN THE DEVICE'S INTERNAL MONTGOMERY RADIX
(tests/golden/adversarial_points.json, made by tests/golden/make_adversarial_points.py: internal x or y tiny, or just
below p).  The lazy arithmetic of csrc/ec.h (values bounded, not reduced; `K p - y` formed limb-wise) is exact only inside
documented bounds, and random points meet the edges of those bounds once in 2^17 .. 2^22 operations: here every such
point goes through doubling (duplicate bases), doubling of the negated point, cancellation, plain accumulation and the
key fold's ladder, for both key kinds, against the big-int oracle."""
import json
import os

import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h

pytestmark = pytest.mark.gpu
CURVES = [o.PALLAS, o.BLS12_381_G1]
FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "adversarial_points.json")))


def points_of(c):
    out = []
    for kind, pts in FIX["curves"][c.name].items():
        out += [(int(x, 16), int(y, 16)) for x, y in pts]
    return out


def neg(c, P):
    return (P[0], (-P[1]) % c.p)


def oracle_msm(c, pts, scalars):
    acc = None
    for P, s in zip(pts, scalars):
        acc = o.add(c, acc, o.mul(c, s % c.r, P))
    return acc


@pytest.fixture(scope="module")
def ctxs():
    from accumulation_amd import Context
    d = {c.name: Context(c.curve_id) for c in CURVES}
    yield d
    for v in d.values():
        v.close()


@pytest.mark.parametrize("flags", [1, 2], ids=["precomputed", "plain"])
@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_duplicates_negations_and_cancellations(ctxs, c, flags):
    from accumulation_amd import CommitterKey, VariableBaseMSM
    ctx = ctxs[c.name]
    pts = points_of(c)
    scal = [15, (1 << 16) - 1, (1 << 17) - 1, c.r - 1, c.r - 2, 3, (1 << 200) - 1, (1 << 128) + 1]
    # one key holding, for every point P: P, P, -P, P  (so equal-digit scalars double, negated-double and cancel)
    key_pts = []
    for P in pts:
        key_pts += [P, P, neg(c, P), P]
    xy, inf = h.points_to_np(c, key_pts)
    ck = CommitterKey.load(ctx, xy, None, flags)
    n = len(key_pts)
    for s in scal:
        for pattern in ((s, s, 0, 0), (s, 0, s, 0), (s, s, s, 0), (0, s, s, s), (s, s, s, s), (s, c.r - s, 0, 0), (s, 1, c.r - 1, s)):
            sc = list(pattern) * len(pts)
            out, oinf = VariableBaseMSM.multi_scalar_mul(ck, ctx.upload(h.scalars_to_np([v % c.r for v in sc])), mont=False)
            assert h.np_to_point(c, out, oinf) == oracle_msm(c, key_pts, sc), (c.name, flags, hex(s), pattern)
    # the same entries one point at a time (tiny MSMs: no other point shares the bucket)
    for k, P in enumerate(pts):
        sub = CommitterKey.load(ctx, xy[4 * k: 4 * k + 4], None, flags)
        for s in scal[:5]:
            sc = [s, s, s, s]
            out, oinf = VariableBaseMSM.multi_scalar_mul(sub, ctx.upload(h.scalars_to_np(sc)), mont=False)
            assert h.np_to_point(c, out, oinf) == oracle_msm(c, key_pts[4 * k: 4 * k + 4], sc), (c.name, flags, k, hex(s))
        sub.free()
    ck.free()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_random_and_top_of_field_scalars_over_the_adversarial_key(ctxs, c):
    from accumulation_amd import CommitterKey, VariableBaseMSM
    ctx = ctxs[c.name]
    pts = points_of(c)
    key_pts = pts + [neg(c, P) for P in pts] + pts[::-1]
    xy, _ = h.points_to_np(c, key_pts)
    n = len(key_pts)
    for flags in (1, 2):
        ck = CommitterKey.load(ctx, xy, None, flags)
        for seed in range(3):
            sc = [o.rng_scalar(0xAD0 + seed, i) % c.r for i in range(n)]
            if seed == 2:
                sc = [[c.r - 1, c.r - 2, 1 << 254, 1, 2, (1 << 17) - 1][i % 6] for i in range(n)]
            out, oinf = VariableBaseMSM.multi_scalar_mul(ck, ctx.upload(h.scalars_to_np(sc)), mont=False)
            assert h.np_to_point(c, out, oinf) == oracle_msm(c, key_pts, sc), (c.name, flags, seed)
        ck.free()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_key_fold_over_adversarial_pairs(ctxs, c):
    """l + x r for every adversarial point as l and as r (paired with itself, its negative and a generic point) and the x that
    steer the ladder into its special cases: r_order - 2 (ends in a negated doubling), r_order - 1 (l - r), 1, 2, 2^128 - 1"""
    from accumulation_amd import CommitterKey
    from accumulation_amd.scalar_field import Fr
    ctx = ctxs[c.name]
    fr = Fr(ctx.curve)
    pts = points_of(c)
    G = o.mul(c, 0xABCDEF, o.generator(c))
    left, right = [], []
    for P in pts:
        left += [P, P, neg(c, P), G, P]
        right += [P, neg(c, P), P, P, G]
    xy, _ = h.points_to_np(c, left + right)
    half = len(left)
    for x, nbits in ((c.r - 2, 255), (c.r - 1, 255), (1, 128), (2, 128), ((1 << 128) - 1, 128), (o.rng_scalar(0xF01D, 0) % c.r, 255)):
        ck = CommitterKey.load(ctx, xy, None, 2)
        f = ck.fold(half, fr.to_limbs(x), nbits)
        got, ginf = f.read()
        for i in range(half):
            want = o.add(c, left[i], o.mul(c, x, right[i]))
            assert h.np_to_point(c, got[i], bool(ginf[i])) == want, (c.name, hex(x), i)
        f.free()
        ck.free()
