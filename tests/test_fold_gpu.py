"""Key / point folds out[i] = l[i] + x r[i] (the `key_l += key_r * xi` of the IPA opening, ark_poly_commit::ipa_pc ext under
src/ipa_pc_as/mod.rs:454) through every formulation the library picks between -- plain NAF ladder, the joint ladder over a
precomputed key's window multiples (k_points_fold_tab), and the GLV split of full-size scalars (host_glv.h) -- against the
big-int oracle on both curves, with edge-case scalars and degenerate points (infinity, equal and opposite halves)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h

pytestmark = pytest.mark.gpu

CURVES = [o.PALLAS, o.BLS12_381_G1]


@pytest.fixture(scope="module")
def ctxs():
    from accumulation_amd import Context
    out = {c.name: Context(c.curve_id) for c in CURVES}
    yield out
    for c in out.values():
        c.close()


def _points(c, n_half):
    g = o.generator(c)
    base = [o.mul(c, 3 + 7 * i, g) for i in range(2 * n_half)]
    # degenerate pairs: r = l, r = -l, l = infinity, r = infinity, both infinity
    base[n_half + 1] = base[1]
    base[n_half + 2] = o.neg(c, base[2])
    base[3] = None
    base[n_half + 4] = None
    base[5] = None
    base[n_half + 5] = None
    return base


def _scalars(c):
    return [(0, 255), (1, 255), (2, 255), ((1 << 17) - 1, 255), (1 << 17, 255), (c.r - 1, 255), (c.r - 2, 255),
            ((1 << 128) - 1, 128), (o.rng_scalar(0xF01D, 0) % (1 << 128), 128), (o.rng_scalar(0xF01D, 1) % c.r, 255),
            (o.rng_scalar(0xF01D, 2) % c.r, 255), (o.rng_scalar(0xF01D, 3) % c.r, 200)]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("flags", [1, 2], ids=["precomputed_key_table_ladder", "plain_key"])
def test_bases_fold_vs_oracle(ctxs, c, flags):
    from accumulation_amd import CommitterKey
    from accumulation_amd.scalar_field import Fr
    ctx = ctxs[c.name]
    fr = Fr(ctx.curve)
    n_half = 24
    pts = _points(c, n_half)
    xy, inf = h.points_to_np(c, pts)
    ck = CommitterKey.load(ctx, xy, inf, flags)
    assert ck.precomputed == (flags == 1)
    for x, nbits in _scalars(c):
        xe = x % (1 << nbits)
        f = ck.fold(n_half, fr.to_limbs(x), nbits)
        got, ginf = f.read()
        for i in range(n_half):
            assert h.np_to_point(c, got[i], bool(ginf[i])) == o.add(c, pts[i], o.mul(c, xe, pts[n_half + i])), (hex(x), nbits, i)
        f.free()
    ck.free()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_points_fold_full_size_scalars_vs_oracle(ctxs, c):
    """amsm_points_fold on caller-visible point vectors: full-size scalars take the GLV split, short ones the plain ladder."""
    from accumulation_amd import PointVector, ffi
    from accumulation_amd.engine import _ptr
    from accumulation_amd.scalar_field import Fr
    ctx = ctxs[c.name]
    fr = Fr(ctx.curve)
    n_half = 24
    pts = _points(c, n_half)
    xy, _ = h.points_to_np(c, pts)
    pv = PointVector(ctx, 2 * n_half)
    ffi.check(ctx._lib.amsm_dev_upload(ctx._h, pv.ptr, _ptr(xy), xy.nbytes), "upload")
    out = PointVector(ctx, n_half)
    for x, nbits in _scalars(c):
        xe = x % (1 << nbits)
        ffi.check(ctx._lib.amsm_points_fold(ctx._h, pv.view(0, n_half).ptr, pv.view(n_half, n_half).ptr, n_half,
                                            _ptr(fr.to_limbs(x)), nbits, out.ptr), "fold")
        got = out.download()
        for i in range(n_half):
            want = o.add(c, pts[i], o.mul(c, xe, pts[n_half + i]))
            assert h.np_to_point(c, got[i], 0) == want or (want is None and not got[i].any()), (hex(x), nbits, i)


def test_large_fold_table_ladder_equals_plain_ladder(ctxs):
    """2^17 + 5 outputs (the batched-inversion path of both kernels) over a generated Pallas key: the fold through the
    window multiples equals the fold of the same generators loaded as a plain key, bit for bit."""
    from accumulation_amd import CommitterKey
    from accumulation_amd.scalar_field import Fr
    c = o.PALLAS
    ctx = ctxs[c.name]
    fr = Fr(ctx.curve)
    n_half = (1 << 17) + 5
    ck = CommitterKey.generate(ctx, 0x5EEDF01D, 2 * n_half)
    assert ck.precomputed
    xy, inf = ck.read()
    plain = CommitterKey.load(ctx, xy, inf, 2)
    for x, nbits in (((o.rng_scalar(0xF01E, 0) % (1 << 128)), 128), (o.rng_scalar(0xF01E, 1) % c.r, 255)):
        a = ck.fold(n_half, fr.to_limbs(x), nbits)
        b = plain.fold(n_half, fr.to_limbs(x), nbits)
        ga, ia = a.read()
        gb, ib = b.read()
        assert np.array_equal(ga, gb) and np.array_equal(ia, ib)
        for i in (0, 1, n_half - 1):
            P, Q = h.np_to_point(c, xy[i], bool(inf[i])), h.np_to_point(c, xy[n_half + i], bool(inf[n_half + i]))
            assert h.np_to_point(c, ga[i], bool(ia[i])) == o.add(c, P, o.mul(c, x % (1 << nbits), Q))
        a.free()
        b.free()
    plain.free()
    ck.free()
