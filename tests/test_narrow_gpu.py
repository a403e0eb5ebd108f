"""Round 4: window widths that add up to exactly 256 bits (MsmGeom::n_narrow: the top W c - 256 windows are c - 1 bits wide,
their digits doubled, their table levels / window sums one doubling short) and what they let onto the bucket-per-lane
pipeline -- PLAIN keys (one bucket set per window, the top window split over two sets, the windows' sums combined on the
host), grouped MSMs and partial ranges over a 20-bit precomputed key -- against the CPU restatement oracle/ark_msm.c and the
big-int oracle, bit for bit.  Replaces ark-ec `VariableBaseMSM::multi_scalar_mul` (ext; the bases are passed per call there,
which is what a plain key is: src/ipa_pc_as/mod.rs:454 folds the key every round, src/hp_as/mod.rs:377 commits over a slice)."""
import os

import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h

pytestmark = pytest.mark.gpu
CURVES = [o.PALLAS, o.BLS12_381_G1]
PLAIN, PRECOMP = 2, 1  # AMSM_BASES_NO_PRECOMPUTE / AMSM_BASES_PRECOMPUTE


def _stats_delta(ctx, before):
    after = ctx.pipeline_stats()
    return after["bucket_per_lane"] - before["bucket_per_lane"], after["fallbacks"] - before["fallbacks"]


@pytest.fixture(scope="module", params=CURVES, ids=lambda c: c.name)
def plain_env(request, cref):
    """a plain (not precomputed) key of 2^20 + 2^19 generators per curve, two-valued shortcut off (constant vectors are
    this file's skewed inputs)"""
    from accumulation_amd import CommitterKey, Context
    c = request.param
    os.environ["AMSM_TWO_VALUED"] = "0"
    try:
        ctx = Context(c.curve_id)
    finally:
        del os.environ["AMSM_TWO_VALUED"]
    n = (1 << 20) + (1 << 19) + 5
    ck = CommitterKey.generate(ctx, 0x5EED4001, n, PLAIN)
    assert not ck.precomputed
    xy, _ = ck.read()
    yield c, ctx, ck, xy
    ck.free()
    ctx.close()


def _check(c, ctx, ck, xy, sc, cref, off=0, expect=None, mont=False):
    from accumulation_amd import VariableBaseMSM
    before = ctx.pipeline_stats()
    got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc, base_off=off, mont=mont)
    took, fell = _stats_delta(ctx, before)
    n = min(len(sc), len(xy) - off)
    ref, rinf = cref.msm(c.curve_id, xy[off:off + n], sc[:n], threads=8)
    assert bool(inf) == bool(rinf) and np.array_equal(got, ref)
    if expect is not None:
        assert (took, fell) == expect, (took, fell)
    return took, fell


@pytest.mark.parametrize("n", [(1 << 17) + 1, 1 << 18, (1 << 18) + 1, 1 << 19, 1 << 20])
def test_plain_key_sizes(plain_env, cref, n):
    """15- and 16-bit windows by size (18 / 16 of them), the edges of both ranges: one bucket-per-lane MSM, no fallback"""
    c, ctx, ck, xy = plain_env
    if c is o.BLS12_381_G1 and n not in ((1 << 17) + 1, 1 << 18, 1 << 20):
        pytest.skip("BLS12-381: one size per window width")
    _check(c, ctx, ck, xy, cref.rng_scalars(0xA000 + n, n), cref, off=12345, expect=(1, 0))


def test_plain_key_at_and_below_2p17_keeps_the_chunked_pipeline(plain_env, cref):
    """8-bit windows on the chunked pipeline: faster there since their widths were re-swept"""
    c, ctx, ck, xy = plain_env
    for n in (1 << 17, (1 << 16) + 1, 1 << 16, 4099, 1):
        _check(c, ctx, ck, xy, cref.rng_scalars(0xA100 + n, n), cref, off=7, expect=(0, 0))


def test_plain_key_longer_than_2p20_runs_as_ranges(plain_env, cref):
    """2^20 + 2^19 + 5 pairs: a range of 2^20 and one of 2^19 + 5, summed on the host"""
    c, ctx, ck, xy = plain_env
    n = len(xy)
    _check(c, ctx, ck, xy, cref.rng_scalars(0xA200, n), cref, expect=(2, 0))


def test_plain_key_uniform_mod_r_scalars(plain_env, cref):
    """the synthetic stream is 254 bits wide; scalars uniform below r reach further into the top window (BLS12-381: r =
    0.45 x 2^256, its top window's partitions run 10 % fuller than the others')"""
    c, ctx, ck, xy = plain_env
    rng = np.random.default_rng(0xA300)
    n = 1 << 20
    raw = rng.integers(0, 1 << 63, (n, 5), dtype=np.uint64)
    vals = [(int(a) | (int(b) << 63) | (int(d) << 126) | (int(e) << 189) | (int(f) << 252)) % c.r for a, b, d, e, f in raw]
    _check(c, ctx, ck, xy, h.scalars_to_np(vals), cref, expect=(1, 0))


def test_plain_key_edge_scalars(plain_env, cref):
    """the largest canonical scalars, powers of two on both sides of every window boundary, zeros -- among uniform ones"""
    c, ctx, ck, xy = plain_env
    n = (1 << 18) + 77
    sc = cref.rng_scalars(0xA400, n)
    edge = [c.r - 1, c.r - 2, 1, 0, 2, (1 << 254) if c.r > (1 << 254) else (1 << 253), (1 << 240) - 1, 1 << 240, (1 << 240) + 1,
            (1 << 15) + 1, 1 << 15, (1 << 16) - 1, 1 << 16, (1 << 128) - 1, (c.r - 1) >> 1, ((c.r - 1) >> 1) + 1]
    for w in range(1, 19):  # window boundaries of the 15-bit walk: 4 regular windows, then 14 of 14 bits
        pos = 15 * w if w <= 4 else 60 + 14 * (w - 4)
        edge += [(1 << pos) - 1, 1 << pos, (1 << (pos - 1)), (1 << (pos - 1)) + 1]
    edge = [e % c.r for e in edge]
    idx = np.linspace(3, n - 3, len(edge)).astype(int)
    sc[idx] = h.scalars_to_np(edge)
    _check(c, ctx, ck, xy, sc, cref, off=999, expect=(1, 0))


def test_plain_key_montgomery_form_scalars(plain_env, cref):
    c, ctx, ck, xy = plain_env
    n = 1 << 19
    sc = cref.rng_scalars(0xA500, n)
    from accumulation_amd import VariableBaseMSM
    d = ctx.upload(cref.fr_to_mont(c.curve_id, sc))
    got, inf = VariableBaseMSM.multi_scalar_mul(ck, d, mont=True)
    ref, rinf = cref.msm(c.curve_id, xy[:n], sc, threads=8)
    assert bool(inf) == bool(rinf) and np.array_equal(got, ref)


def test_plain_key_skewed_vectors_fall_back_and_are_exact(plain_env, cref):
    """vec![x; len] and a vector with a third of its scalars equal: the prep's overflow flag re-runs the MSM chunked"""
    c, ctx, ck, xy = plain_env
    n = 1 << 19
    const = np.tile(h.scalars_to_np([o.rng_scalar(0xA600, 0) % c.r]), (n, 1))
    took, fell = _check(c, ctx, ck, xy, const, cref)
    assert took == 1 and fell == 1
    sc = cref.rng_scalars(0xA601, n)
    sc[np.random.default_rng(3).random(n) < 0.33] = h.scalars_to_np([o.rng_scalar(0xA602, 0) % c.r])[0]
    took, fell = _check(c, ctx, ck, xy, sc, cref)
    assert took == 1 and fell == 1
    _check(c, ctx, ck, xy, cref.rng_scalars(0xA603, n), cref, expect=(1, 0))  # and the context keeps working


def test_plain_key_non_canonical_scalar_is_reported(plain_env, cref):
    from accumulation_amd import VariableBaseMSM, ffi
    c, ctx, ck, xy = plain_env
    n = 1 << 19
    sc = cref.rng_scalars(0xA700, n)
    sc[n // 3] = np.array(o.int_to_limbs((1 << 256) - 189, 4), dtype=np.uint64)
    with pytest.raises(ffi.AmsmError) as e:
        VariableBaseMSM.multi_scalar_mul(ck, sc)
    assert e.value.status == ffi.AMSM_E_SCALAR_RANGE
    _check(c, ctx, ck, xy, cref.rng_scalars(0xA701, n), cref, expect=(1, 0))


def test_plain_key_batch(plain_env, cref):
    """five MSMs in flight on three slots, each with its own host combination of the windows' sums"""
    from accumulation_amd import VariableBaseMSM
    c, ctx, ck, xy = plain_env
    n = 1 << 19
    vecs = [cref.rng_scalars(0xA800 + j, n) for j in range(5)]
    dv = [ctx.upload(v) for v in vecs]
    before = ctx.pipeline_stats()
    pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, dv, mont=False, base_off=4242)
    assert _stats_delta(ctx, before) == (5, 0)
    for j, v in enumerate(vecs):
        ref, rinf = cref.msm(c.curve_id, xy[4242:4242 + n], v, threads=8)
        assert bool(infs[j]) == bool(rinf) and np.array_equal(pts[j], ref), j


@pytest.mark.parametrize("kind", [PLAIN, PRECOMP], ids=["plain_key", "precomputed_20_bit_key"])
@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_grouped_msm(cref, c, kind):
    """amsm_msm_grouped_device (the IPA rounds: two sums over index classes in one pass) -- two bucket sets per window (plain
    key) or two in all (precomputed key) on the bucket-per-lane pipeline"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    ctx = Context(c.curve_id)
    try:
        n = 1 << 20
        ck = CommitterKey.generate(ctx, 0x5EED4002, n, kind)
        if kind == PRECOMP:
            assert ck.window_bits == 20
        xy, _ = ck.read()
        sc = cref.rng_scalars(0xA900, n)
        d = ctx.upload(sc)
        for shift in (0, 7, 19):
            before = ctx.pipeline_stats()
            pts, infs = VariableBaseMSM.multi_scalar_mul_grouped(ck, d, shift, mont=False)
            assert _stats_delta(ctx, before) == (1, 0), shift
            cls = (np.arange(n) >> shift) & 1
            for g in (0, 1):
                ref, rinf = cref.msm(c.curve_id, xy[cls == g], sc[cls == g], threads=8)
                assert bool(infs[g]) == bool(rinf) and np.array_equal(pts[g], ref), (shift, g)
        ck.free()
    finally:
        ctx.close()


def test_plain_key_switch_gives_the_same_points(cref):
    """AMSM_BPL_PLAIN=0 (a documented switch, include/amsm.h): plain keys on the chunked pipeline -- the same points"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.PALLAS
    n = 1 << 20
    sc = cref.rng_scalars(0xAA00, n)
    res = []
    for env in ({}, {"AMSM_BPL_PLAIN": "0"}):
        os.environ.update(env)
        try:
            ctx = Context(c.curve_id)
        finally:
            for k in env:
                del os.environ[k]
        for kind in (PRECOMP, PLAIN):
            ck = CommitterKey.generate(ctx, 0x5EED4003, n, kind)
            res.append(VariableBaseMSM.multi_scalar_mul(ck, sc))
            ck.free()
        st = ctx.pipeline_stats()
        # (the precomputed key's lone HOST slice: two ranges over one bucket set since round 6; the plain key's: one range)
        assert st["bucket_per_lane"] == (2 if "AMSM_BPL_PLAIN" in env else 3) and st["fallbacks"] == 0
        ctx.close()
    for r in res[1:]:
        assert np.array_equal(r[0], res[0][0]) and r[1] == res[0][1]


def test_fold_of_a_narrow_key_through_its_window_multiples(cref):
    """amsm_bases_fold over a 20-bit key whose top four levels stand for 2^(e_w) with e_w off the 20-bit grid: the joint
    ladder cuts x at the e_w (a full-size x reaches them, a 128-bit challenge does not) -- equal to the plain ladder over
    the same generators, and to the big-int oracle at a few indices"""
    from accumulation_amd import CommitterKey, Context
    from accumulation_amd.scalar_field import Fr
    c = o.PALLAS
    ctx = Context(c.curve_id)
    try:
        fr = Fr(ctx.curve)
        n_half = 1 << 19
        ck = CommitterKey.generate(ctx, 0x5EED4004, 2 * n_half)
        assert ck.precomputed and ck.window_bits == 20
        xy, inf = ck.read()
        plain = CommitterKey.load(ctx, xy, inf, PLAIN)
        for x, nbits in ((o.rng_scalar(0xAB00, 0) % (1 << 128), 128), (o.rng_scalar(0xAB00, 1) % c.r, 255), (c.r - 1, 255),
                         ((1 << 199) + (1 << 180) - 1, 255)):
            a = ck.fold(n_half, fr.to_limbs(x), nbits)
            b = plain.fold(n_half, fr.to_limbs(x), nbits)
            ga, ia = a.read()
            gb, ib = b.read()
            assert np.array_equal(ga, gb) and np.array_equal(ia, ib), hex(x)
            for i in (0, 1, n_half - 1):
                P, Q = h.np_to_point(c, xy[i], bool(inf[i])), h.np_to_point(c, xy[n_half + i], bool(inf[n_half + i]))
                assert h.np_to_point(c, ga[i], bool(ia[i])) == o.add(c, P, o.mul(c, x % (1 << nbits), Q))
            a.free()
            b.free()
        plain.free()
        ck.free()
    finally:
        ctx.close()
