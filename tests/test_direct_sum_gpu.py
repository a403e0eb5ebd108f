"""Round 4, the latency regime: precomputed keys of up to 2^15 generators also hold every multiple a 4-bit signed digit can ask
for (j 2^(4w) G_i, j = 1 .. 8, w = 0 .. 63) and an MSM over them is ONE launch that sums table points plus the quad fold
(msm_kernels.h `k_direct_sum`): no buckets, no sort, no dependence on the digit distribution.  Against the CPU restatement
oracle/ark_msm.c and the big-int oracle, bit for bit: every size class, ranges, the digit recoding's corner values (s + 0x0888..8
without a carry chain), Montgomery-form scalars, constant vectors, batches, host slices, keys with points at infinity, scalars
that do not fit.  Replaces ark-ec `VariableBaseMSM::multi_scalar_mul` (ext) at the small shapes of the reference's own tests and
of BASELINE config 0 (`trivial_pc_as`, examples/scaling-as.rs:62-63; src/trivial_pc_as/mod.rs:196 commits)."""
import os

import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h

pytestmark = pytest.mark.gpu
CURVES = [o.PALLAS, o.BLS12_381_G1]
PRECOMP = 1


@pytest.fixture(scope="module", params=CURVES, ids=lambda c: c.name)
def env(request):
    from accumulation_amd import CommitterKey, Context
    c = request.param
    ctx = Context(c.curve_id)
    n = (1 << 14) - 3
    ck = CommitterKey.generate(ctx, 0x5EED6001, n, PRECOMP)
    xy, inf = ck.read()
    yield c, ctx, ck, xy, n
    ck.free()
    ctx.close()


def _msm(c, ctx, ck, xy, sc, cref, off=0, mont=False, direct=1):
    from accumulation_amd import VariableBaseMSM
    before = ctx.pipeline_stats()["direct_sum"]
    got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc, base_off=off, mont=mont)
    assert ctx.pipeline_stats()["direct_sum"] - before == direct
    n = min(len(sc), len(xy) - off)
    raw = cref.fr_from_mont(c.curve_id, sc[:n]) if mont else sc[:n]
    ref, rinf = cref.msm(c.curve_id, xy[off:off + n], raw, threads=8)
    assert bool(inf) == bool(rinf) and np.array_equal(got, ref)
    return got, inf


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 255, 256, 1000, 1024, 1025, 4096, 4097, 8189, 16381])
def test_sizes_and_ranges(env, cref, n):
    """every lane-count class (windows per lane 4 .. 16, one to 256 workgroups), at the start, the middle and the end of the key"""
    c, ctx, ck, xy, N = env
    for off in (0, (N - n) // 2, N - n):
        _msm(c, ctx, ck, xy, cref.rng_scalars(0xC000 + n + off, n), cref, off=off)


def test_longer_vector_is_cut_at_the_key(env, cref):
    c, ctx, ck, xy, N = env
    _msm(c, ctx, ck, xy, cref.rng_scalars(0xC100, 600), cref, off=N - 500)


def test_digit_corner_values(env, cref):
    """s' = s + 0x0888...8 recodes every 4-bit window to [-8, 7] at once: scalars made of the nibbles that sit on the edges (7, 8,
    9, 0, f), carries running the whole length, the top window's unsigned nibble up to 8, r - 1, 0, 1 -- every scalar on its own
    generator AND all of them together"""
    c, ctx, ck, xy, N = env
    pat = lambda nib: int(f"{nib:x}" * 64, 16)  # noqa: E731
    vals = [0, 1, c.r - 1, c.r - 2, (c.r - 1) // 2, pat(7) % c.r, pat(8) % c.r, pat(9) % c.r, pat(0xF) % c.r, pat(1), (1 << 252) - 1,
            (1 << 252), (1 << 253) + (1 << 252) - 1, int("78" * 32, 16) % c.r, int("87" * 32, 16) % c.r, int("0f" * 32, 16),
            int("f0" * 32, 16) % c.r, 8, 9, 7, 15, 16, 0x88, 0x80, (1 << 128) - 1, 1 << 128, int("7" + "8" * 62, 16)]
    vals += [(c.r - 1) - pat(8) % (1 << 200), (1 << 254) % c.r]
    sc = h.scalars_to_np([v % c.r for v in vals])
    _msm(c, ctx, ck, xy, sc, cref, off=17)
    from accumulation_amd import VariableBaseMSM
    for j, v in enumerate(vals):  # one pair at a time against the big-int oracle
        got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc[j:j + 1], base_off=100 + j)
        ref = o.mul(c, v % c.r, h.np_to_point(c, xy[100 + j], False))
        assert h.np_to_point(c, got, bool(inf)) == ref, hex(v)


def test_montgomery_form_scalars(env, cref):
    c, ctx, ck, xy, N = env
    sc = cref.rng_scalars(0xC200, 5000)
    _msm(c, ctx, ck, xy, cref.fr_to_mont(c.curve_id, sc), cref, mont=True)


def test_constant_and_two_valued_vectors_cost_nothing_special(env, cref):
    """no buckets, so no skew: the constant vectors of the reference's DummyCircuit take the same single launch"""
    c, ctx, ck, xy, N = env
    n = 8000
    const = np.tile(np.array(o.int_to_limbs(0x1234567890ABCDEF % c.r, 4), dtype=np.uint64), (n, 1))
    _msm(c, ctx, ck, xy, const, cref)
    two = const.copy()
    two[::3] = 0
    _msm(c, ctx, ck, xy, two, cref)
    _msm(c, ctx, ck, xy, np.zeros((n, 4), dtype=np.uint64), cref)
    ones = np.zeros((n, 4), dtype=np.uint64)
    ones[:, 0] = 1
    _msm(c, ctx, ck, xy, ones, cref)


def test_cancelling_and_doubling_pairs(cref):
    """equal generators with equal / opposite scalars: the mixed additions' doubling and cancellation branches, the tree's too"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.PALLAS
    ctx = Context(c.curve_id)
    try:
        base = CommitterKey.generate(ctx, 0x5EED6002, 8, PRECOMP)
        xy8, _ = base.read()
        xy = np.concatenate([xy8[:1]] * 64 + [xy8[1:2]] * 64)  # 64 copies of G_0, 64 of G_1
        ck = CommitterKey.load(ctx, xy, None, PRECOMP)
        for vals in ([5] * 128, [5] * 64 + [c.r - 5] * 64, [3, c.r - 3] * 64, [1] * 127 + [c.r - 127], [c.r - 1] * 128):
            sc = h.scalars_to_np(vals)
            got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc)
            ref, rinf = cref.msm(c.curve_id, xy, sc, threads=4)
            assert bool(inf) == bool(rinf) and (rinf or np.array_equal(got, ref)), vals[:3]
        assert ctx.pipeline_stats()["direct_sum"] == 5
        ck.free()
        base.free()
    finally:
        ctx.close()


def test_key_with_points_at_infinity(cref):
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.BLS12_381_G1
    ctx = Context(c.curve_id)
    try:
        n = 777
        gen = CommitterKey.generate(ctx, 0x5EED6003, n, PRECOMP)
        xy, _ = gen.read()
        inf = np.zeros(n, dtype=np.uint8)
        inf[[0, 5, 300, n - 1]] = 1
        ck = CommitterKey.load(ctx, xy, inf, PRECOMP)
        sc = cref.rng_scalars(0xC300, n)
        got, i0 = VariableBaseMSM.multi_scalar_mul(ck, sc)
        ref, rinf = cref.msm(c.curve_id, xy, sc, is_inf=inf, threads=4)
        assert bool(i0) == bool(rinf) and np.array_equal(got, ref)
        assert ctx.pipeline_stats()["direct_sum"] == 1
        ck.free()
        gen.free()
    finally:
        ctx.close()


def test_batches_and_host_slices(env, cref):
    from accumulation_amd import VariableBaseMSM
    c, ctx, ck, xy, N = env
    n = 3000
    raws = [cref.rng_scalars(0xC400 + j, n) for j in range(7)]
    vecs = [ctx.upload(s) for s in raws]
    before = ctx.pipeline_stats()["direct_sum"]
    pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, vecs, mont=False, base_off=11)
    hp, hi = VariableBaseMSM.multi_scalar_mul_batch_host(ck, raws, mont=False, base_off=11)
    assert ctx.pipeline_stats()["direct_sum"] - before == 14
    for j in range(7):
        ref, rinf = cref.msm(c.curve_id, xy[11:11 + n], raws[j], threads=8)
        assert bool(infs[j]) == bool(rinf) and np.array_equal(pts[j], ref), j
        assert bool(hi[j]) == bool(rinf) and np.array_equal(hp[j], ref), j


def test_scalars_that_do_not_fit_are_reported(env, cref):
    """the top window takes its nibble unsigned, 0 .. 8: everything below 9 * 2^252 - 0x0888...8 (the canonical scalars of both
    fields and then some) is summed as the integer it is, anything above is AMSM_E_SCALAR_RANGE like on the other pipelines"""
    from accumulation_amd import VariableBaseMSM, ffi
    c, ctx, ck, xy, N = env
    n = 500
    K = int("0" + "8" * 63, 16)
    limit = 9 * (1 << 252) - K  # the first value whose top nibble of s + K exceeds 8
    for bad in (limit, (1 << 256) - 1, (1 << 256) - 189, 1 << 255 | 1 << 254):
        sc = cref.rng_scalars(0xC500, n)
        sc[n // 3] = np.array(o.int_to_limbs(bad, 4), dtype=np.uint64)
        with pytest.raises(ffi.AmsmError) as e:
            VariableBaseMSM.multi_scalar_mul(ck, sc)
        assert e.value.status == ffi.AMSM_E_SCALAR_RANGE, hex(bad)
    # the largest value that fits: an integer multiple, not reduced mod r
    sc = np.zeros((1, 4), dtype=np.uint64)
    sc[0] = np.array(o.int_to_limbs(limit - 1, 4), dtype=np.uint64)
    got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc, base_off=3)
    assert h.np_to_point(c, got, bool(inf)) == o.mul(c, (limit - 1) % c.r, h.np_to_point(c, xy[3], False))
    _msm(c, ctx, ck, xy, cref.rng_scalars(0xC501, n), cref)  # the slot's flag words were left clear


@pytest.mark.parametrize("n,shift", [(2, 0), (4, 1), (8, 0), (64, 5), (96, 4), (256, 7), (512, 0), (512, 7), (1024, 8), (1024, 9), (4000, 3), (4096, 3),
                                     (4096, 0), (8192, 12), (16384 - 512, 8), (16384 - 512, 5)])
def test_grouped_msm_as_two_direct_sums(env, cref, n, shift):
    """amsm_msm_grouped_device (the IPA rounds: two sums by one bit of the scalar's index) over a small key: one launch whose
    two rows of workgroups each sum ONE class, a two-record fold -- whenever n is a multiple of 2 << shift (both classes then hold
    n / 2 indices in the regular pattern)"""
    from accumulation_amd import VariableBaseMSM
    c, ctx, ck, xy, N = env
    sc = cref.rng_scalars(0xC600 + n + shift, n)
    if n == 4096 and shift == 0:
        sc[5] = 0
        sc[6:200] = sc[7]  # runs of equal scalars on neighbouring generators
    before = ctx.pipeline_stats()["direct_sum"]
    pts, infs = VariableBaseMSM.multi_scalar_mul_grouped(ck, ctx.upload(sc), shift, mont=False)
    assert ctx.pipeline_stats()["direct_sum"] - before == (1 if n % (2 << shift) == 0 else 0)
    cls = (np.arange(n) >> shift) & 1
    for g in (0, 1):
        ref, rinf = cref.msm(c.curve_id, xy[:n][cls == g], sc[cls == g], threads=8)
        assert bool(infs[g]) == bool(rinf) and np.array_equal(pts[g], ref), g


def test_grouped_msm_of_other_lengths_keeps_the_windowed_pipelines(env, cref):
    from accumulation_amd import VariableBaseMSM
    c, ctx, ck, xy, N = env
    for n, shift in ((4001, 3), (300, 2), (1024, 10)):
        sc = cref.rng_scalars(0xC680 + n, n)
        before = ctx.pipeline_stats()["direct_sum"]
        pts, infs = VariableBaseMSM.multi_scalar_mul_grouped(ck, ctx.upload(sc), shift, mont=False)
        assert ctx.pipeline_stats()["direct_sum"] == before
        cls = (np.arange(n) >> shift) & 1
        for g in (0, 1):
            ref, rinf = cref.msm(c.curve_id, xy[:n][cls == g], sc[cls == g], threads=8)
            assert bool(infs[g]) == bool(rinf) and (rinf or np.array_equal(pts[g], ref)), (n, shift, g)


def test_switched_off_and_larger_keys(cref):
    """AMSM_DIRECT_SUM_MAX_LOG2=0: no table, the windowed pipelines, the same points; a key of 2^15 + 1 generators has none either;
    amsm_bases_memory counts the table"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.PALLAS
    n = 2000
    sc = cref.rng_scalars(0xC700, n)
    res = []
    for env_, expect in (({}, 1), ({"AMSM_DIRECT_SUM_MAX_LOG2": "0"}, 0), ({"AMSM_DIRECT_SUM_MAX_LOG2": "10"}, 0)):
        os.environ.update(env_)
        try:
            ctx = Context(c.curve_id)
        finally:
            for k in env_:
                del os.environ[k]
        ck = CommitterKey.generate(ctx, 0x5EED6004, n, PRECOMP)
        res.append(VariableBaseMSM.multi_scalar_mul(ck, sc))
        assert ctx.pipeline_stats()["direct_sum"] == expect
        mem = ck.memory()
        per_point = 64  # a Pallas affine point in device memory
        assert (mem["table"] >= 512 * n * per_point) == bool(expect), mem
        ck.free()
        ctx.close()
    for r in res[1:]:
        assert np.array_equal(r[0], res[0][0]) and r[1] == res[0][1]
    ctx = Context(c.curve_id)
    ck = CommitterKey.generate(ctx, 0x5EED6005, (1 << 15) + 1, PRECOMP)
    VariableBaseMSM.multi_scalar_mul(ck, cref.rng_scalars(0xC701, 100))
    assert ctx.pipeline_stats()["direct_sum"] == 0
    ck.free()
    ctx.close()


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_largest_default_key(cref, c):
    """2^15 generators (a 1 / 1.5 GiB table): the whole key, a ragged range, a grouped MSM"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    ctx = Context(c.curve_id)
    try:
        n = 1 << 15
        ck = CommitterKey.generate(ctx, 0x5EED6006, n, PRECOMP)
        xy, _ = ck.read()
        _msm(c, ctx, ck, xy, cref.rng_scalars(0xC800, n), cref)
        _msm(c, ctx, ck, xy, cref.rng_scalars(0xC801, 20001), cref, off=12000)
        sc = cref.rng_scalars(0xC802, n)
        pts, infs = VariableBaseMSM.multi_scalar_mul_grouped(ck, ctx.upload(sc), 14, mont=False)
        assert ctx.pipeline_stats()["direct_sum"] == 3
        cls = (np.arange(n) >> 14) & 1
        for g in (0, 1):
            ref, rinf = cref.msm(c.curve_id, xy[cls == g], sc[cls == g], threads=8)
            assert bool(infs[g]) == bool(rinf) and np.array_equal(pts[g], ref), g
        ck.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("k", [2, 5, 16, 17, 40])
def test_batches_of_ragged_msms_in_one_launch(env, cref, k):
    """amsm_msm_multi: up to 16 MSMs of different lengths and offsets per launch (blockIdx.y), the rest in further launches"""
    from accumulation_amd import VariableBaseMSM
    c, ctx, ck, xy, N = env
    rs = np.random.RandomState(k)
    jobs = []
    for j in range(k):
        off = int(rs.randint(0, N - 1))
        n = int(rs.randint(1, min(N - off, 3000) + 1)) if j % 5 else N - off
        jobs.append((off, cref.rng_scalars(0xC900 + 100 * k + j, n)))
    before = ctx.pipeline_stats()["direct_sum"]
    outs, infs = VariableBaseMSM.multi_scalar_mul_multi(ck, [(off, ctx.upload(v)) for off, v in jobs], mont=False)
    assert ctx.pipeline_stats()["direct_sum"] - before == k
    for j, (off, v) in enumerate(jobs):
        ref, rinf = cref.msm(c.curve_id, xy[off:off + len(v)], v, threads=8)
        assert bool(infs[j]) == bool(rinf) and np.array_equal(outs[j], ref), j


def test_batch_with_an_empty_vector_and_a_bad_scalar(env, cref):
    from accumulation_amd import VariableBaseMSM, ffi
    c, ctx, ck, xy, N = env
    vs = [cref.rng_scalars(0xCA00 + j, 700) for j in range(4)]
    jobs = [(0, ctx.upload(vs[0])), (N, ctx.upload(vs[1])), (5, ctx.upload(vs[2]))]  # the second starts at the key's end: identity
    outs, infs = VariableBaseMSM.multi_scalar_mul_multi(ck, jobs, mont=False)
    assert bool(infs[1])
    for j, off in ((0, 0), (2, 5)):
        ref, rinf = cref.msm(c.curve_id, xy[off:off + 700], vs[j], threads=4)
        assert np.array_equal(outs[j], ref) and not infs[j]
    bad = vs[3].copy()
    bad[123] = np.array(o.int_to_limbs((1 << 256) - 5, 4), dtype=np.uint64)
    with pytest.raises(ffi.AmsmError) as e:
        VariableBaseMSM.multi_scalar_mul_batch(ck, [ctx.upload(vs[0]), ctx.upload(bad), ctx.upload(vs[2])], mont=False)
    assert e.value.status == ffi.AMSM_E_SCALAR_RANGE
    pts, _ = VariableBaseMSM.multi_scalar_mul_batch(ck, [ctx.upload(vs[0]), ctx.upload(vs[2])], mont=False)  # flags left clear
    ref, _ = cref.msm(c.curve_id, xy[:700], vs[2], threads=4)
    assert np.array_equal(pts[1], ref)
