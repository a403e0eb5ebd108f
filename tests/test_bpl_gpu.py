"""The bucket-per-lane accumulation pipeline (round 3: keys of >= 2^20 generators are precomputed for 20-bit windows; MSMs of
(2^18, 2^20] pairs run k_prep_local_t + k_accum_bpl, longer ones as windows of 2^20) against the CPU restatement
oracle/ark_msm.c, bit for bit -- uniform scalars, sizes at both edges of the range, base offsets into a longer key, batches,
BLS12-381 -- and its FALLBACK: scalars whose digits concentrate in a few buckets (the constant vectors of SURVEY.md F8, vectors
with many equal small values, a skew confined to one window) must be detected by the prep and re-run through the chunked
pipeline over the key's 17-bit-window twin with the same result.  amsm_ctx_pipeline_stats says which path ran.
Replaces ark-ec `VariableBaseMSM::multi_scalar_mul` (ext; call sites src/hp_as/mod.rs:196-214,377)."""
import os

import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h

pytestmark = pytest.mark.gpu
C = o.PALLAS
N = 1 << 20


@pytest.fixture(scope="module", params=["probe", "no_probe"])
def env(cref, request):
    """probe: a 1024-sample look at every candidate vector's digits sends skewed vectors straight to the chunked pipeline;
    no_probe (AMSM_BPL_PROBE=0): every candidate is attempted and the prep's overflow flag triggers the re-run -- the safety
    net under the probe, exercised on its own here"""
    from accumulation_amd import CommitterKey, Context
    os.environ["AMSM_BPL_PROBE"] = "1" if request.param == "probe" else "0"
    # constant vectors are this file's skewed inputs: with the two-valued form on they never reach the pipelines under test
    # (tests/test_two_valued_gpu.py covers that form)
    os.environ["AMSM_TWO_VALUED"] = "0"
    try:
        ctx = Context(C.curve_id)
    finally:
        del os.environ["AMSM_BPL_PROBE"]
        del os.environ["AMSM_TWO_VALUED"]
    ctx.probe = request.param == "probe"
    ck = CommitterKey.generate(ctx, 0x5EED1001, N)
    xy, _ = ck.read()
    yield ctx, ck, xy
    ck.free()
    ctx.close()


def check(ctx, ck, xy, sc, cref, off=0, expect_fallback=None, threads=17, probe_sees_it=True):
    """expect_fallback False: the MSM must stay on the bucket-per-lane pipeline; True: it must end up chunked -- re-run after
    the prep's overflow flag, or (probe on, and the skew is one 1024 samples show) sent there without an attempt"""
    from accumulation_amd import VariableBaseMSM
    before = ctx.pipeline_stats()
    got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc, base_off=off)
    after = ctx.pipeline_stats()
    n = min(len(sc), len(xy) - off)
    ref, rinf = cref.msm(C.curve_id, xy[off:off + n], sc[:n], threads=threads)
    assert bool(inf) == bool(rinf) and np.array_equal(got, ref)
    took = after["bucket_per_lane"] - before["bucket_per_lane"]
    fell = after["fallbacks"] - before["fallbacks"]
    if expect_fallback is False:
        assert took >= 1 and fell == 0, (took, fell)
    elif expect_fallback is True:
        if getattr(ctx, "probe", True) and probe_sees_it:
            assert took == 0 and fell == 0, (took, fell)
        else:
            assert took >= 1 and fell >= 1, (took, fell)
    return took, fell


def test_key_is_precomputed_for_20_bit_windows(env):
    ctx, ck, _ = env
    assert ck.precomputed and ck.window_bits == 20


def test_uniform_scalars_take_the_bucket_per_lane_pipeline(env, cref):
    ctx, ck, xy = env
    check(ctx, ck, xy, cref.rng_scalars(0x5EED0001, N), cref, expect_fallback=False)


@pytest.mark.parametrize("n", [(1 << 19) + 1, (1 << 20) - 12345, 1000003])
def test_sizes_inside_the_range(env, cref, n):
    ctx, ck, xy = env
    check(ctx, ck, xy, cref.rng_scalars(77 + n, n), cref, expect_fallback=False)


@pytest.mark.parametrize("n", [1 << 19, (1 << 18) + 1])
def test_ranges_down_to_a_quarter_stay_on_the_pipeline(env, cref, n):
    """round 4 (window widths that add up to 256 bits: no spread top window to overfill a sparser bucket table): any range that
    fills the 2^19 buckets at least a quarter as well as the whole key does"""
    ctx, ck, xy = env
    check(ctx, ck, xy, cref.rng_scalars(5 + n, n), cref, off=(N - n) // 3, expect_fallback=False)


@pytest.mark.parametrize("n", [1 << 18, 4097, 1])
def test_shorter_ranges_run_chunked_over_the_twin_key(env, cref, n):
    """the same key serves every length: below a quarter of the key the 17-bit-window twin and the chunked pipeline take over"""
    ctx, ck, xy = env
    took, fell = check(ctx, ck, xy, cref.rng_scalars(5 + n, n), cref, off=(N - n) // 3)
    assert took == 0 and fell == 0


def test_constant_vector_falls_back_and_is_exact(env, cref):
    """vec![x; len] (examples/scaling-as.rs, the zk provers' hiding vectors): ONE bucket per window holds 2^20 entries"""
    ctx, ck, xy = env
    x = o.rng_scalar(12, 0)
    check(ctx, ck, xy, np.tile(h.scalars_to_np([x]), (N, 1)), cref, expect_fallback=True)
    check(ctx, ck, xy, np.tile(h.scalars_to_np([1]), (N, 1)), cref, expect_fallback=True)
    check(ctx, ck, xy, np.tile(h.scalars_to_np([C.r - 1]), (N, 1)), cref, expect_fallback=True)


def test_all_zero_vector(env, cref):
    ctx, ck, xy = env
    check(ctx, ck, xy, np.zeros((N, 4), dtype=np.uint64), cref)


def test_partly_skewed_vectors(env, cref):
    """30 % of the scalars equal (one bucket of window 0 ... 12 holds 300 k entries): fallback; scalars below 2^19 (only
    window 0 is populated, two entries per bucket): no fallback needed, still exact; scalars below 2^20 (the upper half
    recodes to a negative digit and a carry of ONE into window 1: half a million entries in one bucket): fallback; few
    distinct values: fallback"""
    ctx, ck, xy = env
    rng = np.random.default_rng(7)
    sc = cref.rng_scalars(31, N)
    idx = rng.random(N) < 0.3
    sc[idx] = h.scalars_to_np([o.rng_scalar(32, 0)])[0]
    check(ctx, ck, xy, sc, cref, expect_fallback=True)
    small = np.zeros((N, 4), dtype=np.uint64)
    small[:, 0] = rng.integers(0, 1 << 19, N, dtype=np.uint64)
    check(ctx, ck, xy, small, cref, expect_fallback=False)
    small[:, 0] = rng.integers(0, 1 << 20, N, dtype=np.uint64)
    check(ctx, ck, xy, small, cref, expect_fallback=True)
    few = cref.rng_scalars(33, 16)[rng.integers(0, 16, N)]
    check(ctx, ck, xy, few, cref, expect_fallback=True)


def test_skew_in_one_window_only(env, cref):
    """uniform scalars whose bits 40..59 (window 2) are forced to one value: one bucket with 2^20 entries among uniform ones"""
    ctx, ck, xy = env
    sc = cref.rng_scalars(41, N)
    sc[:, 0] = (sc[:, 0] & ~np.uint64(((1 << 20) - 1) << 40)) | np.uint64(0x5A5A5 << 40)
    check(ctx, ck, xy, sc, cref, expect_fallback=True)


def test_moderately_uneven_buckets_stay_on_the_fast_path(env, cref):
    """a tenth of the scalars share their low 20 bits: one bucket of 10^5 entries is too much (fallback); a hundredth of a
    percent (105 extra entries in one bucket) is absorbed by the padding"""
    ctx, ck, xy = env
    rng = np.random.default_rng(9)
    sc = cref.rng_scalars(51, N)
    idx = rng.random(N) < 1e-4
    sc[idx, 0] = (sc[idx, 0] & ~np.uint64((1 << 20) - 1)) | np.uint64(0x12345)
    check(ctx, ck, xy, sc, cref, expect_fallback=False)


def test_batch_mixes_uniform_and_skewed_vectors(env, cref):
    """five MSMs in flight on three slots, the 2nd and 4th constant: results come back in order, each equal to the CPU's"""
    from accumulation_amd import VariableBaseMSM
    ctx, ck, xy = env
    vecs = [cref.rng_scalars(60 + j, N) for j in range(5)]
    vecs[1] = np.tile(h.scalars_to_np([o.rng_scalar(61, 0)]), (N, 1))
    vecs[3] = np.tile(h.scalars_to_np([3]), (N, 1))
    dv = [ctx.upload(v) for v in vecs]
    before = ctx.pipeline_stats()
    pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, dv, mont=False)
    after = ctx.pipeline_stats()
    took, fell = after["bucket_per_lane"] - before["bucket_per_lane"], after["fallbacks"] - before["fallbacks"]
    assert (took, fell) == ((3, 0) if ctx.probe else (5, 2))
    for j, v in enumerate(vecs):
        ref, rinf = cref.msm(C.curve_id, xy, v, threads=17)
        assert bool(infs[j]) == bool(rinf) and np.array_equal(pts[j], ref), j


def test_windows_of_a_longer_key(cref):
    """a 2^21-generator key: an MSM over all of it runs as two bucket-per-lane windows, one over [off, off + 2^20) as one"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    ctx = Context(C.curve_id)
    try:
        n = 1 << 21
        ck = CommitterKey.generate(ctx, 0x5EED1001, n)
        assert ck.window_bits == 20
        xy, _ = ck.read()
        sc = cref.rng_scalars(71, n)
        before = ctx.pipeline_stats()
        got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc)
        assert ctx.pipeline_stats()["bucket_per_lane"] - before["bucket_per_lane"] == 2
        ref, rinf = cref.msm(C.curve_id, xy, sc, threads=17)
        assert bool(inf) == bool(rinf) and np.array_equal(got, ref)
        off = 777777
        got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc[:N], base_off=off)
        ref, rinf = cref.msm(C.curve_id, xy[off:off + N], sc[:N], threads=17)
        assert bool(inf) == bool(rinf) and np.array_equal(got, ref)
        ck.free()
    finally:
        ctx.close()


def test_same_results_with_the_pipeline_disabled(cref):
    """AMSM_BPL=0: round 2's 17-bit windows + chunked pipeline for every size"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    sc = cref.rng_scalars(81, N)
    res = []
    for bpl in ("1", "0"):
        os.environ["AMSM_BPL"] = bpl
        try:
            ctx = Context(C.curve_id)
        finally:
            del os.environ["AMSM_BPL"]
        ck = CommitterKey.generate(ctx, 0x5EED1001, N)
        assert ck.window_bits == (20 if bpl == "1" else 17)
        res.append(VariableBaseMSM.multi_scalar_mul(ck, sc))
        assert (ctx.pipeline_stats()["bucket_per_lane"] > 0) == (bpl == "1")
        ck.free()
        ctx.close()
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1]


def test_bls12_381_at_2p20(cref):
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.BLS12_381_G1
    ctx = Context(c.curve_id)
    try:
        ck = CommitterKey.generate(ctx, 0x5EED1002, N)
        assert ck.window_bits == 20
        xy, _ = ck.read()
        sc = cref.rng_frs(c.curve_id, 91, N)  # uniform in [0, r): 45 % of these exceed 2^254 -- the recoding's carry lands in the 13th window
        assert 0.40 < float((sc[:, 3] >> np.uint64(62)).astype(bool).mean()) < 0.50
        got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc)
        st = ctx.pipeline_stats()
        # (a lone HOST slice of 2^20 pairs: two ranges over one bucket set since round 6 -- the second half uploads beside the first's MSM)
        assert st["bucket_per_lane"] == 2 and st["shared_bucket_sets"] == 1 and st["fallbacks"] == 0
        ref, rinf = cref.msm(c.curve_id, xy, sc, threads=17)
        assert bool(inf) == bool(rinf) and np.array_equal(got, ref)
        top = np.tile(np.array(o.int_to_limbs(c.r - 2, 4), dtype=np.uint64), (N, 1))
        got, inf = VariableBaseMSM.multi_scalar_mul(ck, top)
        st = ctx.pipeline_stats()
        assert st["bucket_per_lane"] == 2 and st["fallbacks"] == 0  # (still 2: the probe sent the constant vector to the chunked pipeline)
        ref, rinf = cref.msm(c.curve_id, xy, top, threads=17)
        assert bool(inf) == bool(rinf) and np.array_equal(got, ref)
        ck.free()
    finally:
        ctx.close()


def test_non_canonical_scalar_is_reported_not_scattered(env, cref):
    """a scalar >= r whose top digit, spread by MsmGeom::top_shift, would leave the bucket range: AMSM_E_SCALAR_RANGE (the
    reference hands canonical `into_repr()` values only), and the context keeps working"""
    from accumulation_amd import VariableBaseMSM, ffi
    ctx, ck, xy = env
    sc = cref.rng_scalars(95, N)
    sc[N // 2] = np.array(o.int_to_limbs((1 << 255) - 19, 4), dtype=np.uint64)
    with pytest.raises(ffi.AmsmError) as e:
        VariableBaseMSM.multi_scalar_mul(ck, sc)
    assert e.value.status == ffi.AMSM_E_SCALAR_RANGE
    sc[N // 2] = np.array(o.int_to_limbs(C.r - 1, 4), dtype=np.uint64)  # the largest canonical scalar is fine
    check(ctx, ck, xy, sc, cref, expect_fallback=False)
