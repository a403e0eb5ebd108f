"""Vectors with a SHARE of unit scalars -- the boolean wires of an R1CS witness among uniform values -- are not two-valued, but all
their ones would land in bucket 1 of the lowest window.  Round 5: the generators under unit scalars are summed apart (n_ones
additions, what ark-ec's multi_scalar_mul does with them) and the windowed pipelines skip those scalars.  Same canonical results
as the C oracle (oracle/ark_msm.c) in both scalar forms, over precomputed and plain keys, at the sizes of every pipeline
(bucket-split, chunked, bucket-per-lane, ranges over one bucket set), next to vectors that must NOT take that form; views that share
memory with a vector that keeps its ones give the form up."""
import numpy as np
import pytest

from accumulation_amd import ffi
from oracle import cref

pytestmark = pytest.mark.gpu


def _witness(n, frac, seed, ones_only=False, clustered=False):
    """uniform scalars with a fraction replaced by 0 / 1"""
    s = cref.rng_scalars(seed, n).copy()
    rng = np.random.default_rng(seed)
    pick = np.zeros(n, dtype=bool)
    if clustered:
        m = int(n * frac)
        pick[n // 3:n // 3 + m] = True  # (one block of boolean wires)
    else:
        pick = rng.random(n) < frac
    vals = np.zeros((n, 4), dtype=np.uint64)
    vals[:, 0] = 1 if ones_only else rng.integers(0, 2, n)
    s[pick] = vals[pick]
    return s


_ORACLE_CACHE = {}


@pytest.mark.parametrize("curve", [ffi.AMSM_PALLAS, ffi.AMSM_BLS12_381_G1], ids=["pallas", "bls12_381_g1"])
@pytest.mark.parametrize("mont", [False, True], ids=["canonical", "montgomery"])
@pytest.mark.parametrize("log_n", [16, 18])
def test_witness_like_vectors_vs_c_oracle(curve, mont, log_n):
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    ctx = Context(curve)
    try:
        n = (1 << log_n) + 11
        pre = CommitterKey.generate(ctx, 0x0E5 + log_n, n, ffi.AMSM_BASES_PRECOMPUTE | ffi.AMSM_BASES_NO_DIRECT_TABLE)
        xy, inf = pre.read()
        plain = CommitterKey.load(ctx, xy, inf, ffi.AMSM_BASES_NO_PRECOMPUTE)
        cases = {"10pct_booleans": (_witness(n, 0.1, 1), True), "50pct_booleans": (_witness(n, 0.5, 2), True),
                 "90pct_booleans": (_witness(n, 0.9, 3), True), "30pct_ones_only": (_witness(n, 0.3, 4, ones_only=True), True),
                 "2pct_booleans": (_witness(n, 0.02, 5), True), "a_block_of_booleans": (_witness(n, 0.25, 6, clustered=True), True),
                 "uniform": (cref.rng_scalars(77, n), False), "three_ones_in_all": (_witness(n, 3.5 / n, 8, ones_only=True), False)}
        names = list(cases)
        # (the canonical and the Montgomery run of a (curve, size) share inputs: one set of CPU results serves both)
        key_ = (curve, log_n)
        if key_ not in _ORACLE_CACHE:
            _ORACLE_CACHE[key_] = {k: cref.msm(curve, xy, cases[k][0]) for k in names}
        want = _ORACLE_CACHE[key_]
        for key in (pre, plain):
            up = [ctx.upload(cref.fr_to_mont(curve, cases[k][0]) if mont else cases[k][0]) for k in names]
            before = ctx.pipeline_stats()
            out, oinf = VariableBaseMSM.multi_scalar_mul_batch(key, up, mont=mont)
            after = ctx.pipeline_stats()
            assert after["unit_scalar_sums"] - before["unit_scalar_sums"] == sum(1 for k in names if cases[k][1]), (names, before, after)
            for j, k in enumerate(names):
                ref, ref_inf = want[k]
                assert bool(oinf[j]) == bool(ref_inf) and np.array_equal(out[j], ref), (k, mont, key.precomputed, log_n)
            for u in up:
                u.free()
            # HOST slices (what a Rust adapter passes): a 1024-sample look sends a call with such a slice to the device path
            if not mont:
                before = ctx.pipeline_stats()["unit_scalar_sums"]
                out, oinf = VariableBaseMSM.multi_scalar_mul_batch_host(key, [cases["10pct_booleans"][0], cases["uniform"][0]], mont=False)
                assert ctx.pipeline_stats()["unit_scalar_sums"] - before == 1
                for j, k in enumerate(("10pct_booleans", "uniform")):
                    assert np.array_equal(out[j], want[k][0]) and bool(oinf[j]) == bool(want[k][1]), (k, "host slices")
        pre.free()
        plain.free()
    finally:
        ctx.close()


def test_plain_key_bucket_per_lane_and_a_20_bit_table():
    """2^20 pairs: over a PLAIN key (16 bucket sets, bucket-per-lane: the unit scalars used to overflow the prep and send the MSM
    back to the chunked pipeline, 2.3x slower) the form is taken and nothing falls back; over a 20-bit table it is NOT taken (such a
    vector runs chunked over the key's 17-bit twin, within 10 % of a uniform one's time) -- both as the oracle's, also for a window of the key"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    curve = ffi.AMSM_PALLAS
    ctx = Context(curve)
    try:
        n = 1 << 20
        key = CommitterKey.generate(ctx, 0x0E57, n + 12345, ffi.AMSM_BASES_PRECOMPUTE)
        xy, inf = key.read()
        plain = CommitterKey.load(ctx, xy, inf, ffi.AMSM_BASES_NO_PRECOMPUTE)
        w = _witness(n, 0.3, 21)
        a = ctx.upload(w)
        ref0, _ = cref.msm(curve, xy[:n], w)
        ref1, _ = cref.msm(curve, xy[12345:12345 + n], w)
        for k_, taken in ((plain, 2), (key, 0)):
            before = ctx.pipeline_stats()
            out, oinf = VariableBaseMSM.multi_scalar_mul_batch(k_, [a, a], mont=False)
            after = ctx.pipeline_stats()
            assert after["unit_scalar_sums"] - before["unit_scalar_sums"] == taken and after["fallbacks"] == before["fallbacks"]
            if taken:  # (over the 20-bit table the skew probe still sees the ones: such a vector runs chunked over the 17-bit twin)
                assert after["bucket_per_lane"] > before["bucket_per_lane"]
            assert np.array_equal(out[0], ref0) and np.array_equal(out[1], ref0) and not oinf.any()
            one, one_inf = VariableBaseMSM.multi_scalar_mul(k_, a, base_off=12345)  # (a lone call is probed from 2^17 pairs up)
            assert np.array_equal(one, ref1) and not one_inf
            assert ctx.pipeline_stats()["unit_scalar_sums"] - after["unit_scalar_sums"] == taken // 2
    finally:
        ctx.close()


def test_views_that_share_memory_with_a_vector_that_keeps_its_ones():
    """one batch: a boolean-heavy vector, a view that starts 100 scalars before its end and runs on into uniform scalars (40 ones
    in 2^17: it keeps them in the pipelines) and an unrelated witness -- the first gives the form up (the addresses of its scalars
    are how the pipelines would know), the unrelated one keeps it; every result as the oracle's"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    curve = ffi.AMSM_PALLAS
    ctx = Context(curve)
    try:
        n = 1 << 17
        key = CommitterKey.generate(ctx, 0x0E58, n, ffi.AMSM_BASES_PRECOMPUTE | ffi.AMSM_BASES_NO_DIRECT_TABLE)
        xy, _ = key.read()
        buf_h = np.concatenate([_witness(n, 0.4, 31), cref.rng_scalars(33, n)])
        w2 = _witness(n, 0.4, 32)
        buf, c = ctx.upload(buf_h), ctx.upload(w2)
        a, b = buf.view(0, n), buf.view(n - 100, n)
        before = ctx.pipeline_stats()["unit_scalar_sums"]
        out, oinf = VariableBaseMSM.multi_scalar_mul_batch(key, [a, b, c], mont=False)
        assert ctx.pipeline_stats()["unit_scalar_sums"] - before == 1
        for j, vec in enumerate((buf_h[:n], buf_h[n - 100:2 * n - 100], w2)):
            ref, ref_inf = cref.msm(curve, xy, vec)
            assert bool(oinf[j]) == bool(ref_inf) and np.array_equal(out[j], ref), j
        # ... and without the view in the batch the first vector takes the form
        out, oinf = VariableBaseMSM.multi_scalar_mul_batch(key, [a, c], mont=False)
        assert ctx.pipeline_stats()["unit_scalar_sums"] - before == 3
        ref, ref_inf = cref.msm(curve, xy, buf_h[:n])
        assert np.array_equal(out[0], ref)
    finally:
        ctx.close()


def test_witness_like_vectors_over_a_sharded_key():
    """a multi-device context (three shards on GPU 0): every shard probes and sums the unit scalars of ITS slice; same results as
    the single-device context and the oracle"""
    from accumulation_amd import CommitterKey, Context, MultiContext, VariableBaseMSM
    curve = ffi.AMSM_PALLAS
    n = (1 << 18) + 3
    one, multi = Context(curve), MultiContext(curve, (0, 0, 0))
    try:
        k1 = CommitterKey.generate(one, 0x0E59, n, ffi.AMSM_BASES_PRECOMPUTE | ffi.AMSM_BASES_NO_DIRECT_TABLE)
        xy, inf = k1.read()
        kN = CommitterKey.load(multi, xy, inf, ffi.AMSM_BASES_PRECOMPUTE | ffi.AMSM_BASES_NO_DIRECT_TABLE)
        w, u = _witness(n, 0.2, 41), cref.rng_scalars(42, n)
        a = VariableBaseMSM.multi_scalar_mul_batch(k1, [one.upload(w), one.upload(u)], mont=False)
        b = VariableBaseMSM.multi_scalar_mul_batch(kN, [multi.upload(w), multi.upload(u)], mont=False)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        for j, vec in enumerate((w, u)):
            ref, ref_inf = cref.msm(curve, xy, vec)
            assert np.array_equal(b[0][j], ref) and bool(b[1][j]) == bool(ref_inf)
        assert one.pipeline_stats()["unit_scalar_sums"] == 1
    finally:
        one.close()
        multi.close()


def test_a_non_reduced_montgomery_one_stays_in_the_windows():
    """Montgomery-form scalars: R mod r is the unit the separate sum looks for; R mod r + r represents 1 as well (below 2^256) but is
    not that pattern -- the sum does not take it and the pipelines must not drop it (both apply the SAME test to the stored words)"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    from oracle import pyref as o
    curve, c = ffi.AMSM_PALLAS, o.PALLAS
    ctx = Context(curve)
    try:
        n = 1 << 17
        key = CommitterKey.generate(ctx, 0x0E5A, n, ffi.AMSM_BASES_PRECOMPUTE | ffi.AMSM_BASES_NO_DIRECT_TABLE)
        xy, _ = key.read()
        w = _witness(n, 0.3, 51, ones_only=True)
        up = cref.fr_to_mont(curve, w).copy()
        odd = ((1 << 256) % c.r) + c.r
        assert odd < (1 << 256)
        where = [5, 4097, n // 2 + 1, n - 3]
        for i in where:
            up[i] = [(odd >> (64 * k)) & ((1 << 64) - 1) for k in range(4)]
            w[i] = [1, 0, 0, 0]
        before = ctx.pipeline_stats()["unit_scalar_sums"]
        out, oinf = VariableBaseMSM.multi_scalar_mul_batch(key, [ctx.upload(up)], mont=True)
        assert ctx.pipeline_stats()["unit_scalar_sums"] - before == 1
        ref, ref_inf = cref.msm(curve, xy, w)
        assert np.array_equal(out[0], ref) and bool(oinf[0]) == bool(ref_inf)
    finally:
        ctx.close()
