"""The bucket-split pipeline (round 3: k_prep_local_s + k_accum_bps -- every bucket of a small MSM on 2 .. 64 adjacent lanes,
no partial records) against the CPU restatement, bit for bit.  By default only the IPA rounds' grouped MSMs take it
(tests/test_ipa_gpu.py covers those end to end); AMSM_BPS=2 sends every precomputed-key MSM of 2^16 .. 2^17 pairs through it
behind the skew probe -- that mode is exercised here: uniform scalars on both curves, one and two bucket sets, base offsets,
skewed vectors (probe -> chunked; probe off -> overflow flag -> re-run chunked over the same key)."""
import os

import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h

pytestmark = pytest.mark.gpu


def make_ctx(curve_id, **env):
    from accumulation_amd import Context
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return Context(curve_id)
    finally:
        for k in env:
            del os.environ[k]


@pytest.mark.parametrize("c", [o.PALLAS, o.BLS12_381_G1], ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1 << 16, (1 << 16) + 12345, 1 << 17])
def test_uniform_scalars_vs_c_oracle(cref, c, n):
    from accumulation_amd import CommitterKey, VariableBaseMSM
    ctx = make_ctx(c.curve_id, AMSM_BPS=2)
    try:
        ck = CommitterKey.generate(ctx, 21, n + 1000)
        xy, _ = ck.read()
        sc = cref.rng_scalars(22 + n, n)
        got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc, base_off=500)
        ref, rinf = cref.msm(c.curve_id, xy[500:500 + n], sc, threads=8)
        assert bool(inf) == bool(rinf) and np.array_equal(got, ref)
        st = ctx.pipeline_stats()
        assert st["bucket_split"] == 1 and st["bucket_split_fallbacks"] == 0
        ck.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("probe", [1, 0], ids=["probe", "no_probe"])
def test_skewed_vectors_end_up_chunked(cref, probe):
    from accumulation_amd import CommitterKey, VariableBaseMSM
    c = o.PALLAS
    n = 1 << 16
    ctx = make_ctx(c.curve_id, AMSM_BPS=2, AMSM_BPL_PROBE=probe)
    try:
        ck = CommitterKey.generate(ctx, 23, n)
        xy, _ = ck.read()
        rng = np.random.default_rng(3)
        cases = {"constant": np.tile(h.scalars_to_np([o.rng_scalar(24, 0)]), (n, 1)),
                 "all_one": np.tile(h.scalars_to_np([1]), (n, 1)),
                 "few_values": cref.rng_scalars(25, 8)[rng.integers(0, 8, n)]}
        for name, sc in cases.items():
            before = ctx.pipeline_stats()
            got, inf = VariableBaseMSM.multi_scalar_mul(ck, sc)
            after = ctx.pipeline_stats()
            ref, rinf = cref.msm(c.curve_id, xy, sc, threads=8)
            assert bool(inf) == bool(rinf) and np.array_equal(got, ref), name
            took = after["bucket_split"] - before["bucket_split"]
            fell = after["bucket_split_fallbacks"] - before["bucket_split_fallbacks"]
            assert (took, fell) == ((0, 0) if probe else (1, 1)), (name, took, fell)
        ck.free()
    finally:
        ctx.close()


def test_grouped_and_plain_msms_take_it_by_default(cref):
    """the IPA rounds' form: two sums over index classes in one pass (two bucket sets), default settings"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    c = o.PALLAS
    n = 1 << 16
    ctx = Context(c.curve_id)
    try:
        ck = CommitterKey.generate(ctx, 26, n)
        xy, _ = ck.read()
        v = ctx.random_vector(27, n, mont=True)
        sc = cref.fr_from_mont(c.curve_id, v.download())
        pts, infs = VariableBaseMSM.multi_scalar_mul_grouped(ck, v, 5, mont=True)
        assert ctx.pipeline_stats()["bucket_split"] == 1
        for g in (0, 1):
            sel = sc.copy()
            sel[((np.arange(n) >> 5) & 1) != g] = 0
            ref, rinf = cref.msm(c.curve_id, xy, sel, threads=8)
            assert bool(infs[g]) == bool(rinf) and np.array_equal(pts[g], ref), g
        # and so does a plain MSM of the same size (default since late round 3: AMSM_BPS=2, behind the skew probe)
        got, ginf = VariableBaseMSM.multi_scalar_mul(ck, v, mont=True)
        assert ctx.pipeline_stats()["bucket_split"] == 2
        ref, rinf = cref.msm(c.curve_id, xy, sc, threads=8)
        assert bool(ginf) == bool(rinf) and np.array_equal(got, ref)
        ck.free()
    finally:
        ctx.close()
