"""The pipeline an MSM actually TAKES on the GPU equals what the selection table says (csrc/msm_select.h; its edges are checked
without a GPU in tests/test_pipeline_select_cpu.py): the context's counters at the threshold edges, results against the C oracle."""
import numpy as np
import pytest

from oracle import pyref as o

pytestmark = pytest.mark.gpu
C = o.PALLAS
P = lambda k: 1 << k  # noqa: E731


def took(ctx, fn):
    a = ctx.pipeline_stats()
    out = fn()
    b = ctx.pipeline_stats()
    d = {k: b[k] - a[k] for k in ("bucket_per_lane", "bucket_split", "direct_sum", "shared_bucket_sets")}
    name = "direct_sum" if d["direct_sum"] else ("bucket_split" if d["bucket_split"] else ("bucket_per_lane" if d["bucket_per_lane"] else "chunked"))
    return name, d, out


@pytest.mark.parametrize("gens,flags,cases", [
    (P(15), 1, [(1, "direct_sum"), (P(15), "direct_sum")]),
    (P(16), 1, [(P(15), "chunked"), (P(16) - 1, "chunked"), (P(16), "bucket_split")]),
    (P(17), 1, [(P(16) + 1, "bucket_split"), (P(17), "bucket_split")]),
    (P(19), 1, [(P(17) + 1, "chunked"), (P(19), "chunked")]),
    (P(20), 1, [(P(18) + 1, "bucket_per_lane"), (P(19) + 1, "bucket_per_lane"), (P(20), "bucket_per_lane")]),
    (P(20) + 1, 2, [(P(17), "chunked"), (P(17) + 1, "bucket_per_lane"), (P(18), "bucket_per_lane"), (P(18) + 1, "bucket_per_lane"),
                    (P(20), "bucket_per_lane"), (P(20) + 1, "bucket_per_lane")]),
], ids=["direct_2p15", "table_2p16", "table_2p17", "table_2p19", "bpl_2p20", "plain_2p20_plus_1"])
def test_counters_follow_the_table(cref, gens, flags, cases):
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    ctx = Context(C.curve_id)
    try:
        ck = CommitterKey.generate(ctx, 0x5EED6001, gens, flags)
        assert ck.precomputed == (flags == 1)
        xy, _ = ck.read()
        for n, want in cases:
            v = ctx.random_vector(0x5EED6100 + (n & 0xFFFF), n, mont=False)
            name, d, (got, ginf) = took(ctx, lambda: VariableBaseMSM.multi_scalar_mul(ck, v, mont=False))
            assert name == want, (gens, n, d)
            ref, rinf = cref.msm(C.curve_id, xy[:n], v.download(), threads=8)
            assert ginf == rinf and np.array_equal(got, ref), (gens, n)
        ck.free()
    finally:
        ctx.close()


def test_grouped_forms(cref):
    """grouped MSMs (two sums by one bit of the index: the IPA rounds) take the same rows; a short range of the 20-bit key runs over
    its twin; the two sums add up to the ungrouped MSM"""
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    from tests import helpers as h
    ctx = Context(C.curve_id)
    try:
        for gens, cases in ((P(16), [(P(16), "bucket_split"), (P(15), "chunked")]),
                            (P(20), [(P(19) + 2, "bucket_per_lane"), (P(20), "bucket_per_lane"), (P(15), "chunked")])):
            ck = CommitterKey.generate(ctx, 0x5EED6002, gens)
            for n, want in cases:
                v = ctx.random_vector(0x5EED6200 + n % 97, n, mont=True)
                name, d, (got, ginf) = took(ctx, lambda: VariableBaseMSM.multi_scalar_mul_grouped(ck, v, 3, mont=True))
                assert name == want, (gens, n, d)
                a, ai = VariableBaseMSM.multi_scalar_mul(ck, v, mont=True)
                s = o.add(C, *[h.np_to_point(C, got[g], bool(ginf[g])) for g in (0, 1)])
                assert s == h.np_to_point(C, a, bool(ai)), (gens, n)
            if gens == P(20):
                assert ck.tables()["twin"] > 0  # the short range built it
            ck.free()
    finally:
        ctx.close()
