"""The optional tables of a key are a decision, not a side effect (round 5; include/amsm.h: AMSM_BASES_NO_DIRECT_TABLE, AMSM_BASES_NO_TWIN,
amsm_ctx_set_table_budget, amsm_bases_tables, amsm_ctx_tables_denied): a key that was denied a table says so, still computes the
same points on the slower path, and never allocates behind the caller's back."""
import numpy as np
import pytest

from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
from oracle import pyref as o

pytestmark = pytest.mark.gpu
C = o.PALLAS


def test_direct_sum_table_by_flag_and_by_budget(cref):
    ctx = Context(C.curve_id)
    try:
        n = 1 << 12
        sc = cref.rng_scalars(0x7AB1, n)
        ref = None
        for what in ("default", "flag", "budget"):
            if what == "budget":
                ctx.set_table_budget(1 << 20)  # 1 MiB: the 128 MiB table of this key does not fit
            before = ctx.tables_denied()
            ck = CommitterKey.generate(ctx, 0x7AB2, n, ffi.AMSM_BASES_PRECOMPUTE | (ffi.AMSM_BASES_NO_DIRECT_TABLE if what == "flag" else 0))
            t = ck.tables()
            assert ck.precomputed and t["window_table"] == n * 64 * t["levels"]
            if what == "default":
                assert t["direct_sum_table"] == n * 512 * 64 and t["direct_sum_table_denied"] is None
            else:
                assert t["direct_sum_table"] == 0 and t["direct_sum_table_denied"] == what
            # the caller's own flag is not counted as a denial; the budget is
            assert ctx.tables_denied() - before == (1 if what == "budget" else 0)
            d0 = ctx.pipeline_stats()["direct_sum"]
            got = VariableBaseMSM.multi_scalar_mul(ck, sc)
            assert (ctx.pipeline_stats()["direct_sum"] - d0) == (1 if what == "default" else 0)
            if ref is None:
                xy, _ = ck.read()
                r, ri = cref.msm(C.curve_id, xy, sc, threads=4)
                ref = (r, bool(ri))
            assert np.array_equal(got[0], ref[0]) and bool(got[1]) == ref[1], what
            assert ck.memory()["table"] == t["window_table"] + t["direct_sum_table"]
            ck.free()
        ctx.set_table_budget((1 << 64) - 1)
    finally:
        ctx.close()


def test_largest_small_key_builds_its_table_in_slabs_with_bounded_scratch(cref):
    """2^15 generators: 1 GiB of table built through a scratch of at most 256 MiB (it was 2-3 GiB at once: ADVICE r4)"""
    ctx = Context(C.curve_id)
    try:
        n = 1 << 15
        ck = CommitterKey.generate(ctx, 0x7AB3, n)
        assert ck.tables()["direct_sum_table"] == n * 512 * 64
        assert ctx.memory()["workspace_bytes"] <= (300 << 20)
        sc = cref.rng_scalars(0x7AB4, n)
        got, ginf = VariableBaseMSM.multi_scalar_mul(ck, sc)
        xy, _ = ck.read()
        ref, rinf = cref.msm(C.curve_id, xy, sc, threads=8)
        assert ginf == rinf and np.array_equal(got, ref)
        ck.free()
    finally:
        ctx.close()


def test_twin_refused_by_flag_and_by_budget(cref):
    ctx = Context(C.curve_id)
    try:
        n = 1 << 20
        for what in ("flag", "budget"):
            if what == "budget":
                ctx.set_table_budget(900 << 20)  # the 832 MiB table was built before; its 1 GiB twin does not fit
            ck = CommitterKey.generate(ctx, 0x7AB5, n, ffi.AMSM_BASES_PRECOMPUTE | (ffi.AMSM_BASES_NO_TWIN if what == "flag" else 0))
            assert ck.window_bits == 20
            full = ctx.random_vector(1, n, mont=True)
            VariableBaseMSM.multi_scalar_mul(ck, full, mont=True)  # the 20-bit table's own pipeline: fine
            before = ctx.tables_denied()
            short = ctx.random_vector(2, 1 << 16, mont=True)
            with pytest.raises(ffi.AmsmError) as e:  # a range below a quarter of the window needs the twin
                VariableBaseMSM.multi_scalar_mul(ck, short, mont=True)
            assert e.value.status == ffi.AMSM_E_UNSUPPORTED
            t = ck.tables()
            assert t["twin"] == 0 and t["twin_denied"] == what and ctx.tables_denied() - before == 1
            with pytest.raises(ffi.AmsmError):
                VariableBaseMSM.multi_scalar_mul(ck, short, mont=True)
            assert ctx.tables_denied() - before == 1  # counted once per key
            VariableBaseMSM.multi_scalar_mul(ck, full, mont=True)  # and the context is still usable
            ck.free()
        ctx.set_table_budget((1 << 64) - 1)
        ck = CommitterKey.generate(ctx, 0x7AB5, n)
        short = ctx.random_vector(2, 1 << 16, mont=True)
        got, ginf = VariableBaseMSM.multi_scalar_mul(ck, short, mont=True)
        assert ck.tables()["twin"] > 0
        xy, _ = ck.read(0, 1 << 16)
        ref, rinf = cref.msm(C.curve_id, xy, cref.fr_from_mont(C.curve_id, short.download()), threads=8)
        assert ginf == rinf and np.array_equal(got, ref)
        ck.free()
    finally:
        ctx.close()
