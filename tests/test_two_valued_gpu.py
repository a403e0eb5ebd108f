"""Two-valued scalar vectors (every scalar 0 or one value v: the reference harness's `vec![rand; n]` hp_as inputs,
src/hp_as/mod.rs:189-190, the DummyCircuit's A z / B z / C z, src/r1cs_nark_as/mod.rs:1159-1188, boolean witnesses) take
v * (sum of the generators with a non-zero scalar) instead of the windowed pipelines.  Same canonical results as the C oracle
(oracle/ark_msm.c: ark-ec's algorithm) for every shape of such a vector; up to eight scalars may be something else (added one by
one on the host), more than that -- or anything uniform -- must not take the shortcut."""
import numpy as np
import pytest

from accumulation_amd import ffi
from oracle import cref

pytestmark = pytest.mark.gpu

CURVES = [ffi.AMSM_PALLAS, ffi.AMSM_BLS12_381_G1]


def _vectors(curve, n):
    """name -> (canonical scalars (n, 4), expected to take the shortcut)"""
    rnd = cref.rng_scalars(0x7E57 + n, n)
    v = rnd[5].copy()
    out = {}
    const = np.tile(v, (n, 1))
    out["constant"] = (const, True)
    dummy = const.copy()
    dummy[-1] = 0                                   # the DummyCircuit's zero row
    out["constant_with_zero_row"] = (dummy, True)
    ones = np.zeros((n, 4), dtype=np.uint64)
    ones[:, 0] = (np.arange(n) % 3 != 0).astype(np.uint64)   # boolean witness: {0, 1}
    out["boolean"] = (ones, True)
    sparse = np.zeros((n, 4), dtype=np.uint64)
    sparse[7::997] = v                              # few non-zero entries, head mostly zero
    out["sparse_one_value"] = (sparse, True)
    out["all_zero"] = (np.zeros((n, 4), dtype=np.uint64), True)   # the identity, nothing launched
    third = const.copy()
    third[n - 3] = rnd[9]                           # one other value near the end (the DummyCircuit's blinded zero row): an exception
    out["one_exception_at_the_end"] = (third, True)
    off_by_one = const.copy()
    off_by_one[n // 2, 3] ^= np.uint64(1)           # differs in the top word only: an exception too, never mistaken for v
    out["one_scalar_differs_in_its_top_word"] = (off_by_one, True)
    odd_first = const.copy()
    odd_first[0] = rnd[11]                          # the odd value FIRST: v is the majority of the first three non-zero scalars
    out["odd_value_first"] = (odd_first, True)
    eight = dummy.copy()
    for j, pos in enumerate((1, 70, 4099, n // 3, n // 3 + 1, n // 2 + 17, n - 2000, n - 2)):
        eight[pos] = rnd[20 + j]                    # eight exceptions, two of them in one wave
    out["eight_exceptions"] = (eight, True)
    nine = eight.copy()
    nine[n // 5] = rnd[40]                          # a ninth, in another wave: over the limit, regular pipelines
    out["nine_exceptions"] = (nine, False)
    clump = const.copy()
    clump[5000:5012] = rnd[50:62]                   # twelve others inside one wave
    out["twelve_others_in_one_wave"] = (clump, False)
    head_zero = rnd.copy()
    head_zero[:2048] = 0                            # all-zero head: left to the regular pipelines
    out["uniform_behind_a_zero_head"] = (head_zero, False)
    out["uniform"] = (rnd, False)
    return out


@pytest.fixture(scope="module")
def env():
    from accumulation_amd import CommitterKey, Context
    out = {}
    for curve in CURVES:
        ctx = Context(curve)
        n_key = (1 << 16) + 37
        pre = CommitterKey.generate(ctx, 0x5EED7E57, n_key)
        xy, inf = pre.read()
        plain = CommitterKey.load(ctx, xy, inf, 2)
        out[curve] = (ctx, pre, plain, xy)
    yield out
    for ctx, pre, plain, _ in out.values():
        pre.free()
        plain.free()
        ctx.close()


@pytest.mark.parametrize("curve", CURVES, ids=["pallas", "bls12_381_g1"])
@pytest.mark.parametrize("mont", [False, True], ids=["canonical", "montgomery"])
def test_two_valued_vectors_in_a_batch_vs_c_oracle(env, curve, mont):
    from accumulation_amd import VariableBaseMSM
    ctx, pre, plain, xy = env[curve]
    n = (1 << 16) + 37
    vecs = _vectors(curve, n)
    names = list(vecs)
    want = {k: cref.msm(curve, xy[:n], vecs[k][0]) for k in names}
    for key in (pre, plain):
        up = [ctx.upload(cref.fr_to_mont(curve, vecs[k][0]) if mont else vecs[k][0]) for k in names]
        before = ctx.two_valued_msms()
        out, inf = VariableBaseMSM.multi_scalar_mul_batch(key, up, mont=mont)
        taken = ctx.two_valued_msms() - before
        assert taken == sum(1 for k in names if vecs[k][1]), (taken, names)
        for j, k in enumerate(names):
            ref, ref_inf = want[k]
            assert bool(inf[j]) == bool(ref_inf) and np.array_equal(out[j], ref), (k, mont, key.precomputed)


@pytest.mark.parametrize("curve", CURVES, ids=["pallas", "bls12_381_g1"])
def test_host_slices_that_look_two_valued_take_the_device_probe(env, curve):
    """amsm_msm / amsm_msm_batch over HOST slices (the `&[Fr]` shape a Rust adapter passes: the reference harness's constant
    vectors arrive this way): a 1024-sample look on the host sends the call's slices to the device path, whose exact probe decides
    per vector -- the shortcut's vectors take it, the rest the regular pipelines, results as the C oracle's"""
    from accumulation_amd import VariableBaseMSM
    ctx, pre, plain, xy = env[curve]
    n = (1 << 16) + 37
    vecs = _vectors(curve, n)
    names = ["constant", "uniform", "constant_with_zero_row", "eight_exceptions", "nine_exceptions", "boolean", "all_zero"]
    for key in (pre, plain):
        before = ctx.two_valued_msms()
        out, inf = VariableBaseMSM.multi_scalar_mul_batch_host(key, [vecs[k][0] for k in names], mont=False)
        assert ctx.two_valued_msms() - before == sum(1 for k in names if vecs[k][1])
        for j, k in enumerate(names):
            ref, ref_inf = cref.msm(curve, xy[:n], vecs[k][0])
            assert bool(inf[j]) == bool(ref_inf) and np.array_equal(out[j], ref), (k, key.precomputed)
        # a single host call (amsm_msm: uploaded, then the lone device vector's rules -- probed where another probe synchronises
        # anyway or from 2^17 pairs up)
        one, one_inf = VariableBaseMSM.multi_scalar_mul(key, vecs["one_exception_at_the_end"][0])
        ref, ref_inf = cref.msm(curve, xy[:n], vecs["one_exception_at_the_end"][0])
        assert bool(one_inf) == bool(ref_inf) and np.array_equal(one, ref)
        # uniform slices only: nothing looks two-valued, the host path with its overlapped uploads
        before = ctx.two_valued_msms()
        out, inf = VariableBaseMSM.multi_scalar_mul_batch_host(key, [vecs["uniform"][0], vecs["uniform_behind_a_zero_head"][0]])
        assert ctx.two_valued_msms() == before
        for j, k in enumerate(("uniform", "uniform_behind_a_zero_head")):
            ref, ref_inf = cref.msm(curve, xy[:n], vecs[k][0])
            assert bool(inf[j]) == bool(ref_inf) and np.array_equal(out[j], ref), k


@pytest.mark.parametrize("curve", CURVES, ids=["pallas", "bls12_381_g1"])
def test_windows_of_the_key_and_short_vectors(env, curve):
    """base_off != 0 (a window of the key) and vectors below the probe's minimum length (regular pipelines)."""
    from accumulation_amd import VariableBaseMSM
    ctx, pre, _, xy = env[curve]
    n = 1 << 15
    v = cref.rng_scalars(0x7E58, 4)[2]
    const = np.tile(v, (n, 1))
    for off, m, expect in ((1234, n, 2), (n, n // 8, 0)):
        vec = ctx.upload(const[:m])
        before = ctx.two_valued_msms()
        out, inf = VariableBaseMSM.multi_scalar_mul_batch(pre, [vec, vec], mont=False, base_off=off)
        assert ctx.two_valued_msms() - before == expect
        ref, ref_inf = cref.msm(curve, xy[off:off + m], const[:m])
        for j in range(2):
            assert bool(inf[j]) == bool(ref_inf) and np.array_equal(out[j], ref), (off, m, j)


def test_switch_off(env):
    """AMSM_TWO_VALUED=0: the same vectors through the regular pipelines, same results."""
    import os
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    curve = ffi.AMSM_PALLAS
    os.environ["AMSM_TWO_VALUED"] = "0"
    try:
        ctx = Context(curve)
    finally:
        os.environ.pop("AMSM_TWO_VALUED", None)
    n = 1 << 15
    ck = CommitterKey.generate(ctx, 0x5EED7E57, n)
    xy, _ = ck.read()
    const = np.tile(cref.rng_scalars(0x7E59, 1)[0], (n, 1))
    out, inf = VariableBaseMSM.multi_scalar_mul_batch(ck, [ctx.upload(const)] * 2, mont=False)
    assert ctx.two_valued_msms() == 0
    ref, ref_inf = cref.msm(curve, xy, const)
    assert np.array_equal(out[0], ref) and np.array_equal(out[1], ref) and bool(inf[0]) == bool(ref_inf)
    ck.free()
    ctx.close()


@pytest.mark.parametrize("curve", CURVES, ids=["pallas", "bls12_381_g1"])
@pytest.mark.parametrize("log_n", [15, 16], ids=["key_2p15_13_bit_windows", "key_2p16_16_bit_windows"])
def test_scalars_at_or_above_2p255_get_the_windowed_pipelines_verdict(curve, log_n):
    """round-3 ADVICE: a canonical-form (mont = 0) constant vector whose value is 2^255 or more came back as (v mod r) * sum with
    AMSM_OK from the shortcut, whatever the windowed pipelines say about such scalars (AMSM_E_SCALAR_RANGE where the key's
    windows cannot take them -- 16 x 16 bits --, the integer's multiple where they can -- 20 x 13 bits).  Now such values, as the
    vector's value or among its exceptions, never take the shortcut: the outcome equals that of a context with the shortcut
    off.  Values in [r, 2^255) give the same point either way."""
    import os
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM
    from oracle import pyref as o
    c = o.PALLAS if curve == ffi.AMSM_PALLAS else o.BLS12_381_G1
    n = 1 << log_n
    big = np.array(o.int_to_limbs((1 << 256) - 12345, 4), dtype=np.uint64)
    v = cref.rng_scalars(0x7E60, 1)[0]
    exc = np.tile(v, (n, 1))
    exc[n // 2] = big
    cases = {"constant": np.tile(big, (n, 1)), "exception": exc}
    outcome = {}
    for mode in ("1", "0"):
        os.environ["AMSM_TWO_VALUED"] = mode
        try:
            ctx = Context(curve)
        finally:
            os.environ.pop("AMSM_TWO_VALUED", None)
        ck = CommitterKey.generate(ctx, 0x5EED7E61, n)
        xy, _ = ck.read()
        for name, vec in cases.items():
            try:
                out, inf = VariableBaseMSM.multi_scalar_mul_batch(ck, [ctx.upload(vec)] * 2, mont=False)
                outcome[(mode, name)] = ("point", out[0].tolist(), bool(inf[0]))
            except ffi.AmsmError as e:
                outcome[(mode, name)] = ("error", e.status)
        if mode == "1":
            assert ctx.two_valued_msms() == 0  # neither vector's result came from the shortcut
            if c.r + 5 < (1 << 255):  # r <= value < 2^255: within what the windows take -- the point of value mod r, by the shortcut
                above = np.array(o.int_to_limbs(c.r + 5, 4), dtype=np.uint64)
                out, inf = VariableBaseMSM.multi_scalar_mul_batch(ck, [ctx.upload(np.tile(above, (n, 1)))] * 2, mont=False)
                ref, rinf = cref.msm(curve, xy[:n], np.tile(np.array(o.int_to_limbs(5, 4), dtype=np.uint64), (n, 1)))
                assert bool(inf[0]) == bool(rinf) and np.array_equal(out[0], ref) and ctx.two_valued_msms() == 2
            # and the context keeps working
            out, inf = VariableBaseMSM.multi_scalar_mul_batch(ck, [ctx.upload(np.tile(v, (n, 1)))], mont=False)
            ref, rinf = cref.msm(curve, xy[:n], np.tile(v, (n, 1)))
            assert bool(inf[0]) == bool(rinf) and np.array_equal(out[0], ref)
        ck.free()
        ctx.close()
    for name in cases:
        assert outcome[("1", name)] == outcome[("0", name)], name
    if log_n == 16:  # 16 windows of 16 bits: nothing above 2^255 fits
        assert outcome[("1", "constant")] == ("error", ffi.AMSM_E_SCALAR_RANGE)
