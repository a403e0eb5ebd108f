"""amsm_msm_oneshot: the literal `VariableBaseMSM::multi_scalar_mul(&[G], &[BigInt])` call shape (ark-ec ^0.2.0 ext, Cargo.toml:15;
SURVEY.md section 8(b) attachment 1) -- host bases AND host scalars belong to one call -- against the golden cases, the big-integer
oracle and the plain-C restatement: identity bases (flag and (0, 0)), min(len) semantics both ways, empty inputs, Montgomery
scalars, ranges (2^19 pairs each, the generators of range j + 1 uploaded beside the MSM of range j), and the same call on a
resident key.  test_host_oneshot_cpu.py re-collects this module on the host backend."""
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


def oneshot(ctx, xy, sc, inf=None, mont=False):
    from accumulation_amd import VariableBaseMSM
    return VariableBaseMSM.multi_scalar_mul_oneshot(ctx, xy, sc, inf, mont)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_golden_cases_oneshot(ctxs, c):
    for case in h.load_golden()["curves"][c.name]["cases"]:
        pts = [h.pt_from_hex(p) for p in case["points"]]
        xy, inf = h.points_to_np(c, pts)
        sc = h.scalars_to_np([int(s, 16) % c.r for s in case["scalars"]])
        k = min(len(pts), sc.shape[0])
        for flags in (inf, None if not inf[:k].any() else inf):  # identity bases by flag; without flags only where there is none
            out, oinf = oneshot(ctxs[c.name], xy, sc, flags)
            assert [hex(int(v)) for v in out] == case["expected_mont_limbs"], case["name"]
            assert int(oinf) == case["expected_is_inf"], case["name"]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_min_len_identity_bases_and_empty(ctxs, cref, c):
    ctx = ctxs[c.name]
    n = 3000
    xy = cref.rng_points(c.curve_id, 0x0E5, n)
    sc = cref.rng_frs(c.curve_id, 0x0E6, n)
    inf = np.zeros(n, dtype=np.uint8)
    inf[[0, 17, n - 1]] = 1
    xy0 = xy.copy()
    xy0[[5, 99]] = 0  # (0, 0) is the identity as well (ark-ec's GroupAffine::zero() serialises to it)
    keep = np.ones(n, dtype=bool)
    keep[[0, 17, n - 1, 5, 99]] = False
    ref, rinf = cref.msm(c.curve_id, xy[keep], sc[keep], threads=4)
    got, ginf = oneshot(ctx, xy0, sc, inf)
    assert bool(ginf) == bool(rinf) and np.array_equal(got, ref)
    # min(bases.len(), scalars.len()): more bases than scalars, more scalars than bases
    for nb, ns in ((n, 1234), (777, n)):
        k = min(nb, ns)
        ref, rinf = cref.msm(c.curve_id, xy[:k], sc[:k], threads=4)
        got, ginf = oneshot(ctx, xy[:nb], sc[:ns])
        assert bool(ginf) == bool(rinf) and np.array_equal(got, ref), (nb, ns)
    # empty on either side -> the identity, all-zero scalars too
    L = 2 * ctx.fq_limbs
    for a, b in ((xy[:0], sc), (xy, sc[:0]), (xy[:0], sc[:0])):
        got, ginf = oneshot(ctx, a.reshape(-1, L), b.reshape(-1, 4))
        assert ginf and not got.any()
    got, ginf = oneshot(ctx, xy, np.zeros_like(sc))
    assert ginf and not got.any()
    # Montgomery scalars (raw `&[Fr]` memory)
    ref, rinf = cref.msm(c.curve_id, xy, sc, threads=4)
    got, ginf = oneshot(ctx, xy, cref.fr_to_mont(c.curve_id, sc), mont=True)
    assert bool(ginf) == bool(rinf) and np.array_equal(got, ref)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [1, 2, 255, 4097, (1 << 16) + 3])
def test_sizes_vs_c_oracle_and_resident_key(ctxs, cref, c, n):
    from accumulation_amd import CommitterKey, VariableBaseMSM
    ctx = ctxs[c.name]
    xy = cref.rng_points(c.curve_id, 0x0E7 + n, n)
    sc = cref.rng_frs(c.curve_id, 0x0E8 + n, n)
    ref, rinf = cref.msm(c.curve_id, xy, sc, threads=8)
    got, ginf = oneshot(ctx, xy, sc)
    assert bool(ginf) == bool(rinf) and np.array_equal(got, ref)
    ck = CommitterKey.load(ctx, xy, None, 2)
    res, resinf = VariableBaseMSM.multi_scalar_mul(ck, sc)
    ck.free()
    assert bool(resinf) == bool(ginf) and np.array_equal(res, got)


def test_ranges_overlap_upload_and_msm(ctxs, cref):
    """more than 2^19 pairs: ranges whose generators arrive while the previous range computes; skewed and two-valued scalar
    vectors take the other paths over the same temporary key; a second call reuses the buffer"""
    c = o.PALLAS
    ctx = ctxs[c.name]
    n = (1 << 20) + 12345
    xy = cref.rng_points(c.curve_id, 0x0EA, n, threads=16)
    inf = np.zeros(n, dtype=np.uint8)
    inf[[3, (1 << 19) - 1, 1 << 19, n - 2]] = 1
    keep = inf == 0
    vecs = {"uniform": cref.rng_frs(c.curve_id, 0x0EB, n),
            "constant": np.tile(cref.rng_frs(c.curve_id, 0x0EC, 1), (n, 1)),
            "few_values": cref.rng_frs(c.curve_id, 0x0ED, 4)[np.random.default_rng(5).integers(0, 4, n)]}
    for name, sc in vecs.items():
        ref, rinf = cref.msm(c.curve_id, xy[keep], sc[keep], threads=17)
        got, ginf = oneshot(ctx, xy, sc, inf)
        assert bool(ginf) == bool(rinf) and np.array_equal(got, ref), name
    m = ctx.memory()
    assert m["workspace_bytes"] >= n * 64  # the generators passed through the context's one-shot buffer ...
    ctx.trim()
    assert ctx.memory()["workspace_bytes"] < n * 64  # ... which amsm_ctx_trim releases: nothing stays resident
    sc = vecs["uniform"][: 1 << 17]
    ref, rinf = cref.msm(c.curve_id, xy[: 1 << 17][keep[: 1 << 17]], sc[keep[: 1 << 17]], threads=17)
    got, ginf = oneshot(ctx, xy[: 1 << 17], sc, inf[: 1 << 17])
    assert bool(ginf) == bool(rinf) and np.array_equal(got, ref)


def test_oneshot_over_two_shards(cref):
    """a multi-device context splits the pairs over its devices (both shards on GPU 0 here) and folds the sums on the host"""
    from accumulation_amd import MultiContext
    c = o.PALLAS
    n = 70001
    xy = cref.rng_points(c.curve_id, 0x0EE, n)
    sc = cref.rng_frs(c.curve_id, 0x0EF, n)
    ref, rinf = cref.msm(c.curve_id, xy, sc, threads=8)
    ctx = MultiContext(c.curve_id, [0, 0, 0])
    try:
        got, ginf = oneshot(ctx, xy, sc)
        assert bool(ginf) == bool(rinf) and np.array_equal(got, ref)
    finally:
        ctx.close()
