"""GPU parity tests for the scalar-field vector kernels (compute_hp, combine_vectors, scale_vector,
compute_t_vecs) against the restatements in oracle/, bit-exact on Montgomery limbs; plus the reference's
decider identity driven end-to-end on the device."""
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


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n", [0, 1, 255, 256, 257, 5000])
def test_compute_hp(ctxs, cref, c, n):
    from accumulation_amd.hp_as import compute_hp
    ctx = ctxs[c.name]
    a = cref.fr_to_mont(c.curve_id, cref.rng_scalars(1, n))
    b = cref.fr_to_mont(c.curve_id, cref.rng_scalars(2, n + 3))  # zip truncates to the shorter
    got = compute_hp(ctx, ctx.upload(a), ctx.upload(b)).download()
    assert np.array_equal(got, cref.fr_hadamard(c.curve_id, a, b[:n]))


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_combine_vectors_ragged_and_hiding(ctxs, cref, c):
    from accumulation_amd.hp_as import combine_vectors, scale_vector
    ctx = ctxs[c.name]
    lens = [700, 513, 700, 1]
    vecs = [cref.fr_to_mont(c.curve_id, cref.rng_scalars(10 + j, ln)) for j, ln in enumerate(lens)]
    ch = cref.fr_to_mont(c.curve_id, cref.rng_scalars(20, len(lens)))
    hid = cref.fr_to_mont(c.curve_id, cref.rng_scalars(21, 650))
    dv = [ctx.upload(v) for v in vecs]
    got = combine_vectors(ctx, dv, ch, ctx.upload(hid)).download()
    assert np.array_equal(got, cref.fr_combine(c.curve_id, vecs, ch, hiding=hid))
    got = combine_vectors(ctx, dv[:2], ch[:2]).download()
    assert np.array_equal(got, cref.fr_combine(c.curve_id, vecs[:2], ch[:2]))
    got = scale_vector(ctx, dv[0], ch[3]).download()
    assert np.array_equal(got, cref.fr_combine(c.curve_id, vecs[:1], ch[3:4]))
    # python restatement agrees too (small slice)
    exp = o.combine_vectors(c, [h.fr_from_mont_np(c, v[:9]) for v in vecs[:2]], h.fr_from_mont_np(c, ch[:2]))
    assert h.fr_from_mont_np(c, got[:0]) == [] and h.fr_from_mont_np(
        c, combine_vectors(ctx, [ctx.upload(vecs[0][:9]), ctx.upload(vecs[1][:9])], ch[:2]).download()) == exp


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("k", [1, 2, 5, 8, 9, 13, 17])
def test_combine_vectors_any_count_and_unit_coefficients(ctxs, cref, c, k):
    """the reference has no limit on the number of addends (src/hp_as/mod.rs:492-512): more than the 8 of one launch are
    combined in groups; coefficients equal to one (mu_0, beta_0, nu^0) take the no-multiplication branch"""
    from accumulation_amd.hp_as import combine_vectors
    ctx = ctxs[c.name]
    lens = [300 - 7 * j for j in range(k)]
    vecs = [cref.fr_to_mont(c.curve_id, cref.rng_scalars(50 + j, ln)) for j, ln in enumerate(lens)]
    ch = cref.fr_to_mont(c.curve_id, cref.rng_scalars(70, k))
    one = cref.fr_to_mont(c.curve_id, h.scalars_to_np([1]))[0]
    ch[0] = one
    if k > 9:
        ch[9] = one
    hid = cref.fr_to_mont(c.curve_id, cref.rng_scalars(71, 310))
    dv = [ctx.upload(v) for v in vecs]
    assert np.array_equal(combine_vectors(ctx, dv, ch).download(), cref.fr_combine(c.curve_id, vecs, ch))
    assert np.array_equal(combine_vectors(ctx, dv, ch, ctx.upload(hid)).download(),
                          cref.fr_combine(c.curve_id, vecs, ch, hiding=hid))


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("n_in", [1, 2, 3, 4, 8, 9, 13, 16, 17])
@pytest.mark.parametrize("zk", [False, True])
def test_compute_t_vecs(ctxs, c, n_in, zk):
    from accumulation_amd.hp_as import compute_t_vecs
    ctx = ctxs[c.name]
    ln = 77
    a = [o.rng_scalars(100 + j, ln if j != 1 else ln - 5) for j in range(n_in)]  # one ragged witness
    b = [o.rng_scalars(200 + j, ln) for j in range(n_in)]
    mu = [1] + o.rng_scalars(300, n_in)  # n_in + 1 challenges (the last only used when zk)
    hiding = (o.rng_scalars(400, ln), o.rng_scalars(401, ln - 2)) if zk else None
    exp = o.compute_t_vecs(c, a, b, mu, ln, hiding)
    da = [ctx.upload(h.fr_mont_np(c, v)) for v in a]
    db = [ctx.upload(h.fr_mont_np(c, v)) for v in b]
    dh = None if not zk else (ctx.upload(h.fr_mont_np(c, hiding[0])), ctx.upload(h.fr_mont_np(c, hiding[1])))
    got = compute_t_vecs(ctx, da, db, h.fr_mont_np(c, mu), ln, dh)
    assert len(got) == 2 * n_in - 1
    for k in range(2 * n_in - 1):
        assert h.fr_from_mont_np(c, got[k].download()) == exp[k], k
    # the committed subset only (coefficient n-1 is never committed, src/hp_as/mod.rs:373-375)
    got2 = compute_t_vecs(ctx, da, db, h.fr_mont_np(c, mu), ln, dh, skip_uncommitted=True)
    assert got2[n_in - 1] is None
    for k in range(2 * n_in - 1):
        if k != n_in - 1:
            assert np.array_equal(got2[k].download(), got[k].download())


def test_t_vecs_rejects_too_few_challenges(ctxs):
    from accumulation_amd import ffi
    from accumulation_amd.hp_as import compute_t_vecs
    c = o.PALLAS
    ctx = ctxs[c.name]
    v = ctx.upload(h.fr_mont_np(c, [1, 2, 3]))
    with pytest.raises(ffi.AmsmError) as e:  # assert!(n + hiding <= mu.len())  src/hp_as/mod.rs:295
        compute_t_vecs(ctx, [v, v], [v, v], h.fr_mont_np(c, [1]), 3)
    assert e.value.status == ffi.AMSM_E_INVALID_ARG


def test_hp_accumulation_decider_identity(ctxs):
    """One no-zk hp_as accumulation of two inputs with every vector loop and every MSM on the GPU
    (Appendix A.1 of SURVEY.md): the decider's checks commit(a') == C1', commit(b') == C2',
    commit(a' o b') == C3' must hold for the combined instance computed the verifier's way
    (src/hp_as/mod.rs:409-479, 894-925).  Challenges are injected (the sponge stays on the host)."""
    from accumulation_amd import PedersenCommitment, VariableBaseMSM
    from accumulation_amd.hp_as import (combine_vectors, compute_hp, compute_product_poly_comm, compute_t_vecs,
                                        decide_commitments)
    c = o.PALLAS
    ctx = ctxs[c.name]
    ln, n = 1 << 12, 2
    ck = PedersenCommitment.setup(ctx, ln, seed=123)
    a = [ctx.random_vector(500 + j, ln, mont=True) for j in range(n)]
    b = [ctx.random_vector(600 + j, ln, mont=True) for j in range(n)]
    # input instances: (commit(a_j), commit(b_j), commit(a_j o b_j))
    inst = []
    for j in range(n):
        pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, [a[j], b[j], compute_hp(ctx, a[j], b[j])], mont=True)
        inst.append([h.np_to_point(c, pts[i], infs[i]) for i in range(3)])
    mu = [1, o.rng_scalar(700, 0) % (1 << 128)]
    nu1 = o.rng_scalar(701, 0) % (1 << 128)
    nu = [pow(nu1, k, c.r) for k in range(2 * n - 1)]
    t = compute_t_vecs(ctx, a, b, h.fr_mont_np(c, mu), ln, None, skip_uncommitted=True)
    low, high = compute_product_poly_comm(ck, t)
    low = [h.np_to_point(c, p, i) for p, i in low]
    high = [h.np_to_point(c, p, i) for p, i in high]
    chi = [mu[i] * nu[i] % c.r for i in range(n)]
    # verifier side (host, O(n) scalar-muls): combined commitments
    C1 = C2 = C3 = None
    for i in range(n):
        C1 = o.add(c, C1, o.mul(c, chi[i], inst[i][0]))
        C2 = o.add(c, C2, o.mul(c, nu[i], inst[n - 1 - i][1]))
    inner = None
    for i in range(n):
        inner = o.add(c, inner, o.mul(c, mu[i], inst[i][2]))
    for i in range(n - 1):
        C3 = o.add(c, C3, o.mul(c, nu[i], low[i]))
        C3 = o.add(c, C3, o.mul(c, nu[n + i], high[i]))
    C3 = o.add(c, C3, o.mul(c, nu[n - 1], inner))
    # prover side (device): combined openings
    a_new = combine_vectors(ctx, a, h.fr_mont_np(c, chi))
    b_new = combine_vectors(ctx, list(reversed(b)), h.fr_mont_np(c, nu[:n]))
    got = decide_commitments(ck, a_new, b_new)
    got = [h.np_to_point(c, p, i) for p, i in got]
    assert got[0] == C1 and got[1] == C2 and got[2] == C3


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_scalar_field_kernels_on_edge_values(ctxs, c):
    """The scalar-field kernels take Montgomery limbs as they are, so the limb patterns a carry chain cares about can be fed
    directly: 0, 1, r - 1, r - 2, all-ones words, single high bits, 2^k - 1, values just below the modulus in every limb.
    Every pair goes through the Hadamard product, a two-vector combination with edge coefficients, the inner product and
    compute_t_vecs, against Python integers (Montgomery product = a b R^-1 mod r)."""
    from accumulation_amd.hp_as import combine_vectors, compute_hp, compute_t_vecs
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.scalar_field import Fr
    ctx = ctxs[c.name]
    fr = Fr(ctx.curve)
    r = c.r
    R = 1 << 256
    Rinv = pow(R, -1, r)
    w32 = (1 << 32) - 1
    vals = [0, 1, 2, r - 1, r - 2, r - 3, (r - 1) // 2, (r + 1) // 2, R % r, (R * R) % r, (R - 1) % r]
    vals += [(1 << k) % r for k in (31, 32, 33, 63, 64, 65, 127, 128, 191, 192, 223, 224, 253, 254)]
    vals += [((1 << k) - 1) % r for k in (32, 64, 96, 128, 160, 192, 224, 254)]
    vals += [sum(w32 << (32 * j) for j in range(8) if (m >> j) & 1) % r for m in (0x55, 0xAA, 0x0F, 0xF0, 0x7F, 0x81)]
    vals += [r - (1 << k) for k in (1, 32, 64, 128, 200)]
    vals = sorted(set(v % r for v in vals))
    m = len(vals)
    A = [a for a in vals for _ in vals]          # Montgomery representatives, all pairs
    B = [b for _ in vals for b in vals]
    dA, dB = ctx.upload(h.scalars_to_np(A)), ctx.upload(h.scalars_to_np(B))  # limbs as given = Montgomery form
    mont_mul = lambda x, y: x * y * Rinv % r  # noqa: E731
    got = h.np_to_ints(compute_hp(ctx, dA, dB).download())
    assert got == [mont_mul(a, b) for a, b in zip(A, B)]
    for ca, cb in ((vals[3], vals[4]), (R % r, 1), (0, r - 1), (vals[-1], vals[len(vals) // 2])):
        coeffs = h.scalars_to_np([ca, cb])
        got = h.np_to_ints(combine_vectors(ctx, [dA, dB], coeffs).download())
        assert got == [(mont_mul(a, ca) + mont_mul(b, cb)) % r for a, b in zip(A, B)], (hex(ca), hex(cb))
    # a unit first coefficient keeps the combination on the 8 x 32 schedule, where two / three products share ONE Montgomery
    # reduction (fe_dot2 / fe_dot3, fp_mul_gfx950.h): sums just below 2 r and 3 r before the final subtractions
    one = R % r
    for cs in ((r - 1, r - 1), (r - 1, r - 2, r - 1), (vals[3], vals[-1], vals[-2]), (0, r - 1, 1), (r - 1, r - 1, r - 1, r - 1)):
        vs = [dA, dB, dA, dB, dB][:len(cs) + 1]
        hv = [A, B, A, B, B][:len(cs) + 1]
        got = h.np_to_ints(combine_vectors(ctx, vs, h.scalars_to_np([one] + list(cs))).download())
        exp = [(row[0] + sum(mont_mul(x, cf) for x, cf in zip(row[1:], cs))) % r for row in zip(*hv)]
        assert got == exp, [hex(x) for x in cs]
    ip = np.zeros(4, dtype=np.uint64)
    from accumulation_amd import ffi
    from accumulation_amd.engine import _ptr
    ffi.check(ctx._lib.amsm_vec_inner_product(ctx._h, dA.ptr, dB.ptr, len(A), _ptr(ip)), "inner_product")
    assert h.np_to_ints(ip.reshape(1, 4))[0] == sum(mont_mul(a, b) for a, b in zip(A, B)) % r
    # compute_t_vecs with two inputs: t_0 = a_0 o b_1-ish cross terms; compare with the Python restatement on Montgomery ints
    mu = [R % r, vals[4]]
    t = compute_t_vecs(ctx, [dA, dB], [dB, dA], h.scalars_to_np(mu), len(A))
    exp = o.compute_t_vecs(c, [[x * Rinv % r for x in A], [x * Rinv % r for x in B]],
                           [[x * Rinv % r for x in B], [x * Rinv % r for x in A]], [x * Rinv % r for x in mu], len(A), None)
    for k, tv in enumerate(t):
        assert h.np_to_ints(tv.download()) == [x * R % r for x in exp[k]], k
