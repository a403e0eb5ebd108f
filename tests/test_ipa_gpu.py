"""GPU tests for the IPA polynomial-commitment path (ark_poly_commit::ipa_pc restated in
accumulation_amd/ipa_pc.py) and the reference's accumulation-scheme test template for
AtomicASForInnerProductArgPC (src/ipa_pc_as/mod.rs:890-1111: degree 11, zk and no-zk, six scenarios)."""
import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h
from tests.test_hp_as_scheme_gpu import SchemeRng

pytestmark = pytest.mark.gpu

NUM_ITERATIONS = 3
DEGREE = 11  # src/ipa_pc_as/mod.rs:1017 ff. (trimmed up to 15 = next power of two - 1)


@pytest.fixture(scope="module")
def env():
    from accumulation_amd import Context, ffi
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    ctx = Context(ffi.AMSM_PALLAS)
    pp = IpaPC.setup(ctx, DEGREE, seed=0xABCDEF)
    yield ctx, pp
    ctx.close()


def test_ipa_kernels_vs_oracle(env):
    """points_fold, inner product, powers and compute_coeffs against the Python big-int oracle."""
    import ctypes as C
    from accumulation_amd import PointVector, ffi
    from accumulation_amd.engine import _ptr
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC, SuccinctCheckPolynomial
    from accumulation_amd.scalar_field import Fr
    ctx, pp = env
    c = o.PALLAS
    fr = Fr(ctx.curve)
    n = 37
    a, b = o.rng_scalars(1, n), o.rng_scalars(2, n)
    da, db = ctx.upload(h.fr_mont_np(c, a)), ctx.upload(h.fr_mont_np(c, b))
    assert IpaPC._inner_product(ctx, fr, da, db) == sum(x * y for x, y in zip(a, b)) % c.r
    pt = o.rng_scalar(3, 0) % c.r
    z = ctx.vector(n)
    ffi.check(ctx._lib.amsm_vec_powers(ctx._h, _ptr(fr.to_limbs(pt)), n, z.ptr), "powers")
    assert h.fr_from_mont_np(c, z.download()) == [pow(pt, i, c.r) for i in range(n)]
    xi = [o.rng_scalar(4, i) % (1 << 128) for i in range(4)]
    got = h.fr_from_mont_np(c, SuccinctCheckPolynomial(xi).compute_coeffs(ctx).download())
    exp = []
    for p in range(16):
        v = 1
        for i in range(1, 5):
            if (p >> (4 - i)) & 1:
                v = v * xi[i - 1] % c.r
        exp.append(v)
    assert got == exp
    # h(point) evaluated two ways
    hv = sum(cf * pow(pt, i, c.r) for i, cf in enumerate(exp)) % c.r
    assert SuccinctCheckPolynomial(xi).evaluate(fr, pt) == hv
    # key fold l + x*r on points
    pts = o.rng_points(c, 77, 12)
    xy, _ = h.points_to_np(c, pts)
    pv = PointVector(ctx, 12)
    ffi.check(ctx._lib.amsm_dev_upload(ctx._h, pv.ptr, _ptr(xy), xy.nbytes), "upload")
    out = PointVector(ctx, 6)
    x = xi[0]
    ffi.check(ctx._lib.amsm_points_fold(ctx._h, pv.view(0, 6).ptr, pv.view(6, 6).ptr, 6, _ptr(fr.to_limbs(x)), 128,
                                        out.ptr), "fold")
    got = out.download()
    for i in range(6):
        assert h.np_to_point(c, got[i], 0) == o.add(c, pts[i], o.mul(c, x, pts[6 + i]))


@pytest.mark.parametrize("flags", [1, 2], ids=["precomputed", "plain"])
def test_ipa_rounds_without_key_folding_vs_oracle(env, flags):
    """The opening's rounds computed two ways must give the same points: the reference's way -- fold the key every
    round (here with the big-int oracle, and on the device with amsm_bases_fold) -- and the product's way --
    amsm_ipa_round_scalars over the original key + amsm_msm_multi_device, final key from the check polynomial."""
    from accumulation_amd import CommitterKey, VariableBaseMSM, ffi
    from accumulation_amd.engine import _ptr
    from accumulation_amd.ipa_pc import SuccinctCheckPolynomial
    from accumulation_amd.scalar_field import Fr
    ctx, _ = env
    c = o.PALLAS
    fr = Fr(ctx.curve)
    log_n, n = 4, 16
    pts = o.rng_points(c, 91, n)
    xy, _ = h.points_to_np(c, pts)
    key = CommitterKey.load(ctx, xy, None, flags)
    coeffs = [o.rng_scalar(92, i) % c.r for i in range(n)]
    xs = [o.rng_scalar(93, i) % (1 << 128) for i in range(log_n)]
    d_c = ctx.upload(fr.to_limbs_many(coeffs))
    u_l, u_r = ctx.vector(n), ctx.vector(n)
    g, cur, dev_key, owned = list(pts), list(coeffs), key, False
    for j in range(log_n):
        half = len(cur) // 2
        exp_l = o.msm_naive(c, g[:half], cur[half:])   # <c_r, key_l>
        exp_r = o.msm_naive(c, g[half:], cur[:half])   # <c_l, key_r>
        xi = fr.to_limbs_many(xs[:j]) if j else None
        ffi.check(ctx._lib.amsm_ipa_round_scalars(ctx._h, _ptr(xi), j, log_n, d_c.ptr, u_l.ptr, u_r.ptr), "round scalars")
        out, inf = VariableBaseMSM.multi_scalar_mul_multi(key, [(0, u_l), (0, u_r)], mont=True)
        assert h.np_to_point(c, out[0], inf[0]) == exp_l and h.np_to_point(c, out[1], inf[1]) == exp_r, j
        # ... and as ONE vector summed per index class in one pass
        ffi.check(ctx._lib.amsm_ipa_round_scalars(ctx._h, _ptr(xi), j, log_n, d_c.ptr, u_l.ptr, None), "round scalars (1 vector)")
        out1, inf1 = VariableBaseMSM.multi_scalar_mul_grouped(key, u_l, log_n - 1 - j, mont=True)
        assert h.np_to_point(c, out1[0], inf1[0]) == exp_l and h.np_to_point(c, out1[1], inf1[1]) == exp_r, j
        # the same two commitments over the device-folded key (windows of one key: base_off 0 / half)
        cv = ctx.upload(fr.to_limbs_many(cur))
        out2, inf2 = VariableBaseMSM.multi_scalar_mul_multi(dev_key, [(0, cv.view(half, half)), (half, cv.view(0, half))],
                                                            mont=True)
        assert h.np_to_point(c, out2[0], inf2[0]) == exp_l and h.np_to_point(c, out2[1], inf2[1]) == exp_r, j
        # fold: key_l += x * key_r, c_l += x^-1 * c_r
        x = xs[j]
        g = [o.add(c, g[i], o.mul(c, x, g[half + i])) for i in range(half)]
        inv = pow(x, -1, c.r)
        cur = [(cur[i] + inv * cur[half + i]) % c.r for i in range(half)]
        d_c = ctx.upload(fr.to_limbs_many(cur))
        nk = dev_key.fold(half, fr.to_limbs(x), 128)
        if owned:
            dev_key.free()
        dev_key, owned = nk, True
        got, ginf = dev_key.read(0, half)
        assert [h.np_to_point(c, got[i], ginf[i]) for i in range(half)] == g, j
    s_vec = SuccinctCheckPolynomial(xs).compute_coeffs(ctx)
    fk, finf = VariableBaseMSM.multi_scalar_mul(key, s_vec, mont=True)
    assert h.np_to_point(c, fk, finf) == g[0]
    dev_key.free()
    key.free()


@pytest.mark.parametrize("hiding", [False, True], ids=["plain", "hiding"])
def test_ipa_commit_open_check(env, hiding):
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.scalar_field import Fr
    ctx, pp = env
    c = o.PALLAS
    fr = Fr(ctx.curve)
    ck, vk = IpaPC.trim(pp, DEGREE)
    assert ck.supported_degree() == 15
    rng = SchemeRng(99)
    coeffs = [rng.field() % c.r for _ in range(DEGREE + 1)]
    poly = ctx.upload(fr.to_limbs_many(coeffs))
    comm, rand = IpaPC.commit(ck, poly, hiding, rng)
    # the commitment is the MSM the oracle computes
    xy, _ = ck.comm_key.read()
    gens = [h.np_to_point(c, xy[i], 0) for i in range(16)]
    exp = o.msm_naive(c, gens[:12], coeffs)
    if hiding:
        exp = o.add(c, exp, o.mul(c, rand, h.np_to_point(c, ck.s[0], 0)))
    assert h.np_to_point(c, *comm.comm) == exp
    point = rng.field() % c.r
    value = sum(cf * pow(point, i, c.r) for i, cf in enumerate(coeffs)) % c.r
    proof = IpaPC.open(ck, poly, comm, point, rand, hiding, rng)
    assert len(proof.l_vec) == 4
    assert IpaPC.check(vk, comm, point, value, proof)
    assert not IpaPC.check(vk, comm, point, (value + 1) % c.r, proof)
    assert not IpaPC.check(vk, comm, (point + 1) % c.r, value, proof)


def generate_inputs(env, pk, num_inputs, make_zk, rng, degree=None):
    """src/ipa_pc_as/mod.rs:930-1004: random polynomial, commit, random point, evaluate, open"""
    degree = DEGREE if degree is None else degree
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.ipa_pc_as import InputInstance
    from accumulation_amd.scalar_field import Fr
    ctx, pp = env
    fr = Fr(ctx.curve)
    out = []
    for _ in range(num_inputs):
        coeffs = [rng.field() % fr.r for _ in range(degree + 1)]
        poly = ctx.upload(fr.to_limbs_many(coeffs))
        comm, rand = IpaPC.commit(pk.ipa_ck, poly, make_zk, rng)
        point = rng.field() % fr.r
        value = sum(cf * pow(point, i, fr.r) for i, cf in enumerate(coeffs)) % fr.r
        proof = IpaPC.open(pk.ipa_ck, poly, comm, point, rand, make_zk, rng)
        out.append(InputInstance(comm, point, value, proof))
    return out


def run_template(env, num_inputs_per_iteration, make_zk, num_iterations=NUM_ITERATIONS):
    from accumulation_amd.ipa_pc_as import AtomicASForInnerProductArgPC as AS
    ctx, pp = env
    pk, vk, dk = AS.index(pp, DEGREE)
    rng = SchemeRng(4096)
    total = num_iterations * sum(num_inputs_per_iteration)
    inputs = generate_inputs(env, pk, total, make_zk, rng)
    start = 0
    for _ in range(num_iterations):
        old = []
        for k in num_inputs_per_iteration:
            step = inputs[start:start + k]
            start += k
            acc, proof = AS.prove(pk, step, [a.instance for a in old], rng if make_zk else None, None)
            assert AS.verify(ctx, vk, step, [a.instance for a in old], acc.instance, proof, None), "Verify failed"
            old.append(acc)
        assert AS.decide(dk, old[-1], None), "Decide failed"
    return True


@pytest.mark.parametrize("hiding", [False, True], ids=["plain", "hiding"])
def test_open_does_not_depend_on_key_folding(monkeypatch, hiding):
    """Large openings fold the key physically in their first rounds (ipa_pc.IPA_FOLD); the proof must not depend on how
    many rounds do: every split between folded rounds and rounds expressed over the last folded key gives the same
    points, the same final key and the same c (forced here at d + 1 = 64 through AMSM_IPA_FOLD_ABOVE)."""
    from accumulation_amd import Context, ffi
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.scalar_field import Fr
    for curve, c in ((ffi.AMSM_PALLAS, o.PALLAS), (ffi.AMSM_BLS12_381_G1, o.BLS12_381_G1)):
        ctx = Context(curve)
        fr = Fr(ctx.curve)
        pp = IpaPC.setup(ctx, 63, seed=0xF01D)
        ck, vk = IpaPC.trim(pp, 63)
        seen = []
        for t in (99, 5, 4, 2, 1):  # never fold, 1 fold, 2 folds, 4 folds, 5 folds (of 6 rounds)
            monkeypatch.setenv("AMSM_IPA_FOLD_ABOVE", str(t))
            assert IpaPC._fold_rounds(ctx, 6) == max(0, 6 - t)
            rng = SchemeRng(31)
            coeffs = [rng.field() % c.r for _ in range(50)]
            poly = ctx.upload(fr.to_limbs_many(coeffs))
            comm, rand = IpaPC.commit(ck, poly, hiding, rng)
            point = rng.field() % c.r
            value = sum(cf * pow(point, i, c.r) for i, cf in enumerate(coeffs)) % c.r
            proof = IpaPC.open(ck, poly, comm, point, rand, hiding, rng)
            assert IpaPC.check(vk, comm, point, value, proof)
            seen.append(([h.np_to_point(c, *p) for p in proof.l_vec + proof.r_vec + [proof.final_comm_key]], proof.c, proof.rand))
        assert all(x == seen[0] for x in seen[1:])
        monkeypatch.delenv("AMSM_IPA_FOLD_ABOVE")
        ctx.close()


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_ipa_pc_as_bls12_381(make_zk):
    """BASELINE config 2's curve (384-bit base field, 255-bit scalars): the commitment against the oracle, open / check and
    the accumulation template's busiest scenario."""
    from accumulation_amd import Context, ffi
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.scalar_field import Fr
    c = o.BLS12_381_G1
    ctx = Context(ffi.AMSM_BLS12_381_G1)
    fr = Fr(ctx.curve)
    pp = IpaPC.setup(ctx, DEGREE, seed=0xB15)
    ck, vk = IpaPC.trim(pp, DEGREE)
    rng = SchemeRng(1234)
    coeffs = [rng.field() % c.r for _ in range(DEGREE + 1)]
    poly = ctx.upload(fr.to_limbs_many(coeffs))
    comm, rand = IpaPC.commit(ck, poly, make_zk, rng)
    xy, _ = ck.comm_key.read()
    gens = [h.np_to_point(c, xy[i], 0) for i in range(16)]
    exp = o.msm_naive(c, gens[:12], coeffs)
    if make_zk:
        exp = o.add(c, exp, o.mul(c, rand, h.np_to_point(c, ck.s[0], 0)))
    assert h.np_to_point(c, *comm.comm) == exp
    point = rng.field() % c.r
    value = sum(cf * pow(point, i, c.r) for i, cf in enumerate(coeffs)) % c.r
    proof = IpaPC.open(ck, poly, comm, point, rand, make_zk, rng)
    assert IpaPC.check(vk, comm, point, value, proof)
    assert not IpaPC.check(vk, comm, point, (value + 1) % c.r, proof)
    assert run_template((ctx, pp), [1, 1, 2, 3], make_zk, num_iterations=1)
    ctx.close()


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
class TestASForIpaPC:
    def test_single_input_init(self, env, make_zk):
        assert run_template(env, [1], make_zk)

    def test_multiple_inputs_init(self, env, make_zk):
        assert run_template(env, [3], make_zk)

    def test_simple_accumulation(self, env, make_zk):
        assert run_template(env, [1, 1], make_zk)

    def test_multiple_inputs_accumulation(self, env, make_zk):
        assert run_template(env, [1, 1, 2, 3], make_zk, num_iterations=2)

    def test_accumulators_only(self, env, make_zk):
        assert run_template(env, [1, 0, 0, 0], make_zk)

    def test_no_inputs_init(self, env, make_zk):
        assert run_template(env, [0], make_zk, num_iterations=1)


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_simple_accumulation_reference_iteration_count(env, make_zk):
    """the reference runs every scenario NUM_ITERATIONS = 50 times (src/lib.rs:273); one scenario at that count"""
    assert run_template(env, [1, 1], make_zk, num_iterations=50)


def test_sponge_argument_is_refused(env):
    from accumulation_amd.ipa_pc_as import AtomicASForInnerProductArgPC as AS
    from accumulation_amd.sponge import Sha256Sponge
    ctx, pp = env
    pk, vk, dk = AS.index(pp, DEGREE)
    with pytest.raises(NotImplementedError):  # src/ipa_pc_as/mod.rs:566-570
        AS.prove(pk, [], [], None, Sha256Sponge())


def test_ipa_round_entry_point_equals_its_parts(env):
    """amsm_ipa_round (scalar expansion + grouped MSM + both inner products, one synchronisation) against the separate
    entry points it fuses, which are themselves pinned to the oracle above: rounds 0, 1 and 3 of a 64-point key."""
    from accumulation_amd import CommitterKey, VariableBaseMSM, ffi
    from accumulation_amd.engine import _ptr
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.scalar_field import Fr
    ctx, _ = env
    c = o.PALLAS
    fr = Fr(ctx.curve)
    log_key = 6
    key = CommitterKey.load(ctx, h.points_to_np(c, o.rng_points(c, 0x1F, 64))[0], None, ffi.AMSM_BASES_DEFAULT)
    xs_all = [o.rng_scalar(0x2F, i) % (1 << 128) for i in range(3)]
    for j in (0, 1, 3):
        m = 1 << (log_key - j)  # current length of the coefficient / evaluation vectors
        half = m // 2
        coeffs = ctx.upload(h.fr_mont_np(c, [o.rng_scalar(0x3F + j, i) % c.r for i in range(m)]))
        z = ctx.upload(h.fr_mont_np(c, [o.rng_scalar(0x4F + j, i) % c.r for i in range(m)]))
        xi = fr.to_limbs_many(xs_all[:j]) if j else None
        u = ctx.vector(64)
        xy = np.zeros((2, 2 * ctx.fq_limbs), dtype=np.uint64)
        inf = np.zeros((2,), dtype=np.uint8)
        ips = np.zeros((2, 4), dtype=np.uint64)
        ffi.check(ctx._lib.amsm_ipa_round(ctx._h, key._h, _ptr(xi), j, log_key, coeffs.ptr, z.ptr, u.ptr, _ptr(xy), _ptr(inf),
                                          _ptr(ips)), "amsm_ipa_round")
        u2 = ctx.vector(64)
        ffi.check(ctx._lib.amsm_ipa_round_scalars(ctx._h, _ptr(xi), j, log_key, coeffs.ptr, u2.ptr, None), "round_scalars")
        exy, einf = VariableBaseMSM.multi_scalar_mul_grouped(key, u2, log_key - 1 - j, mont=True)
        assert np.array_equal(xy, exy) and np.array_equal(inf, einf), j
        assert fr.from_limbs(ips[0]) == IpaPC._inner_product(ctx, fr, coeffs.view(half, half), z.view(0, half)), j
        assert fr.from_limbs(ips[1]) == IpaPC._inner_product(ctx, fr, coeffs.view(0, half), z.view(half, half)), j
    assert ctx._lib.amsm_ipa_round(ctx._h, key._h, None, 0, 7, coeffs.ptr, z.ptr, u.ptr, _ptr(xy), _ptr(inf), _ptr(ips)) == \
        ffi.AMSM_E_INVALID_ARG  # 2^7 generators asked of a 64-point key
    key.free()


def test_ipa_round_fused_equals_round_plus_host_algebra(env):
    """amsm_ipa_round_fused (previous fold in place + round + h' terms, one normalisation) against amsm_ipa_round followed
    by the driver's own algebra (amsm_vec_combine folds, amsm_host_lincomb for the h' terms): rounds 1 and 3 of a
    64-point key, with and without the fold / the h' term."""
    from accumulation_amd import CommitterKey, ffi
    from accumulation_amd.engine import _ptr
    from accumulation_amd.hp_as import combine_vectors
    from accumulation_amd.ipa_pc import _lincomb
    from accumulation_amd.scalar_field import Fr
    ctx, _ = env
    c = o.PALLAS
    fr = Fr(ctx.curve)
    log_key = 6
    key = CommitterKey.load(ctx, h.points_to_np(c, o.rng_points(c, 0x1F, 64))[0], None, ffi.AMSM_BASES_DEFAULT)
    hp_xy, hp_inf = h.points_to_np(c, o.rng_points(c, 0x5F, 1))
    h_prime = (hp_xy[0], False)
    xs_all = [o.rng_scalar(0x2F, i) % (1 << 128) for i in range(4)]
    one = fr.to_limbs(1)
    for j in (1, 3):
        for with_fold in (False, True):
            for with_h in (False, True):
                m = 1 << (log_key - j)  # length of the round's vectors
                pre = 2 * m if with_fold else m
                cv = [o.rng_scalar(0x6F + j, i) % c.r for i in range(pre)]
                zv = [o.rng_scalar(0x7F + j, i) % c.r for i in range(pre)]
                x = xs_all[j - 1]
                xi = fr.to_limbs_many(xs_all[:j])
                # expected: fold with the vector kernels, plain round, host algebra
                coeffs, z = ctx.upload(h.fr_mont_np(c, cv)), ctx.upload(h.fr_mont_np(c, zv))
                if with_fold:
                    coeffs = combine_vectors(ctx, [coeffs.view(0, m), coeffs.view(m, m)],
                                             np.stack([one, fr.to_limbs(pow(x, -1, c.r))]))
                    z = combine_vectors(ctx, [z.view(0, m), z.view(m, m)], np.stack([one, fr.to_limbs(x)]))
                u = ctx.vector(64)
                exy = np.zeros((2, 2 * ctx.fq_limbs), dtype=np.uint64)
                einf = np.zeros((2,), dtype=np.uint8)
                eips = np.zeros((2, 4), dtype=np.uint64)
                ffi.check(ctx._lib.amsm_ipa_round(ctx._h, key._h, _ptr(xi), j, log_key, coeffs.ptr, z.ptr, u.ptr, _ptr(exy),
                                                  _ptr(einf), _ptr(eips)), "amsm_ipa_round")
                expect = [(exy[g].copy(), bool(einf[g])) for g in range(2)]
                if with_h:
                    expect = [_lincomb(ctx, [expect[g], h_prime], [1, fr.from_limbs(eips[g])], fr) for g in range(2)]
                # fused
                coeffs2, z2 = ctx.upload(h.fr_mont_np(c, cv)), ctx.upload(h.fr_mont_np(c, zv))
                xy = np.zeros((2, 2 * ctx.fq_limbs), dtype=np.uint64)
                inf = np.zeros((2,), dtype=np.uint8)
                ips = np.zeros((2, 4), dtype=np.uint64)
                fx = fr.to_limbs(x) if with_fold else None
                hx = np.ascontiguousarray(h_prime[0]) if with_h else None
                ffi.check(ctx._lib.amsm_ipa_round_fused(ctx._h, key._h, _ptr(xi), j, log_key, coeffs2.ptr, z2.ptr, _ptr(fx),
                                                        _ptr(hx), u.ptr, _ptr(xy), _ptr(inf), _ptr(ips)), "amsm_ipa_round_fused")
                tag = (j, with_fold, with_h)
                assert np.array_equal(ips, eips), tag
                for g in range(2):
                    assert np.array_equal(xy[g], expect[g][0]) and bool(inf[g]) == expect[g][1], tag
                # the buffers now hold the folded vectors in their first m elements, the rest untouched
                assert np.array_equal(coeffs2.view(0, m).download(), coeffs.view(0, m).download()), tag
                assert np.array_equal(z2.view(0, m).download(), z.view(0, m).download()), tag
                if with_fold:
                    assert np.array_equal(coeffs2.view(m, m).download(), h.fr_mont_np(c, cv[m:])), tag
    zero = np.zeros(4, dtype=np.uint64)
    assert ctx._lib.amsm_ipa_round_fused(ctx._h, key._h, _ptr(xi), 3, log_key, coeffs2.ptr, z2.ptr, _ptr(zero), None, u.ptr,
                                         _ptr(xy), _ptr(inf), _ptr(ips)) == ffi.AMSM_E_INVALID_ARG  # x = 0 has no inverse
    key.free()
