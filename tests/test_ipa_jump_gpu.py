"""amsm_ipa_jump_fold (round 6): the key of an IPA opening folded by j challenges in ONE pass -- B_k = sum_t S_t G_(t m0 + k), the
reference's `key_l[i] += key_r[i] * xi` applied j times (ark_poly_commit::ipa_pc::open ext, under src/ipa_pc_as/mod.rs:454) --
against the big-integer oracle's naive sums and against the library's own physical folds; and whole openings with the jump (the last
rounds on the host over the jumped key) equal to openings without it, byte for byte.  test_host_ipa_jump_cpu.py re-collects the
module on the host backend (which takes any key and any m0)."""
import os

import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h

pytestmark = pytest.mark.gpu


def _jump(ctx, ck, log_key, xs, fr):
    from accumulation_amd import ffi
    from accumulation_amd.engine import _ptr
    j = len(xs)
    m0 = (1 << log_key) >> j
    xy = np.zeros((m0, 2 * ctx.fq_limbs), dtype=np.uint64)
    inf = np.zeros((m0,), dtype=np.uint8)
    xi = fr.to_limbs_many(xs)
    rc = ctx._lib.amsm_ipa_jump_fold(ctx._h, ck._h, log_key, _ptr(xi), j, _ptr(xy), _ptr(inf))
    return rc, xy, inf


@pytest.mark.parametrize("c", [o.PALLAS, o.BLS12_381_G1], ids=lambda c: c.name)
@pytest.mark.parametrize("log_key,j", [(7, 1), (10, 4), (12, 6)])
def test_jump_fold_vs_oracle_and_physical_folds(c, log_key, j):
    from accumulation_amd import CommitterKey, Context, ffi
    from accumulation_amd.scalar_field import Fr
    ctx = Context(c.curve_id)
    try:
        fr = Fr(ctx.curve)
        n = 1 << log_key
        ck = CommitterKey.generate(ctx, 0x1F0 + log_key, n, ffi.AMSM_BASES_PRECOMPUTE | ffi.AMSM_BASES_NO_DIRECT_TABLE)
        xs = [o.rng_scalar(0x1F1, r) % (1 << 128) or 1 for r in range(j)]
        rc, xy, inf = _jump(ctx, ck, log_key, xs, fr)
        assert rc == ffi.AMSM_OK
        m0 = n >> j
        # the library's own physical folds, one challenge at a time (k_points_fold: a different code path)
        key, half = ck, n // 2
        for x in xs:
            nxt = key.fold(half, fr.to_limbs(x), 128)
            if key is not ck:
                key.free()
            key, half = nxt, half // 2
        fxy, finf = key.read()
        key.free()
        assert np.array_equal(xy, fxy) and np.array_equal(inf, finf)
        # the oracle's naive sums for a few outputs: S_t = the product of the challenges the bits of t pick (the first one: the top bit)
        gxy, ginf = ck.read()
        gens = [h.np_to_point(c, gxy[i], ginf[i]) for i in range(n)]
        S = []
        for t in range(1 << j):
            s = 1
            for r in range(j):
                if (t >> (j - 1 - r)) & 1:
                    s = s * xs[r] % c.r
            S.append(s)
        for k in (0, 1, m0 // 2, m0 - 1):
            want = o.msm_naive(c, [gens[t * m0 + k] for t in range(1 << j)], S)
            assert h.np_to_point(c, xy[k], inf[k]) == want, k
        ck.free()
    finally:
        ctx.close()


def test_keys_that_do_not_qualify_are_refused_not_miscomputed():
    from accumulation_amd import CommitterKey, Context, ffi
    from accumulation_amd.scalar_field import Fr
    c = o.PALLAS
    ctx = Context(c.curve_id)
    try:
        if ctx.is_host:
            pytest.skip("the host backend takes any key")
        fr = Fr(ctx.curve)
        plain = CommitterKey.generate(ctx, 5, 1 << 8, ffi.AMSM_BASES_NO_PRECOMPUTE)
        assert _jump(ctx, plain, 8, [3, 5], fr)[0] == ffi.AMSM_E_UNSUPPORTED          # no window table
        ds = CommitterKey.generate(ctx, 5, 1 << 8, ffi.AMSM_BASES_PRECOMPUTE)
        assert ds.tables()["direct_sum_table"] > 0
        assert _jump(ctx, ds, 8, [3, 5], fr)[0] == ffi.AMSM_E_UNSUPPORTED             # its rounds are direct sums: faster than the host's
        ds.free()
        pre = CommitterKey.generate(ctx, 5, 1 << 8, ffi.AMSM_BASES_PRECOMPUTE | ffi.AMSM_BASES_NO_DIRECT_TABLE)
        assert _jump(ctx, pre, 8, [3, 5, 7], fr)[0] == ffi.AMSM_E_UNSUPPORTED         # m0 = 32: not a multiple of 64
        assert _jump(ctx, pre, 8, [3, 0], fr)[0] == ffi.AMSM_E_INVALID_ARG            # a zero challenge
        assert _jump(ctx, pre, 8, [3, 5], fr)[0] == ffi.AMSM_OK
        # edges of j: no challenge at all (the key itself, or a refusal -- never something else), more challenges than the key has bits
        rc, xy, inf = _jump(ctx, pre, 8, [], fr)
        assert rc in (ffi.AMSM_OK, ffi.AMSM_E_UNSUPPORTED, ffi.AMSM_E_INVALID_ARG)
        if rc == ffi.AMSM_OK:
            kxy, kinf = pre.read()
            assert np.array_equal(xy, kxy) and np.array_equal(inf, kinf)
        from accumulation_amd.engine import _ptr
        assert ctx._lib.amsm_ipa_jump_fold(ctx._h, pre._h, 8, _ptr(fr.to_limbs_many([3] * 9)), 9, _ptr(xy), _ptr(inf)) == ffi.AMSM_E_INVALID_ARG
        assert _jump(ctx, pre, 9, [3], fr)[0] == ffi.AMSM_E_INVALID_ARG  # log_key beyond the key
        plain.free()
        pre.free()
    finally:
        ctx.close()


@pytest.mark.parametrize("hiding", [False, True], ids=["no_zk", "zk"])
def test_opening_with_and_without_the_jump_same_proof(hiding, monkeypatch):
    """d + 1 = 2^9: three rounds on the device, then the jump to 64 generators and six rounds on the host -- against the same opening
    with AMSM_IPA_JUMP=0 (every round on the device, the final key from the check polynomial's MSM)"""
    from accumulation_amd import Context
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.scalar_field import Fr
    from tests.test_hp_as_scheme_gpu import SchemeRng
    c = o.PALLAS
    monkeypatch.setenv("AMSM_DIRECT_SUM_MAX_LOG2", "0")  # (keys with a direct-sum table keep their rounds on the device: none here)
    ctx = Context(c.curve_id)
    try:
        fr = Fr(ctx.curve)
        n = (1 << 9) if not ctx.is_host else (1 << 5)
        m = "64" if not ctx.is_host else "4"
        pp = IpaPC.setup(ctx, n - 1, seed=0x1F5)
        ck, vk = IpaPC.trim(pp, n - 1)
        poly = ctx.random_vector(0x1F6, n - 3, mont=True)
        point = o.rng_scalar(0x1F7, 0) % c.r
        proofs = []
        for jump in (m, "0"):
            monkeypatch.setenv("AMSM_IPA_JUMP", jump)
            rng = SchemeRng(0x1F8) if hiding else None
            comm, rand = IpaPC.commit(ck, poly, hiding, rng)
            proofs.append((comm, IpaPC.open(ck, poly, comm, point, rand, hiding, rng)))
        (c0, p0), (c1, p1) = proofs
        if not ctx.is_host:  # the jump really ran: the key qualifies
            assert _jump(ctx, ck.comm_key, 9, [3, 5, 7], fr)[0] == 0

        def pts(v):
            return [(np.asarray(a).tolist(), bool(b)) for a, b in v]
        assert pts(p0.l_vec) == pts(p1.l_vec) and pts(p0.r_vec) == pts(p1.r_vec) and len(p0.l_vec) == n.bit_length() - 1
        assert pts([p0.final_comm_key]) == pts([p1.final_comm_key]) and p0.c == p1.c and p0.rand == p1.rand
        coeffs = [fr.from_limbs(v) for v in poly.download()]
        value = sum(cf * pow(point, i, c.r) for i, cf in enumerate(coeffs)) % c.r
        assert IpaPC.check(vk, c0, point, value, p0)
    finally:
        ctx.close()
