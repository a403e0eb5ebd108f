"""Host-side mirror of the inner-product-argument polynomial commitment used by `ipa_pc_as`
(`ark_poly_commit::ipa_pc::InnerProductArgPC`, branch accumulation-experimental -- NOT in the reference tree and
not pinned, Cargo.toml:34).  Restated from the published construction (BCMS20 section 7 / Halo-style IPA) with
the interface the reference calls (src/ipa_pc_as/mod.rs:155,198,400,418,454,525,836):

  commit / cm_commit           one MSM over comm_key[..len]  (+ randomizer * s)
  open                         log2(d+1) rounds: two half-size MSMs, two inner products, folds of the coefficient
                               vector, the evaluation vector and the commitment key (K8 of SURVEY.md section 2.1)
  succinct_check               O(log d) scalar-muls on the host, returns the SuccinctCheckPolynomial
  check                        succinct_check + compute_coeffs (2^k expansion) + ONE (d+1)-point MSM
  SuccinctCheckPolynomial      compute_coeffs (device) / evaluate (host, k multiplications)

Every O(d) loop and every MSM runs on the GPU through the C ABI; challenges come from a caller-supplied sponge
class (see sponge.py).  Self-consistent (prover <-> verifier); byte-level parity with ark-poly-commit is not
checkable here (PARITY UNPINNED).  Single polynomial per call, no degree bounds -- all the
accumulation scheme uses."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import ffi
from .engine import CommitterKey as EngineKey, Context, FrVector, PointVector, VariableBaseMSM, _ptr
from .hp_as import ASForHadamardProducts, _pt_eq, combine_vectors
from .scalar_field import Fr
from .sponge import CryptographicSponge, Sha256Sponge

CHALLENGE_SIZE = 128
# (smallest log2(d+1) whose opening folds at all, fold while the key has more than 2^T generators), measured on MI355X:
# Pallas 2^20 42 -> 34 ms, 2^18 19.0 -> 18.4, 2^16 better without; BLS12-381 2^20 89 -> 70 ms, 2^16 25 -> 23
IPA_FOLD = {ffi.AMSM_PALLAS: (18, 15), ffi.AMSM_BLS12_381_G1: (16, 15)}
_lincomb = ASForHadamardProducts._lincomb


def _zero_pt(ctx):
    return (np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64), True)


@dataclass
class CommitterKey:  # ipa_pc::CommitterKey{comm_key, h, s, max_degree}; VerifierKey is the same type
    comm_key: EngineKey
    h: tuple
    s: tuple
    max_degree: int

    def supported_degree(self) -> int:
        return len(self.comm_key) - 1

    @property
    def svk(self):
        return SuccinctVerifierKey(self.h, self.s, self.supported_degree())


@dataclass
class SuccinctVerifierKey:
    h: tuple
    s: tuple
    _supported_degree: int

    def supported_degree(self) -> int:
        return self._supported_degree


@dataclass
class Commitment:
    comm: tuple
    shifted_comm: Optional[tuple] = None

    @staticmethod
    def default(ctx):
        return Commitment(_zero_pt(ctx), None)


@dataclass
class Proof:
    l_vec: List[tuple]
    r_vec: List[tuple]
    final_comm_key: tuple
    c: int
    hiding_comm: Optional[tuple] = None
    rand: Optional[int] = None


class SuccinctCheckPolynomial:
    def __init__(self, challenges: List[int]):
        self.challenges = list(challenges)

    def compute_coeffs(self, ctx: Context) -> FrVector:
        fr = Fr(ctx.curve)
        k = len(self.challenges)
        out = ctx.vector(1 << k)
        xi = fr.to_limbs_many(self.challenges)
        ffi.check(ctx._lib.amsm_ipa_check_poly_coeffs(ctx._h, _ptr(xi) if k else None, k, out.ptr),
                  "amsm_ipa_check_poly_coeffs")
        return out

    def evaluate(self, fr: Fr, point: int) -> int:
        k = len(self.challenges)
        prod = 1
        for i, ch in enumerate(self.challenges, start=1):
            elem = pow(point, 1 << (k - i), fr.r)
            prod = prod * (1 + elem * ch) % fr.r
        return prod

    def to_bytes(self, fr: Fr) -> bytes:
        return b"".join((c % fr.r).to_bytes(32, "little") for c in self.challenges)


class InnerProductArgPC:
    sponge_cls = Sha256Sponge  # S of the reference's generic parameter; any CryptographicSponge subclass

    # ---- keys -------------------------------------------------------------------------------------
    @classmethod
    def setup(cls, ctx: Context, max_degree: int, seed: int = 0x1BA5EED) -> CommitterKey:
        """UniversalParams: max_degree+1 generators + h + s (synthetic stream; ark-poly-commit hashes to the curve)."""
        n = 1 << (max_degree + 1 - 1).bit_length()  # (max_degree + 1).next_power_of_two()
        tmp = EngineKey.generate(ctx, seed, n + 2, ffi.AMSM_BASES_NO_PRECOMPUTE)
        xy, _ = tmp.read()
        tmp.free()
        key = EngineKey.load(ctx, xy[:n], None, ffi.AMSM_BASES_DEFAULT)
        cls._pp_xy = xy
        return CommitterKey(key, (xy[n].copy(), False), (xy[n + 1].copy(), False), n - 1)

    @classmethod
    def trim(cls, pp: CommitterKey, supported_degree: int):
        """-> (ck, vk) with (supported_degree+1).next_power_of_two() generators (prefix of the universal key)."""
        ctx = pp.comm_key.ctx
        n = 1 << (supported_degree + 1 - 1).bit_length()
        assert n <= len(pp.comm_key)
        if n == len(pp.comm_key):
            ck = pp
        else:
            xy, _ = pp.comm_key.read(0, n)
            ck = CommitterKey(EngineKey.load(ctx, xy, None, ffi.AMSM_BASES_DEFAULT), pp.h, pp.s, pp.max_degree)
        return ck, ck

    # ---- commit -----------------------------------------------------------------------------------
    @staticmethod
    def cm_commit(key: EngineKey, scalars: FrVector, hiding_generator=None, randomizer: Optional[int] = None):
        """msm(key[..len], scalars) (+ randomizer * hiding_generator) -> affine"""
        ctx = key.ctx
        out = VariableBaseMSM.multi_scalar_mul(key, scalars, mont=True)
        if randomizer is not None:
            fr = Fr(ctx.curve)
            out = _lincomb(ctx, [out, hiding_generator], [1, randomizer], fr)
        return out

    @classmethod
    def commit(cls, ck: CommitterKey, polynomial: FrVector, hiding: bool = False, rng=None):
        """-> (Commitment, rand).  hiding <=> LabeledPolynomial::hiding_bound().is_some()"""
        rand = rng.field() if hiding else 0
        comm = cls.cm_commit(ck.comm_key, polynomial, ck.s, rand if hiding else None)
        return Commitment(comm, None), rand

    # ---- challenges ---------------------------------------------------------------------------------
    @classmethod
    def _challenge(cls, fr: Fr, parts) -> int:
        sp = cls.sponge_cls().fork(b"IPA-PC-2020")  # `IpaPCDomain::domain()`, src/ipa_pc_as/data_structures.rs:88-94
        for p in parts:
            if isinstance(p, tuple):
                sp.absorb_point(p)
            elif isinstance(p, bytes):
                sp.absorb_bytes(p)
            else:
                sp.absorb_bytes((int(p) % fr.r).to_bytes(32, "little"))
        return sp.squeeze_field_elements(1, CHALLENGE_SIZE)[0]

    # ---- open ---------------------------------------------------------------------------------------
    @classmethod
    def open(cls, ck: CommitterKey, polynomial: FrVector, commitment: Commitment, point: int, rand: int = 0,
             hiding: bool = False, rng=None) -> Proof:
        """open_individual_opening_challenges for ONE polynomial with opening challenge 1."""
        ctx = ck.comm_key.ctx
        fr = Fr(ctx.curve)
        d = ck.supported_degree()
        n = d + 1
        assert polynomial.n <= n
        one = fr.to_limbs(1)
        z = ctx.vector(n)
        ffi.check(ctx._lib.amsm_vec_powers(ctx._h, _ptr(fr.to_limbs(point)), n, z.ptr), "amsm_vec_powers")
        # coefficient vector padded to d+1
        coeffs = combine_vectors(ctx, [polynomial], one.reshape(1, 4), ctx.fill(fr.to_limbs(0), n))
        combined_comm = commitment.comm
        combined_v = cls._inner_product(ctx, fr, coeffs, z)
        hiding_comm = None
        proof_rand = None
        if hiding:
            assert rng is not None
            # random polynomial that vanishes at `point`
            hp = ctx.upload(fr.to_limbs_many([rng.field() for _ in range(n)]))
            hv = cls._inner_product(ctx, fr, hp, z)
            hp = combine_vectors(ctx, [hp, ctx.upload(fr.to_limbs_many([(-hv) % fr.r]))], np.stack([one, one]))
            hiding_rand = rng.field()
            hiding_comm = cls.cm_commit(ck.comm_key, hp, ck.s, hiding_rand)
            hch = cls._challenge(fr, [combined_comm, point, combined_v, hiding_comm])
            coeffs = combine_vectors(ctx, [coeffs, hp], np.stack([one, fr.to_limbs(hch)]))
            proof_rand = (rand + hch * hiding_rand) % fr.r
            combined_comm = _lincomb(ctx, [combined_comm, hiding_comm, ck.s], [1, hch, (-proof_rand) % fr.r], fr)
        round_challenge = cls._challenge(fr, [combined_comm, point, combined_v])
        h_prime = _lincomb(ctx, [ck.h], [round_challenge], fr)
        # Small and medium openings never fold the key.  Round j's cross commitments L_j = <c_r, key_l>, R_j = <c_l, key_r> are
        # expressed over the ORIGINAL (precomputed, HBM-resident) key: amsm_ipa_round_scalars expands the current
        # coefficients by the products of the previous challenges into ONE vector, and amsm_msm_grouped_device sums
        # its two index classes (the bit that separates key_l from key_r in round j) in one pass.  Folding instead costs n / 2^j 128-bit scalar multiplications
        # with an inversion each per round -- a ~0.8 ms dependency chain per round whatever the size (measured:
        # 13 of 30 ms at d + 1 = 2^16).  The final folded key is one more MSM with the check polynomial's
        # coefficients.  Points are identical to the reference's (ext, under src/ipa_pc_as/mod.rs:454).
        # Large openings fold the key physically in their first rounds (the reference's own `key_l += key_r * xi`,
        # amsm_bases_fold): a fold is n/2 128-bit scalar multiplications, ~9 MSM-rounds' worth of work, but it halves
        # every later round, so it pays while many rounds remain and the halves are large enough to run at throughput
        # (measured thresholds in _fold_rounds).  The remaining rounds use the expansion over the last folded key.
        key = ck.comm_key
        log_n = n.bit_length() - 1
        assert n == 1 << log_n
        # (a key sharded over several devices is never folded: a fold pairs generator i with i + n/2, which live on different devices)
        fold_rounds = 0 if key.num_shards > 1 else cls._fold_rounds(ctx, log_n)
        u_l = ctx.vector(n)
        xs: List[int] = []
        l_vec, r_vec = [], []
        cur_key, log_key = key, log_n  # the key the current round's scalars are expressed over
        # One library call per round (amsm_ipa_round_fused): the previous round's fold of c and z (in place), the scalar
        # expansion, the grouped MSM, both inner products, their h' multiples and one normalisation of L and R.
        hp_xy = None if h_prime[1] else np.ascontiguousarray(h_prime[0], dtype=np.uint64)
        prev_x = None
        jumped = None
        while n > 1:
            half = n // 2
            j = len(xs) - (log_n - log_key)  # challenges since cur_key was formed
            # Round 6 -- the JUMP FOLD: once the vectors are down to JUMP_M entries, the key folded by the j challenges so far
            # (JUMP_M generators) comes from ONE pass over the key's window table (amsm_ipa_jump_fold) and the remaining
            # log2(JUMP_M) rounds -- ~0.36 ms each on the device whatever their logical size -- run on the host over those few
            # generators; the final folded key falls out of the last fold instead of one more full-size MSM.  Keys that do not
            # qualify (AMSM_E_UNSUPPORTED: folded / plain keys, 20-bit tables, sharded keys) keep their rounds on the device.
            if n == cls._jump_m() and j >= 1 and key.num_shards == 1:
                jumped = cls._host_rounds(ctx, fr, cur_key, log_key, xs, j, n, coeffs, z, h_prime, round_challenge, l_vec, r_vec)
                if jumped is not None:
                    break
            xi = fr.to_limbs_many(xs[len(xs) - j:]) if j else None
            xy = np.zeros((2, 2 * ctx.fq_limbs), dtype=np.uint64)
            inf = np.zeros((2,), dtype=np.uint8)
            ips = np.zeros((2, 4), dtype=np.uint64)
            ffi.check(ctx._lib.amsm_ipa_round_fused(ctx._h, cur_key._h, _ptr(xi), j, log_key, coeffs.ptr, z.ptr, _ptr(prev_x),
                                                    _ptr(hp_xy), u_l.ptr, _ptr(xy), _ptr(inf), _ptr(ips)),
                      "amsm_ipa_round_fused")
            l_pt = (xy[0], bool(inf[0]))   # <c_r, key_l> + <c_r, z_l> h'
            r_pt = (xy[1], bool(inf[1]))   # <c_l, key_r> + <c_l, z_r> h'
            l_vec.append(l_pt)
            r_vec.append(r_pt)
            round_challenge = cls._challenge(fr, [round_challenge.to_bytes(16, "little"), l_pt, r_pt])
            xs.append(round_challenge)
            prev_x = fr.to_limbs(round_challenge)
            if len(xs) <= fold_rounds:  # physical fold: the next round sees a key of `half` generators
                assert j == 0
                folded = cur_key.fold(half, prev_x, CHALLENGE_SIZE)
                if cur_key is not key:
                    cur_key.free()
                cur_key, log_key = folded, log_key - 1
            n = half
        if jumped is not None:
            if cur_key is not key:
                cur_key.free()
            final_key, c = jumped
            return Proof(l_vec, r_vec, final_key, c, hiding_comm, proof_rand)
        since = xs[log_n - log_key:]
        if since:
            s_vec = SuccinctCheckPolynomial(since).compute_coeffs(ctx)
            final_key, final_inf = VariableBaseMSM.multi_scalar_mul(cur_key, s_vec, mont=True)
            assert not final_inf
        else:
            fk, _ = cur_key.read(0, 1)
            final_key = fk[0]
        if cur_key is not key:
            cur_key.free()
        # the last fold happens here: c = c_0 + x^-1 c_1 over the two coefficients the last round left
        head = coeffs.view(0, min(2, coeffs.n)).download()
        c = fr.from_limbs(head[0])
        if xs:
            c = (c + pow(xs[-1], -1, fr.r) * fr.from_limbs(head[1])) % fr.r
        return Proof(l_vec, r_vec, (final_key, not final_key.any()), c, hiding_comm, proof_rand)

    @staticmethod
    def _jump_m() -> int:
        """vector length at which an opening jumps to the host (AMSM_IPA_JUMP=0: never; =M: at M entries, a power of two >= 64)"""
        import os
        return int(os.environ.get("AMSM_IPA_JUMP", "64"))

    @classmethod
    def _host_rounds(cls, ctx, fr: Fr, cur_key, log_key: int, xs: List[int], j: int, n: int, coeffs: FrVector, z: FrVector, h_prime,
                     round_challenge: int, l_vec, r_vec):
        """The last log2(n) rounds of `open` on the host over the n generators amsm_ipa_jump_fold returns (the key folded by the j
        challenges since cur_key was formed).  Appends to xs, l_vec, r_vec; returns (final_comm_key, c), or None when the key does
        not qualify.  The same operations as the reference's loop (ark_poly_commit::ipa_pc::open ext): l = <c_r, key_l> + <c_r, z_l> h',
        r = <c_l, key_r> + <c_l, z_r> h', then c_l += x^-1 c_r, z_l += x z_r, key_l += x key_r."""
        w = 2 * ctx.fq_limbs
        xy = np.zeros((n, w), dtype=np.uint64)
        inf = np.zeros((n,), dtype=np.uint8)
        xi = fr.to_limbs_many(xs[len(xs) - j:])
        rc = ctx._lib.amsm_ipa_jump_fold(ctx._h, cur_key._h, log_key, _ptr(xi), j, _ptr(xy), _ptr(inf))
        if rc == ffi.AMSM_E_UNSUPPORTED:
            return None
        ffi.check(rc, "amsm_ipa_jump_fold")
        B = [(xy[k].copy(), bool(inf[k])) for k in range(n)]
        # the fold of c and z by the last challenge is still pending on the device (amsm_ipa_round_fused applies it lazily)
        cz = [[fr.from_limbs(v) for v in vec.view(0, 2 * n).download()] for vec in (coeffs, z)]
        x, xinv = xs[-1], pow(xs[-1], -1, fr.r)
        c = [(cz[0][i] + xinv * cz[0][n + i]) % fr.r for i in range(n)]
        zz = [(cz[1][i] + x * cz[1][n + i]) % fr.r for i in range(n)]
        hp = [] if h_prime[1] else [h_prime]
        while n > 1:
            half = n // 2
            ip_l = sum(c[half + i] * zz[i] for i in range(half)) % fr.r       # <c_r, z_l>
            ip_r = sum(c[i] * zz[half + i] for i in range(half)) % fr.r       # <c_l, z_r>
            l_pt = _lincomb(ctx, B[:half] + hp, c[half:n] + ([ip_l] if hp else []), fr)
            r_pt = _lincomb(ctx, B[half:n] + hp, c[:half] + ([ip_r] if hp else []), fr)
            l_vec.append(l_pt)
            r_vec.append(r_pt)
            round_challenge = cls._challenge(fr, [round_challenge.to_bytes(16, "little"), l_pt, r_pt])
            xs.append(round_challenge)
            x, xinv = round_challenge, pow(round_challenge, -1, fr.r)
            c = [(c[i] + xinv * c[half + i]) % fr.r for i in range(half)]
            zz = [(zz[i] + x * zz[half + i]) % fr.r for i in range(half)]
            B = [_lincomb(ctx, [B[i], B[half + i]], [1, x], fr) for i in range(half)]
            n = half
        return B[0], c[0]

    @staticmethod
    def _fold_rounds(ctx, log_n: int) -> int:
        """Leading rounds that fold the key physically (IPA_FOLD; AMSM_IPA_FOLD_ABOVE=T overrides: fold while the key has
        more than 2^T generators, 99 = never).  The proof does not depend on the choice."""
        import os
        env = int(os.environ.get("AMSM_IPA_FOLD_ABOVE", "0"))
        if env:
            return max(0, log_n - env)
        min_log, t = IPA_FOLD.get(ctx.curve, (99, 99))
        return max(0, log_n - t) if log_n >= min_log else 0

    @staticmethod
    def _inner_product(ctx, fr: Fr, a: FrVector, b: FrVector) -> int:
        out = np.zeros(4, dtype=np.uint64)
        ffi.check(ctx._lib.amsm_vec_inner_product(ctx._h, a.ptr, b.ptr, min(a.n, b.n), _ptr(out)), "amsm_vec_inner_product")
        return fr.from_limbs(out)

    # ---- succinct check / check ------------------------------------------------------------------------
    @classmethod
    def succinct_check(cls, ctx: Context, svk, commitment: Commitment, point: int, value: int, proof: Proof
                       ) -> Optional[SuccinctCheckPolynomial]:
        fr = Fr(ctx.curve)
        d = svk.supported_degree()
        log_d = (d + 1).bit_length() - 1
        if commitment.shifted_comm is not None:
            return None
        if len(proof.l_vec) != len(proof.r_vec) or len(proof.l_vec) != log_d:
            return None
        if (proof.hiding_comm is None) != (proof.rand is None):
            return None
        combined_comm = commitment.comm
        if proof.hiding_comm is not None:
            hch = cls._challenge(fr, [combined_comm, point, value, proof.hiding_comm])
            combined_comm = _lincomb(ctx, [combined_comm, proof.hiding_comm, svk.s], [1, hch, (-proof.rand) % fr.r], fr)
        round_challenge = cls._challenge(fr, [combined_comm, point, value])
        h_prime = _lincomb(ctx, [svk.h], [round_challenge], fr)
        pts, scs = [combined_comm, h_prime], [1, value]
        challenges = []
        for l_pt, r_pt in zip(proof.l_vec, proof.r_vec):
            round_challenge = cls._challenge(fr, [round_challenge.to_bytes(16, "little"), l_pt, r_pt])
            if round_challenge == 0:
                return None
            challenges.append(round_challenge)
            pts += [l_pt, r_pt]
            scs += [pow(round_challenge, -1, fr.r), round_challenge]
        round_commitment = _lincomb(ctx, pts, scs, fr)
        check_poly = SuccinctCheckPolynomial(challenges)
        v_prime = check_poly.evaluate(fr, point) * proof.c % fr.r
        check_commitment = _lincomb(ctx, [proof.final_comm_key, h_prime], [proof.c, v_prime], fr)
        if not _pt_eq(round_commitment, check_commitment):
            return None
        return check_poly

    @classmethod
    def check(cls, vk: CommitterKey, commitment: Commitment, point: int, value: int, proof: Proof) -> bool:
        ctx = vk.comm_key.ctx
        check_poly = cls.succinct_check(ctx, vk.svk, commitment, point, value, proof)
        if check_poly is None:
            return False
        coeffs = check_poly.compute_coeffs(ctx)          # 2^k field multiplications on the device
        final_key = cls.cm_commit(vk.comm_key, coeffs)   # THE (d+1)-point MSM of the decider
        return _pt_eq(final_key, proof.final_comm_key)
