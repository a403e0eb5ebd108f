"""Host-side mirror of the scalar-field vector loops of `ASForHadamardProducts` that produce the MSM
scalars (reference: src/hp_as/mod.rs), running on HBM-resident `FrVector`s through the C ABI.

Function names and argument meaning follow the reference:
  compute_hp(a, b)                                  src/hp_as/mod.rs:278-285
  scale_vector(v, coeff)                            src/hp_as/mod.rs:482-489
  combine_vectors(vectors, challenges, hiding)      src/hp_as/mod.rs:492-512
  compute_t_vecs(a_vecs, b_vecs, mu, len, hiding)   src/hp_as/mod.rs:288-349
  compute_product_poly_comm(ck, t_vecs)             src/hp_as/mod.rs:354-388
  decide_commitments(dk, a, b)                      src/hp_as/mod.rs:894-925 (the three MSMs of `decide`)
Field elements cross this API as Montgomery-form uint64 limbs (raw `Vec<Fr>` memory).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import ffi
from .engine import CommitterKey, Context, FrVector, VariableBaseMSM, _ptr


def compute_hp(ctx: Context, a: FrVector, b: FrVector) -> FrVector:
    n = min(a.n, b.n)  # zip truncates
    out = ctx.vector(n)
    ffi.check(ctx._lib.amsm_vec_hadamard(ctx._h, a.ptr, b.ptr, out.ptr, n), "amsm_vec_hadamard")
    return out


def combine_vectors(ctx: Context, vectors: Sequence[FrVector], challenges_mont: np.ndarray,
                    hiding: Optional[FrVector] = None) -> FrVector:
    k = len(vectors)
    n = max([v.n for v in vectors] + ([hiding.n] if hiding is not None else [0]))
    ch = np.ascontiguousarray(challenges_mont, dtype=np.uint64).reshape(-1, 4)
    assert ch.shape[0] >= k
    ptrs = (C.c_void_p * max(k, 1))(*[v.ptr for v in vectors])
    lens = (C.c_size_t * max(k, 1))(*[v.n for v in vectors])
    out = ctx.vector(n)
    ffi.check(ctx._lib.amsm_vec_combine(ctx._h, ptrs, lens, k, _ptr(ch), hiding.ptr if hiding is not None else None,
                                        hiding.n if hiding is not None else 0, out.ptr, n), "amsm_vec_combine")
    return out


def scale_vector(ctx: Context, v: FrVector, coeff_mont: np.ndarray) -> FrVector:
    return combine_vectors(ctx, [v], np.asarray(coeff_mont, dtype=np.uint64).reshape(1, 4))


def compute_t_vecs(ctx: Context, a_vecs: Sequence[FrVector], b_vecs: Sequence[FrVector], mu_mont: np.ndarray,
                   hp_vec_len: int, hiding: Optional[Tuple[FrVector, FrVector]] = None,
                   skip_uncommitted: bool = False) -> List[Optional[FrVector]]:
    n = len(a_vecs)
    mu = np.ascontiguousarray(mu_mont, dtype=np.uint64).reshape(-1, 4)
    pa = (C.c_void_p * n)(*[v.ptr for v in a_vecs])
    pb = (C.c_void_p * n)(*[v.ptr for v in b_vecs])
    la = (C.c_size_t * n)(*[v.n for v in a_vecs])
    lb = (C.c_size_t * n)(*[v.n for v in b_vecs])
    outs: List[Optional[FrVector]] = []
    for k in range(2 * n - 1):
        outs.append(None if (skip_uncommitted and k == n - 1) else ctx.vector(hp_vec_len))
    pt = (C.c_void_p * (2 * n - 1))(*[(v.ptr if v is not None else None) for v in outs])
    ha, hb = (hiding if hiding is not None else (None, None))
    ffi.check(ctx._lib.amsm_hp_t_vecs(ctx._h, pa, la, pb, lb, n, _ptr(mu), mu.shape[0],
                                      ha.ptr if ha is not None else None, ha.n if ha is not None else 0,
                                      hb.ptr if hb is not None else None, hb.n if hb is not None else 0,
                                      pt, hp_vec_len), "amsm_hp_t_vecs")
    return outs


def compute_product_poly_comm(ck: CommitterKey, t_vecs: Sequence[Optional[FrVector]]):
    """-> (low, high): commitments to t_0..t_{n-2} and t_n..t_{2n-2}; t_{n-1} is skipped (:373-375)."""
    if len(t_vecs) == 0:
        return [], []
    n = (len(t_vecs) + 1) // 2
    todo = [t for i, t in enumerate(t_vecs) if i != n - 1]
    pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, todo, mont=True)
    comms = [(pts[i], bool(infs[i])) for i in range(len(todo))]
    return comms[: n - 1], comms[n - 1:]


def decide_commitments(dk: CommitterKey, a_vec: FrVector, b_vec: FrVector):
    """The decider's three commitments: commit(a), commit(b), commit(a o b) (no hiding)."""
    ctx = dk.ctx
    prod = compute_hp(ctx, a_vec, b_vec)
    pts, infs = VariableBaseMSM.multi_scalar_mul_batch(dk, [a_vec, b_vec, prod], mont=True)
    return [(pts[i], bool(infs[i])) for i in range(3)]


# =====================================================================================================
# The accumulation scheme itself: ASForHadamardProducts {index, prove, verify, decide}
# (src/hp_as/mod.rs:609-925) with every O(len) loop and every MSM on the GPU and the O(#inputs) work on
# the host.  Data structures mirror src/hp_as/data_structures.rs.
# =====================================================================================================
from dataclasses import dataclass, field  # noqa: E402

from .engine import PedersenCommitment  # noqa: E402
from .scalar_field import Fr  # noqa: E402
from .sponge import CryptographicSponge, Sha256Sponge  # noqa: E402

CHALLENGE_SIZE = 128  # src/hp_as/mod.rs:29


class ASError(Exception):
    """src/error.rs:8-20"""


class MalformedAccumulator(ASError):
    pass


class MalformedInput(ASError):
    pass


class MissingRng(ASError):
    pass


def _zero_point(ctx):
    return (np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64), True)


def _pt_eq(p, q) -> bool:
    return bool(p[1]) == bool(q[1]) and (bool(p[1]) or np.array_equal(p[0], q[0]))


@dataclass
class InputInstance:  # data_structures.rs:14-33
    comm_1: tuple
    comm_2: tuple
    comm_3: tuple

    @staticmethod
    def zero(ctx):
        return InputInstance(_zero_point(ctx), _zero_point(ctx), _zero_point(ctx))

    def eq(self, o) -> bool:
        return _pt_eq(self.comm_1, o.comm_1) and _pt_eq(self.comm_2, o.comm_2) and _pt_eq(self.comm_3, o.comm_3)

    def absorb_into(self, sponge):
        for p in (self.comm_1, self.comm_2, self.comm_3):
            sponge.absorb_point(p)


@dataclass
class InputWitnessRandomness:  # data_structures.rs:77-90 (canonical ints)
    rand_1: int
    rand_2: int
    rand_3: int


@dataclass
class InputWitness:  # data_structures.rs:54-74
    a_vec: FrVector
    b_vec: FrVector
    randomness: Optional[InputWitnessRandomness] = None

    @staticmethod
    def zero(ctx, vec_len):
        z = np.zeros(4, dtype=np.uint64)
        return InputWitness(ctx.fill(z, vec_len), ctx.fill(z, vec_len), None)


@dataclass
class ProductPolynomialCommitment:  # data_structures.rs:95-114
    low: list
    high: list


@dataclass
class ProofHidingCommitments:
    comm_1: tuple
    comm_2: tuple
    comm_3: tuple


@dataclass
class Proof:
    product_poly_comm: ProductPolynomialCommitment
    hiding_comms: Optional[ProofHidingCommitments] = None


@dataclass
class Accumulator:
    instance: InputInstance
    witness: InputWitness


class ASForHadamardProducts:
    """Accumulation scheme for Hadamard products over a Pedersen committer key resident in HBM."""

    @staticmethod
    def index(ck: CommitterKey):
        """-> (prover_key, verifier_key, decider_key) = (ck, supported_num_elems, ck)   (:620-644)"""
        return ck, ck.supported_num_elems(), ck

    # ---- helpers --------------------------------------------------------------------------------
    @staticmethod
    def _check_input_witness_structure(w: InputWitness, pk: CommitterKey, vec_len: int, is_acc: bool):
        err = MalformedAccumulator if is_acc else MalformedInput
        if w.a_vec.n == 0 or w.b_vec.n == 0:  # :117-126
            raise err("A vector of the Hadamard Product relation with a length of 0 is unsupported.")
        if w.a_vec.n > pk.supported_num_elems() or w.b_vec.n > pk.supported_num_elems():  # :129-141
            raise err("A vector of the Hadamard Product relation has a length that exceeds the prover key's "
                      "supported length.")
        if w.a_vec.n != w.b_vec.n or w.a_vec.n != vec_len:  # :144-154
            raise err("All of the vectors of the Hadamard Product relation that have or will be accumulated must "
                      "have equal lengths")
        return w

    @staticmethod
    def _check_proof_structure(proof: Proof, num_inputs: int) -> bool:  # :160-176
        assert num_inputs > 0
        ppc = proof.product_poly_comm
        return len(ppc.low) == len(ppc.high) and len(ppc.low) == num_inputs - 1

    @staticmethod
    def _squeeze_mu(sponge, fr: Fr, num_inputs: int, make_zk: bool):  # :233-253
        mu = [1]
        if num_inputs > 1:
            mu += sponge.squeeze_field_elements(num_inputs - 1, CHALLENGE_SIZE)
        if make_zk:
            mu.append(mu[1] * mu[num_inputs - 1] % fr.r)
        return mu

    @staticmethod
    def _squeeze_nu(sponge, fr: Fr, num_inputs: int):  # :256-275
        nu1 = sponge.squeeze_field_elements(1, CHALLENGE_SIZE)[0]
        out, cur = [], 1
        for _ in range(2 * num_inputs - 1):
            out.append(cur)
            cur = cur * nu1 % fr.r
        return out

    @staticmethod
    def _absorb_statement(sponge, num_elems: int, instances, hiding_comms):  # absorb!(...) :753-758, :863-868
        sponge.absorb_u64(num_elems)
        sponge.absorb_len(len(instances))
        for inst in instances:
            inst.absorb_into(sponge)
        if hiding_comms is None:
            sponge.absorb_bytes(b"\x00")
        else:
            sponge.absorb_bytes(b"\x01")
            for p in (hiding_comms.comm_1, hiding_comms.comm_2, hiding_comms.comm_3):
                sponge.absorb_point(p)

    @staticmethod
    def _lincomb(ctx, points, scalars, fr: Fr):
        """host: sum_i scalars[i] * points[i] -> affine  (`combine_commitments`, :391-406)"""
        k = len(points)
        out = np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64)
        inf = C.c_uint8(0)
        if k == 0:
            return (out, True)
        xy = np.stack([np.asarray(p[0], dtype=np.uint64) for p in points])
        infs = np.array([1 if p[1] else 0 for p in points], dtype=np.uint8)
        sc = fr.to_limbs_many([s % fr.r for s in scalars[:k]])
        ffi.check(ctx._lib.amsm_host_lincomb(ctx.curve, _ptr(xy), _ptr(infs), _ptr(sc), k, _ptr(out), C.byref(inf)),
                  "amsm_host_lincomb")
        return (out, bool(inf.value))

    @staticmethod
    def _lincomb_batch(ctx, jobs, fr: Fr):
        """Independent host combinations in one library call (amsm_host_lincomb_batch: host threads, one normalisation);
        jobs = [(points, scalars), ...].  Same results as _lincomb job by job."""
        nj = len(jobs)
        if nj == 0:
            return []
        w = 2 * ctx.fq_limbs
        keep = []
        n_terms = (C.c_size_t * nj)()
        xy_p, inf_p, sc_p = (C.c_void_p * nj)(), (C.c_void_p * nj)(), (C.c_void_p * nj)()
        for j, (points, scalars) in enumerate(jobs):
            k = len(points)
            n_terms[j] = k
            xy = np.stack([np.asarray(p[0], dtype=np.uint64) for p in points]) if k else np.zeros((1, w), dtype=np.uint64)
            infs = np.array([1 if p[1] else 0 for p in points] or [0], dtype=np.uint8)
            sc = fr.to_limbs_many([s % fr.r for s in scalars[:k]]) if k else np.zeros((1, 4), dtype=np.uint64)
            keep.append((xy, infs, sc))
            xy_p[j], inf_p[j], sc_p[j] = xy.ctypes.data, infs.ctypes.data, sc.ctypes.data
        out = np.zeros((nj, w), dtype=np.uint64)
        oinf = np.zeros((nj,), dtype=np.uint8)
        ffi.check(ctx._lib.amsm_host_lincomb_batch(ctx.curve, nj, n_terms, xy_p, inf_p, sc_p, _ptr(out), _ptr(oinf)),
                  "amsm_host_lincomb_batch")
        res = []
        for j in range(nj):
            inf = bool(oinf[j]) or n_terms[j] == 0
            res.append((np.zeros((w,), dtype=np.uint64) if inf else out[j].copy(), inf))
        return res

    @classmethod
    def _compute_combined_hp_commitments(cls, ctx, fr, instances, proof: Proof, mu, nu, chi) -> InputInstance:
        """:409-479 (one host linear combination per output commitment)"""
        n = len(instances)
        hc = proof.hiding_comms
        p1 = [i.comm_1 for i in instances]
        s1 = list(chi[:n])
        p2 = [i.comm_2 for i in reversed(instances)]
        s2 = list(nu[:n])
        low, high = proof.product_poly_comm.low, proof.product_poly_comm.high
        p3 = list(low) + list(high) + [i.comm_3 for i in instances]
        s3 = list(nu[: len(low)]) + list(nu[n: n + len(high)]) + [mu[i] * nu[n - 1] % fr.r for i in range(n)]
        if hc is not None:
            p1.append(hc.comm_1)
            s1.append(mu[n])
            p2.append(hc.comm_2)
            s2.append(mu[1])
            p3.append(hc.comm_3)
            s3.append(mu[n] * nu[n - 1] % fr.r)
        c1, c2, c3 = cls._lincomb_batch(ctx, [(p1, s1), (p2, s2), (p3, s3)], fr)  # three independent combinations
        return InputInstance(c1, c2, c3)

    @staticmethod
    def _generate_prover_randomness(pk: CommitterKey, fr: Fr, hp_vec_len: int, witnesses, rng):
        """:179-230.  The hiding vectors are CONSTANT vectors (`vec![rand; len]`)."""
        ctx = pk.ctx
        a_val, b_val = rng.field(), rng.field()
        a = ctx.fill(fr.to_limbs(a_val), hp_vec_len)
        b = ctx.fill(fr.to_limbs(b_val), hp_vec_len)
        rands = InputWitnessRandomness(rng.field(), rng.field(), rng.field())
        rand_prod_1 = compute_hp(ctx, a, witnesses[0].b_vec)
        rand_prod_2 = compute_hp(ctx, witnesses[-1].a_vec, b)
        one = fr.to_limbs(1)
        rand_prods_sum = combine_vectors(ctx, [rand_prod_1, rand_prod_2], np.stack([one, one]))
        # the three commitments are independent: one pipelined batch (same points as three commit() calls, :196-214)
        comm_1, comm_2, comm_3 = PedersenCommitment.commit_batch(
            pk, [a, b, rand_prods_sum], [fr.to_limbs(rands.rand_1), fr.to_limbs(rands.rand_2), fr.to_limbs(rands.rand_3)])
        return (a, b), rands, ProofHidingCommitments(comm_1, comm_2, comm_3)

    @staticmethod
    def _combine_randomness(fr, rands, challenges, hiding):  # :515-532
        acc = 0
        for i, r in enumerate(rands):
            if r is not None:
                acc = (acc + r * challenges[i]) % fr.r
        if hiding is not None:
            acc = (acc + hiding) % fr.r
        return acc

    # ---- prove ----------------------------------------------------------------------------------
    @classmethod
    def prove(cls, prover_key: CommitterKey, inputs, old_accumulators, rng=None, sponge: Optional[CryptographicSponge] = None):
        """:646-813.  inputs / old_accumulators: sequences of objects with .instance and .witness.
        rng: None = MakeZK::Disabled, else an object with .field() -> random scalar (MakeZK::Enabled).
        Returns (Accumulator, Proof)."""
        ctx = prover_key.ctx
        fr = Fr(ctx.curve)
        sponge = sponge if sponge is not None else Sha256Sponge()
        inputs = list(inputs)
        old_accumulators = list(old_accumulators)
        num_all = len(inputs) + len(old_accumulators)
        make_zk = rng is not None
        if not make_zk and num_all > 0:  # :664-673
            for x in inputs + old_accumulators:
                if x.witness.randomness is not None:
                    raise MissingRng("Accumulating inputs with hiding requires rng.")
        if old_accumulators:  # :676-682
            hp_vec_len = old_accumulators[0].witness.a_vec.n
        elif inputs:
            hp_vec_len = inputs[0].witness.a_vec.n
        else:
            hp_vec_len = prover_key.local_num_elems()
        if num_all == 0:  # default input :685-696
            inputs.append(Accumulator(InputInstance.zero(ctx), InputWitness.zero(ctx, hp_vec_len)))
            num_all += 1
        if make_zk and num_all == 1:  # placeholder for hiding :698-710
            inputs.append(Accumulator(InputInstance.zero(ctx), InputWitness.zero(ctx, hp_vec_len)))
            num_all += 1
        instances = [x.instance for x in inputs] + [x.instance for x in old_accumulators]
        witnesses = [cls._check_input_witness_structure(x.witness, prover_key, hp_vec_len, False) for x in inputs] + \
                    [cls._check_input_witness_structure(x.witness, prover_key, hp_vec_len, True) for x in old_accumulators]
        hiding_vecs = hiding_rands = hiding_comms = None
        if make_zk:  # step 3
            hiding_vecs, hiding_rands, hiding_comms = cls._generate_prover_randomness(prover_key, fr, hp_vec_len,
                                                                                       witnesses, rng)
        cls._absorb_statement(sponge, prover_key.supported_num_elems(), instances, hiding_comms)  # step 4
        mu = cls._squeeze_mu(sponge, fr, num_all, make_zk)
        # steps 5-8: t-vectors on the device (the uncommitted middle one is never materialised), batch commit
        t_vecs = compute_t_vecs(ctx, [w.a_vec for w in witnesses], [w.b_vec for w in witnesses], fr.to_limbs_many(mu),
                                hp_vec_len, hiding_vecs, skip_uncommitted=True)
        low, high = compute_product_poly_comm(prover_key, t_vecs)
        proof = Proof(ProductPolynomialCommitment(low, high), hiding_comms)
        sponge.absorb_points(low)  # step 9
        sponge.absorb_points(high)
        nu = cls._squeeze_nu(sponge, fr, num_all)
        chi = [m * v % fr.r for m, v in zip(mu, nu)]
        acc_instance = cls._compute_combined_hp_commitments(ctx, fr, instances, proof, mu, nu, chi)  # steps 10-12
        # steps 13-15: combined openings (:535-607)
        add1 = scale_vector(ctx, hiding_vecs[0], fr.to_limbs(mu[num_all])) if make_zk else None
        a_open = combine_vectors(ctx, [w.a_vec for w in witnesses], fr.to_limbs_many(chi[:num_all]), add1)
        add2 = scale_vector(ctx, hiding_vecs[1], fr.to_limbs(mu[1])) if make_zk else None
        b_open = combine_vectors(ctx, [w.b_vec for w in reversed(witnesses)], fr.to_limbs_many(nu[:num_all]), add2)
        randomness = None
        if make_zk:
            def rnd(k):
                return [None if w.randomness is None else getattr(w.randomness, k) for w in witnesses]
            a_r = cls._combine_randomness(fr, rnd("rand_1"), chi, hiding_rands.rand_1 * mu[num_all] % fr.r)
            b_r = cls._combine_randomness(fr, list(reversed(rnd("rand_2"))), nu, hiding_rands.rand_2 * mu[1] % fr.r)
            p_r = cls._combine_randomness(fr, rnd("rand_3"), mu, hiding_rands.rand_3 * mu[num_all] % fr.r) * nu[num_all - 1] % fr.r
            randomness = InputWitnessRandomness(a_r, b_r, p_r)
        return Accumulator(acc_instance, InputWitness(a_open, b_open, randomness)), proof

    # ---- verify (host only: no MSM) ---------------------------------------------------------------
    @classmethod
    def verify(cls, ctx, verifier_key: int, input_instances, old_accumulator_instances, new_accumulator_instance,
               proof: Proof, sponge: Optional[CryptographicSponge] = None) -> bool:
        """:815-892"""
        fr = Fr(ctx.curve)
        sponge = sponge if sponge is not None else Sha256Sponge()
        ins = list(input_instances)
        olds = list(old_accumulator_instances)
        num_all = len(ins) + len(olds)
        make_zk = proof.hiding_comms is not None
        if num_all == 0:
            ins.append(InputInstance.zero(ctx))
            num_all += 1
        if make_zk and num_all == 1:
            ins.append(InputInstance.zero(ctx))
            num_all += 1
        if not cls._check_proof_structure(proof, num_all):
            return False
        instances = ins + olds
        cls._absorb_statement(sponge, verifier_key, instances, proof.hiding_comms)
        mu = cls._squeeze_mu(sponge, fr, num_all, make_zk)
        sponge.absorb_points(proof.product_poly_comm.low)
        sponge.absorb_points(proof.product_poly_comm.high)
        nu = cls._squeeze_nu(sponge, fr, num_all)
        chi = [m * v % fr.r for m, v in zip(mu, nu)]
        acc = cls._compute_combined_hp_commitments(ctx, fr, instances, proof, mu, nu, chi)
        return acc.eq(new_accumulator_instance)

    # ---- decide -----------------------------------------------------------------------------------
    @staticmethod
    def decide(decider_key: CommitterKey, accumulator: Accumulator, sponge=None) -> bool:
        """:894-925: three Pedersen commitments of full-length vectors (the MSMs), then equality."""
        ctx = decider_key.ctx
        fr = Fr(ctx.curve)
        w = accumulator.witness
        product = compute_hp(ctx, w.a_vec, w.b_vec)
        rnd = w.randomness
        commit = PedersenCommitment.commit
        if rnd is None:
            c = decide_commitments(decider_key, w.a_vec, w.b_vec)
        else:
            c = PedersenCommitment.commit_batch(decider_key, [w.a_vec, w.b_vec, product],
                                                [fr.to_limbs(rnd.rand_1), fr.to_limbs(rnd.rand_2), fr.to_limbs(rnd.rand_3)])
        inst = accumulator.instance
        return _pt_eq(c[0], inst.comm_1) and _pt_eq(c[1], inst.comm_2) and _pt_eq(c[2], inst.comm_3)
