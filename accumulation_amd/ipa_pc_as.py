"""Host-side mirror of `AtomicASForInnerProductArgPC` (reference: src/ipa_pc_as/mod.rs): index :502-553,
prove :555-676, verify :678-818, decide :820-848, over the IPA polynomial commitment mirror in ipa_pc.py.
The prover's O(d) work -- compute_coeffs of every succinct-check polynomial, their linear combination,
the evaluation and the IPA opening (MSMs, inner products, key folds) -- runs on the GPU; the decider is one
(d+1)-point MSM.  Like the reference (:566-570) the scheme refuses a caller-supplied sponge."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from .engine import FrVector
from .hp_as import ASError, ASForHadamardProducts, MalformedAccumulator, MalformedInput, MissingRng, _pt_eq, combine_vectors
from .ipa_pc import Commitment, CommitterKey, InnerProductArgPC as IpaPC, Proof as IpaProof, SuccinctCheckPolynomial
from .scalar_field import Fr
from .sponge import Sha256Sponge

LINEAR_COMBINATION_CHALLENGE_SIZE = 128  # :42
CHALLENGE_POINT_SIZE = 184               # :43
_lincomb = ASForHadamardProducts._lincomb


@dataclass
class InputInstance:  # data_structures.rs:56-68
    ipa_commitment: Commitment
    point: int
    evaluation: int
    ipa_proof: IpaProof


@dataclass
class Randomness:  # data_structures.rs:71-86 (the scheme's Proof is Option<Randomness>)
    random_linear_polynomial: List[int]  # coefficients, degree <= 1
    random_linear_polynomial_commitment: tuple
    commitment_randomness: int


@dataclass
class VerifierKey:  # data_structures.rs:37-49
    ipa_svk: object
    ipa_ck_linear: CommitterKey
    default_proof: IpaProof


@dataclass
class ProverKey:  # data_structures.rs:27-34
    ipa_ck: CommitterKey
    verifier_key: VerifierKey


@dataclass
class Accumulator:
    instance: InputInstance
    witness: None = None


class AtomicASForInnerProductArgPC:
    sponge_cls = Sha256Sponge

    @classmethod
    def index(cls, pp: CommitterKey, supported_degree_bound: int):
        """:502-553"""
        ctx = pp.comm_key.ctx
        fr = Fr(ctx.curve)
        ipa_ck, ipa_vk = IpaPC.trim(pp, supported_degree_bound)
        zero_poly = ctx.fill(fr.to_limbs(0), 1)
        default_proof = IpaPC.open(ipa_ck, zero_poly, Commitment.default(ctx), 0, 0, False, None)
        ipa_ck_linear, _ = IpaPC.trim(pp, 1)
        vk = VerifierKey(ipa_vk.svk, ipa_ck_linear, default_proof)
        return ProverKey(ipa_ck, vk), vk, ipa_vk

    @classmethod
    def _as_sponge(cls):
        return cls.sponge_cls().fork(b"AS-FOR-IPA-PC-2020")

    @staticmethod
    def _check_proof_structure(proof: Optional[Randomness]) -> bool:  # :130-137
        if proof is not None:
            c = list(proof.random_linear_polynomial)
            while c and c[-1] == 0:
                c.pop()
            return len(c) <= 2
        return True

    @staticmethod
    def _deterministic_ipa_pc_commit(ck_linear: CommitterKey, coeffs: Sequence[int]):  # :147-162 (an MSM of size 2)
        ctx = ck_linear.comm_key.ctx
        fr = Fr(ctx.curve)
        xy, _ = ck_linear.comm_key.read(0, 2)
        return _lincomb(ctx, [(xy[0], False), (xy[1], False)], list(coeffs) + [0] * (2 - len(coeffs)), fr)

    @classmethod
    def _generate_prover_randomness(cls, pk: ProverKey, rng) -> Randomness:  # :165-187
        lin = [rng.field(), rng.field()]
        comm = cls._deterministic_ipa_pc_commit(pk.verifier_key.ipa_ck_linear, lin)
        return Randomness(lin, comm, rng.field())

    @classmethod
    def _succinct_checks(cls, ctx, svk, instances, are_accumulators, out):  # :190-221
        for inst in instances:
            cp = IpaPC.succinct_check(ctx, svk, inst.ipa_commitment, inst.point, inst.evaluation, inst.ipa_proof)
            if cp is None:
                raise (MalformedAccumulator("Succinct check failed on accumulator.") if are_accumulators
                       else MalformedInput("Succinct check failed on input."))
            out.append((cp, inst.ipa_proof.final_comm_key))

    @classmethod
    def _combine(cls, ctx, fr, svk, succinct_checks, proof: Optional[Randomness], sponge):
        """combine_succinct_check_polynomials_and_commitments :254-346"""
        sp = sponge
        if proof is not None:
            co = list(proof.random_linear_polynomial) + [0, 0]
            for i in range(2):
                sp.absorb_bytes((co[i] % fr.r).to_bytes(32, "little"))
            sp.absorb_point(proof.random_linear_polynomial_commitment)
        for cp, comm in succinct_checks:
            sp.absorb_bytes(cp.to_bytes(fr))
            sp.absorb_point(comm)
        chal = sp.squeeze_field_elements(len(succinct_checks), LINEAR_COMBINATION_CHALLENGE_SIZE)
        pts = [comm for _, comm in succinct_checks]
        scs = list(chal)
        if proof is not None:
            pts.append(proof.random_linear_polynomial_commitment)
            scs.append(1)
        combined = _lincomb(ctx, pts, scs, fr)
        randomized = combined
        if proof is not None:
            randomized = _lincomb(ctx, [combined, svk.s], [1, proof.commitment_randomness], fr)
        addends = [(a, cp) for a, (cp, _) in zip(chal, succinct_checks)]
        return combined, Commitment(randomized, None), addends

    @classmethod
    def _new_challenge(cls, fr, sponge, combined_commitment, addends, lin: Optional[Sequence[int]]):  # :349-388
        sp = sponge
        sp.absorb_point(combined_commitment)
        if lin is None:
            sp.absorb_bytes(b"\x00")
        else:
            co = list(lin) + [0, 0]
            # `Option<Vec<u8>>`: the tag is an item of its own (one sponge element), then the byte string (packed by itself)
            sp.absorb_bytes(b"\x01")
            sp.absorb_bytes((co[0] % fr.r).to_bytes(32, "little") + (co[1] % fr.r).to_bytes(32, "little"))
        for a, cp in addends:
            sp.absorb_bytes(int(a).to_bytes((LINEAR_COMBINATION_CHALLENGE_SIZE + 7) // 8, "little"))
            sp.absorb_bytes(cp.to_bytes(fr))
        return sp.squeeze_field_elements(1, CHALLENGE_POINT_SIZE)[0]

    # ---- prove ------------------------------------------------------------------------------------
    @classmethod
    def prove(cls, pk: ProverKey, inputs: Sequence[InputInstance], old_accumulators: Sequence[InputInstance], rng=None,
              sponge=None):
        if sponge is not None:
            raise NotImplementedError("ASForIpaPC is unable to accept sponge objects until IpaPC gets updated to "
                                      "accept them.")  # :566-570
        ipa_ck = pk.ipa_ck
        ctx = ipa_ck.comm_key.ctx
        fr = Fr(ctx.curve)
        ins = list(inputs)
        olds = list(old_accumulators)
        for group, err in ((ins, MalformedInput), (olds, MalformedAccumulator)):  # check_input_instance_structure :112-128
            for x in group:
                if x.ipa_commitment.shifted_comm is not None:
                    raise err("Explicit degree bounds not supported.")
        make_zk = rng is not None
        if not make_zk:
            for x in ins + olds:
                if x.ipa_proof.hiding_comm is not None or x.ipa_proof.rand is not None:
                    raise MissingRng("Accumulating inputs with hiding requires rng.")
        if not make_zk and not ins and not olds:  # default instance :599-609
            ins.append(InputInstance(Commitment.default(ctx), 0, 0, pk.verifier_key.default_proof))
        proof = cls._generate_prover_randomness(pk, rng) if make_zk else None
        checks = []
        cls._succinct_checks(ctx, pk.verifier_key.ipa_svk, ins, False, checks)
        cls._succinct_checks(ctx, pk.verifier_key.ipa_svk, olds, True, checks)
        as_sponge = cls._as_sponge()
        combined, randomized, addends = cls._combine(ctx, fr, pk.verifier_key.ipa_svk, checks, proof, as_sponge.fork(b""))
        # combined check polynomial on the device: sum_i alpha_i * h_i (+ random linear polynomial)  :391-404
        n = ipa_ck.supported_degree() + 1
        vecs = [cp.compute_coeffs(ctx) for _, cp in addends]
        lin = None
        if proof is not None:
            lin = ctx.upload(fr.to_limbs_many(proof.random_linear_polynomial))
        if vecs:
            poly = combine_vectors(ctx, vecs, fr.to_limbs_many([a for a, _ in addends]), lin)
        else:
            poly = lin if lin is not None else ctx.fill(fr.to_limbs(0), 1)
        challenge = cls._new_challenge(fr, as_sponge.fork(b""), combined, addends,
                                       proof.random_linear_polynomial if proof is not None else None)
        # compute_new_accumulator :424-472: evaluate, then ONE IPA opening of the combined polynomial
        z = ctx.vector(poly.n)
        from . import ffi
        from .engine import _ptr
        ffi.check(ctx._lib.amsm_vec_powers(ctx._h, _ptr(fr.to_limbs(challenge)), poly.n, z.ptr), "amsm_vec_powers")
        evaluation = IpaPC._inner_product(ctx, fr, poly, z)
        ipa_proof = IpaPC.open(ipa_ck, poly, randomized, challenge, proof.commitment_randomness if proof else 0,
                               hiding=proof is not None, rng=rng)
        return Accumulator(InputInstance(randomized, challenge, evaluation, ipa_proof)), proof

    # ---- verify (host only apart from nothing: no MSM) -------------------------------------------------
    @classmethod
    def verify(cls, ctx, vk: VerifierKey, input_instances, old_accumulator_instances, new_acc: InputInstance,
               proof: Optional[Randomness], sponge=None) -> bool:
        if sponge is not None:
            raise NotImplementedError("ASForIpaPC is unable to accept sponge objects until IpaPC gets updated to "
                                      "accept them.")
        fr = Fr(ctx.curve)
        ins = list(input_instances)
        olds = list(old_accumulator_instances)
        if any(x.ipa_commitment.shifted_comm is not None for x in ins + olds):
            return False
        if not cls._check_proof_structure(proof):
            return False
        make_zk = proof is not None
        if not make_zk and not ins and not olds:
            ins.append(InputInstance(Commitment.default(ctx), 0, 0, vk.default_proof))
        checks = []
        try:
            cls._succinct_checks(ctx, vk.ipa_svk, ins, False, checks)
            cls._succinct_checks(ctx, vk.ipa_svk, olds, True, checks)
        except ASError:
            return False
        if proof is not None:
            lc = cls._deterministic_ipa_pc_commit(vk.ipa_ck_linear, proof.random_linear_polynomial)
            if not _pt_eq(lc, proof.random_linear_polynomial_commitment):
                return False
        as_sponge = cls._as_sponge()
        combined, randomized, addends = cls._combine(ctx, fr, vk.ipa_svk, checks, proof, as_sponge.fork(b""))
        if not _pt_eq(randomized.comm, new_acc.ipa_commitment.comm):
            return False
        challenge = cls._new_challenge(fr, as_sponge.fork(b""), combined, addends,
                                       proof.random_linear_polynomial if proof is not None else None)
        if challenge % fr.r != new_acc.point % fr.r:
            return False
        ev = 0
        if proof is not None:
            co = list(proof.random_linear_polynomial) + [0, 0]
            ev = (co[0] + co[1] * challenge) % fr.r
        for a, cp in addends:  # evaluate_combined_succinct_check_polynomials :407-421
            ev = (ev + cp.evaluate(fr, challenge) * a) % fr.r
        return ev == new_acc.evaluation % fr.r

    # ---- decide -------------------------------------------------------------------------------------
    @classmethod
    def decide(cls, dk: CommitterKey, accumulator: Accumulator, sponge=None) -> bool:
        if sponge is not None:
            raise NotImplementedError("ASForIpaPC is unable to accept sponge objects until IpaPC gets updated to "
                                      "accept them.")
        a = accumulator.instance
        return IpaPC.check(dk, a.ipa_commitment, a.point, a.evaluation, a.ipa_proof)  # :836-845
