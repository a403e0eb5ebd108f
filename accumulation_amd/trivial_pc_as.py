"""Host-side mirror of `ASForTrivialPC` (reference: src/trivial_pc_as/mod.rs): index :310-330, prove :332-468,
verify :470-609, decide :611-632, over `ark_poly_commit::trivial_pc::TrivialPC` (ext), whose commitment is the
Pedersen commitment of the coefficient vector and whose opening "proof" is the polynomial itself.

SURVEY.md section 8(a) row a10 (BASELINE config 1): every commitment the scheme issues is an MSM of at most d+1 <=
2^10 pairs through the same ABI as the large ones (`amsm_pedersen_commit_device`); the O(d) polynomial work per
claim -- the quotient (p(X) - v) / (X - z), evaluations at the challenge point, the linear combination of the
witness polynomials -- is small sequential host arithmetic in the reference too and stays on the host here
(Python integers).  The sponge is pluggable like in the other mirrors (the SHA-256 stand-in, or sponge.PoseidonSponge: the reference's
sponge with ark-sponge's parameters restated as recalled); what is absorbed and squeezed, and in which order, follows the reference's
`absorb!` lists and is compared with their restatement outside the product in tests/test_transcripts_vs_oracle.py.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from .engine import CommitterKey, Context, PedersenCommitment, VariableBaseMSM
from .hp_as import ASForHadamardProducts, MalformedAccumulator, MalformedInput, _pt_eq
from .scalar_field import Fr
from .sponge import CryptographicSponge, Sha256Sponge

LINEAR_COMBINATION_CHALLENGE_SIZE = 126  # :31
CHALLENGE_POINT_SIZE = 184               # :32
_lincomb = ASForHadamardProducts._lincomb


@dataclass
class LabeledPolynomial:  # ark_poly_commit::LabeledPolynomial over DensePolynomial (ext)
    coeffs: List[int]                      # little-endian coefficients, trailing zeros allowed
    degree_bound: Optional[int] = None
    hiding_bound: Optional[int] = None

    def degree(self) -> int:
        d = len(self.coeffs) - 1
        while d > 0 and self.coeffs[d] == 0:
            d -= 1
        return max(d, 0)

    def evaluate(self, fr: Fr, x: int) -> int:
        acc = 0
        for c in reversed(self.coeffs):
            acc = (acc * x + c) % fr.r
        return acc


@dataclass
class LabeledCommitment:  # LabeledCommitment<trivial_pc::Commitment<G>> (ext)
    elem: tuple                            # (xy Montgomery limbs, is_inf)
    degree_bound: Optional[int] = None


@dataclass
class InputInstance:  # data_structures.rs:11-21
    commitment: LabeledCommitment
    point: int
    eval: int

    @staticmethod
    def zero(ctx: Context) -> "InputInstance":  # data_structures.rs:24-35
        return InputInstance(LabeledCommitment((np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64), True)), 0, 0)


@dataclass
class SingleProof:  # data_structures.rs (Proof<G> = Vec<SingleProof<G>>)
    witness_commitment: LabeledCommitment
    witness_eval: int
    eval: int


@dataclass
class Input:
    instance: InputInstance
    witness: LabeledPolynomial


@dataclass
class Accumulator:
    instance: InputInstance
    witness: LabeledPolynomial


class TrivialPC:
    """ark_poly_commit::trivial_pc::TrivialPC (ext): setup / trim / commit / check."""

    @staticmethod
    def setup(ctx: Context, max_degree: int, seed: int = 0x7121A1) -> CommitterKey:
        return PedersenCommitment.setup(ctx, max_degree + 1, seed)

    @staticmethod
    def trim(pp: CommitterKey, supported_degree: int) -> Tuple[CommitterKey, CommitterKey]:
        """-> (ck, vk); both are the committer key truncated to supported_degree + 1 generators."""
        ctx = pp.ctx
        xy, inf = pp.read(0, supported_degree + 1)
        ck = CommitterKey.load(ctx, xy, inf, hiding_generator=pp.hiding_generator)
        return ck, ck

    @staticmethod
    def supported_degree(ck: CommitterKey) -> int:
        return len(ck) - 1

    @staticmethod
    def commit(ck: CommitterKey, poly: LabeledPolynomial) -> LabeledCommitment:
        """Pedersen commitment of the coefficient vector (no hiding: :145-158 rejects hiding bounds)."""
        ctx = ck.ctx
        fr = Fr(ctx.curve)
        n = min(len(poly.coeffs), len(ck))
        if n == 0:
            return LabeledCommitment((np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64), True))
        vec = ctx.upload(fr.to_limbs_many([c % fr.r for c in poly.coeffs[:n]]))
        out = PedersenCommitment.commit(ck, vec, None)
        vec.free()
        return LabeledCommitment(out)

    @staticmethod
    def commit_many(ck: CommitterKey, polys) -> list:
        """several commitments under one key in ONE library call (amsm_msm_multi_device: small keys sum them in a single launch);
        the same points as len(polys) calls of commit()"""
        ctx = ck.ctx
        fr = Fr(ctx.curve)
        out = [LabeledCommitment((np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64), True)) for _ in polys]
        jobs, which = [], []
        for k, poly in enumerate(polys):
            n = min(len(poly.coeffs), len(ck))
            if n == 0:
                continue
            jobs.append((0, ctx.upload(fr.to_limbs_many([c % fr.r for c in poly.coeffs[:n]]))))
            which.append(k)
        if jobs:
            pts, infs = VariableBaseMSM.multi_scalar_mul_multi(ck, jobs, mont=True)
            for j, k in enumerate(which):
                out[k] = LabeledCommitment((pts[j].copy(), bool(infs[j])))
            for _, v in jobs:
                v.free()
        return out

    @classmethod
    def check(cls, vk: CommitterKey, commitment: LabeledCommitment, point: int, value: int,
              polynomial: LabeledPolynomial) -> bool:
        """check_individual_opening_challenges with one commitment and opening challenge 1: the proof IS the
        polynomial: recommit and re-evaluate."""
        fr = Fr(vk.ctx.curve)
        if polynomial.degree() > cls.supported_degree(vk):
            return False
        return _pt_eq(cls.commit(vk, polynomial).elem, commitment.elem) and polynomial.evaluate(fr, point) == value % fr.r


def _poly_div_linear(fr: Fr, coeffs: Sequence[int], v: int, z: int) -> List[int]:
    """(p(X) - v) / (X - z) by synthetic division (the remainder p(z) - v is dropped, as `Div` does)."""
    n = len(coeffs)
    if n <= 1:
        return [0]
    q = [0] * (n - 1)
    carry = 0
    for i in range(n - 1, 0, -1):
        carry = (coeffs[i] + carry * z) % fr.r
        q[i - 1] = carry
    return q


class ASForTrivialPC:
    sponge_cls = Sha256Sponge

    @classmethod
    def index(cls, pp: CommitterKey, predicate_index: int):
        """:310-330 -> (prover key, verifier key = supported degree, decider key)"""
        ck, vk = TrivialPC.trim(pp, predicate_index)
        return ck, predicate_index, vk

    # ---- structure checks (:101-178) -----------------------------------------------------------------
    @staticmethod
    def _check_instance(inst: InputInstance, is_acc: bool) -> InputInstance:
        if inst.commitment.degree_bound is not None:
            raise (MalformedAccumulator if is_acc else MalformedInput)("Degree bounds on instances are unsupported.")
        return inst

    @staticmethod
    def _check_witness(w: LabeledPolynomial, pk: CommitterKey, is_acc: bool) -> LabeledPolynomial:
        err = MalformedAccumulator if is_acc else MalformedInput
        if w.degree_bound is not None:
            raise err("Degree bounds on witnesses are unsupported.")
        if w.hiding_bound is not None:
            raise err("Hiding bounds on witnesses are unsupported.")
        if w.degree() > TrivialPC.supported_degree(pk):
            raise err(f"A witness of degree {w.degree()} is unsupported for this prover key")
        return w

    # ---- sponge plumbing --------------------------------------------------------------------------------
    @staticmethod
    def _absorb_instance(sp: CryptographicSponge, fr: Fr, inst: InputInstance) -> None:  # data_structures.rs:38-55
        sp.absorb_point(inst.commitment.elem)
        sp.absorb_bytes((inst.point % fr.r).to_bytes(32, "little"))
        sp.absorb_bytes((inst.eval % fr.r).to_bytes(32, "little"))

    @classmethod
    def _challenge_point(cls, fr: Fr, sponge: CryptographicSponge, supported_degree: int, instances, wit_comms) -> int:
        sp = sponge.fork(b"")
        sp.absorb_u64(supported_degree)
        for inst, wc in zip(instances, wit_comms):
            cls._absorb_instance(sp, fr, inst)
            sp.absorb_point(wc.elem)
        return sp.squeeze_field_elements(1, CHALLENGE_POINT_SIZE)[0]

    @staticmethod
    def _lc_challenges(fr: Fr, sponge: CryptographicSponge, challenge_point: int, proof: Sequence[SingleProof]) -> List[int]:
        sp = sponge
        sp.absorb_bytes((challenge_point % fr.r).to_bytes(32, "little")[: (CHALLENGE_POINT_SIZE + 7) // 8])
        for p in proof:
            sp.absorb_bytes((p.eval % fr.r).to_bytes(32, "little"))
            sp.absorb_bytes((p.witness_eval % fr.r).to_bytes(32, "little"))
        return sp.squeeze_field_elements(2 * len(proof), LINEAR_COMBINATION_CHALLENGE_SIZE)

    # ---- prove (:332-468) ---------------------------------------------------------------------------------
    @classmethod
    def prove(cls, pk: CommitterKey, inputs: Sequence[Input], old_accumulators: Sequence[Accumulator], rng=None,
              sponge: Optional[CryptographicSponge] = None):
        ctx = pk.ctx
        fr = Fr(ctx.curve)
        sponge = sponge if sponge is not None else cls.sponge_cls()
        inputs = list(inputs)
        accs = list(old_accumulators)
        if not inputs and not accs:  # default input (:349-364)
            inputs = [Input(InputInstance.zero(ctx), LabeledPolynomial([0]))]
        instances = [cls._check_instance(i.instance, False) for i in inputs] + \
                    [cls._check_instance(a.instance, True) for a in accs]
        witnesses = [cls._check_witness(i.witness, pk, False) for i in inputs] + \
                    [cls._check_witness(a.witness, pk, True) for a in accs]
        # steps 1c-1d: witness polynomials w = (p - v) / (X - z) and their commitments (:181-222)
        wit_polys = [LabeledPolynomial(_poly_div_linear(fr, w.coeffs, inst.eval, inst.point))
                     for inst, w in zip(instances, witnesses)]
        wit_comms = TrivialPC.commit_many(pk, wit_polys)  # (the reference commits one by one, :198)
        # step 2: challenge point
        z = cls._challenge_point(fr, sponge, TrivialPC.supported_degree(pk), instances, wit_comms)
        # steps 3-4: evaluations at the challenge point, linear-combination challenges
        proof = [SingleProof(wc, wp.evaluate(fr, z), w.evaluate(fr, z)) for w, wp, wc in zip(witnesses, wit_polys, wit_comms)]
        ch = cls._lc_challenges(fr, sponge, z, proof)
        # step 5-7: combined polynomial / evaluation / commitment
        polys = witnesses + wit_polys
        width = max(len(p.coeffs) for p in polys)
        combined = [0] * width
        for c, p in zip(ch, polys):
            for i, cf in enumerate(p.coeffs):
                combined[i] = (combined[i] + c * cf) % fr.r
        combined_poly = LabeledPolynomial(combined)
        combined_eval = combined_poly.evaluate(fr, z)
        comms = [i.commitment.elem for i in instances] + [wc.elem for wc in wit_comms]
        combined_comm = _lincomb(ctx, comms, ch, fr)
        acc = Accumulator(InputInstance(LabeledCommitment(combined_comm), z, combined_eval), combined_poly)
        return acc, proof

    # ---- verify (:470-609) --------------------------------------------------------------------------------
    @classmethod
    def verify(cls, ctx: Context, vk: int, input_instances: Sequence[InputInstance],
               old_accumulator_instances: Sequence[InputInstance], new_acc: InputInstance, proof: Sequence[SingleProof],
               sponge: Optional[CryptographicSponge] = None) -> bool:
        fr = Fr(ctx.curve)
        sponge = sponge if sponge is not None else cls.sponge_cls()
        try:
            instances = [cls._check_instance(i, False) for i in input_instances] + \
                        [cls._check_instance(a, True) for a in old_accumulator_instances]
            if not instances:
                instances = [InputInstance.zero(ctx)]
            cls._check_instance(new_acc, True)
        except (MalformedInput, MalformedAccumulator):
            return False
        if len(proof) != len(instances):
            return False
        # step 4: eval - v == w(z') * (z' - z) for every claim
        for inst, p in zip(instances, proof):
            if (p.eval - inst.eval) % fr.r != p.witness_eval * (new_acc.point - inst.point) % fr.r:
                return False
        # step 3: the challenge point
        z = cls._challenge_point(fr, sponge, vk, instances, [p.witness_commitment for p in proof])
        if z != new_acc.point % fr.r:
            return False
        # steps 5-7
        ch = cls._lc_challenges(fr, sponge, z, proof)
        evals = [p.eval for p in proof] + [p.witness_eval for p in proof]
        if sum(c * e for c, e in zip(ch, evals)) % fr.r != new_acc.eval % fr.r:
            return False
        comms = [i.commitment.elem for i in instances] + [p.witness_commitment.elem for p in proof]
        return _pt_eq(_lincomb(ctx, comms, ch, fr), new_acc.commitment.elem)

    # ---- decide (:611-632) --------------------------------------------------------------------------------
    @classmethod
    def decide(cls, dk: CommitterKey, accumulator: Accumulator, sponge=None) -> bool:
        return TrivialPC.check(dk, accumulator.instance.commitment, accumulator.instance.point, accumulator.instance.eval,
                               accumulator.witness)
