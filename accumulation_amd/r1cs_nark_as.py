"""Host-side mirror of `ASForR1CSNark` (reference: src/r1cs_nark_as/mod.rs): prove :713-926, verify :928-1029,
decide :1031-1112, with the SpMVs, the witness linear combinations, the nested Hadamard-product
accumulation and every Pedersen commitment on the GPU; O(#inputs) point / challenge algebra on the host."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from .engine import FrVector, PedersenCommitment
from .hp_as import (ASError, ASForHadamardProducts, Accumulator as HPAccumulator, InputInstance as HPInputInstance,
                    InputWitness as HPInputWitness, InputWitnessRandomness as HPInputWitnessRandomness,
                    MalformedAccumulator, MalformedInput, MissingRng, Proof as HPProof, _pt_eq, combine_vectors)
from .r1cs_nark import (FirstRoundMessage, IndexProverKey, SecondRoundMessage, compute_challenge, hash_matrices,
                        matrix_vec_mul)
from .scalar_field import Fr
from .sponge import CryptographicSponge, Sha256Sponge

PROTOCOL_NAME = b"AS-FOR-R1CS-NARK-2020"  # :37
HP_AS_PROTOCOL_NAME = b"AS-FOR-HP-2020"   # src/hp_as
NARK_PROTOCOL_NAME = b"R1CS-NARK-2020"
CHALLENGE_SIZE = 128                      # :41


@dataclass
class InputInstance:  # data_structures.rs:78-99
    r1cs_input: List[int]
    first_round_message: FirstRoundMessage

    @staticmethod
    def zero(ctx, input_len, make_zk):
        return InputInstance([0] * input_len, FirstRoundMessage.zero(ctx, make_zk))

    def absorb_into(self, sponge, fr):
        sponge.absorb_bytes(b"".join((x % fr.r).to_bytes(32, "little") for x in self.r1cs_input))
        self.first_round_message.absorb_into(sponge)


InputWitness = SecondRoundMessage  # data_structures.rs:125


@dataclass
class AccumulatorInstance:  # data_structures.rs:156-171
    r1cs_input: List[int]
    comm_a: tuple
    comm_b: tuple
    comm_c: tuple
    hp_instance: HPInputInstance

    def absorb_into(self, sponge, fr):
        sponge.absorb_bytes(b"".join((x % fr.r).to_bytes(32, "little") for x in self.r1cs_input))
        for p in (self.comm_a, self.comm_b, self.comm_c):
            sponge.absorb_point(p)
        self.hp_instance.absorb_into(sponge)


@dataclass
class AccumulatorWitnessRandomness:
    sigma_a: int
    sigma_b: int
    sigma_c: int


@dataclass
class AccumulatorWitness:  # data_structures.rs:218-227
    r1cs_blinded_witness: FrVector
    hp_witness: HPInputWitness
    randomness: Optional[AccumulatorWitnessRandomness] = None


@dataclass
class ProofRandomness:  # data_structures.rs:250-262
    r1cs_r_input: List[int]
    comm_r_a: tuple
    comm_r_b: tuple
    comm_r_c: tuple


@dataclass
class Proof:
    hp_proof: HPProof
    randomness: Optional[ProofRandomness] = None


@dataclass
class Accumulator:
    instance: AccumulatorInstance
    witness: AccumulatorWitness


@dataclass
class Input:
    instance: InputInstance
    witness: SecondRoundMessage


@dataclass
class ProverKey:  # data_structures.rs:27-34
    nark_pk: IndexProverKey
    as_matrices_hash: bytes


@dataclass
class VerifierKey:  # data_structures.rs:37-50
    num_instance_variables: int
    num_constraints: int
    nark_matrices_hash: bytes
    as_matrices_hash: bytes


class ASForR1CSNark:
    @staticmethod
    def index(ipk: IndexProverKey):
        """:664-711 -> (ProverKey, VerifierKey, DeciderKey = the NARK index key)"""
        as_hash = hash_matrices(PROTOCOL_NAME, ipk.a, ipk.b, ipk.c)
        info = ipk.index_info
        return (ProverKey(ipk, as_hash),
                VerifierKey(info.num_instance_variables, info.num_constraints, info.matrices_hash, as_hash), ipk)

    # sponge forks :112-125
    @staticmethod
    def _sponges(sponge):
        return sponge.fork(NARK_PROTOCOL_NAME), sponge.fork(PROTOCOL_NAME), sponge.fork(HP_AS_PROTOCOL_NAME)

    # ---- structure checks :128-217 ----------------------------------------------------------------
    @staticmethod
    def _check_input_instance(inst: InputInstance, r1cs_input_len: int):
        if len(inst.r1cs_input) != r1cs_input_len:
            raise MalformedInput("All R1CS input lengths must be equal and supported by the index prover key.")

    @staticmethod
    def _check_input(inp: Input, r1cs_input_len: int, r1cs_witness_len: int):
        ASForR1CSNark._check_input_instance(inp.instance, r1cs_input_len)
        if inp.witness.blinded_witness.n != r1cs_witness_len:
            raise MalformedInput("All R1CS witness lengths must be equal and supported by the index prover key.")
        if (inp.instance.first_round_message.randomness is None) != (inp.witness.randomness is None):
            raise MalformedInput("The existence of the first round message randomness and the second round message "
                                 "randomness must be equal.")

    @staticmethod
    def _check_acc_instance(inst: AccumulatorInstance, r1cs_input_len: int):
        if len(inst.r1cs_input) != r1cs_input_len:
            raise MalformedAccumulator("All R1CS input lengths must be equal and supported by the index prover key.")

    @staticmethod
    def _check_acc_witness(w: AccumulatorWitness, r1cs_witness_len: int):
        if w.r1cs_blinded_witness.n != r1cs_witness_len:
            raise MalformedAccumulator("All R1CS witness lengths must be equal and supported by the index prover key.")

    # ---- shared by prover and verifier --------------------------------------------------------------
    @staticmethod
    def _compute_blinded_commitments(ctx, fr, nark_matrices_hash, input_instances, nark_sponge):
        """:220-286"""
        # the four combinations of every input that carries randomness are independent: one batched library call
        jobs, job_of = [], {}
        for k, inst in enumerate(input_instances):
            m = inst.first_round_message
            if m.randomness is None:
                continue
            g = compute_challenge(fr, nark_matrices_hash, inst.r1cs_input, m, nark_sponge.fork(b""))
            r = m.randomness
            job_of[k] = len(jobs)
            jobs += [([m.comm_a, r.comm_r_a], [1, g]), ([m.comm_b, r.comm_r_b], [1, g]), ([m.comm_c, r.comm_r_c], [1, g]),
                     ([m.comm_c, r.comm_1, r.comm_2], [1, g, g * g % fr.r])]
        res = ASForHadamardProducts._lincomb_batch(ctx, jobs, fr)
        A, B, Cc, P = [], [], [], []
        for k, inst in enumerate(input_instances):
            m = inst.first_round_message
            j = job_of.get(k)
            A.append(m.comm_a if j is None else res[j])
            B.append(m.comm_b if j is None else res[j + 1])
            Cc.append(m.comm_c if j is None else res[j + 2])
            P.append(m.comm_c if j is None else res[j + 3])
        return A, B, Cc, P

    @staticmethod
    def _compute_beta_challenges(fr, num, as_matrices_hash, acc_instances, input_instances, proof_randomness, as_sponge):
        """:423-448"""
        s = as_sponge
        s.absorb_bytes(as_matrices_hash)
        s.absorb_len(len(acc_instances))
        for a in acc_instances:
            a.absorb_into(s, fr)
        s.absorb_len(len(input_instances))
        for i in input_instances:
            i.absorb_into(s, fr)
        if proof_randomness is None:
            s.absorb_bytes(b"\x00")
        else:
            # `Option<ProofRandomness>`: the tag is an item of its own (one sponge element), then `to_bytes!(r1cs_r_input)` packed by itself
            s.absorb_bytes(b"\x01")
            s.absorb_bytes(b"".join((x % fr.r).to_bytes(32, "little") for x in proof_randomness.r1cs_r_input))
            for p in (proof_randomness.comm_r_a, proof_randomness.comm_r_b, proof_randomness.comm_r_c):
                s.absorb_point(p)
        return [1] + s.squeeze_field_elements(num - 1, CHALLENGE_SIZE)

    @staticmethod
    def _instance_components(ctx, fr, input_instances, A, B, Cc, acc_instances, beta, proof_randomness):
        """:452-542: accumulators first, then (blinded) inputs, then the prover's randomness"""
        r1cs_inputs = [a.r1cs_input for a in acc_instances] + [i.r1cs_input for i in input_instances]
        ca = [a.comm_a for a in acc_instances] + list(A)
        cb = [a.comm_b for a in acc_instances] + list(B)
        cc = [a.comm_c for a in acc_instances] + list(Cc)
        if proof_randomness is not None:
            r1cs_inputs.append(proof_randomness.r1cs_r_input)
            ca.append(proof_randomness.comm_r_a)
            cb.append(proof_randomness.comm_r_b)
            cc.append(proof_randomness.comm_r_c)
        assert len(ca) <= len(beta)
        n_in = max(len(v) for v in r1cs_inputs)
        combined = [0] * n_in
        for j, v in enumerate(r1cs_inputs):
            for i, x in enumerate(v):
                combined[i] = (combined[i] + beta[j] * x) % fr.r
        oa, ob, oc = ASForHadamardProducts._lincomb_batch(ctx, [(ca, beta), (cb, beta), (cc, beta)], fr)
        return combined, oa, ob, oc

    # ---- prove ------------------------------------------------------------------------------------
    @classmethod
    def prove(cls, pk: ProverKey, inputs: Sequence[Input], old_accumulators: Sequence[Accumulator], rng=None,
              sponge: Optional[CryptographicSponge] = None):
        ipk = pk.nark_pk
        ctx = ipk.ck.ctx
        fr = Fr(ctx.curve)
        sponge = sponge if sponge is not None else Sha256Sponge()
        nark_sponge, as_sponge, hp_sponge = cls._sponges(sponge)
        info = ipk.index_info
        in_len = info.num_instance_variables
        wit_len = info.num_variables - in_len
        old_accumulators = list(old_accumulators)
        inputs = list(inputs)
        for acc in old_accumulators:
            cls._check_acc_instance(acc.instance, in_len)
            cls._check_acc_witness(acc.witness, wit_len)
        for inp in inputs:
            cls._check_input(inp, in_len, wit_len)
        if not inputs and not old_accumulators:  # default input :761-768
            z = np.zeros(4, dtype=np.uint64)
            inputs.append(Input(InputInstance.zero(ctx, in_len, False), SecondRoundMessage(ctx.fill(z, wit_len), None)))
        make_zk = rng is not None
        if not make_zk:
            if any(i.witness.randomness is not None for i in inputs):
                raise MissingRng("Accumulating inputs with hiding requires rng.")
            if any(a.witness.randomness is not None for a in old_accumulators):
                raise MissingRng("Accumulating accumulators with hiding requires rng.")
        # step 4 (:793-811, generate_prover_randomness :366-420): constant vectors, 3 SpMV + 3 commits
        proof_randomness = prover_wit_rand = None
        if make_zk:
            r_in_val, r_wit_val = rng.field(), rng.field()
            r1, r2, r3 = rng.field(), rng.field(), rng.field()
            d_rin = ctx.fill(fr.to_limbs(r_in_val), in_len)
            d_rwit = ctx.fill(fr.to_limbs(r_wit_val), wit_len)
            cra, crb, crc = PedersenCommitment.commit_batch(
                ipk.ck, [matrix_vec_mul(m, d_rin, d_rwit) for m in (ipk.a, ipk.b, ipk.c)],
                [fr.to_limbs(r1), fr.to_limbs(r2), fr.to_limbs(r3)])
            proof_randomness = ProofRandomness([r_in_val] * in_len, cra, crb, crc)
            prover_wit_rand = (d_rwit, r1, r2, r3)
        input_instances = [i.instance for i in inputs]
        acc_instances = [a.instance for a in old_accumulators]
        # steps 1-2
        A, B, Cc, P = cls._compute_blinded_commitments(ctx, fr, info.matrices_hash, input_instances, nark_sponge)
        hp_inputs = []
        for inst, inp, a, b, p in zip(input_instances, inputs, A, B, P):  # compute_hp_input_witnesses :316-363
            d_in = ctx.upload(fr.to_limbs_many(inst.r1cs_input))
            a_vec = matrix_vec_mul(ipk.a, d_in, inp.witness.blinded_witness)
            b_vec = matrix_vec_mul(ipk.b, d_in, inp.witness.blinded_witness)
            rnd = inp.witness.randomness
            hp_rnd = None if rnd is None else HPInputWitnessRandomness(rnd.sigma_a, rnd.sigma_b, rnd.sigma_o)
            hp_inputs.append(HPAccumulator(HPInputInstance(a, b, p), HPInputWitness(a_vec, b_vec, hp_rnd)))
        hp_accs = [HPAccumulator(a.instance.hp_instance, a.witness.hp_witness) for a in old_accumulators]
        # step 3: nested Hadamard-product accumulation over the NARK's committer key
        hp_acc, hp_proof = ASForHadamardProducts.prove(ipk.ck, hp_inputs, hp_accs, rng, hp_sponge)
        # step 5
        num_addends = len(input_instances) + len(acc_instances) + (1 if make_zk else 0)
        beta = cls._compute_beta_challenges(fr, num_addends, pk.as_matrices_hash, acc_instances, input_instances,
                                            proof_randomness, as_sponge)
        # step 6
        r1cs_input, ca, cb, cc = cls._instance_components(ctx, fr, input_instances, A, B, Cc, acc_instances, beta,
                                                          proof_randomness)
        acc_instance = AccumulatorInstance(r1cs_input, ca, cb, cc, hp_acc.instance)
        # step 7 (:546-658)
        wits = [a.witness.r1cs_blinded_witness for a in old_accumulators] + [i.witness.blinded_witness for i in inputs]
        sig = [a.witness.randomness for a in old_accumulators] + [i.witness.randomness for i in inputs]
        sa = [None if s is None else s.sigma_a for s in sig]
        sb = [None if s is None else s.sigma_b for s in sig]
        sc = [None if s is None else s.sigma_c for s in sig]
        if make_zk:
            wits.append(prover_wit_rand[0])
            sa.append(prover_wit_rand[1])
            sb.append(prover_wit_rand[2])
            sc.append(prover_wit_rand[3])
        blinded = combine_vectors(ctx, wits, fr.to_limbs_many(beta[: len(wits)]))
        randomness = None
        if make_zk:
            comb = ASForHadamardProducts._combine_randomness
            randomness = AccumulatorWitnessRandomness(comb(fr, sa, beta, None), comb(fr, sb, beta, None),
                                                      comb(fr, sc, beta, None))
        acc = Accumulator(acc_instance, AccumulatorWitness(blinded, hp_acc.witness, randomness))
        return acc, Proof(hp_proof, proof_randomness)

    # ---- verify -------------------------------------------------------------------------------------
    @classmethod
    def verify(cls, ctx, vk: VerifierKey, input_instances, old_accumulator_instances, new_acc: AccumulatorInstance,
               proof: Proof, sponge: Optional[CryptographicSponge] = None) -> bool:
        fr = Fr(ctx.curve)
        sponge = sponge if sponge is not None else Sha256Sponge()
        nark_sponge, as_sponge, hp_sponge = cls._sponges(sponge)
        make_zk = proof.randomness is not None
        in_len = vk.num_instance_variables
        ins = list(input_instances)
        olds = list(old_accumulator_instances)
        try:
            for i in ins:
                cls._check_input_instance(i, in_len)
            for a in olds:
                cls._check_acc_instance(a, in_len)
        except ASError:
            return False
        if not ins and not olds:
            ins.append(InputInstance.zero(ctx, in_len, False))
        A, B, Cc, P = cls._compute_blinded_commitments(ctx, fr, vk.nark_matrices_hash, ins, nark_sponge)
        hp_ins = [HPInputInstance(a, b, p) for a, b, p in zip(A, B, P)]
        hp_ok = ASForHadamardProducts.verify(ctx, vk.num_constraints, hp_ins, [a.hp_instance for a in olds],
                                             new_acc.hp_instance, proof.hp_proof, hp_sponge)
        num_addends = len(ins) + len(olds) + (1 if make_zk else 0)
        beta = cls._compute_beta_challenges(fr, num_addends, vk.as_matrices_hash, olds, ins, proof.randomness, as_sponge)
        r1cs_input, ca, cb, cc = cls._instance_components(ctx, fr, ins, A, B, Cc, olds, beta, proof.randomness)
        return bool(hp_ok and [x % fr.r for x in r1cs_input] == [x % fr.r for x in new_acc.r1cs_input]
                    and _pt_eq(ca, new_acc.comm_a) and _pt_eq(cb, new_acc.comm_b) and _pt_eq(cc, new_acc.comm_c))

    # ---- decide -------------------------------------------------------------------------------------
    @classmethod
    def decide(cls, dk: IndexProverKey, accumulator: Accumulator, sponge=None) -> bool:
        ctx = dk.ck.ctx
        fr = Fr(ctx.curve)
        inst, wit = accumulator.instance, accumulator.witness
        in_len = dk.index_info.num_instance_variables
        wit_len = dk.index_info.num_variables - in_len
        try:
            cls._check_acc_instance(inst, in_len)
            cls._check_acc_witness(wit, wit_len)
        except ASError:
            return False
        d_in = ctx.upload(fr.to_limbs_many(inst.r1cs_input))
        za = matrix_vec_mul(dk.a, d_in, wit.r1cs_blinded_witness)
        zb = matrix_vec_mul(dk.b, d_in, wit.r1cs_blinded_witness)
        zc = matrix_vec_mul(dk.c, d_in, wit.r1cs_blinded_witness)
        rnd = wit.randomness
        ca, cb, cc = PedersenCommitment.commit_batch(
            dk.ck, [za, zb, zc], [None if rnd is None else fr.to_limbs(v) for v in
                                  ((rnd.sigma_a, rnd.sigma_b, rnd.sigma_c) if rnd is not None else (0, 0, 0))])
        comm_check = _pt_eq(ca, inst.comm_a) and _pt_eq(cb, inst.comm_b) and _pt_eq(cc, inst.comm_c)
        return bool(comm_check and ASForHadamardProducts.decide(dk.ck, HPAccumulator(inst.hp_instance, wit.hp_witness), None))
