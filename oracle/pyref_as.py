"""TEST INFRASTRUCTURE ONLY -- big-integer restatement of the ACCUMULATION layers above the MSM hot path.

PARITY UNPINNED (see oracle/pyref.py): the reference holds no vectors for these functions and cannot be built here.
What this file adds over pyref.py is an implementation of the reference's accumulation-prover algebra that shares
no code with the product (affine group law + Python ints; the product runs HIP kernels and a C++ host field), so the
`-m gpu` tests can compare accumulator instances, witnesses and proofs bit for bit instead of only checking that
prove/verify/decide agree with each other.

Fiat-Shamir challenges are INJECTED (the callers pass the values the product's sponge produced): the sponge is
O(#inputs) host hashing outside the accelerated path, and injecting keeps this file independent of any absorb encoding.

Reference functions restated (file:line under /root/reference):
  hp_as      generate_prover_randomness          src/hp_as/mod.rs:179-230
             compute_mu / compute_nu challenges   src/hp_as/mod.rs:233-275   (derivation from the squeezed values)
             compute_combined_hp_commitments      src/hp_as/mod.rs:409-479
             compute_combined_hp_openings         src/hp_as/mod.rs:535-607
             prove / decide                       src/hp_as/mod.rs:646-813, 894-925
  r1cs_nark_as  compute_blinded_commitments       src/r1cs_nark_as/mod.rs:220-286
             compute_hp_input_witnesses           src/r1cs_nark_as/mod.rs:316-363
             generate_prover_randomness           src/r1cs_nark_as/mod.rs:366-420
             compute_accumulator_instance_components  src/r1cs_nark_as/mod.rs:452-542
             compute_accumulator_witness_components   src/r1cs_nark_as/mod.rs:546-658
             prove / decide                       src/r1cs_nark_as/mod.rs:713-926, 1031-1112
  ipa_pc_as  combine_succinct_check_polynomials_and_commitments  src/ipa_pc_as/mod.rs:254-346
             combine_succinct_check_polynomials / evaluate_...   src/ipa_pc_as/mod.rs:391-421
Only tests/ may import this."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

from . import pyref as o

Point = o.Point


class PyOps:
    """The O(len) steps on Python-int lists (oracle/pyref.py): the default.  The config-size tests (2^18 constraints, 2^22
    elements) swap in oracle/fastref.py's NumpyOps -- the same steps over Montgomery limb arrays through the C restatement
    oracle/ark_msm.c -- with `use_ops`; the accumulation algebra in this file is the same code either way."""
    pedersen_commit = staticmethod(o.pedersen_commit)
    compute_hp = staticmethod(o.compute_hp)
    combine_vectors = staticmethod(o.combine_vectors)
    scale_vector = staticmethod(o.scale_vector)
    compute_t_vecs = staticmethod(o.compute_t_vecs)
    matrix_vec_mul = staticmethod(o.matrix_vec_mul)

    @staticmethod
    def const(c, value: int, n: int) -> List[int]:
        """`vec![value; n]` (src/hp_as/mod.rs:187-188, src/r1cs_nark_as/mod.rs:378-379)"""
        return [value % c.r] * n

    # ---- the O(len) steps of the IPA opening (ipa_open below): vectors are int lists, keys are Point lists ----
    @staticmethod
    def vlen(v) -> int:
        return len(v)

    @staticmethod
    def split(v):
        half = len(v) // 2
        return v[:half], v[half:]

    @staticmethod
    def scalar_at(c, v, i: int) -> int:
        return v[i] % c.r

    @staticmethod
    def powers(c, point: int, n: int) -> List[int]:
        out, cur = [], 1
        for _ in range(n):
            out.append(cur)
            cur = cur * point % c.r
        return out

    @staticmethod
    def inner_product(c, a, b) -> int:
        return sum(x * y for x, y in zip(a, b)) % c.r

    @staticmethod
    def axpy(c, a, coeff: int, b):
        """a + coeff * b, elementwise (the folds of the coefficient and evaluation vectors)"""
        return [(x + coeff * y) % c.r for x, y in zip(a, b)]

    @staticmethod
    def pad(c, v, n: int):
        return list(v) + [0] * (n - len(v))

    @staticmethod
    def cm_commit(c, key, v) -> Point:
        """PedersenCommitment-style `cm_commit(comm_key, scalars, None, None)`: the MSM over the first len(v) generators"""
        return o.msm_naive(c, key[:len(v)], [x % c.r for x in v])

    @staticmethod
    def fold_points(c, key_l, key_r, x: int):
        """`key_l += key_r * x`"""
        return [o.add(c, P, o.mul(c, x % c.r, Q)) for P, Q in zip(key_l, key_r)]

    @staticmethod
    def point_at(c, key, i: int) -> Point:
        return key[i]

    @staticmethod
    def check_poly_coeffs(c, xi):
        return check_poly_coeffs(c, xi)


ops = PyOps()


class use_ops:
    """with use_ops(NumpyOps(...)): ...  -- vector backend of this module for the duration of the block"""

    def __init__(self, new):
        self.new = new

    def __enter__(self):
        global ops
        self.old, ops = ops, self.new
        return self.new

    def __exit__(self, *exc):
        global ops
        ops = self.old
        return False


# --------------------------------------------------------------------------------------------------------------
# hp_as
# --------------------------------------------------------------------------------------------------------------
def hp_mu_challenges(c, squeezed: Sequence[int], num_inputs: int, make_zk: bool) -> List[int]:
    """src/hp_as/mod.rs:233-253: [1, squeezed..., (zk) mu[1] * mu[num_inputs - 1]]"""
    mu = [1] + [int(x) for x in squeezed[: num_inputs - 1]]
    if make_zk:
        mu.append(mu[1] * mu[num_inputs - 1] % c.r)
    return mu


def hp_nu_challenges(c, nu1: int, num_inputs: int) -> List[int]:
    """src/hp_as/mod.rs:256-275: powers 1, nu, nu^2, ... (2 * num_inputs - 1 of them)"""
    out, cur = [], 1
    for _ in range(2 * num_inputs - 1):
        out.append(cur)
        cur = cur * nu1 % c.r
    return out


def combine_commitments(c, comms: Sequence[Point], challenges: Sequence[int], hiding: Point = None) -> Point:
    """src/hp_as/mod.rs:391-406"""
    acc: Point = None
    for i, P in enumerate(comms):
        acc = o.add(c, acc, o.mul(c, challenges[i] % c.r, P))
    if hiding is not None:
        acc = o.add(c, acc, hiding)
    return acc


def combine_randomness(c, rands: Sequence[Optional[int]], challenges: Sequence[int], hiding: Optional[int]) -> int:
    """src/hp_as/mod.rs:515-532"""
    acc = 0
    for i, r in enumerate(rands):
        if r is not None:
            acc = (acc + r * challenges[i]) % c.r
    if hiding is not None:
        acc = (acc + hiding) % c.r
    return acc


def hp_combined_commitments(c, instances: Sequence[Tuple[Point, Point, Point]], low: Sequence[Point],
                            high: Sequence[Point], hiding_comms: Optional[Tuple[Point, Point, Point]],
                            mu: Sequence[int], nu: Sequence[int], chi: Sequence[int]) -> Tuple[Point, Point, Point]:
    """src/hp_as/mod.rs:409-479"""
    n = len(instances)
    h1 = h2 = h3 = None
    if hiding_comms is not None:
        h1 = o.mul(c, mu[n], hiding_comms[0])
        h2 = o.mul(c, mu[1], hiding_comms[1])
        h3 = o.mul(c, mu[n], hiding_comms[2])
    comm_1 = combine_commitments(c, [i[0] for i in instances], chi, h1)
    comm_2 = combine_commitments(c, [i[1] for i in reversed(instances)], nu, h2)
    low_addend = combine_commitments(c, low, nu)
    high_addend = combine_commitments(c, high, nu[n:])
    comm_3_addend = o.mul(c, nu[n - 1], combine_commitments(c, [i[2] for i in instances], mu, h3))
    comm_3 = o.add(c, o.add(c, low_addend, high_addend), comm_3_addend)
    return comm_1, comm_2, comm_3


def hp_combined_openings(c, witnesses: Sequence[dict], mu, nu, chi, hiding_vecs, hiding_rands) -> dict:
    """src/hp_as/mod.rs:535-607.  witness = {"a": [...], "b": [...], "rand": None | (r1, r2, r3)}"""
    n = len(witnesses)
    add1 = ops.scale_vector(c, hiding_vecs[0], mu[n]) if hiding_vecs is not None else None
    a_open = ops.combine_vectors(c, [w["a"] for w in witnesses], chi, add1)
    add2 = ops.scale_vector(c, hiding_vecs[1], mu[1]) if hiding_vecs is not None else None
    b_open = ops.combine_vectors(c, [w["b"] for w in reversed(witnesses)], nu, add2)
    rand = None
    if hiding_rands is not None:
        def col(k):
            return [None if w["rand"] is None else w["rand"][k] for w in witnesses]
        a_r = combine_randomness(c, col(0), chi, hiding_rands[0] * mu[n] % c.r)
        b_r = combine_randomness(c, list(reversed(col(1))), nu, hiding_rands[1] * mu[1] % c.r)
        p_r = combine_randomness(c, col(2), mu, hiding_rands[2] * mu[n] % c.r) * nu[n - 1] % c.r
        rand = (a_r, b_r, p_r)
    return {"a": a_open, "b": b_open, "rand": rand}


def hp_prove(c, gens: Sequence[Point], H: Point, inputs: Sequence[dict], accs: Sequence[dict], make_zk: bool,
             rnd: Optional[dict], mu_squeezed: Sequence[int], nu1: int, supported: Optional[int] = None) -> dict:
    """src/hp_as/mod.rs:646-813.  inputs / accs: {"inst": (c1, c2, c3), "wit": {"a","b","rand"}}.
    rnd (zk): {"a": value of the constant hiding vector a, "b": ..., "rand_1", "rand_2", "rand_3"} in the order the
    reference draws them (:187-193).  mu_squeezed: the num_all - 1 squeezed mu values; nu1: the squeezed nu."""
    inputs, accs = list(inputs), list(accs)
    num_all = len(inputs) + len(accs)
    if accs:
        hp_vec_len = len(accs[0]["wit"]["a"])
    elif inputs:
        hp_vec_len = len(inputs[0]["wit"]["a"])
    else:
        hp_vec_len = supported if supported is not None else len(gens)

    def zero_input():
        return {"inst": (None, None, None), "wit": {"a": ops.const(c, 0, hp_vec_len), "b": ops.const(c, 0, hp_vec_len), "rand": None}}
    if num_all == 0:
        inputs.append(zero_input())
        num_all += 1
    if make_zk and num_all == 1:
        inputs.append(zero_input())
        num_all += 1
    all_ = inputs + accs
    instances = [x["inst"] for x in all_]
    witnesses = [x["wit"] for x in all_]
    hiding_vecs = hiding_rands = hiding_comms = None
    if make_zk:  # generate_prover_randomness :179-230
        a = ops.const(c, rnd["a"], hp_vec_len)
        b = ops.const(c, rnd["b"], hp_vec_len)
        hiding_rands = (rnd["rand_1"], rnd["rand_2"], rnd["rand_3"])
        comm_1 = ops.pedersen_commit(c, gens, H, a, hiding_rands[0])
        comm_2 = ops.pedersen_commit(c, gens, H, b, hiding_rands[1])
        p1 = ops.compute_hp(c, a, witnesses[0]["b"])
        p2 = ops.compute_hp(c, witnesses[-1]["a"], b)
        comm_3 = ops.pedersen_commit(c, gens, H, ops.combine_vectors(c, [p1, p2], [1, 1]), hiding_rands[2])
        hiding_vecs, hiding_comms = (a, b), (comm_1, comm_2, comm_3)
    mu = hp_mu_challenges(c, mu_squeezed, num_all, make_zk)
    t = ops.compute_t_vecs(c, [w["a"] for w in witnesses], [w["b"] for w in witnesses], mu, hp_vec_len, hiding_vecs)
    low = [ops.pedersen_commit(c, gens, H, t[i], None) for i in range(num_all - 1)]
    high = [ops.pedersen_commit(c, gens, H, t[i], None) for i in range(num_all, 2 * num_all - 1)]
    nu = hp_nu_challenges(c, nu1, num_all)
    chi = [m * v % c.r for m, v in zip(mu, nu)]
    inst = hp_combined_commitments(c, instances, low, high, hiding_comms, mu, nu, chi)
    wit = hp_combined_openings(c, witnesses, mu, nu, chi, hiding_vecs, hiding_rands)
    return {"inst": inst, "wit": wit, "proof": {"low": low, "high": high, "hiding_comms": hiding_comms},
            "mu": mu, "nu": nu}


def hp_decide(c, gens, H, acc: dict) -> bool:
    """src/hp_as/mod.rs:894-925"""
    w = acc["wit"]
    r = w["rand"] if w["rand"] is not None else (None, None, None)
    prod = ops.compute_hp(c, w["a"], w["b"])
    c1 = ops.pedersen_commit(c, gens, H, w["a"], r[0])
    c2 = ops.pedersen_commit(c, gens, H, w["b"], r[1])
    c3 = ops.pedersen_commit(c, gens, H, prod, r[2])
    return (c1, c2, c3) == tuple(acc["inst"])


# --------------------------------------------------------------------------------------------------------------
# r1cs_nark_as
# --------------------------------------------------------------------------------------------------------------
def nark_as_blinded_commitments(c, input_instances: Sequence[dict], gammas: Sequence[Optional[int]]):
    """src/r1cs_nark_as/mod.rs:220-286.  instance = {"r1cs_input", "first_msg": {comm_a, comm_b, comm_c, randomness}};
    gammas[i] = compute_challenge of instance i (None / ignored when it has no randomness).
    NOTE the reference starts comm_prod from comm_c (:236)."""
    A, B, Cc, P = [], [], [], []
    for inst, g in zip(input_instances, gammas):
        m = inst["first_msg"]
        a, b, cc, prod = m["comm_a"], m["comm_b"], m["comm_c"], m["comm_c"]
        r = m.get("randomness")
        if r is not None:
            g %= c.r
            a = o.add(c, a, o.mul(c, g, r["comm_r_a"]))
            b = o.add(c, b, o.mul(c, g, r["comm_r_b"]))
            cc = o.add(c, cc, o.mul(c, g, r["comm_r_c"]))
            prod = o.add(c, o.add(c, prod, o.mul(c, g, r["comm_1"])), o.mul(c, g * g % c.r, r["comm_2"]))
        A.append(a)
        B.append(b)
        Cc.append(cc)
        P.append(prod)
    return A, B, Cc, P


def nark_as_beta_challenges(squeezed: Sequence[int], num: int) -> List[int]:
    """src/r1cs_nark_as/mod.rs:423-448: [1, squeezed...]"""
    return [1] + [int(x) for x in squeezed[: num - 1]]


def nark_as_instance_components(c, input_instances, A, B, Cc, acc_instances, beta, proof_randomness):
    """src/r1cs_nark_as/mod.rs:452-542: accumulators first, then the blinded inputs, then the prover's randomness"""
    r1cs_inputs = [a["r1cs_input"] for a in acc_instances] + [i["r1cs_input"] for i in input_instances]
    ca = [a["comm_a"] for a in acc_instances] + list(A)
    cb = [a["comm_b"] for a in acc_instances] + list(B)
    cc = [a["comm_c"] for a in acc_instances] + list(Cc)
    if proof_randomness is not None:
        r1cs_inputs.append(proof_randomness["r1cs_r_input"])
        ca.append(proof_randomness["comm_r_a"])
        cb.append(proof_randomness["comm_r_b"])
        cc.append(proof_randomness["comm_r_c"])
    assert len(ca) <= len(beta)
    return (o.combine_vectors(c, r1cs_inputs, beta), combine_commitments(c, ca, beta), combine_commitments(c, cb, beta),
            combine_commitments(c, cc, beta))


def nark_as_witness_components(c, input_witnesses, acc_witnesses, beta, prover_witness_randomness):
    """src/r1cs_nark_as/mod.rs:546-658.  input witness = {"blinded_witness", "randomness": None | (sa, sb, sc, so)} (the
    NARK's SecondRoundMessageRandomness, r1cs_nark/data_structures.rs:150-167: the beta combination takes sigma_a, sigma_b,
    sigma_c; sigma_o only feeds the Hadamard-product witness);
    accumulator witness = {"r1cs_blinded_witness", "randomness": None | (sa, sb, sc)};
    prover_witness_randomness = None | (r_witness, rand_1, rand_2, rand_3)."""
    wits = [w["r1cs_blinded_witness"] for w in acc_witnesses] + [w["blinded_witness"] for w in input_witnesses]
    sig = [w["randomness"] for w in acc_witnesses] + [w["randomness"] for w in input_witnesses]
    cols = [[None if s is None else s[k] for s in sig] for k in range(3)]
    if prover_witness_randomness is not None:
        wits.append(prover_witness_randomness[0])
        for k in range(3):
            cols[k].append(prover_witness_randomness[1 + k])
    blinded = ops.combine_vectors(c, wits, beta)
    rand = None
    if prover_witness_randomness is not None:
        rand = tuple(combine_randomness(c, cols[k], beta, None) for k in range(3))
    return blinded, rand


def nark_as_prove(c, A_m, B_m, C_m, gens, H, num_input: int, num_witness: int, inputs: Sequence[dict],
                  accs: Sequence[dict], make_zk: bool, rnd: Optional[dict], chal: dict) -> dict:
    """src/r1cs_nark_as/mod.rs:713-926.
    inputs: {"inst": {"r1cs_input", "first_msg"}, "wit": {"blinded_witness", "randomness": None | (sa, sb, sc, so)}};
    accs:   {"inst": {"r1cs_input", "comm_a", "comm_b", "comm_c", "hp_instance"}, "wit": {"r1cs_blinded_witness",
             "hp_witness", "randomness"}}.
    rnd (zk), in the order the reference draws (:378-384, then the nested hp_as :187-193): r_input, r_witness (values of
    the CONSTANT vectors `vec![rand; len]`), rand_1..3, then "hp": the nested scheme's rnd.
    chal: {"gammas": [...], "hp_mu": [...], "hp_nu": int, "beta": [...]} -- the squeezed values."""
    inputs, accs = list(inputs), list(accs)
    if not inputs and not accs:  # default input :761-768
        zero_msg = {"comm_a": None, "comm_b": None, "comm_c": None, "randomness": None}
        inputs.append({"inst": {"r1cs_input": [0] * num_input, "first_msg": zero_msg},
                       "wit": {"blinded_witness": ops.const(c, 0, num_witness), "randomness": None}})
    proof_randomness = prover_wit_rand = None
    if make_zk:  # generate_prover_randomness :366-420
        r_in = [rnd["r_input"] % c.r] * num_input
        r_wit = ops.const(c, rnd["r_witness"], num_witness)
        r1, r2, r3 = rnd["rand_1"], rnd["rand_2"], rnd["rand_3"]
        proof_randomness = {
            "r1cs_r_input": r_in,
            "comm_r_a": ops.pedersen_commit(c, gens, H, ops.matrix_vec_mul(c, A_m, r_in, r_wit), r1),
            "comm_r_b": ops.pedersen_commit(c, gens, H, ops.matrix_vec_mul(c, B_m, r_in, r_wit), r2),
            "comm_r_c": ops.pedersen_commit(c, gens, H, ops.matrix_vec_mul(c, C_m, r_in, r_wit), r3),
        }
        prover_wit_rand = (r_wit, r1, r2, r3)
    in_insts = [x["inst"] for x in inputs]
    acc_insts = [x["inst"] for x in accs]
    A, B, Cc, P = nark_as_blinded_commitments(c, in_insts, chal["gammas"])
    hp_inputs = []
    for x, a, b, p in zip(inputs, A, B, P):  # compute_hp_input_instances / _witnesses :289-363
        w = x["wit"]
        a_vec = ops.matrix_vec_mul(c, A_m, x["inst"]["r1cs_input"], w["blinded_witness"])
        b_vec = ops.matrix_vec_mul(c, B_m, x["inst"]["r1cs_input"], w["blinded_witness"])
        s = w["randomness"]  # HPInputWitnessRandomness{rand_1: sigma_a, rand_2: sigma_b, rand_3: sigma_o}  (:341-352)
        hp_inputs.append({"inst": (a, b, p), "wit": {"a": a_vec, "b": b_vec, "rand": None if s is None else (s[0], s[1], s[3])}})
    hp_accs = [{"inst": x["inst"]["hp_instance"], "wit": x["wit"]["hp_witness"]} for x in accs]
    hp = hp_prove(c, gens, H, hp_inputs, hp_accs, make_zk, rnd["hp"] if make_zk else None, chal["hp_mu"], chal["hp_nu"])
    num_addends = len(in_insts) + len(acc_insts) + (1 if make_zk else 0)
    beta = nark_as_beta_challenges(chal["beta"], num_addends)
    r1cs_input, ca, cb, cc = nark_as_instance_components(c, in_insts, A, B, Cc, acc_insts, beta, proof_randomness)
    blinded, rand = nark_as_witness_components(c, [x["wit"] for x in inputs], [x["wit"] for x in accs], beta, prover_wit_rand)
    return {
        "inst": {"r1cs_input": r1cs_input, "comm_a": ca, "comm_b": cb, "comm_c": cc, "hp_instance": hp["inst"]},
        "wit": {"r1cs_blinded_witness": blinded, "hp_witness": hp["wit"], "randomness": rand},
        "proof": {"hp_proof": hp["proof"], "randomness": proof_randomness},
        "beta": beta,
    }


def nark_as_decide(c, A_m, B_m, C_m, gens, H, acc: dict) -> bool:
    """src/r1cs_nark_as/mod.rs:1031-1112"""
    inst, wit = acc["inst"], acc["wit"]
    s = wit["randomness"] if wit["randomness"] is not None else (None, None, None)
    za, zb, zc = (ops.matrix_vec_mul(c, M, inst["r1cs_input"], wit["r1cs_blinded_witness"]) for M in (A_m, B_m, C_m))
    ok = (ops.pedersen_commit(c, gens, H, za, s[0]) == inst["comm_a"] and
          ops.pedersen_commit(c, gens, H, zb, s[1]) == inst["comm_b"] and
          ops.pedersen_commit(c, gens, H, zc, s[2]) == inst["comm_c"])
    return ok and hp_decide(c, gens, H, {"inst": inst["hp_instance"], "wit": wit["hp_witness"]})


# --------------------------------------------------------------------------------------------------------------
# ipa_pc_as
# --------------------------------------------------------------------------------------------------------------
def check_poly_coeffs(c, xi: Sequence[int]) -> List[int]:
    """SuccinctCheckPolynomial::compute_coeffs (ark-poly-commit ipa_pc, ext; call site src/ipa_pc_as/mod.rs:400):
    coefficients of prod_{i=1..k} (1 + xi_i X^(2^(k-i)))."""
    k = len(xi)
    coeffs = [1]
    for i in range(k):  # multiply by (1 + xi_{k-i} X^(2^i)): the LAST challenge carries X^1
        x = xi[k - 1 - i] % c.r
        coeffs = coeffs + [v * x % c.r for v in coeffs]
    return coeffs


def check_poly_evaluate(c, xi: Sequence[int], point: int) -> int:
    """SuccinctCheckPolynomial::evaluate (ext): prod_i (1 + xi_i * point^(2^(k-i)))"""
    k = len(xi)
    out, p = 1, point % c.r
    for i in range(k):
        out = out * (1 + xi[k - 1 - i] * p) % c.r
        p = p * p % c.r
    return out


def ipa_open(c, comm_key, h_gen: Point, s_gen: Point, polynomial, commitment: Point, point: int, challenges: Sequence[int],
             hiding: Optional[dict] = None) -> dict:
    """`InnerProductArgPC::open_individual_opening_challenges` for ONE polynomial with opening challenge 1 (ark-poly-commit
    ipa_pc, branch accumulation-experimental: ext -- restated from the published construction, BCMS20 section 7; the call
    site is src/ipa_pc_as/mod.rs:454, the accumulation prover's opening of the combined check polynomial).

    comm_key: the d + 1 generators (a power of two); polynomial: its coefficients (<= d + 1, low degree first).
    challenges: the Fiat-Shamir values IN THE ORDER THE OPENING DRAWS THEM -- [hiding challenge (only with `hiding`), the
    first round challenge (the multiplier of h), then one per round] -- injected like every challenge in this file.
    hiding = None | {"polynomial": the random polynomial as drawn (before its value at `point` is taken out), "rand": its
    commitment's randomness, "poly_rand": the randomness of `commitment`}.
    Per round, over the CURRENT halves:  L = <c_r, key_l> + <c_r, z_l> h',  R = <c_l, key_r> + <c_l, z_r> h',
    then  c_l += x^-1 c_r,  z_l += x z_r,  key_l += x key_r  -- the key is folded EVERY round, the definition; the product
    expresses the rounds over the original key or folds only the first few, with the same points.
    -> {"l_vec", "r_vec", "final_comm_key", "c", "hiding_comm", "rand"}"""
    n = ops.vlen(comm_key)
    assert n & (n - 1) == 0 and ops.vlen(polynomial) <= n
    ch = list(challenges)
    coeffs = ops.pad(c, polynomial, n)
    z = ops.powers(c, point % c.r, n)
    combined_comm = commitment
    combined_v = ops.inner_product(c, coeffs, z)
    hiding_comm, proof_rand = None, None
    if hiding is not None:
        hp = ops.pad(c, hiding["polynomial"], n)
        hv = ops.inner_product(c, hp, z)
        hp = ops.axpy(c, hp, (-hv) % c.r, ops.pad(c, ops.powers(c, 0, 1), n))  # - hv on the constant term: hp(point) = 0
        hiding_comm = o.add(c, ops.cm_commit(c, comm_key, hp), o.mul(c, hiding["rand"] % c.r, s_gen))
        hch = ch.pop(0) % c.r
        coeffs = ops.axpy(c, coeffs, hch, hp)
        proof_rand = (hiding["poly_rand"] + hch * hiding["rand"]) % c.r
        combined_comm = o.add(c, o.add(c, combined_comm, o.mul(c, hch, hiding_comm)), o.mul(c, (-proof_rand) % c.r, s_gen))
    h_prime = o.mul(c, ch.pop(0) % c.r, h_gen)
    key = comm_key
    l_vec, r_vec = [], []
    x = None
    while n > 1:
        c_l, c_r = ops.split(coeffs)
        z_l, z_r = ops.split(z)
        k_l, k_r = ops.split(key)
        l_vec.append(o.add(c, ops.cm_commit(c, k_l, c_r), o.mul(c, ops.inner_product(c, c_r, z_l), h_prime)))
        r_vec.append(o.add(c, ops.cm_commit(c, k_r, c_l), o.mul(c, ops.inner_product(c, c_l, z_r), h_prime)))
        x = ch.pop(0) % c.r
        coeffs = ops.axpy(c, c_l, pow(x, -1, c.r), c_r)
        z = ops.axpy(c, z_l, x, z_r)
        key = ops.fold_points(c, k_l, k_r, x)
        n //= 2
    assert not ch, "more challenges than the opening draws"
    return {"l_vec": l_vec, "r_vec": r_vec, "final_comm_key": ops.point_at(c, key, 0), "c": ops.scalar_at(c, coeffs, 0),
            "hiding_comm": hiding_comm, "rand": proof_rand, "combined_v": combined_v}


def ipa_check(c, comm_key, h_gen: Point, s_gen: Point, commitment: Point, point: int, value: int, proof: dict,
              challenges: Sequence[int]) -> bool:
    """`InnerProductArgPC::check` for one commitment (ext; call site src/ipa_pc_as/mod.rs:836, the decider): the succinct check
    -- round commitment  C' + sum_j (x_j^-1 L_j + x_j R_j)  against  c final_key + c h(point) h'  -- and THE (d + 1)-point MSM
    of the check polynomial's coefficients against the proof's final key.  challenges as in ipa_open."""
    ch = list(challenges)
    combined = commitment
    if proof["hiding_comm"] is not None:
        hch = ch.pop(0) % c.r
        combined = o.add(c, o.add(c, combined, o.mul(c, hch, proof["hiding_comm"])), o.mul(c, (-proof["rand"]) % c.r, s_gen))
    h_prime = o.mul(c, ch.pop(0) % c.r, h_gen)
    round_comm = o.add(c, combined, o.mul(c, value % c.r, h_prime))
    xs = [x % c.r for x in ch]
    if len(xs) != len(proof["l_vec"]) or len(xs) != len(proof["r_vec"]):
        return False
    for L, R, x in zip(proof["l_vec"], proof["r_vec"], xs):
        round_comm = o.add(c, round_comm, o.add(c, o.mul(c, pow(x, -1, c.r), L), o.mul(c, x, R)))
    v_prime = check_poly_evaluate(c, xs, point) * proof["c"] % c.r
    check_comm = o.add(c, o.mul(c, proof["c"] % c.r, proof["final_comm_key"]), o.mul(c, v_prime, h_prime))
    if round_comm != check_comm:
        return False
    return ops.cm_commit(c, comm_key, ops.check_poly_coeffs(c, xs)) == proof["final_comm_key"]


def ipa_as_combine(c, final_comm_keys: Sequence[Point], chal: Sequence[int], s_gen: Point,
                   randomness: Optional[dict]) -> Tuple[Point, Point]:
    """combine_succinct_check_polynomials_and_commitments, src/ipa_pc_as/mod.rs:254-346 (the point algebra):
    -> (combined_commitment, randomized_combined_commitment).  randomness = None | {"lin_comm", "commitment_randomness"}."""
    comb: Point = randomness["lin_comm"] if randomness is not None else None
    for P, a in zip(final_comm_keys, chal):
        comb = o.add(c, comb, o.mul(c, a % c.r, P))
    rand = comb
    if randomness is not None:
        rand = o.add(c, comb, o.mul(c, randomness["commitment_randomness"] % c.r, s_gen))
    return comb, rand


def ipa_as_combined_polynomial(c, xis: Sequence[Sequence[int]], chal: Sequence[int], lin: Optional[Sequence[int]]) -> List[int]:
    """combine_succinct_check_polynomials, src/ipa_pc_as/mod.rs:391-404"""
    out = list(lin) if lin is not None else []
    for xi, a in zip(xis, chal):
        co = check_poly_coeffs(c, xi)
        if len(co) > len(out):
            out += [0] * (len(co) - len(out))
        for i, v in enumerate(co):
            out[i] = (out[i] + a * v) % c.r
    return out


def ipa_as_evaluate_combined(c, xis, chal, point: int, lin: Optional[Sequence[int]]) -> int:
    """evaluate_combined_succinct_check_polynomials, src/ipa_pc_as/mod.rs:407-421"""
    ev = 0
    if lin is not None:
        co = list(lin) + [0, 0]
        ev = (co[0] + co[1] * point) % c.r
    for xi, a in zip(xis, chal):
        ev = (ev + check_poly_evaluate(c, xi, point) * a) % c.r
    return ev
