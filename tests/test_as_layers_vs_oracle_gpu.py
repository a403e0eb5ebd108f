"""Accumulation LAYERS against the big-int oracle (oracle/pyref_as.py), bit for bit: the product's prove() of
ASForHadamardProducts (src/hp_as/mod.rs:646-813), ASForR1CSNark (src/r1cs_nark_as/mod.rs:713-926: blinded commitments
:220-286, beta combinations :452-658) and the combine step of AtomicASForInnerProductArgPC (src/ipa_pc_as/mod.rs:254-346,
391-421) run on the GPU through the C ABI; the oracle recomputes the same accumulator from the same inputs, the same prover
randomness and the SAME Fiat-Shamir challenges (recorded from the product's sponge and injected -- the sponge is host
hashing, not the path under test).  Instances (affine points), witnesses (vectors, randomness) and proofs must be equal."""
import numpy as np
import pytest

from oracle import pyref as o
from oracle import pyref_as as oa
from tests import helpers as h
from tests.test_hp_as_scheme_gpu import SchemeRng
from tests.test_r1cs_nark_gpu import RecordingRng, dummy_circuit

pytestmark = pytest.mark.gpu
C = o.PALLAS


class RecordingSponge:
    """Wraps a product sponge; logs every squeeze as (fork path, values).  fork(b"") (a clone) extends the path with a
    running clone index so that the per-instance NARK challenges can be told apart."""

    def __init__(self, inner, log=None, path=()):
        self.inner, self.log, self.path, self.clones = inner, (log if log is not None else []), path, 0

    def absorb_bytes(self, b):
        self.inner.absorb_bytes(b)

    def absorb_u64(self, x):
        self.inner.absorb_u64(x)

    def absorb_len(self, n):
        self.inner.absorb_len(n)

    def absorb_point(self, p):
        self.inner.absorb_point(p)

    def absorb_points(self, p):
        self.inner.absorb_points(p)

    def squeeze_bits(self, n):
        return self.inner.squeeze_bits(n)

    def squeeze_field_elements(self, n, n_bits=128):
        v = self.inner.squeeze_field_elements(n, n_bits)
        self.log.append((self.path, list(v)))
        return v

    def fork(self, domain):
        if domain:
            return RecordingSponge(self.inner.fork(domain), self.log, self.path + (domain,))
        self.clones += 1
        return RecordingSponge(self.inner.fork(domain), self.log, self.path + (self.clones - 1,))

    def squeezed(self, *path):
        return [v for p, v in self.log if p == path]


def pt(p):
    return h.np_to_point(C, p[0], p[1])


def vec(v):
    return h.fr_from_mont_np(C, v.download())


# ---------------------------------------------------------------------------------------------------------------
# hp_as
# ---------------------------------------------------------------------------------------------------------------
def hp_to_oracle(x):
    r = x.witness.randomness
    return {"inst": (pt(x.instance.comm_1), pt(x.instance.comm_2), pt(x.instance.comm_3)),
            "wit": {"a": vec(x.witness.a_vec), "b": vec(x.witness.b_vec),
                    "rand": None if r is None else (r.rand_1, r.rand_2, r.rand_3)}}


def assert_hp_acc_equal(acc, proof, ref):
    got = hp_to_oracle(acc)
    assert got["inst"] == tuple(ref["inst"]), "hp accumulator instance"
    assert got["wit"]["a"] == ref["wit"]["a"] and got["wit"]["b"] == ref["wit"]["b"], "hp accumulator witness vectors"
    assert got["wit"]["rand"] == ref["wit"]["rand"], "hp accumulator witness randomness"
    assert [pt(p) for p in proof.product_poly_comm.low] == ref["proof"]["low"]
    assert [pt(p) for p in proof.product_poly_comm.high] == ref["proof"]["high"]
    hc = proof.hiding_comms
    assert (None if hc is None else (pt(hc.comm_1), pt(hc.comm_2), pt(hc.comm_3))) == ref["proof"]["hiding_comms"]


@pytest.fixture(scope="module")
def hp_env():
    from accumulation_amd import Context, PedersenCommitment, ffi
    ctx = Context(ffi.AMSM_PALLAS)
    n = 23
    ck = PedersenCommitment.setup(ctx, n, seed=777)
    xy, _ = ck.read()
    gens = [h.np_to_point(C, xy[i], 0) for i in range(n)]
    H = h.np_to_point(C, ck.hiding_generator, 0)
    yield ctx, ck, gens, H, n
    ctx.close()


def hp_inputs(ctx, ck, n, count, make_zk, seed):
    """random (not constant) vectors so that a transposed index would show"""
    from accumulation_amd import PedersenCommitment
    from accumulation_amd.hp_as import Accumulator, InputInstance, InputWitness, InputWitnessRandomness, compute_hp
    from accumulation_amd.scalar_field import Fr
    fr = Fr(ctx.curve)
    rng = SchemeRng(seed)
    out = []
    for k in range(count):
        a = ctx.upload(h.fr_mont_np(C, o.rng_scalars(seed + 10 * k + 1, n)))
        b = ctx.upload(h.fr_mont_np(C, o.rng_scalars(seed + 10 * k + 2, n)))
        rnd = InputWitnessRandomness(rng.field(), rng.field(), rng.field()) if make_zk else None
        lim = (lambda v: fr.to_limbs(v))
        c1 = PedersenCommitment.commit(ck, a, lim(rnd.rand_1) if rnd else None)
        c2 = PedersenCommitment.commit(ck, b, lim(rnd.rand_2) if rnd else None)
        c3 = PedersenCommitment.commit(ck, compute_hp(ctx, a, b), lim(rnd.rand_3) if rnd else None)
        out.append(Accumulator(InputInstance(c1, c2, c3), InputWitness(a, b, rnd)))
    return out


def hp_case(ctx, ck, gens, H, n, make_zk, shape, seed=100):
    """One hp_as prove over `shape` = (inputs, old accumulators) compared with the oracle (also driven by tools/fuzz_schemes.py
    with random shapes, lengths and seeds)."""
    from accumulation_amd.hp_as import ASForHadamardProducts as AS
    from accumulation_amd.sponge import Sha256Sponge
    n_in, n_acc = shape
    ins = hp_inputs(ctx, ck, n, n_in, make_zk, seed)
    # old accumulators are real accumulators (outputs of earlier proves), as in the reference's template
    olds = []
    for k in range(n_acc):
        a, _ = AS.prove(ck, hp_inputs(ctx, ck, n, 2, make_zk, seed + 400 + 50 * k), [], SchemeRng(seed - 60 + k) if make_zk else None, None)
        olds.append(a)
    rng = RecordingRng(seed - 1) if make_zk else None
    sp = RecordingSponge(Sha256Sponge())
    acc, proof = AS.prove(ck, ins, olds, rng, sp)
    sq = sp.squeezed()
    num_all = max(n_in + n_acc, 1) + (1 if (make_zk and n_in + n_acc <= 1) else 0)
    mu_sq = sq[0] if num_all > 1 else []
    nu1 = sq[-1][0]
    rnd = None
    if make_zk:
        d = rng.draws
        rnd = {"a": d[0], "b": d[1], "rand_1": d[2], "rand_2": d[3], "rand_3": d[4]}
    ref = oa.hp_prove(C, gens, H, [hp_to_oracle(x) for x in ins], [hp_to_oracle(x) for x in olds], make_zk, rnd, mu_sq, nu1,
                      supported=n)
    assert_hp_acc_equal(acc, proof, ref)
    assert oa.hp_decide(C, gens, H, ref) and AS.decide(ck, acc, None)


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
@pytest.mark.parametrize("shape", [(1, 0), (3, 0), (2, 1), (0, 2), (0, 0), (7, 2), (9, 8), (16, 0), (3, 20)],
                         ids=lambda s: f"in{s[0]}_acc{s[1]}")
def test_hp_as_prove_vs_oracle(hp_env, make_zk, shape):
    """the reference has no limit on inputs + accumulators (src/hp_as/mod.rs:288-349): beyond the fused kernel's eight the
    t-vectors come from block products (api_schemes.inc: t_vecs_blocked) -- 9, 16, 17 and 23 here, zk and not"""
    ctx, ck, gens, H, n = hp_env
    hp_case(ctx, ck, gens, H, n, make_zk, shape)


# ---------------------------------------------------------------------------------------------------------------
# r1cs_nark_as
# ---------------------------------------------------------------------------------------------------------------
N_IN, N_CON = 5, 12


@pytest.fixture(scope="module")
def nark_env():
    from accumulation_amd import Context, ffi
    from accumulation_amd import r1cs_nark as nark
    ctx = Context(ffi.AMSM_PALLAS)
    A, B, C_, _, _ = dummy_circuit(N_IN, N_CON, 2, 3, C.r)
    ipk = nark.index(ctx, A, B, C_, N_IN + 1, N_IN + 3, key_seed=4711)
    xy, _ = ipk.ck.read()
    gens = [h.np_to_point(C, xy[i], 0) for i in range(N_CON)]
    H = h.np_to_point(C, ipk.ck.hiding_generator, 0)
    yield ctx, ipk, (A, B, C_), gens, H
    ctx.close()


def first_msg_to_oracle(m):
    r = m.randomness
    return {"comm_a": pt(m.comm_a), "comm_b": pt(m.comm_b), "comm_c": pt(m.comm_c),
            "randomness": None if r is None else {k: pt(getattr(r, k)) for k in
                                                  ("comm_r_a", "comm_r_b", "comm_r_c", "comm_1", "comm_2")}}


def nark_input_to_oracle(x):
    s = x.witness.randomness
    return {"inst": {"r1cs_input": [v % C.r for v in x.instance.r1cs_input],
                     "first_msg": first_msg_to_oracle(x.instance.first_round_message)},
            "wit": {"blinded_witness": vec(x.witness.blinded_witness),
                    "randomness": None if s is None else (s.sigma_a, s.sigma_b, s.sigma_c, s.sigma_o)}}


def nark_acc_to_oracle(a):
    from accumulation_amd.hp_as import Accumulator as HPAcc
    i, w = a.instance, a.witness
    hp = hp_to_oracle(HPAcc(i.hp_instance, w.hp_witness))
    s = w.randomness
    return {"inst": {"r1cs_input": [v % C.r for v in i.r1cs_input], "comm_a": pt(i.comm_a), "comm_b": pt(i.comm_b),
                     "comm_c": pt(i.comm_c), "hp_instance": hp["inst"]},
            "wit": {"r1cs_blinded_witness": vec(w.r1cs_blinded_witness), "hp_witness": hp["wit"],
                    "randomness": None if s is None else (s.sigma_a, s.sigma_b, s.sigma_c)}}


def nark_inputs(nark_env, count, make_zk, rng, satisfiable=True):
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS, Input, InputInstance
    from accumulation_amd.sponge import Sha256Sponge
    ctx, ipk, _, _, _ = nark_env
    out = []
    for _ in range(count):
        a, b = rng.field() % C.r, rng.field() % C.r
        _, _, _, inst, w = dummy_circuit(N_IN, N_CON, a, b, C.r)
        nark_sponge, _, _ = AS._sponges(Sha256Sponge())
        proof = nark.prove(ipk, inst, ctx.upload(h.fr_mont_np(C, w)), make_zk, nark_sponge, rng if make_zk else None)
        out.append(Input(InputInstance(inst, proof.first_msg), proof.second_msg))
    return out


def nark_as_step(nark_env, ins, olds, make_zk, seed):
    """one product prove + the oracle's restatement of it -> (accumulator, proof, reference dict)"""
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS, HP_AS_PROTOCOL_NAME, NARK_PROTOCOL_NAME, PROTOCOL_NAME
    from accumulation_amd.sponge import Sha256Sponge
    ctx, ipk, (A, B, C_), gens, H = nark_env
    pk, vk, dk = AS.index(ipk)
    rng = RecordingRng(seed) if make_zk else None
    sp = RecordingSponge(Sha256Sponge())
    acc, proof = AS.prove(pk, ins, olds, rng, sp)
    assert AS.verify(ctx, vk, [x.instance for x in ins], [x.instance for x in olds], acc.instance, proof, None)
    # challenges, in the order the prover consumed them
    n_inputs = max(len(ins), 0 if olds else 1)
    gam = [v[0] for p, v in sp.log if p[:1] == (NARK_PROTOCOL_NAME,)]
    gammas, gi = [], 0
    ins_or_default = ins if (ins or olds) else [None]
    for x in ins_or_default:
        if x is not None and x.instance.first_round_message.randomness is not None:
            gammas.append(gam[gi])
            gi += 1
        else:
            gammas.append(None)
    hp_sq = sp.squeezed(HP_AS_PROTOCOL_NAME)
    num_all = n_inputs + len(olds)
    hp_num = num_all + (1 if (make_zk and num_all == 1) else 0)
    beta_sq = sp.squeezed(PROTOCOL_NAME)
    chal = {"gammas": gammas, "hp_mu": hp_sq[0] if hp_num > 1 else [], "hp_nu": hp_sq[-1][0],
            "beta": beta_sq[0] if beta_sq else []}
    rnd = None
    if make_zk:
        d = rng.draws
        rnd = {"r_input": d[0], "r_witness": d[1], "rand_1": d[2], "rand_2": d[3], "rand_3": d[4],
               "hp": {"a": d[5], "b": d[6], "rand_1": d[7], "rand_2": d[8], "rand_3": d[9]}}
    ref = oa.nark_as_prove(C, A, B, C_, gens, H, N_IN + 1, 2, [nark_input_to_oracle(x) for x in ins],
                           [nark_acc_to_oracle(x) for x in olds], make_zk, rnd, chal)
    return acc, proof, ref, dk


def assert_nark_acc_equal(acc, proof, ref):
    from accumulation_amd.hp_as import Accumulator as HPAcc
    got = nark_acc_to_oracle(acc)
    for k in ("r1cs_input", "comm_a", "comm_b", "comm_c", "hp_instance"):
        assert got["inst"][k] == (tuple(ref["inst"][k]) if k == "hp_instance" else ref["inst"][k]), f"instance.{k}"
    assert got["wit"]["r1cs_blinded_witness"] == ref["wit"]["r1cs_blinded_witness"], "blinded witness (beta combination)"
    assert got["wit"]["randomness"] == ref["wit"]["randomness"], "sigma combination"
    assert_hp_acc_equal(HPAcc(acc.instance.hp_instance, acc.witness.hp_witness), proof.hp_proof,
                        {"inst": ref["inst"]["hp_instance"], "wit": ref["wit"]["hp_witness"], "proof": ref["proof"]["hp_proof"]})
    pr = proof.randomness
    if ref["proof"]["randomness"] is None:
        assert pr is None
    else:
        rr = ref["proof"]["randomness"]
        assert [v % C.r for v in pr.r1cs_r_input] == rr["r1cs_r_input"]
        for k in ("comm_r_a", "comm_r_b", "comm_r_c"):
            assert pt(getattr(pr, k)) == rr[k], k


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_r1cs_nark_as_prove_vs_oracle(nark_env, make_zk):
    """[2 inputs] -> acc1; [1 input + acc1] -> acc2; [0 inputs + acc1 + acc2] -> acc3; every accumulator equals the
    oracle's and the oracle's decider accepts it."""
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS
    ctx, ipk, (A, B, C_), gens, H = nark_env
    rng = SchemeRng(31)
    ins = nark_inputs(nark_env, 3, make_zk, rng)
    acc1, p1, ref1, dk = nark_as_step(nark_env, ins[:2], [], make_zk, 7)
    assert_nark_acc_equal(acc1, p1, ref1)
    acc2, p2, ref2, _ = nark_as_step(nark_env, ins[2:], [acc1], make_zk, 8)
    assert_nark_acc_equal(acc2, p2, ref2)
    acc3, p3, ref3, _ = nark_as_step(nark_env, [], [acc1, acc2], make_zk, 9)
    assert_nark_acc_equal(acc3, p3, ref3)
    assert oa.nark_as_decide(C, A, B, C_, gens, H, ref3) and AS.decide(dk, acc3, None)


def test_r1cs_nark_as_default_input_vs_oracle(nark_env):
    """no inputs, no accumulators: the default input of :761-768"""
    acc, proof, ref, dk = nark_as_step(nark_env, [], [], False, 1)
    assert_nark_acc_equal(acc, proof, ref)


def test_blinded_commitments_start_comm_prod_from_comm_c(nark_env):
    """the reference's quirk at src/r1cs_nark_as/mod.rs:236 (comm_prod starts from comm_c) is kept"""
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS
    from accumulation_amd.scalar_field import Fr
    from accumulation_amd.sponge import Sha256Sponge
    ctx, ipk, _, _, _ = nark_env
    ins = nark_inputs(nark_env, 2, True, SchemeRng(77))
    sp = RecordingSponge(Sha256Sponge())
    A, B, Cc, P = AS._compute_blinded_commitments(ctx, Fr(ctx.curve), ipk.index_info.matrices_hash,
                                                  [x.instance for x in ins], sp)
    gam = [v[0] for _, v in sp.log]
    rA, rB, rC, rP = oa.nark_as_blinded_commitments(C, [nark_input_to_oracle(x)["inst"] for x in ins], gam)
    assert [pt(p) for p in A] == rA and [pt(p) for p in B] == rB and [pt(p) for p in Cc] == rC and [pt(p) for p in P] == rP


# ---------------------------------------------------------------------------------------------------------------
# ipa_pc_as: the combine step
# ---------------------------------------------------------------------------------------------------------------
def ipa_as_case(ctx, degree, n_first, n_second, make_zk, seed):
    """two chained ipa_pc_as proves -- n_first inputs, then n_second more plus the first accumulator -- with the combine step of
    the FIRST prove (src/ipa_pc_as/mod.rs:254-421) recomputed by the big-int oracle from the recorded Fiat-Shamir challenges; the
    second accumulation must verify and decide.  (tools/fuzz_schemes.py draws the shape; the test below fixes one.)"""
    from accumulation_amd import ipa_pc_as as M
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.sponge import Sha256Sponge
    from tests.test_ipa_gpu import generate_inputs
    AS = M.AtomicASForInnerProductArgPC
    pp = IpaPC.setup(ctx, degree, seed=11 + seed)
    pk, vk, dk = AS.index(pp, degree)
    env = (ctx, pp)
    rng = SchemeRng(3 + seed)
    ins = generate_inputs(env, pk, n_first + n_second, make_zk, rng, degree=min(degree, 11) if degree == 15 else degree)
    log = []

    class Rec(RecordingSponge):
        def __init__(self):
            super().__init__(Sha256Sponge(), log)
    old_cls = AS.sponge_cls
    AS.sponge_cls = Rec
    try:
        zk_rng = RecordingRng(5 + seed) if make_zk else None
        acc, proof = AS.prove(pk, ins[:n_first], [], zk_rng, None)
        n_log = len(log)
        acc2, proof2 = AS.prove(pk, ins[n_first:], [acc.instance], RecordingRng(6 + seed) if make_zk else None, None)
    finally:
        AS.sponge_cls = old_cls
    assert AS.verify(ctx, vk, ins[n_first:], [acc.instance], acc2.instance, proof2, None) and AS.decide(dk, acc2, None)
    # first prove: squeezes = [linear-combination challenges (n_first), challenge point (1)] on the AS fork's two clones
    assert n_first >= 2  # (with one input the two squeezes have the same length: the shape below would be ambiguous)
    first = [v for _, v in log[:n_log]]
    i_lc = next(i for i, v in enumerate(first) if len(v) == n_first)
    lc = first[i_lc]
    point = next(v for v in first[i_lc + 1:] if len(v) == 1)[0]
    checks = []
    AS._succinct_checks(ctx, pk.verifier_key.ipa_svk, ins[:n_first], False, checks)
    xis = [[int(x) for x in cp.challenges] for cp, _ in checks]
    fks = [pt(fk) for _, fk in checks]
    rnd = None
    lin = None
    if make_zk:
        lin = [v % C.r for v in proof.random_linear_polynomial]
        rnd = {"lin_comm": pt(proof.random_linear_polynomial_commitment), "commitment_randomness": proof.commitment_randomness}
    comb, randomized = oa.ipa_as_combine(C, fks, lc, pt((pk.verifier_key.ipa_svk.s[0], pk.verifier_key.ipa_svk.s[1])), rnd)
    assert pt(acc.instance.ipa_commitment.comm) == randomized
    assert acc.instance.point % C.r == point % C.r
    assert acc.instance.evaluation % C.r == oa.ipa_as_evaluate_combined(C, xis, lc, point, lin)
    poly = oa.ipa_as_combined_polynomial(C, xis, lc, lin)
    ev = 0
    for k in reversed(poly):
        ev = (ev * point + k) % C.r
    assert ev == acc.instance.evaluation % C.r


@pytest.mark.parametrize("make_zk", [False, True], ids=["no_zk", "zk"])
def test_ipa_pc_as_combine_vs_oracle(make_zk):
    from accumulation_amd import Context, ffi
    ctx = Context(ffi.AMSM_PALLAS)
    try:
        ipa_as_case(ctx, 15, 2, 1, make_zk, 0)
    finally:
        ctx.close()
