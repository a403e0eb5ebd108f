"""The Fiat-Shamir TRANSCRIPTS of the four schemes against oracle/pyref_transcript.py: the challenges the product's drivers squeeze
(recorded through a wrapper around the product's Poseidon sponge) must equal the challenges the oracle derives from the PUBLIC data
alone -- the reference's `absorb!` item lists restated over the big-integer sponge of oracle/pyref_poseidon.py.  The layer tests
(tests/test_as_layers_vs_oracle_gpu.py) check the schemes' algebra with the product's challenges as inputs; this module closes the
other half: what is absorbed, in which order, what is squeezed (src/hp_as/mod.rs:753-780, src/r1cs_nark_as/mod.rs:423-448,
r1cs_nark/mod.rs:49-72, src/ipa_pc_as/mod.rs:267-296,349-388, src/trivial_pc_as/mod.rs:372-428).  Host backend, both curves, no GPU
(the transcripts are host code on every backend)."""
import pytest

from oracle import pyref as o
from oracle import pyref_transcript as ot
from tests import helpers as h
from tests.test_as_layers_vs_oracle_gpu import RecordingSponge
from tests.test_hp_as_scheme_gpu import SchemeRng

CURVES = [o.PALLAS, o.BLS12_381_G1]


@pytest.fixture(params=CURVES, ids=lambda c: c.name)
def env(request, built_lib):
    from accumulation_amd import Context, ffi
    c = request.param
    ctx = Context(c.curve_id, device=ffi.AMSM_DEVICE_HOST)
    yield c, ctx
    ctx.close()


def recording(ctx):
    from accumulation_amd.sponge import PoseidonSponge
    return RecordingSponge(PoseidonSponge(ctx.curve))


def P(c, p):
    """product point (xy limbs, is_inf) -> the oracle's (x, y) / None"""
    return h.np_to_point(c, p[0], p[1])


# ---- hp_as -----------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_in,n_acc,make_zk", [(2, 0, False), (1, 1, False), (0, 0, False), (1, 0, True), (0, 2, True), (3, 2, True)],
                         ids=["in2", "in1_acc1", "default_input", "in1_zk_placeholder", "acc2_zk", "in3_acc2_zk"])
def test_hp_as_transcript(env, n_in, n_acc, make_zk):
    from accumulation_amd import PedersenCommitment
    from accumulation_amd.hp_as import ASForHadamardProducts as AS
    from tests.test_hp_as_scheme_gpu import VECTOR_LEN, generate_inputs
    c, ctx = env
    ck = PedersenCommitment.setup(ctx, VECTOR_LEN, seed=4242)
    every = generate_inputs(ctx, ck, n_in + 2 * n_acc, make_zk)
    ins, rest = every[:n_in], every[n_in:]
    olds = [AS.prove(ck, rest[2 * k:2 * k + 2], [], SchemeRng(50 + k) if make_zk else None, None)[0] for k in range(n_acc)]
    sp = recording(ctx)
    acc, proof = AS.prove(ck, ins, olds, SchemeRng(9) if make_zk else None, sp)
    # the oracle's view: instances in the order the prover processes them -- inputs (the default / placeholder zero instances appended
    # to them, :685-710), then accumulators
    inst = [(P(c, x.instance.comm_1), P(c, x.instance.comm_2), P(c, x.instance.comm_3)) for x in ins]
    num_all = n_in + n_acc
    if num_all == 0:
        inst.append((None, None, None))
        num_all += 1
    if make_zk and num_all == 1:
        inst.append((None, None, None))
        num_all += 1
    inst += [(P(c, x.instance.comm_1), P(c, x.instance.comm_2), P(c, x.instance.comm_3)) for x in olds]
    hc = proof.hiding_comms
    mu, nu = ot.hp_as(c, ot.base_sponge(c), VECTOR_LEN, inst, None if hc is None else (P(c, hc.comm_1), P(c, hc.comm_2), P(c, hc.comm_3)),
                      [P(c, p) for p in proof.product_poly_comm.low], [P(c, p) for p in proof.product_poly_comm.high], num_all)
    log = sp.squeezed()
    got_mu = list(log[0]) if num_all > 1 else []
    assert [int(v) for v in got_mu] == mu and int(log[-1][0]) == nu and len(log) == (2 if num_all > 1 else 1)
    ck.free()


# ---- trivial_pc_as ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_in,n_acc", [(2, 0), (1, 2), (0, 0)], ids=["in2", "in1_acc2", "default_input"])
def test_trivial_pc_as_transcript(env, n_in, n_acc):
    from accumulation_amd.trivial_pc_as import ASForTrivialPC as AS, TrivialPC
    from tests.test_trivial_pc_as_scheme_gpu import generate_inputs
    c, ctx = env
    degree = 5
    pp = TrivialPC.setup(ctx, degree)
    ck, _ = TrivialPC.trim(pp, degree)
    pk, vk, dk = AS.index(pp, degree)
    rng = SchemeRng(77)
    every = generate_inputs((ctx, pp), ck, n_in + n_acc, rng)
    ins = every[:n_in]
    olds = [AS.prove(pk, [x], [], None, None)[0] for x in every[n_in:]]
    sp = recording(ctx)
    acc, proof = AS.prove(pk, ins, olds, None, sp)
    instances = [x.instance for x in ins] + [a.instance for a in olds]
    if not instances:
        instances = [None]  # the default input (:349-364): the zero instance
    oi = [(None, 0, 0) if i is None else (P(c, i.commitment.elem), i.point, i.eval) for i in instances]
    z, lc = ot.trivial_as(c, ot.base_sponge(c), degree, oi, [P(c, p.witness_commitment.elem) for p in proof],
                          lambda _z: [(p.eval, p.witness_eval) for p in proof])
    assert z == acc.instance.point
    log = sp.log
    assert [int(v) for v in log[-1][1]] == lc and int(log[0][1][0]) == z and len(log) == 2
    assert AS.verify(ctx, vk, [x.instance for x in ins], [a.instance for a in olds], acc.instance, proof, recording(ctx))


# ---- r1cs_nark_as ----------------------------------------------------------------------------------------------------------------------
def _first_msg(c, m):
    r = m.randomness
    return (P(c, m.comm_a), P(c, m.comm_b), P(c, m.comm_c),
            None if r is None else tuple(P(c, getattr(r, k)) for k in ("comm_r_a", "comm_r_b", "comm_r_c", "comm_1", "comm_2")))


@pytest.mark.parametrize("n_in,n_acc,make_zk", [(2, 0, False), (2, 1, True), (0, 2, True), (0, 0, False), (1, 0, True)],
                         ids=["in2", "in2_acc1_zk", "acc2_zk", "default_input", "in1_zk"])
def test_r1cs_nark_as_transcript(env, n_in, n_acc, make_zk):
    """the per-input NARK challenges gamma (only inputs that carry randomness have one, :220-286), the beta challenges, and the nested
    hp_as transcript over the blinded commitments the oracle recomputes from the gammas"""
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS, HP_AS_PROTOCOL_NAME, NARK_PROTOCOL_NAME, PROTOCOL_NAME, Input, InputInstance
    from accumulation_amd.sponge import PoseidonSponge
    from tests.test_r1cs_nark_gpu import dummy_circuit
    c, ctx = env
    n_inp, n_con = 4, 9
    A, B, C_, _, _ = dummy_circuit(n_inp, n_con, 2, 3, c.r)
    ipk = nark.index(ctx, A, B, C_, n_inp + 1, n_inp + 3, key_seed=4711)
    pk, vk, dk = AS.index(ipk)
    rng = SchemeRng(31)

    def inputs(count):
        out = []
        for _ in range(count):
            a, b = rng.field() % c.r, rng.field() % c.r
            _, _, _, inst, w = dummy_circuit(n_inp, n_con, a, b, c.r)
            nark_sponge, _, _ = AS._sponges(PoseidonSponge(ctx.curve))
            proof = nark.prove(ipk, inst, ctx.upload(h.fr_mont_np(c, w)), make_zk, nark_sponge, rng if make_zk else None)
            out.append(Input(InputInstance(inst, proof.first_msg), proof.second_msg))
        return out

    ins = inputs(n_in)
    olds = [AS.prove(pk, inputs(2), [], rng if make_zk else None, PoseidonSponge(ctx.curve))[0] for _ in range(n_acc)]
    sp = recording(ctx)
    acc, proof = AS.prove(pk, ins, olds, SchemeRng(5) if make_zk else None, sp)
    assert AS.verify(ctx, vk, [x.instance for x in ins], [x.instance for x in olds], acc.instance, proof, PoseidonSponge(ctx.curve))
    # ---- the oracle's derivation from the public data ----
    o_nark, o_as, o_hp = ot.nark_as_sponges(c, ot.base_sponge(c))
    nark_hash = ot.hash_matrices(c, b"R1CS-NARK-2020", A, B, C_)        # (r1cs_nark/mod.rs:104)
    as_hash = ot.hash_matrices(c, b"AS-FOR-R1CS-NARK-2020", A, B, C_)   # (src/r1cs_nark_as/mod.rs:694)
    assert nark_hash == bytes(ipk.index_info.matrices_hash) and as_hash == bytes(pk.as_matrices_hash)
    in_inst = [([v % c.r for v in x.instance.r1cs_input], _first_msg(c, x.instance.first_round_message)) for x in ins]
    if not ins and not olds:  # the default input (:761-768): zero r1cs input, identity commitments, no randomness
        in_inst = [([0] * (n_inp + 1), (None, None, None, None))]
    gammas = [None if m[3] is None else ot.nark_gamma(c, o_nark.clone(), nark_hash, r, m) for r, m in in_inst]
    got_g = [int(v[0]) for p, v in sp.log if p[:1] == (NARK_PROTOCOL_NAME,)]
    assert got_g == [g for g in gammas if g is not None]
    hp_of = lambda a: (P(c, a.comm_1), P(c, a.comm_2), P(c, a.comm_3))  # noqa: E731
    acc_inst = [([v % c.r for v in a.instance.r1cs_input], P(c, a.instance.comm_a), P(c, a.instance.comm_b), P(c, a.instance.comm_c),
                 hp_of(a.instance.hp_instance)) for a in olds]
    pr = proof.randomness
    o_pr = None if pr is None else ([v % c.r for v in pr.r1cs_r_input], P(c, pr.comm_r_a), P(c, pr.comm_r_b), P(c, pr.comm_r_c))
    num_addends = len(in_inst) + len(acc_inst) + (1 if make_zk else 0)
    beta = ot.nark_as_beta(c, o_as, as_hash, acc_inst, in_inst, o_pr, num_addends)
    got_beta = sp.squeezed(PROTOCOL_NAME)
    assert ([int(v) for v in got_beta[0]] if got_beta else []) == beta
    # the nested hp_as: its input instances are the blinded commitments (:220-286) comm_a + g comm_r_a, comm_b + g comm_r_b,
    # comm_c + g comm_1 + g^2 comm_2 -- recomputed here with the ORACLE's gammas and curve arithmetic
    hp_inst = []
    for (r, m), g in zip(in_inst, gammas):
        if g is None:
            hp_inst.append((m[0], m[1], m[2]))
        else:
            ra, rb, rc, c1, c2 = m[3]
            hp_inst.append((o.add(c, m[0], o.mul(c, g, ra)), o.add(c, m[1], o.mul(c, g, rb)),
                            o.add(c, o.add(c, m[2], o.mul(c, g, c1)), o.mul(c, g * g % c.r, c2))))
    hp_num = len(hp_inst) + len(olds)
    if make_zk and hp_num == 1:
        hp_inst.append((None, None, None))
        hp_num += 1
    hp_inst += [a[4] for a in acc_inst]
    hpp = proof.hp_proof
    hc = hpp.hiding_comms
    mu, nu = ot.hp_as(c, o_hp, n_con, hp_inst, None if hc is None else (P(c, hc.comm_1), P(c, hc.comm_2), P(c, hc.comm_3)),
                      [P(c, p) for p in hpp.product_poly_comm.low], [P(c, p) for p in hpp.product_poly_comm.high], hp_num)
    hp_log = sp.squeezed(HP_AS_PROTOCOL_NAME)
    assert ([int(v) for v in hp_log[0]] if hp_num > 1 else []) == mu and int(hp_log[-1][0]) == nu


# ---- ipa_pc_as -------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n_in,n_acc,make_zk", [(2, 0, False), (1, 1, True), (0, 0, False), (0, 2, True)],
                         ids=["in2", "in1_acc1_zk", "default_input", "acc2_zk"])
def test_ipa_pc_as_transcript(env, n_in, n_acc, make_zk):
    """the linear-combination challenges and the new accumulator's point (the two clones of the domain-separated sponge,
    src/ipa_pc_as/mod.rs:641,652); the succinct check polynomials come from the product's IPA-PC check (ark-poly-commit's own
    transcript: ext, not restated here), the combined commitment from the oracle's curve arithmetic over the ORACLE's alphas"""
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.ipa_pc_as import AtomicASForInnerProductArgPC as AS, Commitment, InputInstance
    from accumulation_amd.sponge import PoseidonSponge
    from tests.test_ipa_gpu import generate_inputs
    c, ctx = env
    degree = 7
    log = []
    old_cls = AS.sponge_cls, IpaPC.sponge_cls
    AS.sponge_cls = IpaPC.sponge_cls = lambda: RecordingSponge(PoseidonSponge(ctx.curve), log)
    try:
        pp = IpaPC.setup(ctx, degree, seed=0xABCDEF)
        pk, vk, dk = AS.index(pp, degree)
        rng = SchemeRng(4096)
        ins = generate_inputs((ctx, pp), pk, n_in, make_zk, rng, degree=degree)
        olds = [AS.prove(pk, generate_inputs((ctx, pp), pk, 1, make_zk, rng, degree=degree), [], rng if make_zk else None, None)[0].instance
                for _ in range(n_acc)]
        del log[:]
        acc, proof = AS.prove(pk, ins, olds, SchemeRng(11) if make_zk else None, None)
        prove_log = list(log)
        assert AS.verify(ctx, vk, ins, olds, acc.instance, proof, None)
        every = list(ins) + list(olds)
        if not make_zk and not every:  # the default instance (:599-609)
            every = [InputInstance(Commitment.default(ctx), 0, 0, pk.verifier_key.default_proof)]
        checks = []
        for x in every:
            cp = IpaPC.succinct_check(ctx, pk.verifier_key.ipa_svk, x.ipa_commitment, x.point, x.evaluation, x.ipa_proof)
            assert cp is not None
            checks.append(([int(v) % c.r for v in cp.challenges], P(c, x.ipa_proof.final_comm_key)))
    finally:
        AS.sponge_cls, IpaPC.sponge_cls = old_cls
    dom = b"AS-FOR-IPA-PC-2020"
    got_alphas = [[int(v) for v in vals] for p, vals in prove_log if p == (dom, 0)]
    got_point = [int(vals[0]) for p, vals in prove_log if p == (dom, 1)]
    assert len(got_alphas) == 1 and len(got_point) == 1
    o_as = ot.ipa_as_sponge(c)
    rnd = None if proof is None else ([v % c.r for v in proof.random_linear_polynomial], P(c, proof.random_linear_polynomial_commitment))
    alphas = ot.ipa_as_alphas(c, o_as, checks, rnd)
    assert got_alphas[0] == alphas
    combined = None  # sum_i alpha_i final_comm_key_i (+ the random linear polynomial's commitment), :303-321
    for a, (_, key) in zip(alphas, checks):
        combined = o.add(c, combined, o.mul(c, a, key))
    if rnd is not None:
        combined = o.add(c, combined, rnd[1])
    z = ot.ipa_as_challenge_point(c, o_as, combined, alphas, [poly for poly, _ in checks], None if rnd is None else rnd[0])
    assert got_point[0] == z and z == acc.instance.point % c.r
