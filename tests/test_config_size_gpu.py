"""BASELINE.json configs 4 and 5 at their STATED sizes, bit for bit against the CPU restatements (round-2 verdict, item 1):

  * the scalar-field vector kernels at 2^22 elements (grid-stride + double-buffer path of csrc/vec_kernels.h) against
    oracle/ark_msm.c: compute_hp (src/hp_as/mod.rs:278-285), combine_vectors for 2 and 3 addends (:492-512),
    compute_t_vecs for 2 and 3 inputs with and without hiding vectors (:288-349);
  * matrix_vec_mul at 2^18 rows (src/r1cs_nark_as/r1cs_nark/mod.rs:443-462);
  * ASForR1CSNark::prove / decide at 2^18 constraints (src/r1cs_nark_as/mod.rs:713-926, 1031-1112) in the n_all = 2 shape
    of SURVEY.md 8(d) and in the reference harness's shape (1 input + the same accumulator twice, zk:
    examples/scaling-as.rs:91-104): the whole accumulator -- every commitment an MSM of the C restatement over vectors the
    C restatement computed -- from oracle/pyref_as.py on oracle/fastref.py's array backend, challenges injected;
  * ASForHadamardProducts::prove / decide at 2^22 elements, n_all = 2 (src/hp_as/mod.rs:646-813, 894-925): low / high are
    the C restatement's MSMs of its own t-vectors;
  * the sharded forms at config size on the one GPU of the test box: 8 shards behind amsm_ctx_create_multi (device 0 listed
    eight times) and 2 ranks over gloo, equal to the unsharded accumulator.
What this cannot cover is N > 1 PHYSICAL GPUs (no such box in reach): the RCCL exchange itself stays unrun."""
import numpy as np
import pytest

from oracle import fastref
from oracle import pyref as o
from oracle import pyref_as as oa
from tests import helpers as h
from tests.test_as_layers_vs_oracle_gpu import RecordingSponge, first_msg_to_oracle, pt
from tests.test_hp_as_scheme_gpu import SchemeRng
from tests.test_r1cs_nark_gpu import RecordingRng

pytestmark = pytest.mark.gpu
C = o.PALLAS
LOG_VEC = 22   # config 5: 2^22-element vectors
LOG_CON = 18   # config 4: 2^18 constraints


@pytest.fixture(scope="module")
def ctx():
    from accumulation_amd import Context
    c = Context(C.curve_id)
    yield c
    c.close()


def mont1(x):
    return h.fr_mont_np(C, [x])


# ---------------------------------------------------------------------------------------------------------------------
# vector kernels at 2^22 elements
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def big_vectors(ctx):
    """device vector + its downloaded limbs; ragged lengths around 2^22 so the tails of the grid-stride loops are hit"""
    n = 1 << LOG_VEC
    lens = [n, n - 3, n, n - 257, n, n, n, n - 1]
    out = []
    for j, ln in enumerate(lens):
        v = ctx.random_vector(0x5EED2001 + j, ln, mont=True)
        out.append((v, v.download()))
    return out


def test_compute_hp_at_2p22(ctx, cref, big_vectors):
    from accumulation_amd.hp_as import compute_hp
    (a, a_h), (b, b_h) = big_vectors[0], big_vectors[1]
    got = compute_hp(ctx, a, b).download()
    assert got.shape[0] == min(len(a_h), len(b_h))
    assert np.array_equal(got, cref.fr_hadamard(C.curve_id, a_h, b_h))


@pytest.mark.parametrize("k", [2, 3])
@pytest.mark.parametrize("unit_first", [True, False], ids=["mu0_is_one", "arbitrary"])
@pytest.mark.parametrize("hiding", [False, True], ids=["plain", "hiding"])
def test_combine_vectors_at_2p22(ctx, cref, big_vectors, k, unit_first, hiding):
    from accumulation_amd.hp_as import combine_vectors
    vs = big_vectors[:k]
    ch = [1 if unit_first else o.rng_scalar(71, 0)] + [o.rng_scalar(71, 1 + j) % (1 << 128) for j in range(k - 1)]
    hid = big_vectors[3] if hiding else None
    got = combine_vectors(ctx, [v for v, _ in vs], h.fr_mont_np(C, ch), hid[0] if hid else None).download()
    exp = cref.fr_combine(C.curve_id, [a for _, a in vs], h.fr_mont_np(C, ch), hiding=hid[1] if hid else None)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("n_in", [2, 3])
@pytest.mark.parametrize("zk", [False, True], ids=["no_zk", "zk"])
def test_compute_t_vecs_at_2p22(ctx, cref, big_vectors, n_in, zk):
    from accumulation_amd.hp_as import compute_t_vecs
    n = 1 << LOG_VEC
    a, b = big_vectors[:n_in], big_vectors[n_in:2 * n_in]
    hid = (big_vectors[6], big_vectors[7]) if zk else None
    mu = [1] + [o.rng_scalar(72, j) % (1 << 128) for j in range(n_in - 1)]
    if zk:
        mu.append(mu[1] * mu[n_in - 1] % C.r)  # src/hp_as/mod.rs:246-250
    got = compute_t_vecs(ctx, [v for v, _ in a], [v for v, _ in b], h.fr_mont_np(C, mu), n,
                         (hid[0][0], hid[1][0]) if zk else None)
    exp = cref.fr_t_vecs(C.curve_id, [x for _, x in a], [x for _, x in b], h.fr_mont_np(C, mu), n,
                         (hid[0][1], hid[1][1]) if zk else None)
    assert len(got) == 2 * n_in - 1
    for k in range(2 * n_in - 1):
        assert np.array_equal(got[k].download(), exp[k]), k
    del got


# ---------------------------------------------------------------------------------------------------------------------
# R1CS: SpMV and ASForR1CSNark at 2^18 constraints
# ---------------------------------------------------------------------------------------------------------------------
N_IN = 5  # public inputs of the reference's DummyCircuit (examples/scaling-nark.rs:21-56)


def scaled_dummy_circuit(n_con, a, b):
    """The reference's DummyCircuit (a * b = c, repeated; src/r1cs_nark_as/mod.rs:1159-1188) with per-row coefficients so
    that A z, B z, C z are not constant vectors (a transposed or dropped row shows): row r is (k_r a) (l_r b) = k_r l_r c.
    instance = [1, a b, a, ..., a], witness = [a, b]; the last row is empty like the reference's."""
    n_inst = N_IN + 1
    pool = [1, 2, 3, C.r - 1, 0x1234567, (1 << 127) + 5, C.r - 7, 11]
    A = [[(pool[r % 8], n_inst)] for r in range(n_con - 1)] + [[]]
    B = [[(pool[(r // 8) % 8], n_inst + 1)] for r in range(n_con - 1)] + [[]]
    Cm = [[(pool[r % 8] * pool[(r // 8) % 8] % C.r, 1)] for r in range(n_con - 1)] + [[]]
    inst = [1, a * b % C.r] + [a] * (N_IN - 1)
    return A, B, Cm, inst, [a, b]


def test_matrix_vec_mul_at_2p18(ctx, cref):
    """2^18 rows, 0..3 entries per row over 2^16 + 6 columns"""
    from accumulation_amd.r1cs_nark import Matrix, matrix_vec_mul
    n_rows, n_in, n_wit = 1 << LOG_CON, 6, 1 << 16
    rng = np.random.default_rng(1234)
    nnz = rng.integers(0, 4, n_rows)
    cols = rng.integers(0, n_in + n_wit, int(nnz.sum()))
    sel = rng.integers(0, 8, int(nnz.sum()))
    pool = [1, C.r - 1, 2, 0x1234567, (1 << 200) + 17, C.r - 5, 3, (1 << 64) - 1]
    rows, k = [], 0
    for r in range(n_rows):
        rows.append([(pool[int(sel[k + t])], int(cols[k + t])) for t in range(int(nnz[r]))])
        k += int(nnz[r])
    M = Matrix(ctx, rows)
    inp = ctx.random_vector(81, n_in, mont=True)
    wit = ctx.random_vector(82, n_wit, mont=True)
    got = matrix_vec_mul(M, inp, wit).download()
    csr = fastref.csr_from_rows(C, rows)
    exp = cref.fr_spmv(C.curve_id, csr["row_ptr"], csr["col"], csr["coeff"], inp.download(), wit.download())
    assert np.array_equal(got, exp)
    M.free()


@pytest.fixture(scope="module")
def nark_env(ctx):
    from accumulation_amd import r1cs_nark as nark
    n_con = 1 << LOG_CON
    A, B, Cm, _, _ = scaled_dummy_circuit(n_con, 2, 3)
    ipk = nark.index(ctx, A, B, Cm, N_IN + 1, N_IN + 3, key_seed=0x5EED1001)
    xy, _ = ipk.ck.read()
    H = h.np_to_point(C, ipk.ck.hiding_generator, 0)
    csr = [fastref.csr_from_rows(C, M) for M in (A, B, Cm)]
    return ipk, xy, H, csr


def hp_to_oracle_np(x):
    r = x.witness.randomness
    return {"inst": (pt(x.instance.comm_1), pt(x.instance.comm_2), pt(x.instance.comm_3)),
            "wit": {"a": x.witness.a_vec.download(), "b": x.witness.b_vec.download(),
                    "rand": None if r is None else (r.rand_1, r.rand_2, r.rand_3)}}


def nark_input_to_oracle_np(x):
    s = x.witness.randomness
    return {"inst": {"r1cs_input": [v % C.r for v in x.instance.r1cs_input],
                     "first_msg": first_msg_to_oracle(x.instance.first_round_message)},
            "wit": {"blinded_witness": x.witness.blinded_witness.download(),
                    "randomness": None if s is None else (s.sigma_a, s.sigma_b, s.sigma_c, s.sigma_o)}}


def nark_acc_to_oracle_np(a):
    from accumulation_amd.hp_as import Accumulator as HPAcc
    i, w = a.instance, a.witness
    hp = hp_to_oracle_np(HPAcc(i.hp_instance, w.hp_witness))
    s = w.randomness
    return {"inst": {"r1cs_input": [v % C.r for v in i.r1cs_input], "comm_a": pt(i.comm_a), "comm_b": pt(i.comm_b),
                     "comm_c": pt(i.comm_c), "hp_instance": hp["inst"]},
            "wit": {"r1cs_blinded_witness": w.r1cs_blinded_witness.download(), "hp_witness": hp["wit"],
                    "randomness": None if s is None else (s.sigma_a, s.sigma_b, s.sigma_c)}}


def assert_hp_equal_np(got, proof, ref):
    assert got["inst"] == tuple(ref["inst"]), "hp accumulator instance"
    assert np.array_equal(got["wit"]["a"], ref["wit"]["a"]) and np.array_equal(got["wit"]["b"], ref["wit"]["b"])
    assert got["wit"]["rand"] == ref["wit"]["rand"]
    assert [pt(p) for p in proof.product_poly_comm.low] == ref["proof"]["low"]
    assert [pt(p) for p in proof.product_poly_comm.high] == ref["proof"]["high"]
    hc = proof.hiding_comms
    assert (None if hc is None else (pt(hc.comm_1), pt(hc.comm_2), pt(hc.comm_3))) == ref["proof"]["hiding_comms"]


def nark_as_step_np(ctx, nark_env, ins, olds, make_zk, seed, ops):
    """one product prove at 2^18 constraints + the oracle's recomputation of the whole accumulator"""
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS, HP_AS_PROTOCOL_NAME, NARK_PROTOCOL_NAME, PROTOCOL_NAME
    from accumulation_amd.sponge import Sha256Sponge
    ipk, xy, H, csr = nark_env
    pk, vk, dk = AS.index(ipk)
    rng = RecordingRng(seed) if make_zk else None
    sp = RecordingSponge(Sha256Sponge())
    acc, proof = AS.prove(pk, ins, olds, rng, sp)
    assert AS.verify(ctx, vk, [x.instance for x in ins], [x.instance for x in olds], acc.instance, proof, None)
    gam = [v[0] for p, v in sp.log if p[:1] == (NARK_PROTOCOL_NAME,)]
    gammas, gi = [], 0
    for x in ins:
        if x.instance.first_round_message.randomness is not None:
            gammas.append(gam[gi])
            gi += 1
        else:
            gammas.append(None)
    hp_sq = sp.squeezed(HP_AS_PROTOCOL_NAME)
    num_all = len(ins) + len(olds)
    beta_sq = sp.squeezed(PROTOCOL_NAME)
    chal = {"gammas": gammas, "hp_mu": hp_sq[0] if num_all > 1 else [], "hp_nu": hp_sq[-1][0],
            "beta": beta_sq[0] if beta_sq else []}
    rnd = None
    if make_zk:
        d = rng.draws
        rnd = {"r_input": d[0], "r_witness": d[1], "rand_1": d[2], "rand_2": d[3], "rand_3": d[4],
               "hp": {"a": d[5], "b": d[6], "rand_1": d[7], "rand_2": d[8], "rand_3": d[9]}}
    with oa.use_ops(ops):
        ref = oa.nark_as_prove(C, csr[0], csr[1], csr[2], xy, H, N_IN + 1, 2, [nark_input_to_oracle_np(x) for x in ins],
                               [nark_acc_to_oracle_np(x) for x in olds], make_zk, rnd, chal)
    return acc, proof, ref, dk


def assert_nark_acc_equal_np(acc, proof, ref):
    from accumulation_amd.hp_as import Accumulator as HPAcc
    got = nark_acc_to_oracle_np(acc)
    for k in ("r1cs_input", "comm_a", "comm_b", "comm_c"):
        assert got["inst"][k] == ref["inst"][k], f"instance.{k}"
    assert np.array_equal(got["wit"]["r1cs_blinded_witness"], ref["wit"]["r1cs_blinded_witness"])
    assert got["wit"]["randomness"] == ref["wit"]["randomness"]
    assert_hp_equal_np(hp_to_oracle_np(HPAcc(acc.instance.hp_instance, acc.witness.hp_witness)), proof.hp_proof,
                       {"inst": ref["inst"]["hp_instance"], "wit": ref["wit"]["hp_witness"], "proof": ref["proof"]["hp_proof"]})
    pr = proof.randomness
    if ref["proof"]["randomness"] is None:
        assert pr is None
    else:
        rr = ref["proof"]["randomness"]
        assert [v % C.r for v in pr.r1cs_r_input] == rr["r1cs_r_input"]
        for k in ("comm_r_a", "comm_r_b", "comm_r_c"):
            assert pt(getattr(pr, k)) == rr[k], k


@pytest.mark.parametrize("shape", ["n2_no_zk", "harness_zk"])
def test_r1cs_nark_as_at_2p18(ctx, cref, nark_env, shape):
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS, Input, InputInstance
    from accumulation_amd.sponge import Sha256Sponge
    ipk, xy, H, csr = nark_env
    make_zk = shape == "harness_zk"
    ops = fastref.NumpyOps(C)
    rng = SchemeRng(0xC0FFEE)
    n_con = 1 << LOG_CON

    def make_input(seed):
        a, b = rng.field() % C.r, rng.field() % C.r
        _, _, _, inst, w = scaled_dummy_circuit(8, a, b)  # only the assignment depends on (a, b)
        rrng = RecordingRng(seed) if make_zk else None
        proof = nark.prove(ipk, inst, ctx.upload(h.fr_mont_np(C, w)), make_zk, AS._sponges(Sha256Sponge())[0], rrng)
        # the NARK's own commitments: comm_a/b/c = commit(M z [, blinder]) with M z from the C restatement's SpMV; the
        # blinders are the prover's 2nd..4th draws after r (r1cs_nark/mod.rs:147-213 draws r, then a/b/c blinders)
        za, zb, zc = (ops.matrix_vec_mul(C, M, inst, h.fr_mont_np(C, w)) for M in csr)
        assert za.shape[0] == n_con
        if not make_zk:
            m = proof.first_msg
            assert pt(m.comm_a) == ops.pedersen_commit(C, xy, H, za, None)
            assert pt(m.comm_b) == ops.pedersen_commit(C, xy, H, zb, None)
            assert pt(m.comm_c) == ops.pedersen_commit(C, xy, H, zc, None)
        assert nark.verify(ipk, inst, proof, AS._sponges(Sha256Sponge())[0])
        return Input(InputInstance(inst, proof.first_msg), proof.second_msg)

    # an old accumulator that is a real accumulator: the output of an earlier prove over two inputs
    acc0, p0, ref0, dk = nark_as_step_np(ctx, nark_env, [make_input(11), make_input(12)], [], make_zk, 21, ops)
    assert_nark_acc_equal_np(acc0, p0, ref0)
    olds = [acc0] if shape == "n2_no_zk" else [acc0, acc0]  # the harness passes the same accumulator twice
    acc, proof, ref, dk = nark_as_step_np(ctx, nark_env, [make_input(13)], olds, make_zk, 22, ops)
    assert_nark_acc_equal_np(acc, proof, ref)
    with oa.use_ops(ops):
        assert oa.nark_as_decide(C, csr[0], csr[1], csr[2], xy, H, ref)
    assert AS.decide(dk, acc, None)


# ---------------------------------------------------------------------------------------------------------------------
# hp_as at 2^22 elements
# ---------------------------------------------------------------------------------------------------------------------
def hp_inputs_big(ctx, ck, n, count, make_zk, seed):
    from accumulation_amd import PedersenCommitment
    from accumulation_amd.hp_as import Accumulator, InputInstance, InputWitness, InputWitnessRandomness, compute_hp
    from accumulation_amd.scalar_field import Fr
    fr = Fr(ctx.curve)
    rng = SchemeRng(seed)
    out = []
    for k in range(count):
        a = ctx.random_vector(seed + 10 * k + 1, n, mont=True)
        b = ctx.random_vector(seed + 10 * k + 2, n, mont=True)
        rnd = InputWitnessRandomness(rng.field(), rng.field(), rng.field()) if make_zk else None
        lim = fr.to_limbs
        c1 = PedersenCommitment.commit(ck, a, lim(rnd.rand_1) if rnd else None)
        c2 = PedersenCommitment.commit(ck, b, lim(rnd.rand_2) if rnd else None)
        c3 = PedersenCommitment.commit(ck, compute_hp(ctx, a, b), lim(rnd.rand_3) if rnd else None)
        out.append(Accumulator(InputInstance(c1, c2, c3), InputWitness(a, b, rnd)))
    return out


def test_hp_as_at_2p22_n2(ctx, cref):
    """1 input + 1 old accumulator, no zk (the n_all = 2 shape of SURVEY.md 8(d) cfg5): 2 MSMs of 2^22 in prove"""
    from accumulation_amd import PedersenCommitment
    from accumulation_amd.hp_as import ASForHadamardProducts as AS
    from accumulation_amd.sponge import Sha256Sponge
    n = 1 << LOG_VEC
    ck = PedersenCommitment.setup(ctx, n, seed=0x5EED1001)
    xy, _ = ck.read()
    H = h.np_to_point(C, ck.hiding_generator, 0)
    ops = fastref.NumpyOps(C)
    ins = hp_inputs_big(ctx, ck, n, 3, False, 300)
    old, _ = AS.prove(ck, ins[:2], [], None, None)
    sp = RecordingSponge(Sha256Sponge())
    acc, proof = AS.prove(ck, ins[2:], [old], None, sp)
    sq = sp.squeezed()
    with oa.use_ops(ops):
        ref = oa.hp_prove(C, xy, H, [hp_to_oracle_np(x) for x in ins[2:]], [hp_to_oracle_np(old)], False, None, sq[0], sq[-1][0],
                          supported=n)
        assert_hp_equal_np(hp_to_oracle_np(acc), proof, ref)
        assert oa.hp_decide(C, xy, H, ref)
    assert AS.verify(ctx, n, [x.instance for x in ins[2:]], [old.instance], acc.instance, proof, None)
    assert AS.decide(ck, acc, None)
    ck.free()


# ---------------------------------------------------------------------------------------------------------------------
# the sharded forms at config size on one GPU
# ---------------------------------------------------------------------------------------------------------------------
def test_hp_as_at_2p22_on_8_shards_of_a_multi_device_context():
    """amsm_ctx_create_multi with device 0 listed eight times: 2^19 generators per shard (config 5's 8-GPU layout), the
    scheme driver unchanged; accumulator and proof equal the single-device run's."""
    from accumulation_amd import Context, MultiContext, PedersenCommitment, ffi
    from accumulation_amd.hp_as import ASForHadamardProducts as AS
    n = 1 << LOG_VEC
    res = []
    for make in (lambda: Context(ffi.AMSM_PALLAS), lambda: MultiContext(ffi.AMSM_PALLAS, (0,) * 8)):
        c = make()
        try:
            ck = PedersenCommitment.setup(c, n, seed=0x5EED1001)
            if isinstance(c, MultiContext):
                assert c._lib.amsm_bases_num_shards(ck._h) == 8
            ins = hp_inputs_big(c, ck, n, 2, False, 700)
            a1, p1 = AS.prove(ck, ins[:1], [], None, None)
            a2, p2 = AS.prove(ck, ins[1:], [a1], None, None)
            assert AS.verify(c, n, [ins[1].instance], [a1.instance], a2.instance, p2, None) and AS.decide(ck, a2, None)
            g = hp_to_oracle_np(a2)
            res.append((g["inst"], [pt(p) for p in p2.product_poly_comm.low], [pt(p) for p in p2.product_poly_comm.high],
                        g["wit"]["a"], g["wit"]["b"]))
            ck.free()
        finally:
            c.close()
    assert res[0][:3] == res[1][:3]
    assert np.array_equal(res[0][3], res[1][3]) and np.array_equal(res[0][4], res[1][4])


def test_r1cs_nark_as_at_2p18_on_8_shards_of_a_multi_device_context():
    """config 4's layout behind the C ABI: the 2^18-generator key in eight shards of 2^15 (all on GPU 0 here)"""
    from accumulation_amd import Context, MultiContext, ffi
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS, Input, InputInstance
    from accumulation_amd.sponge import Sha256Sponge
    n_con = 1 << LOG_CON
    A, B, Cm, _, _ = scaled_dummy_circuit(n_con, 2, 3)
    res = []
    for make in (lambda: Context(ffi.AMSM_PALLAS), lambda: MultiContext(ffi.AMSM_PALLAS, (0,) * 8)):
        c = make()
        try:
            ipk = nark.index(c, A, B, Cm, N_IN + 1, N_IN + 3, key_seed=0x5EED1001)
            if isinstance(c, MultiContext):
                assert c._lib.amsm_bases_num_shards(ipk.ck._h) == 8
            pk, vk, dk = AS.index(ipk)
            rng = SchemeRng(5)
            ins = []
            for _ in range(2):
                a, b = rng.field() % C.r, rng.field() % C.r
                _, _, _, inst, w = scaled_dummy_circuit(8, a, b)
                proof = nark.prove(ipk, inst, c.upload(h.fr_mont_np(C, w)), False, AS._sponges(Sha256Sponge())[0], None)
                ins.append(Input(InputInstance(inst, proof.first_msg), proof.second_msg))
            a1, p1 = AS.prove(pk, ins[:1], [], None, None)
            a2, p2 = AS.prove(pk, ins[1:], [a1], None, None)
            assert AS.verify(c, vk, [ins[1].instance], [a1.instance], a2.instance, p2, None) and AS.decide(dk, a2, None)
            g = nark_acc_to_oracle_np(a2)
            res.append(g)
        finally:
            c.close()
    x, y = res
    assert x["inst"] == y["inst"] and x["wit"]["randomness"] == y["wit"]["randomness"]
    assert np.array_equal(x["wit"]["r1cs_blinded_witness"], y["wit"]["r1cs_blinded_witness"])
    assert np.array_equal(x["wit"]["hp_witness"]["a"], y["wit"]["hp_witness"]["a"])
    assert np.array_equal(x["wit"]["hp_witness"]["b"], y["wit"]["hp_witness"]["b"])


def test_hp_as_at_2p22_two_ranks_over_gloo(built_lib):
    from tests.test_hp_as_sharded_gpu import run_sharded_vs_unsharded
    run_sharded_vs_unsharded(False, 2, N=1 << LOG_VEC, digest=True, timeout=1200)


def test_r1cs_nark_as_at_2p18_two_ranks_over_gloo(built_lib):
    from tests.test_r1cs_nark_as_sharded_gpu import run_sharded_vs_unsharded
    run_sharded_vs_unsharded(False, 2, NUM_CONSTRAINTS=1 << LOG_CON, digest=True, timeout=1200)
