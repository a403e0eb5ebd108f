"""GPU parity tests for the R1CS NARK path: SpMV (matrix_vec_mul) and the full commit sequence of
R1CSNark::prove (src/r1cs_nark_as/r1cs_nark/mod.rs:127-332) against the Python restatement
(oracle/pyref.py:nark_prove), plus prove -> verify round trips (:335-419, the reference's own
`test_simple_circuit`, :509-556)."""
import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h
from tests.test_hp_as_scheme_gpu import SchemeRng

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from accumulation_amd import Context
    c = Context(o.PALLAS.curve_id)
    yield c
    c.close()


def random_matrix(seed, n_rows, n_cols, max_nnz, c):
    rows = []
    for r in range(n_rows):
        k = o.rng_word(seed, 3 * r) % (max_nnz + 1)
        row = []
        for t in range(k):
            idx = o.rng_word(seed, 1000003 * r + 7 * t + 1) % n_cols
            sel = o.rng_word(seed, 1000003 * r + 7 * t + 2) % 4
            coeff = 1 if sel == 0 else (c.r - 1 if sel == 1 else o.rng_scalar(seed + 1, 64 * r + t) % c.r)
            row.append((coeff, idx))
        rows.append(row)
    return rows


class RecordingRng(SchemeRng):
    def __init__(self, seed):
        super().__init__(seed)
        self.draws = []

    def field(self):
        v = super().field()
        self.draws.append(v)
        return v


@pytest.mark.parametrize("n_rows,n_in,n_wit", [(1, 1, 1), (100, 6, 10), (3000, 33, 700)])
def test_matrix_vec_mul(ctx, n_rows, n_in, n_wit):
    from accumulation_amd.r1cs_nark import Matrix, matrix_vec_mul
    c = o.PALLAS
    rows = random_matrix(5, n_rows, n_in + n_wit, 6, c)
    M = Matrix(ctx, rows)
    inp = o.rng_scalars(6, n_in)
    wit = o.rng_scalars(7, n_wit)
    got = matrix_vec_mul(M, ctx.upload(h.fr_mont_np(c, inp)), ctx.upload(h.fr_mont_np(c, wit))).download()
    assert h.fr_from_mont_np(c, got) == o.matrix_vec_mul(c, rows, inp, wit)


@pytest.mark.parametrize("make_zk", [False, True])
def test_nark_prove_matches_restatement_and_verifies(ctx, make_zk):
    from accumulation_amd import PedersenCommitment
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.scalar_field import Fr
    from accumulation_amd.sponge import Sha256Sponge
    c = o.PALLAS
    fr = Fr(ctx.curve)
    n_con, n_in, n_wit = 64, 6, 20
    # parity of the prover's data flow does not need a satisfiable instance: random sparse matrices
    A = random_matrix(11, n_con, n_in + n_wit, 4, c)
    B = random_matrix(12, n_con, n_in + n_wit, 4, c)
    C_ = random_matrix(13, n_con, n_in + n_wit, 4, c)
    inp = [1] + o.rng_scalars(14, n_in - 1)
    wit = o.rng_scalars(15, n_wit)
    ipk = nark.index(ctx, A, B, C_, n_in, n_in + n_wit, key_seed=99)
    xy, _ = ipk.ck.read()
    gens = [h.np_to_point(c, xy[i], 0) for i in range(n_con)]
    H = h.np_to_point(c, ipk.ck.hiding_generator, 0)
    rng = RecordingRng(555) if make_zk else None
    proof = nark.prove(ipk, inp, ctx.upload(h.fr_mont_np(c, wit)), make_zk, Sha256Sponge(), rng)
    gamma = nark.compute_challenge(fr, ipk.index_info.matrices_hash, inp, proof.first_msg, Sha256Sponge())
    rnd = None
    if make_zk:
        d = rng.draws
        rnd = {"r": d[:n_wit]}
        for i, k in enumerate(["a_blinder", "b_blinder", "c_blinder", "r_a_blinder", "r_b_blinder", "r_c_blinder",
                               "blinder_1", "blinder_2"]):
            rnd[k] = d[n_wit + i]
    ref = o.nark_prove(c, A, B, C_, gens, H, inp, wit, make_zk, rnd, lambda first: gamma)
    f = proof.first_msg
    for k in ("comm_a", "comm_b", "comm_c"):
        assert h.np_to_point(c, *getattr(f, k)) == ref["first_msg"][k], k
    if make_zk:
        for k, v in ref["first_msg"]["randomness"].items():
            assert h.np_to_point(c, *getattr(f.randomness, k)) == v, k
        s = proof.second_msg.randomness
        g = gamma
        assert s.sigma_a == (rnd["a_blinder"] + g * rnd["r_a_blinder"]) % c.r
        assert s.sigma_o == (rnd["c_blinder"] + g * rnd["blinder_1"] + g * g * rnd["blinder_2"]) % c.r
    assert h.fr_from_mont_np(c, proof.second_msg.blinded_witness.download()) == ref["blinded_witness"]
    # the verifier's first three checks hold for any matrices (:365-393); the product check needs Az o Bz = Cz
    from accumulation_amd.hp_as import ASForHadamardProducts as HP, _pt_eq
    for M, cm, crm, sig in ((A, "comm_a", "comm_r_a", "sigma_a"), (B, "comm_b", "comm_r_b", "sigma_b")):
        zp = o.matrix_vec_mul(c, M, inp, ref["blinded_witness"])
        if make_zk:
            lhs = o.pedersen_commit(c, gens, H, zp, getattr(proof.second_msg.randomness, sig))
            rhs = o.add(c, ref["first_msg"][cm], o.mul(c, gamma, ref["first_msg"]["randomness"][crm]))
        else:
            lhs, rhs = o.pedersen_commit(c, gens, H, zp, None), ref["first_msg"][cm]
        assert lhs == rhs


def dummy_circuit(num_inputs, num_constraints, a, b, r):
    """The reference's DummyCircuit (src/r1cs_nark_as/mod.rs:1159-1188) as R1CS matrices + assignment in
    ark-relations' layout: instance = [1, a*b, a, ..., a] (num_inputs public inputs), witness = [a, b];
    num_constraints - 1 copies of a * b = c and one empty constraint."""
    n_inst = num_inputs + 1
    ia, ib, ic = n_inst + 0, n_inst + 1, 1
    A = [[(1, ia)] for _ in range(num_constraints - 1)] + [[]]
    B = [[(1, ib)] for _ in range(num_constraints - 1)] + [[]]
    C_ = [[(1, ic)] for _ in range(num_constraints - 1)] + [[]]
    inst = [1, a * b % r] + [a] * (num_inputs - 1)
    return A, B, C_, inst, [a, b]


@pytest.mark.parametrize("make_zk", [False, True])
def test_simple_circuit_prove_verify(ctx, make_zk):
    """src/r1cs_nark_as/r1cs_nark/mod.rs:509-556: honest proofs verify; a wrong public input does not."""
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.sponge import Sha256Sponge
    c = o.PALLAS
    A, B, C_, _, _ = dummy_circuit(5, 100, 2, 3, c.r)
    ipk = nark.index(ctx, A, B, C_, 6, 8, key_seed=7)
    rng = SchemeRng(9)
    for _ in range(3):
        a, b = rng.field() % c.r, rng.field() % c.r
        _, _, _, inst, w = dummy_circuit(5, 100, a, b, c.r)
        proof = nark.prove(ipk, inst, ctx.upload(h.fr_mont_np(c, w)), make_zk, Sha256Sponge(), rng if make_zk else None)
        assert nark.verify(ipk, inst, proof, Sha256Sponge())
        bad = list(inst)
        bad[1] = (bad[1] + 1) % c.r
        assert not nark.verify(ipk, bad, proof, Sha256Sponge())
