"""GPU parity tests for the R1CS NARK prover path: SpMV (matrix_vec_mul) and the full commit sequence of
R1CSNark::prove (src/r1cs_nark_as/r1cs_nark/mod.rs:127-332) against the Python restatement, plus the
verifier's algebraic checks (:356-417) evaluated with the oracle."""
import hashlib

import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h

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


def to_dev_matrix(ctx, c, rows):
    from accumulation_amd.r1cs_nark import Matrix
    return Matrix(ctx, [[(o.int_to_limbs(o.fr_to_mont(c, cf), 4), idx) for cf, idx in row] for row in rows])


def gamma_from(first, c):
    """Deterministic stand-in for the Poseidon challenge (NOT the reference's sponge): SHA-256 of the
    commitments, truncated to 128 bits like CHALLENGE_SIZE."""
    hsh = hashlib.sha256()
    def absorb(P):
        hsh.update(repr(P).encode())
    for k in ("comm_a", "comm_b", "comm_c"):
        absorb(first[k])
    if first["randomness"]:
        for k in ("comm_r_a", "comm_r_b", "comm_r_c", "comm_1", "comm_2"):
            absorb(first["randomness"][k])
    return int.from_bytes(hsh.digest()[:16], "little")


@pytest.mark.parametrize("n_rows,n_in,n_wit", [(1, 1, 1), (100, 6, 10), (3000, 33, 700)])
def test_matrix_vec_mul(ctx, n_rows, n_in, n_wit):
    from accumulation_amd.r1cs_nark import matrix_vec_mul
    c = o.PALLAS
    rows = random_matrix(5, n_rows, n_in + n_wit, 6, c)
    M = to_dev_matrix(ctx, c, rows)
    inp = o.rng_scalars(6, n_in)
    wit = o.rng_scalars(7, n_wit)
    got = matrix_vec_mul(M, ctx.upload(h.fr_mont_np(c, inp)), ctx.upload(h.fr_mont_np(c, wit))).download()
    assert h.fr_from_mont_np(c, got) == o.matrix_vec_mul(c, rows, inp, wit)


@pytest.mark.parametrize("make_zk", [False, True])
def test_nark_prove_matches_restatement_and_verifies(ctx, make_zk):
    from accumulation_amd import PedersenCommitment
    from accumulation_amd.r1cs_nark import IndexProverKey, prove
    c = o.PALLAS
    n_con, n_in, n_wit = 64, 6, 20
    # a satisfiable instance is not needed for parity of the prover's data flow; random sparse matrices
    A = random_matrix(11, n_con, n_in + n_wit, 4, c)
    B = random_matrix(12, n_con, n_in + n_wit, 4, c)
    C_ = random_matrix(13, n_con, n_in + n_wit, 4, c)
    inp = [1] + o.rng_scalars(14, n_in - 1)
    wit = o.rng_scalars(15, n_wit)
    ck = PedersenCommitment.setup(ctx, n_con, seed=99)
    xy, _ = ck.read()
    gens = [h.np_to_point(c, xy[i], 0) for i in range(n_con)]
    H = h.np_to_point(c, ck.hiding_generator, 0)
    names = ["a_blinder", "b_blinder", "c_blinder", "r_a_blinder", "r_b_blinder", "r_c_blinder", "blinder_1", "blinder_2"]
    rnd = {k: o.rng_scalar(200 + i, 0) % c.r for i, k in enumerate(names)}
    rnd["r"] = o.rng_scalars(300, n_wit)
    ref = o.nark_prove(c, A, B, C_, gens, H, inp, wit, make_zk, rnd, lambda first: gamma_from(first, c))

    def dev_gamma(first):
        conv = {k: h.np_to_point(c, *first[k]) for k in ("comm_a", "comm_b", "comm_c")}
        conv["randomness"] = None
        if first["randomness"]:
            conv["randomness"] = {k: h.np_to_point(c, *v) for k, v in first["randomness"].items()}
        return h.fr_mont_np(c, [gamma_from(conv, c)])[0]

    ipk = IndexProverKey(to_dev_matrix(ctx, c, A), to_dev_matrix(ctx, c, B), to_dev_matrix(ctx, c, C_), ck, n_in)
    rnd_dev = {k: h.fr_mont_np(c, [v])[0] for k, v in rnd.items() if k != "r"}
    rnd_dev["r"] = h.fr_mont_np(c, rnd["r"])
    got = prove(ipk, ctx.upload(h.fr_mont_np(c, inp)), ctx.upload(h.fr_mont_np(c, wit)), make_zk,
                rnd_dev if make_zk else None, dev_gamma)
    for k in ("comm_a", "comm_b", "comm_c"):
        assert h.np_to_point(c, *got["first_msg"][k]) == ref["first_msg"][k], k
    if make_zk:
        for k, v in ref["first_msg"]["randomness"].items():
            assert h.np_to_point(c, *got["first_msg"]["randomness"][k]) == v, k
    assert h.fr_from_mont_np(c, got["gamma"].reshape(1, 4)) == [ref["gamma"]]
    assert h.fr_from_mont_np(c, got["second_msg"]["blinded_witness"].download()) == ref["blinded_witness"]
    # verifier's first three checks (:365-393): commit(M (x||w'); sigma_M) == C_M + gamma C_rM
    g = ref["gamma"]
    for M, cm, crm, bl, rbl in ((A, "comm_a", "comm_r_a", "a_blinder", "r_a_blinder"),
                                (B, "comm_b", "comm_r_b", "b_blinder", "r_b_blinder")):
        zp = o.matrix_vec_mul(c, M, inp, ref["blinded_witness"])
        if make_zk:
            sigma = (rnd[bl] + g * rnd[rbl]) % c.r
            lhs = o.pedersen_commit(c, gens, H, zp, sigma)
            rhs = o.add(c, ref["first_msg"][cm], o.mul(c, g, ref["first_msg"]["randomness"][crm]))
        else:
            lhs = o.pedersen_commit(c, gens, H, zp, None)
            rhs = ref["first_msg"][cm]
        assert lhs == rhs
