"""CPU test (no GPU): the accumulation-layer oracle gives the same accumulators on its two vector backends -- Python-int
lists (oracle/pyref.py, the small-size GPU tests) and Montgomery limb arrays through the C restatement
(oracle/fastref.py, the config-size GPU tests) -- for hp_as and r1cs_nark_as proves and deciders, zk and not."""
import numpy as np
import pytest

from oracle import fastref
from oracle import pyref as o
from oracle import pyref_as as oa
from tests import helpers as h

C = o.PALLAS
N = 9


def _gens(n, seed):
    g = o.generator(C)
    pts = [o.mul(C, k, g) for k in o.rng_scalars(seed, n + 1)]
    xy, _ = h.points_to_np(C, pts[:n])
    return pts[:n], xy, pts[n]


def _hp_item(seed, zk, np_side, gens):
    a, b = o.rng_scalars(seed, N), o.rng_scalars(seed + 1, N)
    r = tuple(o.rng_scalars(seed + 2, 3)) if zk else None
    pts = tuple(gens[k] for k in (seed % 5, (seed + 1) % 5, (seed + 2) % 5))  # any points: the algebra does not check them
    wit = {"a": h.fr_mont_np(C, a) if np_side else a, "b": h.fr_mont_np(C, b) if np_side else b, "rand": r}
    return {"inst": pts, "wit": wit}


def _same_vec(x, y):
    return h.fr_from_mont_np(C, x) == list(y)


@pytest.mark.parametrize("zk", [False, True], ids=["no_zk", "zk"])
@pytest.mark.parametrize("shape", [(2, 1), (1, 0), (0, 2), (3, 0)], ids=lambda s: f"in{s[0]}_acc{s[1]}")
def test_hp_prove_same_on_both_backends(zk, shape, cref):
    gens, xy, H = _gens(N, 41)
    rnd = None
    if zk:
        d = o.rng_scalars(77, 5)
        rnd = {"a": d[0], "b": d[1], "rand_1": d[2], "rand_2": d[3], "rand_3": d[4]}
    mu_sq, nu1 = [v % (1 << 128) for v in o.rng_scalars(78, 4)], o.rng_scalar(79, 0) % (1 << 128)
    ins = lambda side: [_hp_item(100 + 10 * k, zk, side, gens) for k in range(shape[0])]   # noqa: E731
    accs = lambda side: [_hp_item(500 + 10 * k, zk, side, gens) for k in range(shape[1])]  # noqa: E731
    ref = oa.hp_prove(C, gens, H, ins(False), accs(False), zk, rnd, mu_sq, nu1, supported=N)
    with oa.use_ops(fastref.NumpyOps(C, threads=2)):
        got = oa.hp_prove(C, xy, H, ins(True), accs(True), zk, rnd, mu_sq, nu1, supported=N)
    assert tuple(got["inst"]) == tuple(ref["inst"]) and got["proof"] == ref["proof"]
    assert _same_vec(got["wit"]["a"], ref["wit"]["a"]) and _same_vec(got["wit"]["b"], ref["wit"]["b"])
    assert got["wit"]["rand"] == ref["wit"]["rand"]


@pytest.mark.parametrize("zk", [False, True], ids=["no_zk", "zk"])
def test_hp_decide_accepts_a_valid_accumulator_on_the_array_backend(zk, cref):
    gens, xy, H = _gens(N, 43)
    ops = fastref.NumpyOps(C, threads=2)
    a, b = o.rng_scalars(1, N), o.rng_scalars(2, N)
    r = tuple(o.rng_scalars(3, 3)) if zk else (None, None, None)
    inst = (o.pedersen_commit(C, gens, H, a, r[0]), o.pedersen_commit(C, gens, H, b, r[1]),
            o.pedersen_commit(C, gens, H, o.compute_hp(C, a, b), r[2]))
    acc = {"inst": inst, "wit": {"a": h.fr_mont_np(C, a), "b": h.fr_mont_np(C, b), "rand": r if zk else None}}
    with oa.use_ops(ops):
        assert oa.hp_decide(C, xy, H, acc)
        acc["wit"]["a"][3, 0] ^= 1
        assert not oa.hp_decide(C, xy, H, acc)
    assert oa.ops.__class__ is oa.PyOps  # restored


@pytest.mark.parametrize("zk", [False, True], ids=["no_zk", "zk"])
def test_nark_as_prove_same_on_both_backends(zk, cref):
    from tests.test_r1cs_nark_gpu import dummy_circuit
    n_in, n_con = 3, N
    A, B, Cm, _, _ = dummy_circuit(n_in, n_con, 2, 3, C.r)
    num_input, num_wit = n_in + 1, n_in + 3
    gens, xy, H = _gens(n_con, 47)
    csr = [fastref.csr_from_rows(C, M) for M in (A, B, Cm)]

    def nark_in(seed, side):
        w = o.rng_scalars(seed, num_wit)
        x = [1] + o.rng_scalars(seed + 1, num_input - 1)
        msg = {"comm_a": gens[seed % 4], "comm_b": gens[(seed + 1) % 4], "comm_c": gens[(seed + 2) % 4], "randomness": None}
        s = None
        if zk:
            msg["randomness"] = {k: gens[(seed + 3 + j) % 7] for j, k in enumerate(("comm_r_a", "comm_r_b", "comm_r_c", "comm_1", "comm_2"))}
            s = tuple(o.rng_scalars(seed + 2, 4))
        return {"inst": {"r1cs_input": x, "first_msg": msg},
                "wit": {"blinded_witness": h.fr_mont_np(C, w) if side else w, "randomness": s}}

    def acc_in(seed, side):
        hp = _hp_item(seed, zk, side, gens)
        w = o.rng_scalars(seed + 5, num_wit)
        return {"inst": {"r1cs_input": [1] + o.rng_scalars(seed + 6, num_input - 1), "comm_a": gens[1], "comm_b": gens[2],
                         "comm_c": gens[3], "hp_instance": hp["inst"]},
                "wit": {"r1cs_blinded_witness": h.fr_mont_np(C, w) if side else w, "hp_witness": hp["wit"],
                        "randomness": tuple(o.rng_scalars(seed + 7, 3)) if zk else None}}
    rnd = None
    if zk:
        d = o.rng_scalars(90, 10)
        rnd = {"r_input": d[0], "r_witness": d[1], "rand_1": d[2], "rand_2": d[3], "rand_3": d[4],
               "hp": {"a": d[5], "b": d[6], "rand_1": d[7], "rand_2": d[8], "rand_3": d[9]}}
    sq = [v % (1 << 128) for v in o.rng_scalars(91, 12)]
    chal = {"gammas": [sq[0], sq[1]], "hp_mu": sq[2:5], "hp_nu": sq[5], "beta": sq[6:10]}
    ref = oa.nark_as_prove(C, A, B, Cm, gens, H, num_input, num_wit, [nark_in(10, False), nark_in(20, False)],
                           [acc_in(30, False)], zk, rnd, chal)
    with oa.use_ops(fastref.NumpyOps(C, threads=2)):
        got = oa.nark_as_prove(C, csr[0], csr[1], csr[2], xy, H, num_input, num_wit, [nark_in(10, True), nark_in(20, True)],
                               [acc_in(30, True)], zk, rnd, chal)
    assert got["inst"]["r1cs_input"] == ref["inst"]["r1cs_input"]
    for k in ("comm_a", "comm_b", "comm_c"):
        assert got["inst"][k] == ref["inst"][k]
    assert tuple(got["inst"]["hp_instance"]) == tuple(ref["inst"]["hp_instance"])
    assert _same_vec(got["wit"]["r1cs_blinded_witness"], ref["wit"]["r1cs_blinded_witness"])
    assert _same_vec(got["wit"]["hp_witness"]["a"], ref["wit"]["hp_witness"]["a"])
    assert _same_vec(got["wit"]["hp_witness"]["b"], ref["wit"]["hp_witness"]["b"])
    assert got["wit"]["hp_witness"]["rand"] == ref["wit"]["hp_witness"]["rand"] and got["wit"]["randomness"] == ref["wit"]["randomness"]
    assert got["proof"]["hp_proof"] == ref["proof"]["hp_proof"]
    assert got["proof"]["randomness"] == ref["proof"]["randomness"]


@pytest.mark.parametrize("curve", [o.PALLAS, o.BLS12_381_G1], ids=lambda c: c.name)
@pytest.mark.parametrize("hiding", [False, True], ids=["no_zk", "zk"])
def test_ipa_open_and_check_same_on_both_backends(curve, hiding):
    """oracle/pyref_as.py ipa_open / ipa_check (ark-poly-commit ipa_pc open / check, ext; call sites src/ipa_pc_as/mod.rs:454,
    :836): the big-int backend (affine law, Python ints) and the array backend (oracle/ark_msm.c: MSMs, `key_l += x key_r` by
    plain double-and-add, inner products) give the same proof; the check accepts it and rejects a changed one.  The config-size
    GPU test (tests/test_ipa_open_vs_oracle_gpu.py) uses the array backend."""
    c = curve
    n = 32
    pts = o.rng_points(c, 77, n + 2)
    key, hg, sg = pts[:n], pts[n], pts[n + 1]
    poly = [v % c.r for v in o.rng_scalars(78, n - 3)]
    point = o.rng_scalar(79, 0) % c.r
    hid = None
    if hiding:
        hid = {"polynomial": [v % c.r for v in o.rng_scalars(80, n)], "rand": o.rng_scalar(81, 0) % c.r,
               "poly_rand": o.rng_scalar(81, 1) % c.r}
    comm = o.msm_naive(c, key[:len(poly)], poly)
    if hiding:
        comm = o.add(c, comm, o.mul(c, hid["poly_rand"], sg))
    ch = [o.rng_scalar(82, i) % (1 << 128) for i in range((1 if hiding else 0) + 1 + 5)]
    a = oa.ipa_open(c, key, hg, sg, poly, comm, point, ch, hid)
    assert a["combined_v"] == sum(v * pow(point, i, c.r) for i, v in enumerate(poly)) % c.r
    assert oa.ipa_check(c, key, hg, sg, comm, point, a["combined_v"], a, ch)
    ops = fastref.NumpyOps(c, threads=2)
    xy, _ = h.points_to_np(c, key)
    with oa.use_ops(ops):
        hid2 = None if hid is None else dict(hid, polynomial=ops.mont(hid["polynomial"]))
        b = oa.ipa_open(c, xy, hg, sg, ops.mont(poly), comm, point, ch, hid2)
        assert oa.ipa_check(c, xy, hg, sg, comm, point, b["combined_v"], b, ch)
    assert a == b
    for field, bad in (("c", (a["c"] + 1) % c.r), ("final_comm_key", o.add(c, a["final_comm_key"], hg)),
                       ("l_vec", [a["l_vec"][0]] + [a["l_vec"][0]] + a["l_vec"][2:])):
        assert not oa.ipa_check(c, key, hg, sg, comm, point, a["combined_v"], dict(a, **{field: bad}), ch), field
    assert not oa.ipa_check(c, key, hg, sg, comm, point, (a["combined_v"] + 1) % c.r, a, ch)
