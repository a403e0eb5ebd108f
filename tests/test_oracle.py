"""CPU tests (no GPU): the two oracles against the curve KATs, the committed golden vectors and each
other.  The oracle is test infrastructure; PARITY UNPINNED w.r.t. the reference (oracle/pyref.py)."""
import re
import os

import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h

CURVES = [o.PALLAS, o.BLS12_381_G1]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_curve_kats(c):
    g = o.generator(c)
    assert o.is_on_curve(c, g)
    assert o.mul(c, c.r, g) is None
    assert o.mul(c, c.r - 1, g) == o.neg(c, g)
    kat = h.load_golden()["curves"][c.name]["kat"]
    assert h.pt_from_hex(kat["2G"]) == o.add(c, g, g)
    assert h.pt_from_hex(kat["3G"]) == o.add(c, o.add(c, g, g), g)
    assert kat["rG_is_inf"] is True


def test_pallas_published_constants():
    # SURVEY.md Appendix B (verified there with sympy): 2G and 3G of Pallas, generator (-1, 2)
    c = o.PALLAS
    assert o.generator(c) == (c.p - 1, 2)
    assert o.mul(c, 2, o.generator(c)) == (
        0x1C0000000000000000000000000000000EFEE2EE4411ACFC1303C567B0000003,
        0x2B00000000000000000000000000000017076EC9563FB75E8AEA5CDF3BFFFFFC)
    assert o.mul(c, 3, o.generator(c)) == (
        0x08E7566FBAA967EDB84C45A7474EDF4CFFF647DE5AF5FC5CB7F08A3BEB32D263,
        0x301D0A4CC182E0F43897D34A1F5EF0CBC7C89E18DE142DF1187FFB7B17EB87C5)
    assert o.mont_constants(c.p, 4)["INV64"] == 0x992D30ECFFFFFFFF
    assert o.mont_constants(c.r, 4)["INV64"] == 0x8C46EB20FFFFFFFF
    assert o.mont_constants(o.BLS12_381_G1.p, 6)["INV64"] == 0x89F3FFFCFFFCFFFD


def test_device_field_tables_match_oracle():
    """The modulus / R / R^2 / INV tables compiled into the HIP code (csrc/fp.h) are the oracle's."""
    src = open(os.path.join(os.path.dirname(__file__), "..", "accumulation_amd", "csrc", "fp.h")).read()
    fields = {"PallasFq": (o.PALLAS.p, 8), "PallasFr": (o.PALLAS.r, 8),
              "Bls12381Fq": (o.BLS12_381_G1.p, 12), "Bls12381Fr": (o.BLS12_381_G1.r, 8)}
    for name, (m, L) in fields.items():
        blk = src[src.index("struct " + name):]
        blk = blk[:blk.index("};")]

        def tab(t):
            mm = re.search(r"AMSM_TABLE\(" + t + r", \d+, ([^)]*)\)", blk, re.S)
            vals = [int(x.strip().rstrip("u"), 16) for x in mm.group(1).replace("\n", " ").split(",")]
            assert len(vals) == L
            return sum(v << (32 * i) for i, v in enumerate(vals))

        R = 1 << (32 * L)
        assert tab("mod") == m
        assert tab("one") == R % m
        assert tab("r2") == R * R % m
        inv = int(re.search(r"INV = (0x[0-9a-f]+)u", blk).group(1), 16)
        assert inv == (-pow(m, -1, 1 << 32)) % (1 << 32)


@pytest.mark.parametrize("pack,modulus", [("PallasFqU", o.PALLAS.p), ("Bls12381FqU", o.BLS12_381_G1.p)])
def test_device_unsaturated_field_tables_match_oracle(pack, modulus):
    """The B-bit-limb tables of the unsaturated base fields (csrc/fpu.h): modulus, R' = 2^(B L), and the two constants
    that convert between the C-ABI Montgomery radix 2^(32 W) and R'."""
    src = open(os.path.join(os.path.dirname(__file__), "..", "accumulation_amd", "csrc", "fpu.h")).read()
    blk = src[src.index("struct " + pack + " {"):]
    blk = blk[:blk.index("};")]
    L = int(re.search(r"int L = (\d+);", blk).group(1))
    B = int(re.search(r"int B = (\d+);", blk).group(1))
    W = int(re.search(r"int W = (\d+);", blk).group(1))
    m = modulus

    def tab(t):
        mm = re.search(r"AMSM_TABLE\(" + t + r", \d+, ([^)]*)\)", blk, re.S)
        vals = [int(x.strip().rstrip("u"), 16) for x in mm.group(1).replace("\n", " ").split(",")]
        assert len(vals) == L and all(v < (1 << B) for v in vals)
        return sum(v << (B * i) for i, v in enumerate(vals))

    R_abi, R_dev = 1 << (32 * W), 1 << (B * L)
    assert tab("mod") == m
    assert tab("one") == R_dev % m
    assert tab("k_import") == R_dev * R_dev * pow(R_abi, -1, m) % m   # mont_mul(x R_abi, k) = x R_dev
    assert tab("k_export") == R_abi % m                               # mont_mul(x R_dev, k) = x R_abi
    ninv = int(re.search(r"NINV = (0x[0-9a-f]+)u", blk).group(1), 16)
    assert ninv == (-pow(m, -1, 1 << B)) % (1 << B)
    # headroom the bound comments in ec.h rely on: 2^(B L) >= 127 p; column sums of two products fit 64 bits
    assert R_dev // m >= 127
    assert 2 * L * (1 << (2 * B + 1)) + L * (1 << (2 * B)) < (1 << 64)


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_python_oracle_matches_golden(c):
    g = h.load_golden()["curves"][c.name]
    for case in g["cases"]:
        pts = [h.pt_from_hex(p) for p in case["points"]]
        sc = [int(s, 16) for s in case["scalars"]]
        exp = h.pt_from_hex(case["expected_affine"])
        assert o.msm_naive(c, pts, sc) == exp, case["name"]
        assert o.msm_pippenger(c, pts, sc) == exp, case["name"]
        xy, inf = o.point_to_mont_limbs(c, exp)
        assert [hex(v) for v in xy] == case["expected_mont_limbs"] and inf == case["expected_is_inf"]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("threads", [1, 4])
def test_c_oracle_matches_golden(c, threads, cref):
    g = h.load_golden()["curves"][c.name]
    for case in g["cases"]:
        pts = [h.pt_from_hex(p) for p in case["points"]]
        sc = [int(s, 16) % c.r for s in case["scalars"]]
        xy, inf = h.points_to_np(c, pts)
        out, oinf = cref.msm(c.curve_id, xy, h.scalars_to_np(sc), is_inf=inf, threads=threads)
        assert [hex(int(v)) for v in out] == case["expected_mont_limbs"], case["name"]
        assert int(oinf) == case["expected_is_inf"], case["name"]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_seeded_streams_agree_and_match_golden(c, cref):
    """Python and C generate identical synthetic inputs; MSM over them matches the committed result."""
    g = h.load_golden()["curves"][c.name]
    assert cref.rng_scalars(77, 9).tolist() == h.scalars_to_np(o.rng_scalars(77, 9)).tolist()
    # the scalar stream (round 6): uniform in [0, r) by rejection, the same in Python and C; every value below r, and the top bit
    # (2^254) set about as often as r / 2^255 says (0.906 of BLS12-381's scalars lie below 2^255 * 0.906; Pallas: never set)
    fr = cref.rng_frs(c.curve_id, 77, 4096)
    assert fr[:64].tolist() == h.scalars_to_np(o.rng_frs(c, 77, 64)).tolist()
    vals = h.np_to_ints(fr)
    assert max(vals) < c.r
    top = sum(v >> 254 for v in vals) / len(vals)
    assert abs(top - (c.r - (1 << 254)) / c.r) < 0.03
    assert abs(sum(vals) / len(vals) / c.r - 0.5) < 0.02
    pts_c = cref.rng_points(c.curve_id, 0x5EED1001, 12)
    pts_p, _ = h.points_to_np(c, o.rng_points(c, 0x5EED1001, 12))
    assert np.array_equal(pts_c, pts_p)
    for case in g["seeded"]:
        xy = cref.rng_points(c.curve_id, case["seed_points"], case["n"])
        sc = cref.rng_frs(c.curve_id, case["seed_scalars"], case["n"])
        out, oinf = cref.msm(c.curve_id, xy, sc, threads=4)
        assert [hex(int(v)) for v in out] == case["expected_mont_limbs"], case["name"]


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_c_oracle_vs_python_random(c, cref):
    n = 300
    xy = cref.rng_points(c.curve_id, 5, n)
    sc = cref.rng_scalars(6, n)
    pts = [h.np_to_point(c, xy[i], 0) for i in range(n)]
    exp = o.msm_pippenger(c, pts, h.np_to_ints(sc))
    out, oinf = cref.msm(c.curve_id, xy, sc, threads=1)
    assert h.np_to_point(c, out, oinf) == exp
    # linearity (the verifier/prover agreement the reference checks, src/hp_as/mod.rs:883-891)
    sc2 = cref.rng_scalars(7, n)
    s_sum = h.scalars_to_np([(a + b) % c.r for a, b in zip(h.np_to_ints(sc), h.np_to_ints(sc2))])
    o1, i1 = cref.msm(c.curve_id, xy, sc2)
    o12, i12 = cref.msm(c.curve_id, xy, s_sum)
    assert h.np_to_point(c, o12, i12) == o.add(c, exp, h.np_to_point(c, o1, i1))


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_fr_vector_oracles_agree(c, cref):
    """C scalar-field loops == Python restatement of compute_hp / combine_vectors (src/hp_as/mod.rs:278-285,492-512)."""
    n = 50
    a = o.rng_scalars(1, n)
    b = o.rng_scalars(2, n - 7)
    am, bm = h.fr_mont_np(c, a), h.fr_mont_np(c, b)
    assert h.fr_from_mont_np(c, cref.fr_hadamard(c.curve_id, am, bm)) == o.compute_hp(c, a, b)
    ch = o.rng_scalars(3, 2)
    hid = o.rng_scalars(4, n + 3)
    got = cref.fr_combine(c.curve_id, [am, bm], h.fr_mont_np(c, ch), hiding=h.fr_mont_np(c, hid))
    assert h.fr_from_mont_np(c, got) == o.combine_vectors(c, [a, b], ch, hid)
    assert h.np_to_ints(cref.fr_from_mont(c.curve_id, am)) == [x % c.r for x in a]
    assert np.array_equal(cref.fr_to_mont(c.curve_id, h.scalars_to_np(a)), am)


def test_t_vecs_identity():
    """sum_k nu^k t_k == a' o b' (property P3 of SURVEY.md Appendix A.1) for the Python restatement."""
    c = o.PALLAS
    n_in, ln = 3, 11
    a = [o.rng_scalars(10 + j, ln) for j in range(n_in)]
    b = [o.rng_scalars(20 + j, ln) for j in range(n_in)]
    mu = [1] + o.rng_scalars(30, n_in - 1)
    nu = o.rng_scalar(31, 0)
    t = o.compute_t_vecs(c, a, b, mu, ln)
    assert len(t) == 2 * n_in - 1
    nus = [pow(nu, k, c.r) for k in range(2 * n_in - 1)]
    a_comb = o.combine_vectors(c, a, [mu[i] * nus[i] % c.r for i in range(n_in)])
    b_comb = o.combine_vectors(c, list(reversed(b)), nus[:n_in])
    lhs = o.combine_vectors(c, t, nus)
    assert lhs == o.compute_hp(c, a_comb, b_comb)
    # the uncommitted middle coefficient is sum_j mu_j (a_j o b_j)   (src/hp_as/mod.rs:373-375)
    mid = [sum(mu[j] * a[j][li] * b[j][li] for j in range(n_in)) % c.r for li in range(ln)]
    assert t[n_in - 1] == mid


@pytest.mark.parametrize("c", CURVES, ids=lambda c: c.name)
def test_c_t_vecs_and_spmv_match_python_oracle(c, cref):
    """the C restatements the config-size GPU tests check against (2^22-element t-vectors, 2^18-row SpMV) equal the
    big-int restatements of src/hp_as/mod.rs:288-349 and src/r1cs_nark_as/r1cs_nark/mod.rs:443-462 on ragged inputs"""
    import random
    for n in (1, 2, 3, 4):
        for zk in (False, True):
            ln = 19
            a = [o.rng_scalars(100 + j, ln if j != 1 else ln - 5) for j in range(n)]
            b = [o.rng_scalars(200 + j, ln if j != 0 else ln - 2) for j in range(n)]
            mu = [1] + o.rng_scalars(300, n)
            hid = (o.rng_scalars(400, ln), o.rng_scalars(401, ln - 2)) if zk else None
            exp = o.compute_t_vecs(c, a, b, mu, ln, hid)
            got = cref.fr_t_vecs(c.curve_id, [h.fr_mont_np(c, v) for v in a], [h.fr_mont_np(c, v) for v in b],
                                 h.fr_mont_np(c, mu), ln,
                                 None if not zk else (h.fr_mont_np(c, hid[0]), h.fr_mont_np(c, hid[1])))
            assert len(got) == 2 * n - 1
            for k in range(2 * n - 1):
                assert h.fr_from_mont_np(c, got[k]) == exp[k], (n, zk, k)
    rnd = random.Random(5)
    rows = [[(rnd.randrange(c.r), rnd.randrange(7)) for _ in range(rnd.randrange(4))] for _ in range(11)]
    inp, wit = o.rng_scalars(1, 3), o.rng_scalars(2, 4)
    rp, col, co = [0], [], []
    for r in rows:
        for cf, i in r:
            col.append(i)
            co.append(cf)
        rp.append(len(col))
    got = cref.fr_spmv(c.curve_id, np.array(rp), np.array(col, dtype=np.uint64), h.fr_mont_np(c, co), h.fr_mont_np(c, inp),
                       h.fr_mont_np(c, wit))
    assert h.fr_from_mont_np(c, got) == o.matrix_vec_mul(c, rows, inp, wit)
