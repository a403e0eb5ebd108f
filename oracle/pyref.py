"""TEST INFRASTRUCTURE ONLY -- pure-Python big-integer oracle for the MSM hot path.

PARITY UNPINNED: the reference (arkworks-rs/accumulation) keeps the arithmetic of this
path in un-vendored, un-pinned crates (ark-ec/ark-ff `^0.2.0`, Cargo.toml:15-16;
ark-poly-commit @ git branch `accumulation-experimental`, Cargo.toml:34) and its own
tests hold no golden vectors (src/lib.rs:334-395 are randomized accept/accept runs).
This oracle therefore restates the *mathematical definition* the reference's call
sites rely on -- prime-field arithmetic, the short-Weierstrass group law and
`sum_i s_i * G_i` -- with an implementation that shares no code with the product
(affine formulas + `pow(x, -1, p)`; the product uses XYZZ/Montgomery limbs).  Because
every consumer in the reference stores MSM results as *affine* points
(src/hp_as/data_structures.rs:14-23) the canonical affine (x, y, infinity) triple with
fully-reduced Montgomery limbs is algorithm-independent, which is what is compared.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Reference call sites restated here (file:line under /root/reference):
  * PedersenCommitment::commit(ck, v, r)        src/hp_as/mod.rs:196,197,214,377,911-918
  * compute_hp                                  src/hp_as/mod.rs:278-285
  * compute_t_vecs                              src/hp_as/mod.rs:288-349
  * scale_vector / combine_vectors              src/hp_as/mod.rs:482-512
  * compute_product_poly_comm                   src/hp_as/mod.rs:354-388
  * decide (decider identity)                   src/hp_as/mod.rs:894-925
  * ark_ec::msm::VariableBaseMSM::multi_scalar_mul  (ark-ec ^0.2.0, not in tree)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Optional, Sequence, Tuple

MASK64 = (1 << 64) - 1

# --------------------------------------------------------------------------------------
# Curves (constants re-derived/verified in tests/test_oracle.py: primality is assumed,
# on-curve / r*G = infinity / Montgomery constants are checked).
# --------------------------------------------------------------------------------------


@dataclass(frozen=True)
class Curve:
    name: str
    curve_id: int  # matches AMSM_CURVE_* in include/amsm.h
    p: int  # base field modulus (coordinates)
    r: int  # scalar field modulus (group order)
    b: int  # y^2 = x^3 + b
    gx: int
    gy: int
    limbs: int  # u64 limbs of a base-field element (4 | 6)

    @property
    def R(self) -> int:  # Montgomery radix of the base field
        return 1 << (64 * self.limbs)

    @property
    def Rr(self) -> int:  # Montgomery radix of the scalar field (always 4 limbs here)
        return 1 << 256


PALLAS = Curve(
    name="pallas",
    curve_id=0,
    p=0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,
    r=0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,
    b=5,
    gx=0x40000000000000000000000000000000224698FC094CF91B992D30ED00000000,  # -1
    gy=2,
    limbs=4,
)

BLS12_381_G1 = Curve(
    name="bls12_381_g1",
    curve_id=1,
    p=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
    r=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
    b=4,
    gx=0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    gy=0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    limbs=6,
)

CURVES = {c.name: c for c in (PALLAS, BLS12_381_G1)}
CURVES_BY_ID = {c.curve_id: c for c in (PALLAS, BLS12_381_G1)}

Point = Optional[Tuple[int, int]]  # None = point at infinity

# --------------------------------------------------------------------------------------
# Group law, affine, straight from the definition.
# --------------------------------------------------------------------------------------


def is_on_curve(c: Curve, P: Point) -> bool:
    if P is None:
        return True
    x, y = P
    return (y * y - (x * x * x + c.b)) % c.p == 0


def neg(c: Curve, P: Point) -> Point:
    if P is None:
        return None
    return (P[0], (-P[1]) % c.p)


def add(c: Curve, P: Point, Q: Point) -> Point:
    if P is None:
        return Q
    if Q is None:
        return P
    p = c.p
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    y3 = (lam * (x1 - x3) - y1) % p
    return (x3, y3)


def mul(c: Curve, k: int, P: Point) -> Point:
    """k*P by left-to-right double-and-add (k taken as a non-negative integer)."""
    if k < 0:
        return mul(c, -k, neg(c, P))
    acc: Point = None
    for bit in bin(k)[2:] if k else "":
        acc = add(c, acc, acc)
        if bit == "1":
            acc = add(c, acc, P)
    return acc


def generator(c: Curve) -> Point:
    return (c.gx, c.gy)


def msm_naive(c: Curve, bases: Sequence[Point], scalars: Sequence[int]) -> Point:
    """sum_i scalars[i]*bases[i] over min(len) pairs -- the semantics of
    ark_ec::msm::VariableBaseMSM::multi_scalar_mul."""
    acc: Point = None
    for P, s in zip(bases, scalars):
        acc = add(c, acc, mul(c, s % c.r, P))
    return acc


# Jacobian helpers keep the Python Pippenger usable up to 2^16 (no inversion per add).
def _jac_add_affine(c: Curve, J, P: Point):
    if P is None:
        return J
    p = c.p
    X1, Y1, Z1 = J
    if Z1 == 0:
        return (P[0], P[1], 1)
    x2, y2 = P
    Z1Z1 = Z1 * Z1 % p
    U2 = x2 * Z1Z1 % p
    S2 = y2 * Z1 * Z1Z1 % p
    if U2 == X1:
        if S2 == Y1:
            return _jac_double(c, J)
        return (1, 1, 0)
    H = (U2 - X1) % p
    Rr = (S2 - Y1) % p
    HH = H * H % p
    HHH = H * HH % p
    V = X1 * HH % p
    X3 = (Rr * Rr - HHH - 2 * V) % p
    Y3 = (Rr * (V - X3) - Y1 * HHH) % p
    Z3 = Z1 * H % p
    return (X3, Y3, Z3)


def _jac_double(c: Curve, J):
    p = c.p
    X1, Y1, Z1 = J
    if Z1 == 0 or Y1 == 0:
        return (1, 1, 0)
    A = X1 * X1 % p
    B = Y1 * Y1 % p
    C = B * B % p
    D = 2 * ((X1 + B) * (X1 + B) - A - C) % p
    E = 3 * A % p
    F = E * E % p
    X3 = (F - 2 * D) % p
    Y3 = (E * (D - X3) - 8 * C) % p
    Z3 = 2 * Y1 * Z1 % p
    return (X3, Y3, Z3)


def _jac_add(c: Curve, J1, J2):
    p = c.p
    if J1[2] == 0:
        return J2
    if J2[2] == 0:
        return J1
    X1, Y1, Z1 = J1
    X2, Y2, Z2 = J2
    Z1Z1 = Z1 * Z1 % p
    Z2Z2 = Z2 * Z2 % p
    U1 = X1 * Z2Z2 % p
    U2 = X2 * Z1Z1 % p
    S1 = Y1 * Z2 * Z2Z2 % p
    S2 = Y2 * Z1 * Z1Z1 % p
    if U1 == U2:
        if S1 == S2:
            return _jac_double(c, J1)
        return (1, 1, 0)
    H = (U2 - U1) % p
    Rr = (S2 - S1) % p
    HH = H * H % p
    HHH = H * HH % p
    V = U1 * HH % p
    X3 = (Rr * Rr - HHH - 2 * V) % p
    Y3 = (Rr * (V - X3) - S1 * HHH) % p
    Z3 = Z1 * Z2 * H % p
    return (X3, Y3, Z3)


def _jac_to_affine(c: Curve, J) -> Point:
    X, Y, Z = J
    if Z == 0:
        return None
    zi = pow(Z, -1, c.p)
    zi2 = zi * zi % c.p
    return (X * zi2 % c.p, Y * zi2 * zi % c.p)


def msm_pippenger(c: Curve, bases: Sequence[Point], scalars: Sequence[int], window: Optional[int] = None) -> Point:
    """Windowed-bucket MSM with the structure recalled for ark-ec 0.2:
    unsigned c-bit windows, 2^c - 1 buckets, running-sum reduction, Horner combine."""
    n = min(len(bases), len(scalars))
    if n == 0:
        return None
    if window is None:
        window = 3 if n < 32 else (max(n - 1, 1).bit_length() * 69 // 100) + 2
    cbits = window
    nbits = c.r.bit_length()
    INF = (1, 1, 0)
    total = INF
    starts = list(range(0, nbits, cbits))
    for w_start in reversed(starts):
        for _ in range(cbits):
            total = _jac_double(c, total)
        buckets = [INF] * ((1 << cbits) - 1)
        for i in range(n):
            s = scalars[i] % c.r
            d = (s >> w_start) & ((1 << cbits) - 1)
            if d:
                buckets[d - 1] = _jac_add_affine(c, buckets[d - 1], bases[i])
        running = INF
        res = INF
        for bkt in reversed(buckets):
            running = _jac_add(c, running, bkt)
            res = _jac_add(c, res, running)
        total = _jac_add(c, total, res)
    return _jac_to_affine(c, total)


# --------------------------------------------------------------------------------------
# Montgomery limb encodings (the memory format of ark-ff Fp256/Fp384: little-endian u64
# limbs of x*R mod m; SURVEY.md section 8 preamble).
# --------------------------------------------------------------------------------------


def int_to_limbs(x: int, n: int) -> List[int]:
    return [(x >> (64 * i)) & MASK64 for i in range(n)]


def limbs_to_int(limbs: Iterable[int]) -> int:
    out = 0
    for i, l in enumerate(limbs):
        out |= int(l) << (64 * i)
    return out


def fq_to_mont(c: Curve, x: int) -> int:
    return x * c.R % c.p


def fq_from_mont(c: Curve, x: int) -> int:
    return x * pow(c.R, -1, c.p) % c.p


def fr_to_mont(c: Curve, x: int) -> int:
    return x * c.Rr % c.r


def fr_from_mont(c: Curve, x: int) -> int:
    return x * pow(c.Rr, -1, c.r) % c.r


def point_to_mont_limbs(c: Curve, P: Point) -> Tuple[List[int], int]:
    """-> (2*limbs u64 words [x_mont | y_mont], is_inf).  Infinity is encoded (0, 0, 1)."""
    if P is None:
        return [0] * (2 * c.limbs), 1
    return int_to_limbs(fq_to_mont(c, P[0]), c.limbs) + int_to_limbs(fq_to_mont(c, P[1]), c.limbs), 0


def point_from_mont_limbs(c: Curve, words: Sequence[int], is_inf: int) -> Point:
    if is_inf:
        return None
    L = c.limbs
    return (fq_from_mont(c, limbs_to_int(words[:L])), fq_from_mont(c, limbs_to_int(words[L : 2 * L])))


def mont_constants(m: int, limbs: int) -> dict:
    R = 1 << (64 * limbs)
    return {
        "R": R % m,
        "R2": R * R % m,
        "INV64": (-pow(m, -1, 1 << 64)) % (1 << 64),
        "INV32": (-pow(m, -1, 1 << 32)) % (1 << 32),
    }


# --------------------------------------------------------------------------------------
# Deterministic synthetic inputs.  Counter-based splitmix64 so that Python (numpy),
# C (oracle/ark_msm.c) and HIP (accumulation_amd/csrc/gen.hip) produce identical streams.
#   word(seed, j) = mix(seed * 0xD1342543DE82EF95 + j * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019)
#   scalar_i     = words 4i..4i+3 as LE u64 limbs, top limb masked to 62 bits (< 2^254 < r): the multiplier stream;
#   rng_fr       = the scalar stream, uniform in [0, r) (round 6)
# --------------------------------------------------------------------------------------


def _mix64(z: int) -> int:
    z &= MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def rng_word(seed: int, j: int) -> int:
    return _mix64(seed * 0xD1342543DE82EF95 + j * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019)


def rng_scalar(seed: int, i: int) -> int:
    limbs = [rng_word(seed, 4 * i + k) for k in range(4)]
    limbs[3] &= (1 << 62) - 1
    return limbs_to_int(limbs)


def rng_scalars(seed: int, n: int) -> List[int]:
    return [rng_scalar(seed, i) for i in range(n)]


def rng_fr(c: "Curve", seed: int, i: int) -> int:
    """The SCALAR stream of amsm_vec_random since round 6 (accumulation_amd/csrc/rng.h:rng_scalar_fr): uniform in [0, r) by
    rejection -- candidate t of scalar i = words (t << 40) + 4 i .. + 3 masked to 255 bits, the first one below r wins; after 64
    rejections candidate 63 with bit 254 cleared.  rng_scalar above stays the 254-bit MULTIPLIER stream of rng_points."""
    v = 0
    for t in range(64):
        limbs = [rng_word(seed, (t << 40) + 4 * i + k) for k in range(4)]
        limbs[3] &= (1 << 63) - 1
        v = limbs_to_int(limbs)
        if v < c.r:
            return v
    return v & ((1 << 254) - 1)


def rng_frs(c: "Curve", seed: int, n: int) -> List[int]:
    return [rng_fr(c, seed, i) for i in range(n)]


def rng_points(c: Curve, seed: int, n: int) -> List[Point]:
    """P_i = k_i * G with k_i = rng_scalar(seed, i) -- the same definition the device-side
    `amsm_pedersen_setup` uses for its synthetic committer key."""
    g = generator(c)
    # fixed-base doubling table keeps this usable for a few thousand points
    table = [g]
    for _ in range(254):
        table.append(add(c, table[-1], table[-1]))
    out: List[Point] = []
    for i in range(n):
        k = rng_scalar(seed, i)
        J = (1, 1, 0)
        b = 0
        while k:
            if k & 1:
                J = _jac_add_affine(c, J, table[b])
            k >>= 1
            b += 1
        out.append(_jac_to_affine(c, J))
    return out


# --------------------------------------------------------------------------------------
# Restatement of the reference's scalar-field vector loops (all arithmetic mod r).
# --------------------------------------------------------------------------------------


def pedersen_commit(c: Curve, generators: Sequence[Point], hiding_generator: Point,
                    elems: Sequence[int], randomizer: Optional[int]) -> Point:
    """PedersenCommitment::commit as used at src/hp_as/mod.rs:196,377,911:
    msm(generators[..len], elems) (+ randomizer * hiding_generator)."""
    out = msm_pippenger(c, generators[: len(elems)], elems) if len(elems) >= 64 else msm_naive(
        c, generators[: len(elems)], elems)
    if randomizer is not None:
        out = add(c, out, mul(c, randomizer % c.r, hiding_generator))
    return out


def compute_hp(c: Curve, a: Sequence[int], b: Sequence[int]) -> List[int]:
    """src/hp_as/mod.rs:278-285 -- zip truncates to the shorter vector."""
    return [(x * y) % c.r for x, y in zip(a, b)]


def scale_vector(c: Curve, v: Sequence[int], coeff: int) -> List[int]:
    """src/hp_as/mod.rs:482-489."""
    return [(x * coeff) % c.r for x in v]


def combine_vectors(c: Curve, vectors: Sequence[Sequence[int]], challenges: Sequence[int],
                    hiding: Optional[Sequence[int]] = None) -> List[int]:
    """src/hp_as/mod.rs:492-512 -- ragged vectors allowed; output grows to the longest."""
    out = list(hiding) if hiding is not None else []
    for ni, vec in enumerate(vectors):
        for li, e in enumerate(vec):
            prod = challenges[ni] * e % c.r
            if li >= len(out):
                out.append(prod)
            else:
                out[li] = (out[li] + prod) % c.r
    return out


def compute_t_vecs(c: Curve, a_vecs: Sequence[Sequence[int]], b_vecs: Sequence[Sequence[int]],
                   mu: Sequence[int], hp_vec_len: int,
                   hiding: Optional[Tuple[Sequence[int], Sequence[int]]] = None) -> List[List[int]]:
    """src/hp_as/mod.rs:288-349.  a_vecs[j], b_vecs[j] are the j-th input witness's vectors
    (missing entries read as zero, :306-318); returns 2n-1 coefficient vectors."""
    n = len(a_vecs)
    assert n + (1 if hiding is not None else 0) <= len(mu)
    t = [[0] * hp_vec_len for _ in range(2 * n - 1)]
    for li in range(hp_vec_len):
        ac = [(mu[j] * a_vecs[j][li]) % c.r if li < len(a_vecs[j]) else 0 for j in range(n)]
        bc = [b_vecs[j][li] if li < len(b_vecs[j]) else 0 for j in range(n)]
        bc.reverse()
        if hiding is not None:
            ha, hb = hiding
            if li < len(ha):
                ac[0] = (ac[0] + ha[li] * mu[n]) % c.r
            if li < len(hb):
                bc[0] = (bc[0] + hb[li] * mu[1]) % c.r
        for i in range(n):
            for j in range(n):
                t[i + j][li] = (t[i + j][li] + ac[i] * bc[j]) % c.r
    return t


# --------------------------------------------------------------------------------------
# R1CS NARK prover restatement (src/r1cs_nark_as/r1cs_nark/mod.rs:127-332, 443-462), canonical ints.
# --------------------------------------------------------------------------------------


def matrix_vec_mul(c: Curve, matrix: Sequence[Sequence[Tuple[int, int]]], inp: Sequence[int],
                   wit: Sequence[int]) -> List[int]:
    """matrix[r] = [(coeff, index), ...];  z = inp || wit  (:443-462)."""
    out = []
    for row in matrix:
        acc = 0
        for coeff, i in row:
            tmp = inp[i] if i < len(inp) else wit[i - len(inp)]
            acc = (acc + tmp * coeff) % c.r
        out.append(acc)
    return out


def nark_prove(c: Curve, A, B, C_, generators: Sequence[Point], hiding_generator: Point, inp: Sequence[int],
               wit: Sequence[int], make_zk: bool, rnd: Optional[dict], gamma_fn) -> dict:
    """Same data flow as R1CSNark::prove; `rnd` holds the prover's random field elements (canonical ints),
    gamma_fn(first_msg) the injected Fiat-Shamir challenge."""
    def commit(v, blinder):
        return pedersen_commit(c, generators, hiding_generator, v, blinder)
    z_a, z_b, z_c = (matrix_vec_mul(c, M, inp, wit) for M in (A, B, C_))
    if not make_zk:
        first = {"comm_a": commit(z_a, None), "comm_b": commit(z_b, None), "comm_c": commit(z_c, None), "randomness": None}
        return {"first_msg": first, "blinded_witness": list(wit), "gamma": gamma_fn(first)}
    zeros = [0] * len(inp)
    r = rnd["r"]
    r_a, r_b, r_c = (matrix_vec_mul(c, M, zeros, r) for M in (A, B, C_))
    first = {
        "comm_a": commit(z_a, rnd["a_blinder"]), "comm_b": commit(z_b, rnd["b_blinder"]),
        "comm_c": commit(z_c, rnd["c_blinder"]),
        "randomness": {
            "comm_r_a": commit(r_a, rnd["r_a_blinder"]), "comm_r_b": commit(r_b, rnd["r_b_blinder"]),
            "comm_r_c": commit(r_c, rnd["r_c_blinder"]),
            "comm_1": commit([(x * y + u * v) % c.r for x, y, u, v in zip(z_a, r_b, z_b, r_a)], rnd["blinder_1"]),
            "comm_2": commit([(x * y) % c.r for x, y in zip(r_a, r_b)], rnd["blinder_2"]),
        },
    }
    gamma = gamma_fn(first)
    blinded = [(w + gamma * ri) % c.r for w, ri in zip(wit, r)]
    return {"first_msg": first, "blinded_witness": blinded, "gamma": gamma}
