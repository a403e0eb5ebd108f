"""Host-side mirror of the R1CS NARK (reference: src/r1cs_nark_as/r1cs_nark/mod.rs), with every SpMV, vector
loop and Pedersen commitment on the GPU.

  Matrix / matrix_vec_mul(matrix, input, witness)      :443-462
  index(matrices, ...) -> IndexProverKey               :78-124  (data_structures.rs:33-48)
  compute_challenge(matrices_hash, input, first_msg, sponge)   :49-72
  prove(ipk, input, witness, make_zk, sponge, rng)     :127-332
      no zk: 3 SpMV + 3 commits;  zk: 6 SpMV + 8 commits + cross terms + blinded witness
  verify(ivk, input, proof, sponge)                    :335-419  (4 commits)

Constraint synthesis (ark-relations) is outside this path: callers hand over the R1CS matrices and the
assignment (input = instance variables incl. the leading one, witness).  Scalars on the host are canonical
Python ints; vectors of length O(#constraints) / O(#witness) live in HBM as FrVector (Montgomery)."""
from __future__ import annotations

import ctypes as C
import hashlib
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import ffi
from .engine import CommitterKey, Context, FrVector, PedersenCommitment, VariableBaseMSM, _ptr
from .hp_as import ASForHadamardProducts, combine_vectors, compute_hp
from .scalar_field import Fr
from .sponge import CryptographicSponge, Sha256Sponge

PROTOCOL_NAME = b"R1CS-NARK-2020"  # :27
CHALLENGE_SIZE = 128


class Matrix:
    """Row-sparse matrix `Vec<Vec<(F, usize)>>` resident in HBM as CSR."""

    def __init__(self, ctx: Context, rows: Sequence[Sequence[Tuple[int, int]]], row_range: Optional[Tuple[int, int]] = None):
        """rows[r] = [(coeff as canonical int, column index), ...].  row_range = (lo, hi): only those constraints are
        loaded to THIS rank's device (the constraint-sharded layout of dist.ShardedCommitterKey: matrix_vec_mul then yields
        the rank's slice of M z); the digest still covers every row."""
        self.ctx = ctx
        fr = Fr(ctx.curve)
        self.rows = [[(int(cf) % fr.r, int(i)) for cf, i in row] for row in rows]
        lo, hi = row_range if row_range is not None else (0, len(rows))
        dev_rows = self.rows[lo:hi]
        row_ptr = np.zeros(len(dev_rows) + 1, dtype=np.uint32)
        cols: List[int] = []
        vals = []
        cache = {}
        for r, row in enumerate(dev_rows):
            for coeff, idx in row:
                if coeff not in cache:
                    cache[coeff] = fr.to_limbs(coeff)
                vals.append(cache[coeff])
                cols.append(idx)
            row_ptr[r + 1] = len(cols)
        col = np.array(cols, dtype=np.uint32)
        val = np.array(vals, dtype=np.uint64).reshape(-1, 4)
        h = C.c_void_p()
        ffi.check(ctx._lib.amsm_matrix_load(ctx._h, _ptr(row_ptr), _ptr(col) if len(cols) else None,
                                            _ptr(val) if len(cols) else None, len(dev_rows), len(cols), C.byref(h)),
                  "amsm_matrix_load")
        self._h = h
        self.n_rows = len(dev_rows)      # rows resident on this rank
        self.n_rows_global = len(rows)

    def serialize(self) -> bytes:
        out = [len(self.rows).to_bytes(8, "little")]
        for row in self.rows:
            out.append(len(row).to_bytes(8, "little"))
            for cf, i in row:
                out.append(cf.to_bytes(32, "little") + i.to_bytes(8, "little"))
        return b"".join(out)

    def free(self):
        if self._h is not None:
            self.ctx._lib.amsm_matrix_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def matrix_vec_mul(matrix: Matrix, input: FrVector, witness: FrVector) -> FrVector:
    ctx = matrix.ctx
    out = ctx.vector(matrix.n_rows)
    ffi.check(ctx._lib.amsm_matrix_vec_mul(ctx._h, matrix._h, input.ptr, input.n, witness.ptr, witness.n, out.ptr),
              "amsm_matrix_vec_mul")
    return out


def hash_matrices(domain_separator: bytes, a: Matrix, b: Matrix, c: Matrix) -> bytes:
    """:422-440: Blake2b-256 over the domain separator and the three matrices in ark-serialize 0.2's layout of `Vec<Vec<(F, usize)>>`
    (Matrix.serialize: u64 little-endian lengths, 32-byte canonical little-endian coefficients, column indices as u64)."""
    h = hashlib.blake2b(digest_size=32)
    h.update(domain_separator)
    for m in (a, b, c):
        h.update(m.serialize())
    return h.digest()


@dataclass
class IndexInfo:  # data_structures.rs:14-30
    num_variables: int
    num_constraints: int
    num_instance_variables: int
    matrices_hash: bytes


@dataclass
class IndexProverKey:  # data_structures.rs:33-48;  IndexVerifierKey is the same type (:51)
    index_info: IndexInfo
    a: Matrix
    b: Matrix
    c: Matrix
    ck: CommitterKey


@dataclass
class FirstRoundMessageRandomness:
    comm_r_a: tuple
    comm_r_b: tuple
    comm_r_c: tuple
    comm_1: tuple
    comm_2: tuple


@dataclass
class FirstRoundMessage:  # data_structures.rs:101-113
    comm_a: tuple
    comm_b: tuple
    comm_c: tuple
    randomness: Optional[FirstRoundMessageRandomness] = None

    @staticmethod
    def zero(ctx, make_zk):
        z = lambda: (np.zeros((2 * ctx.fq_limbs,), dtype=np.uint64), True)  # noqa: E731
        rnd = FirstRoundMessageRandomness(z(), z(), z(), z(), z()) if make_zk else None
        return FirstRoundMessage(z(), z(), z(), rnd)

    def absorb_into(self, sponge):
        for p in (self.comm_a, self.comm_b, self.comm_c):
            sponge.absorb_point(p)
        if self.randomness is None:
            sponge.absorb_bytes(b"\x00")
        else:
            sponge.absorb_bytes(b"\x01")
            r = self.randomness
            for p in (r.comm_r_a, r.comm_r_b, r.comm_r_c, r.comm_1, r.comm_2):
                sponge.absorb_point(p)


@dataclass
class SecondRoundMessageRandomness:
    sigma_a: int
    sigma_b: int
    sigma_c: int
    sigma_o: int


@dataclass
class SecondRoundMessage:  # data_structures.rs:171-177
    blinded_witness: FrVector
    randomness: Optional[SecondRoundMessageRandomness] = None


@dataclass
class Proof:
    first_msg: FirstRoundMessage
    second_msg: SecondRoundMessage


def index(ctx: Context, a_rows, b_rows, c_rows, num_instance_variables: int, num_variables: int,
          ck: Optional[CommitterKey] = None, key_seed: int = 0x5EED1001) -> IndexProverKey:
    """R1CSNark::index (:78-124): matrices + a Pedersen key with num_constraints generators.  With a
    dist.ShardedCommitterKey every rank keeps the constraints of its key slice: the prover, the verifier and the
    accumulation scheme then run unchanged on slices of every constraint-length vector (the assignment is replicated)."""
    rr = (ck.lo, ck.hi) if hasattr(ck, "sharded") else None
    a, b, c = Matrix(ctx, a_rows, rr), Matrix(ctx, b_rows, rr), Matrix(ctx, c_rows, rr)
    n_con = a.n_rows_global
    if ck is None:
        ck = PedersenCommitment.setup(ctx, n_con, seed=key_seed)
    info = IndexInfo(num_variables, n_con, num_instance_variables, hash_matrices(PROTOCOL_NAME, a, b, c))
    return IndexProverKey(info, a, b, c, ck)


def compute_challenge(fr: Fr, matrices_hash: bytes, input: Sequence[int], msg: FirstRoundMessage,
                      sponge: CryptographicSponge) -> int:
    """:49-72"""
    sponge.absorb_bytes(matrices_hash)
    sponge.absorb_bytes(b"".join((int(x) % fr.r).to_bytes(32, "little") for x in input))
    msg.absorb_into(sponge)
    return sponge.squeeze_field_elements(1, CHALLENGE_SIZE)[0]


def prove(ipk: IndexProverKey, input: Sequence[int], witness: FrVector, make_zk: bool,
          sponge: Optional[CryptographicSponge] = None, rng=None) -> Proof:
    """R1CSNark::prove (:127-332).  input: instance assignment (canonical ints, incl. the leading one);
    witness: FrVector.  rng (zk only): object with .field()."""
    ctx = ipk.ck.ctx
    fr = Fr(ctx.curve)
    commit = PedersenCommitment.commit
    d_input = ctx.upload(fr.to_limbs_many(list(input)))
    z_a = matrix_vec_mul(ipk.a, d_input, witness)
    z_b = matrix_vec_mul(ipk.b, d_input, witness)
    z_c = matrix_vec_mul(ipk.c, d_input, witness)
    sponge = sponge if sponge is not None else Sha256Sponge()
    if not make_zk:
        pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ipk.ck, [z_a, z_b, z_c], mont=True)
        first = FirstRoundMessage((pts[0], bool(infs[0])), (pts[1], bool(infs[1])), (pts[2], bool(infs[2])), None)
        compute_challenge(fr, ipk.index_info.matrices_hash, input, first, sponge)  # gamma is squeezed but unused
        return Proof(first, SecondRoundMessage(witness, None))
    assert rng is not None
    r = ctx.upload(fr.to_limbs_many([rng.field() for _ in range(witness.n)]))  # :168-172
    zeros = ctx.upload(np.zeros((len(input), 4), dtype=np.uint64))
    r_a = matrix_vec_mul(ipk.a, zeros, r)
    r_b = matrix_vec_mul(ipk.b, zeros, r)
    r_c = matrix_vec_mul(ipk.c, zeros, r)
    a_bl, b_bl, c_bl = rng.field(), rng.field(), rng.field()
    ra_bl, rb_bl, rc_bl = rng.field(), rng.field(), rng.field()
    one = fr.to_limbs(1)
    cross = combine_vectors(ctx, [compute_hp(ctx, z_a, r_b), compute_hp(ctx, z_b, r_a)], np.stack([one, one]))
    bl1 = rng.field()
    bl2 = rng.field()
    # the eight commitments are independent: one pipelined MSM batch (same points as eight commit() calls, :216-261)
    comm_a, comm_b, comm_c, comm_r_a, comm_r_b, comm_r_c, comm_1, comm_2 = PedersenCommitment.commit_batch(
        ipk.ck, [z_a, z_b, z_c, r_a, r_b, r_c, cross, compute_hp(ctx, r_a, r_b)],
        [fr.to_limbs(v) for v in (a_bl, b_bl, c_bl, ra_bl, rb_bl, rc_bl, bl1, bl2)])
    first = FirstRoundMessage(comm_a, comm_b, comm_c,
                              FirstRoundMessageRandomness(comm_r_a, comm_r_b, comm_r_c, comm_1, comm_2))
    gamma = compute_challenge(fr, ipk.index_info.matrices_hash, input, first, sponge)
    blinded = combine_vectors(ctx, [witness, r], np.stack([one, fr.to_limbs(gamma)]))  # w + gamma r  (:294-296)
    rnd = SecondRoundMessageRandomness((a_bl + gamma * ra_bl) % fr.r, (b_bl + gamma * rb_bl) % fr.r,
                                       (c_bl + gamma * rc_bl) % fr.r, (c_bl + gamma * bl1 + gamma * gamma * bl2) % fr.r)
    return Proof(first, SecondRoundMessage(blinded, rnd))


def verify(ivk: IndexProverKey, input: Sequence[int], proof: Proof, sponge: Optional[CryptographicSponge] = None) -> bool:
    """R1CSNark::verify (:335-419): 3 SpMV + 4 commitments on the GPU, O(1) point arithmetic on the host."""
    ctx = ivk.ck.ctx
    fr = Fr(ctx.curve)
    commit = PedersenCommitment.commit
    sponge = sponge if sponge is not None else Sha256Sponge()
    first, second = proof.first_msg, proof.second_msg
    if (first.randomness is None) != (second.randomness is None):
        return False
    gamma = compute_challenge(fr, ivk.index_info.matrices_hash, input, first, sponge)
    d_input = ctx.upload(fr.to_limbs_many(list(input)))
    za = matrix_vec_mul(ivk.a, d_input, second.blinded_witness)
    zb = matrix_vec_mul(ivk.b, d_input, second.blinded_witness)
    zc = matrix_vec_mul(ivk.c, d_input, second.blinded_witness)
    rnd = second.randomness
    lim = (lambda v: fr.to_limbs(v)) if rnd is not None else (lambda v: None)
    lhs = PedersenCommitment.commit_batch(  # four independent commitments, one pipelined batch (:375-403)
        ivk.ck, [za, zb, zc, compute_hp(ctx, za, zb)],
        [lim(rnd.sigma_a) if rnd else None, lim(rnd.sigma_b) if rnd else None, lim(rnd.sigma_c) if rnd else None,
         lim(rnd.sigma_o) if rnd else None])
    L = ASForHadamardProducts._lincomb
    from .hp_as import _pt_eq
    if rnd is None:
        rhs = [first.comm_a, first.comm_b, first.comm_c, first.comm_c]
    else:
        fr_ = first.randomness
        g2 = gamma * gamma % fr.r
        rhs = [L(ctx, [first.comm_a, fr_.comm_r_a], [1, gamma], fr), L(ctx, [first.comm_b, fr_.comm_r_b], [1, gamma], fr),
               L(ctx, [first.comm_c, fr_.comm_r_c], [1, gamma], fr),
               L(ctx, [first.comm_c, fr_.comm_1, fr_.comm_2], [1, gamma, g2], fr)]
    return all(_pt_eq(x, y) for x, y in zip(lhs, rhs))
