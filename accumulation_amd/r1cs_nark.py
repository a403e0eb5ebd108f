"""Host-side mirror of the R1CS NARK prover's data-parallel part (reference:
src/r1cs_nark_as/r1cs_nark/mod.rs), with every SpMV, vector loop and Pedersen commitment on the GPU.

  Matrix / matrix_vec_mul(matrix, input, witness)      :443-462
  IndexProverKey{a, b, c, ck}                          data_structures.rs:33-48, index() :78-124
  prove(ipk, input, witness, make_zk, randomness, gamma_fn)   :127-332
      no zk: 3 SpMV + 3 commits;  zk: 6 SpMV + 8 commits + cross terms + blinded witness

The Fiat-Shamir challenge gamma (`compute_challenge`, :49-72, a Poseidon sponge over the first message) is
host-side hashing outside this path: the caller supplies `gamma_fn(first_msg) -> int`.  Likewise the
prover's random field elements are supplied by the caller (the reference draws them from its RngCore).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import ffi
from .engine import CommitterKey, Context, FrVector, PedersenCommitment, VariableBaseMSM, _ptr
from .hp_as import combine_vectors, compute_hp


class Matrix:
    """Row-sparse matrix `Vec<Vec<(F, usize)>>` resident in HBM as CSR."""

    def __init__(self, ctx: Context, rows_mont: Sequence[Sequence[Tuple[Sequence[int], int]]]):
        """rows_mont[r] = [(coeff as 4 u64 Montgomery limbs, column index), ...]"""
        self.ctx = ctx
        row_ptr = np.zeros(len(rows_mont) + 1, dtype=np.uint32)
        cols: List[int] = []
        vals: List[Sequence[int]] = []
        for r, row in enumerate(rows_mont):
            for coeff, idx in row:
                vals.append(coeff)
                cols.append(idx)
            row_ptr[r + 1] = len(cols)
        col = np.array(cols, dtype=np.uint32)
        val = np.array(vals, dtype=np.uint64).reshape(-1, 4)
        h = C.c_void_p()
        ffi.check(ctx._lib.amsm_matrix_load(ctx._h, _ptr(row_ptr), _ptr(col) if len(cols) else None,
                                            _ptr(val) if len(cols) else None, len(rows_mont), len(cols), C.byref(h)),
                  "amsm_matrix_load")
        self._h = h
        self.n_rows = len(rows_mont)

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


@dataclass
class IndexProverKey:
    a: Matrix
    b: Matrix
    c: Matrix
    ck: CommitterKey  # num_constraints generators (+ hiding generator)
    num_input_variables: int


def prove(ipk: IndexProverKey, input: FrVector, witness: FrVector, make_zk: bool,
          randomness: Optional[Dict[str, np.ndarray]], gamma_fn: Callable[[dict], np.ndarray]) -> dict:
    """R1CSNark::prove (:127-332).  `randomness` (zk only): Montgomery limbs for
    r (|witness| x 4), a_blinder, b_blinder, c_blinder, r_a_blinder, r_b_blinder, r_c_blinder, blinder_1,
    blinder_2.  gamma_fn(first_msg) -> (4,) uint64 Montgomery limbs of the challenge.
    Returns {'first_msg': {...}, 'second_msg': {...}} with affine commitments as (xy, is_inf)."""
    ctx = ipk.ck.ctx
    z_a = matrix_vec_mul(ipk.a, input, witness)
    z_b = matrix_vec_mul(ipk.b, input, witness)
    z_c = matrix_vec_mul(ipk.c, input, witness)
    rnd = randomness or {}
    if make_zk:
        r = ctx.upload(rnd["r"])
        zeros = ctx.upload(np.zeros((ipk.num_input_variables, 4), dtype=np.uint64))
        r_a = matrix_vec_mul(ipk.a, zeros, r)
        r_b = matrix_vec_mul(ipk.b, zeros, r)
        r_c = matrix_vec_mul(ipk.c, zeros, r)
        commit = PedersenCommitment.commit
        comm_a = commit(ipk.ck, z_a, rnd["a_blinder"])
        comm_b = commit(ipk.ck, z_b, rnd["b_blinder"])
        comm_c = commit(ipk.ck, z_c, rnd["c_blinder"])
        comm_r_a = commit(ipk.ck, r_a, rnd["r_a_blinder"])
        comm_r_b = commit(ipk.ck, r_b, rnd["r_b_blinder"])
        comm_r_c = commit(ipk.ck, r_c, rnd["r_c_blinder"])
        one = _one_mont(ctx)
        cross = combine_vectors(ctx, [compute_hp(ctx, z_a, r_b), compute_hp(ctx, z_b, r_a)], np.stack([one, one]))
        comm_1 = commit(ipk.ck, cross, rnd["blinder_1"])
        comm_2 = commit(ipk.ck, compute_hp(ctx, r_a, r_b), rnd["blinder_2"])
        first = {"comm_a": comm_a, "comm_b": comm_b, "comm_c": comm_c,
                 "randomness": {"comm_r_a": comm_r_a, "comm_r_b": comm_r_b, "comm_r_c": comm_r_c,
                                "comm_1": comm_1, "comm_2": comm_2}}
        gamma = np.asarray(gamma_fn(first), dtype=np.uint64).reshape(4)
        blinded = combine_vectors(ctx, [witness, r], np.stack([one, gamma]))  # w + gamma * r   (:294-296)
        return {"first_msg": first, "second_msg": {"blinded_witness": blinded}, "gamma": gamma}
    pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ipk.ck, [z_a, z_b, z_c], mont=True)
    first = {"comm_a": (pts[0], bool(infs[0])), "comm_b": (pts[1], bool(infs[1])), "comm_c": (pts[2], bool(infs[2])),
             "randomness": None}
    gamma = np.asarray(gamma_fn(first), dtype=np.uint64).reshape(4)
    return {"first_msg": first, "second_msg": {"blinded_witness": witness}, "gamma": gamma}


_ONE_CACHE: Dict[int, np.ndarray] = {}


def _one_mont(ctx: Context) -> np.ndarray:
    """Montgomery form of 1 in the scalar field (R mod r)."""
    if ctx.curve not in _ONE_CACHE:
        r = {ffi.AMSM_PALLAS: 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,
             ffi.AMSM_BLS12_381_G1: 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001}[ctx.curve]
        one = (1 << 256) % r
        _ONE_CACHE[ctx.curve] = np.array([(one >> (64 * i)) & ((1 << 64) - 1) for i in range(4)], dtype=np.uint64)
    return _ONE_CACHE[ctx.curve]
