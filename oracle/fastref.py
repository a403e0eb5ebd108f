"""TEST INFRASTRUCTURE ONLY -- the O(len) steps of the accumulation-layer oracle (oracle/pyref_as.py) over Montgomery limb
arrays through the plain-C restatement oracle/ark_msm.c, so that the config-size GPU tests (r1cs_nark_as at 2^18
constraints, hp_as at 2^22 elements: BASELINE.json configs 4 and 5) can recompute whole accumulators on the CPU in seconds.

PARITY UNPINNED (see oracle/ark_msm.c).  Same restrictions as the rest of oracle/: only tests/ may import this.

A vector is an (n, 4) uint64 array of Montgomery-form scalar-field elements (the memory of a Rust `Vec<Fr>`, and what
`DeviceVector.download()` returns); scalars and challenges are Python ints (canonical); points are oracle/pyref.py
Points.  `gens` is the committer key as its (n, 2L) uint64 Montgomery affine array.

Reference functions (file:line under /root/reference) the methods stand in for:
  pedersen_commit  ark_poly_commit::trivial_pc::PedersenCommitment::commit (ext) at src/hp_as/mod.rs:196,197,214,377,911-918
  compute_hp       src/hp_as/mod.rs:278-285      combine_vectors  :492-512      scale_vector  :482-489
  compute_t_vecs   src/hp_as/mod.rs:288-349      matrix_vec_mul   src/r1cs_nark_as/r1cs_nark/mod.rs:443-462
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import numpy as np

from . import cref
from . import pyref as o


def csr_from_rows(c, rows) -> dict:
    """`Matrix = Vec<Vec<(F, usize)>>` (canonical ints) -> CSR arrays with Montgomery coefficients"""
    rp, col, co = [0], [], []
    for r in rows:
        for cf, i in r:
            col.append(i)
            co.append(o.fr_to_mont(c, int(cf) % c.r))
        rp.append(len(col))
    coeff = np.array([o.int_to_limbs(v, 4) for v in co], dtype=np.uint64).reshape(-1, 4)
    return {"row_ptr": np.array(rp, dtype=np.uint64), "col": np.array(col, dtype=np.uint64), "coeff": coeff}


class NumpyOps:
    def __init__(self, c, threads: Optional[int] = None):
        self.c = c
        self.threads = threads or min(os.cpu_count() or 1, 20)

    # ---- conversions -------------------------------------------------------------------------------------------
    def mont(self, vals: Sequence[int]) -> np.ndarray:
        c = self.c
        return np.array([o.int_to_limbs(o.fr_to_mont(c, int(v) % c.r), 4) for v in vals], dtype=np.uint64).reshape(-1, 4)

    def vec(self, v) -> np.ndarray:
        if isinstance(v, np.ndarray):
            return np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
        return self.mont(v)  # a list of canonical ints (short vectors: r1cs_input)

    def point(self, xy: np.ndarray, is_inf: bool):
        return o.point_from_mont_limbs(self.c, [int(v) for v in np.asarray(xy).reshape(-1)], int(bool(is_inf)))

    # ---- the steps pyref_as calls ------------------------------------------------------------------------------
    def const(self, c, value: int, n: int) -> np.ndarray:
        return np.tile(self.mont([value]), (n, 1))

    def pedersen_commit(self, c, gens: np.ndarray, H, v, blinder: Optional[int]):
        v = self.vec(v)
        n = min(v.shape[0], gens.shape[0])
        xy, inf = cref.msm(c.curve_id, gens[:n], cref.fr_from_mont(c.curve_id, v[:n]), threads=self.threads)
        P = self.point(xy, inf)
        if blinder is not None:
            P = o.add(c, P, o.mul(c, blinder % c.r, H))
        return P

    def compute_hp(self, c, a, b) -> np.ndarray:
        return cref.fr_hadamard(c.curve_id, self.vec(a), self.vec(b))

    def combine_vectors(self, c, vecs, challenges, hiding=None) -> np.ndarray:
        k = min(len(vecs), len(challenges))  # zip
        vs = [self.vec(v) for v in vecs[:k]]
        return cref.fr_combine(c.curve_id, vs, self.mont(challenges[:k]), None if hiding is None else self.vec(hiding))

    def scale_vector(self, c, v, coeff: int) -> np.ndarray:
        return cref.fr_combine(c.curve_id, [self.vec(v)], self.mont([coeff]))

    def compute_t_vecs(self, c, a_vecs, b_vecs, mu, hp_len: int, hiding=None):
        hid = None if hiding is None else (self.vec(hiding[0]), self.vec(hiding[1]))
        return cref.fr_t_vecs(c.curve_id, [self.vec(v) for v in a_vecs], [self.vec(v) for v in b_vecs], self.mont(mu), hp_len,
                              hid)

    # ---- the O(len) steps of the IPA opening (pyref_as.ipa_open): vectors (n, 4) Montgomery arrays, keys (n, 2L) arrays ----
    def vlen(self, v) -> int:
        return int(np.asarray(v).shape[0])

    def split(self, v):
        half = v.shape[0] // 2
        return v[:half], v[half:]

    def scalar_at(self, c, v, i: int) -> int:
        return o.fr_from_mont(c, o.limbs_to_int([int(x) for x in self.vec(v)[i]]))

    def powers(self, c, point: int, n: int) -> np.ndarray:
        return cref.fr_powers(c.curve_id, self.mont([point])[0], n)

    def inner_product(self, c, a, b) -> int:
        return o.fr_from_mont(c, o.limbs_to_int([int(x) for x in cref.fr_inner_product(c.curve_id, self.vec(a), self.vec(b))]))

    def axpy(self, c, a, coeff: int, b) -> np.ndarray:
        return cref.fr_combine(c.curve_id, [self.vec(a), self.vec(b)], self.mont([1, coeff]))

    def pad(self, c, v, n: int) -> np.ndarray:
        v = self.vec(v)
        out = np.zeros((n, 4), dtype=np.uint64)
        out[:v.shape[0]] = v
        return out

    def cm_commit(self, c, key: np.ndarray, v):
        return self.pedersen_commit(c, key, None, v, None)

    def fold_points(self, c, key_l: np.ndarray, key_r: np.ndarray, x: int) -> np.ndarray:
        return cref.points_fold(c.curve_id, key_l, key_r, x % c.r, threads=self.threads)

    def point_at(self, c, key: np.ndarray, i: int):
        return self.point(key[i], not key[i].any())

    def check_poly_coeffs(self, c, xi) -> np.ndarray:
        """SuccinctCheckPolynomial::compute_coeffs over limb arrays: doubling steps, the LAST challenge on X^1"""
        k = len(xi)
        co = self.mont([1])
        for i in range(k):
            co = np.concatenate([co, cref.fr_combine(c.curve_id, [co], self.mont([xi[k - 1 - i]]))])
        return co

    def matrix_vec_mul(self, c, M: dict, inp, wit) -> np.ndarray:
        return cref.fr_spmv(c.curve_id, M["row_ptr"], M["col"], M["coeff"], self.vec(inp), self.vec(wit))
