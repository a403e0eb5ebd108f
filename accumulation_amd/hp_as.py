"""Host-side mirror of the scalar-field vector loops of `ASForHadamardProducts` that produce the MSM
scalars (reference: src/hp_as/mod.rs), running on HBM-resident `FrVector`s through the C ABI.

Function names and argument meaning follow the reference:
  compute_hp(a, b)                                  src/hp_as/mod.rs:278-285
  scale_vector(v, coeff)                            src/hp_as/mod.rs:482-489
  combine_vectors(vectors, challenges, hiding)      src/hp_as/mod.rs:492-512
  compute_t_vecs(a_vecs, b_vecs, mu, len, hiding)   src/hp_as/mod.rs:288-349
  compute_product_poly_comm(ck, t_vecs)             src/hp_as/mod.rs:354-388
  decide_commitments(dk, a, b)                      src/hp_as/mod.rs:894-925 (the three MSMs of `decide`)
Field elements cross this API as Montgomery-form uint64 limbs (raw `Vec<Fr>` memory).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import ffi
from .engine import CommitterKey, Context, FrVector, VariableBaseMSM, _ptr


def compute_hp(ctx: Context, a: FrVector, b: FrVector) -> FrVector:
    n = min(a.n, b.n)  # zip truncates
    out = ctx.vector(n)
    ffi.check(ctx._lib.amsm_vec_hadamard(ctx._h, a.ptr, b.ptr, out.ptr, n), "amsm_vec_hadamard")
    return out


def combine_vectors(ctx: Context, vectors: Sequence[FrVector], challenges_mont: np.ndarray,
                    hiding: Optional[FrVector] = None) -> FrVector:
    k = len(vectors)
    n = max([v.n for v in vectors] + ([hiding.n] if hiding is not None else [0]))
    ch = np.ascontiguousarray(challenges_mont, dtype=np.uint64).reshape(-1, 4)
    assert ch.shape[0] >= k
    ptrs = (C.c_void_p * max(k, 1))(*[v.ptr for v in vectors])
    lens = (C.c_size_t * max(k, 1))(*[v.n for v in vectors])
    out = ctx.vector(n)
    ffi.check(ctx._lib.amsm_vec_combine(ctx._h, ptrs, lens, k, _ptr(ch), hiding.ptr if hiding is not None else None,
                                        hiding.n if hiding is not None else 0, out.ptr, n), "amsm_vec_combine")
    return out


def scale_vector(ctx: Context, v: FrVector, coeff_mont: np.ndarray) -> FrVector:
    return combine_vectors(ctx, [v], np.asarray(coeff_mont, dtype=np.uint64).reshape(1, 4))


def compute_t_vecs(ctx: Context, a_vecs: Sequence[FrVector], b_vecs: Sequence[FrVector], mu_mont: np.ndarray,
                   hp_vec_len: int, hiding: Optional[Tuple[FrVector, FrVector]] = None,
                   skip_uncommitted: bool = False) -> List[Optional[FrVector]]:
    n = len(a_vecs)
    mu = np.ascontiguousarray(mu_mont, dtype=np.uint64).reshape(-1, 4)
    pa = (C.c_void_p * n)(*[v.ptr for v in a_vecs])
    pb = (C.c_void_p * n)(*[v.ptr for v in b_vecs])
    la = (C.c_size_t * n)(*[v.n for v in a_vecs])
    lb = (C.c_size_t * n)(*[v.n for v in b_vecs])
    outs: List[Optional[FrVector]] = []
    for k in range(2 * n - 1):
        outs.append(None if (skip_uncommitted and k == n - 1) else ctx.vector(hp_vec_len))
    pt = (C.c_void_p * (2 * n - 1))(*[(v.ptr if v is not None else None) for v in outs])
    ha, hb = (hiding if hiding is not None else (None, None))
    ffi.check(ctx._lib.amsm_hp_t_vecs(ctx._h, pa, la, pb, lb, n, _ptr(mu), mu.shape[0],
                                      ha.ptr if ha is not None else None, ha.n if ha is not None else 0,
                                      hb.ptr if hb is not None else None, hb.n if hb is not None else 0,
                                      pt, hp_vec_len), "amsm_hp_t_vecs")
    return outs


def compute_product_poly_comm(ck: CommitterKey, t_vecs: Sequence[Optional[FrVector]]):
    """-> (low, high): commitments to t_0..t_{n-2} and t_n..t_{2n-2}; t_{n-1} is skipped (:373-375)."""
    if len(t_vecs) == 0:
        return [], []
    n = (len(t_vecs) + 1) // 2
    todo = [t for i, t in enumerate(t_vecs) if i != n - 1]
    pts, infs = VariableBaseMSM.multi_scalar_mul_batch(ck, todo, mont=True)
    comms = [(pts[i], bool(infs[i])) for i in range(len(todo))]
    return comms[: n - 1], comms[n - 1:]


def decide_commitments(dk: CommitterKey, a_vec: FrVector, b_vec: FrVector):
    """The decider's three commitments: commit(a), commit(b), commit(a o b) (no hiding)."""
    ctx = dk.ctx
    prod = compute_hp(ctx, a_vec, b_vec)
    pts, infs = VariableBaseMSM.multi_scalar_mul_batch(dk, [a_vec, b_vec, prod], mont=True)
    return [(pts[i], bool(infs[i])) for i in range(3)]
