"""TEST INFRASTRUCTURE ONLY -- ctypes loader for oracle/libark_msm.so (built from oracle/ark_msm.c by
oracle/Makefile).  Same restrictions as oracle/pyref.py: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this.  PARITY UNPINNED (see oracle/ark_msm.c header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libark_msm.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ark_msm.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libark_msm.so"])
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(LIB_PATH)
        vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
        lib.ark_msm.restype = C.c_int
        lib.ark_msm.argtypes = [C.c_int, vp, vp, vp, sz, C.c_int, vp, vp]
        lib.ark_msm_window_bits.restype = C.c_int
        lib.ark_msm_window_bits.argtypes = [sz]
        lib.ark_rng_scalars.restype = None
        lib.ark_rng_scalars.argtypes = [u64, sz, vp]
        lib.ark_rng_scalars_fr.restype = C.c_int
        lib.ark_rng_scalars_fr.argtypes = [C.c_int, u64, sz, vp]
        lib.ark_rng_points.restype = C.c_int
        lib.ark_rng_points.argtypes = [C.c_int, u64, sz, C.c_int, vp]
        lib.ark_fr_hadamard.restype = C.c_int
        lib.ark_fr_hadamard.argtypes = [C.c_int, vp, vp, sz, vp]
        lib.ark_fr_combine.restype = C.c_int
        lib.ark_fr_combine.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(sz), sz, vp, vp, sz, sz, vp]
        lib.ark_fr_to_mont.restype = C.c_int
        lib.ark_fr_to_mont.argtypes = [C.c_int, vp, sz, vp]
        lib.ark_fr_from_mont.restype = C.c_int
        lib.ark_fr_from_mont.argtypes = [C.c_int, vp, sz, vp]
        lib.ark_fr_t_vecs.restype = C.c_int
        lib.ark_fr_t_vecs.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), sz, vp, sz, vp, sz,
                                      vp, sz, C.POINTER(vp)]
        lib.ark_fr_spmv.restype = C.c_int
        lib.ark_fr_spmv.argtypes = [C.c_int, vp, vp, vp, sz, vp, sz, vp, sz, vp]
        lib.ark_points_fold.restype = C.c_int
        lib.ark_points_fold.argtypes = [C.c_int, vp, vp, sz, vp, C.c_int, vp]
        lib.ark_fr_inner_product.restype = C.c_int
        lib.ark_fr_inner_product.argtypes = [C.c_int, vp, vp, sz, vp]
        lib.ark_fr_powers.restype = C.c_int
        lib.ark_fr_powers.argtypes = [C.c_int, vp, sz, vp]
        _lib = lib
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fq_limbs(curve_id: int) -> int:
    return 4 if curve_id == 0 else 6


def msm(curve_id: int, bases_xy: np.ndarray, scalars: np.ndarray, is_inf: Optional[np.ndarray] = None,
        threads: Optional[int] = None) -> Tuple[np.ndarray, bool]:
    """ark-ec-style CPU MSM.  bases_xy (n, 2L) uint64 Montgomery affine; scalars (n, 4) canonical.
    threads: window-parallel workers (the result does not depend on it); None = min(cores, 20) -- the checker should not be the slow
    part of a parity test (round 6: the GPU suite spent most of its 13 minutes in single-threaded oracle MSMs); 1 = ark-ec's default
    features, what bench.py's `single_thread_value` times."""
    if threads is None:
        threads = max(1, min(os.cpu_count() or 1, 20))
    L = fq_limbs(curve_id)
    b = np.ascontiguousarray(bases_xy, dtype=np.uint64).reshape(-1, 2 * L)
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    n = min(b.shape[0], s.shape[0])
    inf = None if is_inf is None else np.ascontiguousarray(is_inf, dtype=np.uint8)
    out = np.zeros((2 * L,), dtype=np.uint64)
    oinf = C.c_uint8(0)
    rc = load().ark_msm(curve_id, _p(b), _p(inf), _p(s), n, threads, _p(out), C.byref(oinf))
    assert rc == 0
    return out, bool(oinf.value)


def rng_scalars(seed: int, n: int) -> np.ndarray:
    out = np.empty((n, 4), dtype=np.uint64)
    load().ark_rng_scalars(seed, n, _p(out))
    return out


def rng_frs(curve_id: int, seed: int, n: int) -> np.ndarray:
    """the scalar stream of amsm_vec_random: uniform in [0, r) (oracle/pyref.py:rng_fr)"""
    out = np.empty((n, 4), dtype=np.uint64)
    assert load().ark_rng_scalars_fr(curve_id, seed, n, _p(out)) == 0
    return out


def rng_points(curve_id: int, seed: int, n: int, threads: int = 8) -> np.ndarray:
    out = np.zeros((n, 2 * fq_limbs(curve_id)), dtype=np.uint64)
    rc = load().ark_rng_points(curve_id, seed, n, threads, _p(out))
    assert rc == 0
    return out


def fr_hadamard(curve_id: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 4)
    n = min(a.shape[0], b.shape[0])
    out = np.empty((n, 4), dtype=np.uint64)
    assert load().ark_fr_hadamard(curve_id, _p(a), _p(b), n, _p(out)) == 0
    return out


def fr_combine(curve_id: int, vecs: Sequence[np.ndarray], coeffs: np.ndarray,
               hiding: Optional[np.ndarray] = None) -> np.ndarray:
    vecs = [np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4) for v in vecs]
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    n = max([v.shape[0] for v in vecs] + ([hiding.shape[0]] if hiding is not None else [0]))
    ptrs = (C.c_void_p * max(len(vecs), 1))(*[v.ctypes.data for v in vecs])
    lens = (C.c_size_t * max(len(vecs), 1))(*[v.shape[0] for v in vecs])
    h = None if hiding is None else np.ascontiguousarray(hiding, dtype=np.uint64).reshape(-1, 4)
    out = np.empty((n, 4), dtype=np.uint64)
    assert load().ark_fr_combine(curve_id, ptrs, lens, len(vecs), _p(coeffs), _p(h), 0 if h is None else h.shape[0], n,
                                 _p(out)) == 0
    return out


def points_fold(curve_id: int, l_xy: np.ndarray, r_xy: np.ndarray, x: int, threads: int = 8) -> np.ndarray:
    """out[i] = l[i] + x * r[i] (affine Montgomery arrays, (0, 0) = identity; x a canonical Python int)"""
    L = fq_limbs(curve_id)
    l = np.ascontiguousarray(l_xy, dtype=np.uint64).reshape(-1, 2 * L)
    r = np.ascontiguousarray(r_xy, dtype=np.uint64).reshape(-1, 2 * L)
    assert l.shape == r.shape
    xw = np.array([(int(x) >> (64 * k)) & ((1 << 64) - 1) for k in range(4)], dtype=np.uint64)
    out = np.zeros_like(l)
    assert load().ark_points_fold(curve_id, _p(l), _p(r), l.shape[0], _p(xw), threads, _p(out)) == 0
    return out


def fr_inner_product(curve_id: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a, b = _m4(a), _m4(b)
    n = min(a.shape[0], b.shape[0])
    out = np.zeros((4,), dtype=np.uint64)
    assert load().ark_fr_inner_product(curve_id, _p(a), _p(b), n, _p(out)) == 0
    return out


def fr_powers(curve_id: int, point_mont: np.ndarray, n: int) -> np.ndarray:
    pt = np.ascontiguousarray(point_mont, dtype=np.uint64).reshape(4)
    out = np.empty((n, 4), dtype=np.uint64)
    assert load().ark_fr_powers(curve_id, _p(pt), n, _p(out)) == 0
    return out


def fr_to_mont(curve_id: int, a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(a)
    assert load().ark_fr_to_mont(curve_id, _p(a), a.shape[0], _p(out)) == 0
    return out


def fr_from_mont(curve_id: int, a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    out = np.empty_like(a)
    assert load().ark_fr_from_mont(curve_id, _p(a), a.shape[0], _p(out)) == 0
    return out


def _m4(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)


def fr_t_vecs(curve_id: int, a_vecs: Sequence[np.ndarray], b_vecs: Sequence[np.ndarray], mu: np.ndarray, hp_len: int,
              hiding: Optional[Tuple[np.ndarray, np.ndarray]] = None) -> list:
    """compute_t_vecs (src/hp_as/mod.rs:288-349) over Montgomery (len, 4) uint64 arrays -> 2n-1 arrays of hp_len."""
    a_vecs, b_vecs = [_m4(v) for v in a_vecs], [_m4(v) for v in b_vecs]
    n = len(a_vecs)
    assert n == len(b_vecs) and n >= 1
    mu = _m4(mu)
    assert mu.shape[0] >= n + (1 if hiding is not None else 0)
    ap = (C.c_void_p * n)(*[v.ctypes.data for v in a_vecs])
    al = (C.c_size_t * n)(*[v.shape[0] for v in a_vecs])
    bp = (C.c_void_p * n)(*[v.ctypes.data for v in b_vecs])
    bl = (C.c_size_t * n)(*[v.shape[0] for v in b_vecs])
    out = [np.empty((hp_len, 4), dtype=np.uint64) for _ in range(2 * n - 1)]
    op = (C.c_void_p * (2 * n - 1))(*[v.ctypes.data for v in out])
    ha = hb = None
    if hiding is not None:
        ha, hb = _m4(hiding[0]), _m4(hiding[1])
    rc = load().ark_fr_t_vecs(curve_id, ap, al, bp, bl, n, _p(mu), hp_len, _p(ha), 0 if ha is None else ha.shape[0], _p(hb),
                              0 if hb is None else hb.shape[0], op)
    assert rc == 0
    return out


def fr_spmv(curve_id: int, row_ptr: np.ndarray, col: np.ndarray, coeff: np.ndarray, inp: np.ndarray,
            wit: np.ndarray) -> np.ndarray:
    """matrix_vec_mul (src/r1cs_nark_as/r1cs_nark/mod.rs:443-462) on a CSR matrix; coeff / inp / wit Montgomery (k, 4)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
    col = np.ascontiguousarray(col, dtype=np.uint64)
    coeff, inp, wit = _m4(coeff), _m4(inp), _m4(wit)
    n_rows = row_ptr.shape[0] - 1
    out = np.empty((n_rows, 4), dtype=np.uint64)
    rc = load().ark_fr_spmv(curve_id, _p(row_ptr), _p(col), _p(coeff), n_rows, _p(inp), inp.shape[0], _p(wit), wit.shape[0],
                            _p(out))
    assert rc == 0, rc
    return out
