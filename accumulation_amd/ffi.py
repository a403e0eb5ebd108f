"""ctypes binding of the C ABI in include/amsm.h (libamsm.so, built in-tree by build.py).

Nothing falls back implicitly: importing works without a GPU, a GPU context cannot be created there
(AMSM_E_NO_DEVICE), and `load()` raises if the shared library itself is missing.  The library's host
backend is a device of its own (AMSM_DEVICE_HOST) that a caller has to ask for.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# AMSM_LIB_PATH: load another build of the same library (A/B of compile-time variants, tools/build_variant.sh)
LIB_PATH = os.environ.get("AMSM_LIB_PATH") or os.path.join(_HERE, "libamsm.so")

AMSM_PALLAS = 0
AMSM_BLS12_381_G1 = 1

AMSM_OK = 0
AMSM_E_INVALID_ARG = -1
AMSM_E_OOM = -2
AMSM_E_HIP = -3
AMSM_E_UNSUPPORTED = -4
AMSM_E_NO_DEVICE = -5
AMSM_E_SCALAR_RANGE = -6
AMSM_E_RCCL = -7

AMSM_DEVICE_HOST = -1

AMSM_BASES_DEFAULT = 0
AMSM_BASES_PRECOMPUTE = 1
AMSM_BASES_NO_PRECOMPUTE = 2
AMSM_BASES_NO_DIRECT_TABLE = 4
AMSM_BASES_NO_TWIN = 8
AMSM_BASES_REPLICATE = 16

_vp = C.c_void_p
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_sz = C.c_size_t

# name -> (restype, argtypes); must list every symbol include/amsm.h declares (tests check this)
SIGNATURES = {
    "amsm_strerror": (C.c_char_p, [C.c_int]),
    "amsm_device_count": (C.c_int, []),
    "amsm_ctx_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, _vp]),
    "amsm_ctx_is_host": (C.c_int, [_vp]),
    "amsm_ctx_create_multi": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(C.c_int), C.c_int]),
    "amsm_ctx_num_devices": (C.c_int, [_vp]),
    "amsm_ctx_shard": (_vp, [_vp, C.c_int]),
    "amsm_ctx_collective": (C.c_char_p, [_vp]),
    "amsm_ctx_collectives": (C.c_ulonglong, [_vp]),
    "amsm_ctx_two_valued_msms": (C.c_ulonglong, [_vp]),
    "amsm_ctx_unit_scalar_msms": (C.c_ulonglong, [_vp]),
    "amsm_ctx_direct_sum_msms": (C.c_ulonglong, [_vp]),
    "amsm_ctx_shared_bucket_msms": (C.c_ulonglong, [_vp]),
    "amsm_ctx_pipeline_stats": (C.c_int, [_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "amsm_ctx_pipeline_stats_small": (C.c_int, [_vp, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "amsm_ctx_destroy": (None, [_vp]),
    "amsm_ctx_curve": (C.c_int, [_vp]),
    "amsm_ctx_fq_limbs": (C.c_int, [_vp]),
    "amsm_ctx_set_window": (C.c_int, [_vp, C.c_int]),
    "amsm_ctx_synchronize": (C.c_int, [_vp]),
    "amsm_ctx_memory": (C.c_int, [_vp, C.POINTER(_sz), C.POINTER(_sz), C.POINTER(_sz)]),
    "amsm_ctx_trim": (C.c_int, [_vp]),
    "amsm_ctx_set_profiling": (C.c_int, [_vp, C.c_int]),
    "amsm_stage_count": (C.c_int, []),
    "amsm_stage_name": (C.c_char_p, [C.c_int]),
    "amsm_ctx_stage_ms": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_float)]),
    "amsm_bases_load": (C.c_int, [_vp, _vp, _vp, _sz, C.c_uint, C.POINTER(_vp)]),
    "amsm_bases_generate": (C.c_int, [_vp, C.c_uint64, _sz, C.c_uint, C.POINTER(_vp)]),
    "amsm_bases_read": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "amsm_bases_len": (_sz, [_vp]),
    "amsm_bases_num_shards": (C.c_int, [_vp]),
    "amsm_bases_shard_range": (C.c_int, [_vp, C.c_int, C.POINTER(_sz), C.POINTER(_sz)]),
    "amsm_bases_precomputed": (C.c_int, [_vp]),
    "amsm_bases_window_bits": (C.c_int, [_vp]),
    "amsm_bases_free": (None, [_vp]),
    "amsm_msm": (C.c_int, [_vp, _vp, _sz, _vp, _sz, C.c_int, _vp, _vp]),
    "amsm_msm_device": (C.c_int, [_vp, _vp, _sz, _vp, _sz, C.c_int, _vp, _vp]),
    "amsm_msm_batch": (C.c_int, [_vp, _vp, _sz, C.POINTER(_vp), _sz, _sz, C.c_int, _vp, _vp]),
    "amsm_pedersen_commit_batch": (C.c_int, [_vp, _vp, C.POINTER(_vp), C.POINTER(_sz), _sz, C.POINTER(_vp), _vp, _vp, _vp]),
    "amsm_msm_batch_device": (C.c_int, [_vp, _vp, _sz, C.POINTER(_vp), _sz, _sz, C.c_int, _vp, _vp]),
    "amsm_msm_batch_sharded_device": (C.c_int, [_vp, _vp, C.POINTER(_vp), _sz, C.c_int, _vp, _vp]),
    "amsm_msm_multi_device": (C.c_int, [_vp, _vp, _sz, C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_sz), C.c_int, _vp, _vp]),
    "amsm_msm_grouped_device": (C.c_int, [_vp, _vp, _sz, _vp, _sz, C.c_int, C.c_uint, _vp, _vp]),
    "amsm_partial_bytes": (_sz, [_vp]),
    "amsm_msm_partial_device": (C.c_int, [_vp, _vp, _sz, _vp, _sz, C.c_int, _vp]),
    "amsm_partials_combine": (C.c_int, [_vp, _vp, _sz, _vp, _vp]),
    "amsm_msm_partial_batch_device": (C.c_int, [_vp, _vp, _sz, C.POINTER(_vp), _sz, _sz, C.c_int, _vp]),
    "amsm_partials_combine_batch": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "amsm_pedersen_commit": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "amsm_pedersen_commit_device": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp]),
    "amsm_host_lincomb": (C.c_int, [C.c_int, _vp, _vp, _vp, _sz, _vp, _vp]),
    "amsm_host_lincomb_batch": (C.c_int, [C.c_int, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amsm_host_threads": (C.c_int, []),
    "amsm_fr_mul": (C.c_int, [C.c_int, _vp, _vp, _sz, _vp]),
    "amsm_fr_add": (C.c_int, [C.c_int, _vp, _vp, _sz, _vp]),
    "amsm_fr_sub": (C.c_int, [C.c_int, _vp, _vp, _sz, _vp]),
    "amsm_fr_inv": (C.c_int, [C.c_int, _vp, _sz, _vp]),
    "amsm_fr_to_mont": (C.c_int, [C.c_int, _vp, _sz, _vp]),
    "amsm_fr_from_mont": (C.c_int, [C.c_int, _vp, _sz, _vp]),
    "amsm_bases_memory": (C.c_int, [_vp, _vp, _vp, _vp]),
    "amsm_bases_prebuild_twin": (C.c_int, [_vp, _vp]),
    "amsm_bases_tables": (C.c_int, [_vp, _vp]),
    "amsm_ctx_set_table_budget": (C.c_int, [_vp, _sz]),
    "amsm_ctx_tables_denied": (C.c_ulonglong, [_vp]),
    "amsm_host_register": (C.c_int, [_vp, _sz]),
    "amsm_host_unregister": (C.c_int, [_vp]),
    "amsm_host_is_pinned": (C.c_int, [_vp]),
    "amsm_fr_serialized_size": (_sz, [C.c_int]),
    "amsm_point_serialized_size": (_sz, [C.c_int, C.c_int]),
    "amsm_fr_serialize": (C.c_int, [C.c_int, _vp, _sz, _vp]),
    "amsm_fr_deserialize": (C.c_int, [C.c_int, _vp, _sz, _vp]),
    "amsm_points_serialize": (C.c_int, [C.c_int, _vp, _vp, _sz, C.c_int, _vp]),
    "amsm_points_deserialize": (C.c_int, [C.c_int, _vp, _sz, C.c_int, _vp, _vp]),
    "amsm_poseidon_new": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "amsm_poseidon_clone": (C.c_int, [_vp, C.POINTER(_vp)]),
    "amsm_poseidon_free": (None, [_vp]),
    "amsm_poseidon_fork": (C.c_int, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "amsm_poseidon_absorb_native": (C.c_int, [_vp, _vp, _sz]),
    "amsm_poseidon_absorb_u64": (C.c_int, [_vp, C.c_uint64]),
    "amsm_poseidon_absorb_bytes": (C.c_int, [_vp, _vp, _sz]),
    "amsm_poseidon_absorb_points": (C.c_int, [_vp, _vp, _vp, _sz]),
    "amsm_poseidon_squeeze_native": (C.c_int, [_vp, _sz, _vp]),
    "amsm_poseidon_squeeze_bits": (C.c_int, [_vp, _sz, _vp]),
    "amsm_poseidon_squeeze_nonnative": (C.c_int, [_vp, C.c_uint, _sz, _vp]),
    "amsm_poseidon_permute": (C.c_int, [C.c_int, _vp]),
    "amsm_poseidon_round_constants": (C.c_int, [C.c_int, _vp]),
    "amsm_vec_fill": (C.c_int, [_vp, _vp, _sz, _vp]),
    "amsm_dev_alloc": (C.c_int, [_vp, _sz, C.POINTER(_vp)]),
    "amsm_dev_free": (C.c_int, [_vp, _vp]),
    "amsm_dev_upload": (C.c_int, [_vp, _vp, _vp, _sz]),
    "amsm_dev_download": (C.c_int, [_vp, _vp, _vp, _sz]),
    "amsm_ctx_set_replicate_below": (C.c_int, [_vp, _sz]),
    "amsm_bases_replicas": (C.c_int, [_vp]),
    "amsm_ctx_replicated_msms": (C.c_ulonglong, [_vp]),
    "amsm_ipa_jump_fold": (C.c_int, [_vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "amsm_msm_oneshot": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _sz, C.c_int, _vp, _vp]),
    "amsm_vec_random": (C.c_int, [_vp, C.c_uint64, _sz, C.c_int, _vp]),
    "amsm_vec_hadamard": (C.c_int, [_vp, _vp, _vp, _vp, _sz]),
    "amsm_vec_combine": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_sz), _sz, _vp, _vp, _sz, _vp, _sz]),
    "amsm_bases_from_device": (C.c_int, [_vp, _vp, _sz, C.c_uint, C.POINTER(_vp)]),
    "amsm_bases_device_ptr": (_vp, [_vp]),
    "amsm_points_fold": (C.c_int, [_vp, _vp, _vp, _sz, _vp, C.c_uint, _vp]),
    "amsm_bases_fold": (C.c_int, [_vp, _vp, _sz, _vp, C.c_uint, C.POINTER(_vp)]),
    "amsm_vec_inner_product": (C.c_int, [_vp, _vp, _vp, _sz, _vp]),
    "amsm_vec_powers": (C.c_int, [_vp, _vp, _sz, _vp]),
    "amsm_ipa_check_poly_coeffs": (C.c_int, [_vp, _vp, _sz, _vp]),
    "amsm_ipa_round_scalars": (C.c_int, [_vp, _vp, _sz, _sz, _vp, _vp, _vp]),
    "amsm_ipa_round": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amsm_ipa_round_fused": (C.c_int, [_vp, _vp, _vp, _sz, _sz, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "amsm_matrix_load": (C.c_int, [_vp, _vp, _vp, _vp, _sz, _sz, C.POINTER(_vp)]),
    "amsm_matrix_rows": (_sz, [_vp]),
    "amsm_matrix_free": (None, [_vp]),
    "amsm_matrix_vec_mul": (C.c_int, [_vp, _vp, _vp, _sz, _vp, _sz, _vp]),
    "amsm_hp_t_vecs": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_sz), _sz, _vp, _sz,
                                 _vp, _sz, _vp, _sz, C.POINTER(_vp), _sz]),
}

_lib: Optional[C.CDLL] = None


class AmsmError(RuntimeError):
    def __init__(self, status: int, where: str):
        self.status = status
        msg = load().amsm_strerror(status)
        super().__init__(f"{where}: {msg.decode() if msg else status} ({status})")


def load() -> C.CDLL:
    """Load libamsm.so (raises if it has not been built -- no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()' or python -m accumulation_amd.build)")
    # PyTorch ships its own copy of the HIP runtime (same soname as /opt/rocm's): whichever is loaded first serves
    # both, and torch only finds its GPUs through its own.  Load torch's first when torch is installed, so that
    # device tensors (RCCL all-gather buffers of dist.py) and this library can live in one process in either order.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, where: str) -> None:
    if status != AMSM_OK:
        raise AmsmError(status, where)
