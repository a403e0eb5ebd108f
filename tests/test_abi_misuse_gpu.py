"""The C ABI with a LIVE context and nothing else: every context-taking entry point called with NULL keys / buffers /
outputs and zero sizes must come back with a status (invalid argument, or ok for the calls that are no-ops at size 0)
-- never dereference.  Runs in a child process so that a crash is a test failure."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import ctypes as C
from accumulation_amd import ffi
lib = ffi.load()
NOT_CTX_FIRST = {"amsm_bases_len", "amsm_bases_free", "amsm_bases_precomputed", "amsm_bases_num_shards", "amsm_bases_window_bits",
                 "amsm_bases_shard_range", "amsm_bases_device_ptr", "amsm_bases_memory", "amsm_bases_replicas", "amsm_matrix_rows", "amsm_matrix_free", "amsm_ctx_destroy"}
PREFIXES = ("amsm_msm", "amsm_ctx", "amsm_bases", "amsm_vec", "amsm_dev", "amsm_ipa", "amsm_matrix", "amsm_hp", "amsm_pedersen",
            "amsm_partials", "amsm_points_fold")
for curve in (ffi.AMSM_PALLAS, ffi.AMSM_BLS12_381_G1):
    h = C.c_void_p()
    assert lib.amsm_ctx_create(C.byref(h), curve, 0, None) == 0
    n = 0
    for name, (restype, argtypes) in ffi.SIGNATURES.items():
        if not argtypes or argtypes[0] is not C.c_void_p or name in NOT_CTX_FIRST or not name.startswith(PREFIXES):
            continue
        args = [h] + [t(0) if t in (C.c_int, C.c_uint, C.c_size_t, C.c_uint64) else None for t in argtypes[1:]]
        r = getattr(lib, name)(*args)
        if restype is C.c_int and name not in ("amsm_ctx_num_devices", "amsm_ctx_curve", "amsm_ctx_fq_limbs"):
            assert r in (ffi.AMSM_OK, ffi.AMSM_E_INVALID_ARG, ffi.AMSM_E_UNSUPPORTED), (name, r)
        n += 1
    assert lib.amsm_ctx_synchronize(h) == 0  # the context is still usable
    lib.amsm_ctx_destroy(h)
print("survived", n)
"""


def test_context_entry_points_survive_null_arguments():
    p = subprocess.run([sys.executable, "-c", PROBE], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
    assert "survived" in p.stdout
