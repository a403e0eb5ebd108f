"""The host backend of the C ABI (include/amsm.h: AMSM_DEVICE_HOST; SURVEY.md section 8(b) "CPU fallback selected by n_dev == 0",
BASELINE.json config 1 "plumbing, no GPU").  This directory re-collects the parity tests the HIP path passes on the GPU box with
every Context on the host device (conftest.py) -- MSMs against the golden fixtures and the C / big-int oracles, the vector kernels,
key folds, the IPA opening, the scheme mirrors' six-scenario templates and their accumulation layers against oracle/pyref_as.py --
so they run without a GPU (-m "not gpu"); tests/test_host_backend_gpu.py checks on the GPU box that both backends return the
same bytes.  This module: what is particular to the backend itself."""
import numpy as np
import pytest

from accumulation_amd import ffi
from accumulation_amd.engine import Context


def test_context_is_host_and_never_implicit(built_lib):
    ctx = Context(ffi.AMSM_PALLAS)
    assert built_lib.amsm_ctx_is_host(ctx._h) == 1 and built_lib.amsm_ctx_num_devices(ctx._h) == 1
    st = ctx.pipeline_stats()
    assert st["bucket_per_lane"] == 0 and st["direct_sum"] == 0  # no HIP pipeline ran
    ctx.close()
    import ctypes as C
    h = C.c_void_p()
    # n_dev == 0 is the other spelling; a stream makes no sense on the host
    assert built_lib.amsm_ctx_create_multi(C.byref(h), ffi.AMSM_PALLAS, None, 0) == ffi.AMSM_OK
    assert built_lib.amsm_ctx_is_host(h) == 1
    built_lib.amsm_ctx_destroy(h)
    assert built_lib.amsm_ctx_create(C.byref(h), ffi.AMSM_PALLAS, ffi.AMSM_DEVICE_HOST, C.c_void_p(1)) == ffi.AMSM_E_INVALID_ARG
    if built_lib.amsm_device_count() == 0:  # a GPU context is never granted on the host's behalf
        assert built_lib.amsm_ctx_create(C.byref(h), ffi.AMSM_PALLAS, 0, None) == ffi.AMSM_E_NO_DEVICE


def test_host_keys_are_plain_and_do_not_mix_with_gpu_contexts(built_lib):
    from accumulation_amd import CommitterKey
    ctx = Context(ffi.AMSM_PALLAS)
    ck = CommitterKey.generate(ctx, 7, 64, ffi.AMSM_BASES_PRECOMPUTE)  # a hint on this backend
    assert not ck.precomputed and built_lib.amsm_bases_window_bits(ck._h) == 0
    assert built_lib.amsm_bases_device_ptr(ck._h)  # host memory, C-ABI radix
    import ctypes as C
    raw = np.ctypeslib.as_array((C.c_uint64 * (64 * 8)).from_address(built_lib.amsm_bases_device_ptr(ck._h)))
    assert np.array_equal(np.asarray(raw).reshape(64, 8), ck.read()[0].reshape(64, 8))
    ck.free()
    ctx.close()


