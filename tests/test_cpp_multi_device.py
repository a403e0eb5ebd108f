"""Multi-device context behind the C ABI (amsm_ctx_create_multi, include/amsm.h): tests/cpp/multi_device_check.cpp uses
ONLY amsm.h and must get bit-identical results from a sharded key and from the single-device key, through every entry point
that accepts a sharded key.  On the 1-GPU test box all shards map to device 0 (peer-copy exchange); on a node the distinct
devices and RCCL are used."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "multi_device_check.cpp")
EXE = os.path.join(ROOT, "build", "multi_device_check")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "accumulation_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
                           "-L", libdir, "-l:libamsm.so", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])


def test_multi_device_check_compiles(built_lib):
    build()
    assert os.path.exists(EXE)


def test_multi_context_without_gpu_fails_loudly(built_lib):
    """nothing falls back implicitly: creation over device ids reports AMSM_E_NO_DEVICE on a box without a GPU"""
    import ctypes as C
    from accumulation_amd import ffi
    lib = built_lib
    if lib.amsm_device_count() > 0:
        pytest.skip("has a GPU")
    h = C.c_void_p()
    ids = (C.c_int * 2)(0, 0)
    assert lib.amsm_ctx_create_multi(C.byref(h), ffi.AMSM_PALLAS, ids, 2) == ffi.AMSM_E_NO_DEVICE
    assert lib.amsm_ctx_create_multi(C.byref(h), ffi.AMSM_PALLAS, None, 2) == ffi.AMSM_E_INVALID_ARG


@pytest.mark.gpu
@pytest.mark.parametrize("shards,n,curve", [(1, 5000, 0), (2, 50000, 0), (3, 1000, 0), (4, 1 << 17, 0), (2, 20000, 1),
                                            (5, 3, 0)])
def test_sharded_key_is_bit_identical(built_lib, shards, n, curve):
    build()
    out = subprocess.run([EXE, str(shards), str(n), str(curve)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert f"ok shards={shards} n={n} curve={curve}" in out.stdout
    assert "collective" in out.stdout
