"""CPU tests: the C-ABI library builds, loads and exports every symbol include/amsm.h declares; with no GPU a GPU context
cannot be created (AMSM_E_NO_DEVICE, nothing falls back implicitly), while a context of the host backend (AMSM_DEVICE_HOST,
explicit) computes the same results through the same entry points."""
import ctypes as C
import os
import re

import pytest

from accumulation_amd import ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "amsm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(amsm_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(built_lib):
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(built_lib, s), f"libamsm.so does not export {s}"


def test_ffi_signatures_cover_header():
    assert sorted(ffi.SIGNATURES) == header_symbols()


def test_header_cites_reference_interfaces():
    """Every MSM / commit / vector entry point names the reference interface it replaces (file:line)."""
    src = open(os.path.join(ROOT, "include", "amsm.h")).read()
    for needle in ("src/hp_as/mod.rs:377", "src/hp_as/mod.rs:278-285", "src/hp_as/mod.rs:482-512",
                   "src/hp_as/mod.rs:288-349", "src/r1cs_nark_as/r1cs_nark/mod.rs", "src/ipa_pc_as/mod.rs:836",
                   "VariableBaseMSM::multi_scalar_mul", "PedersenCommitment::commit"):
        assert needle in src, needle


def test_strerror_and_stage_names(built_lib):
    assert built_lib.amsm_strerror(0) == b"ok"
    assert b"never chosen implicitly" in built_lib.amsm_strerror(ffi.AMSM_E_NO_DEVICE)
    n = built_lib.amsm_stage_count()
    names = [built_lib.amsm_stage_name(i).decode() for i in range(n)]
    assert "accum_l0" in names and "prep_chain" in names


def test_no_gpu_fails_loudly(built_lib, have_gpu):
    if have_gpu:
        pytest.skip("GPU present: the no-device path cannot be exercised")
    assert built_lib.amsm_device_count() == 0
    h = C.c_void_p()
    rc = built_lib.amsm_ctx_create(C.byref(h), ffi.AMSM_PALLAS, 0, None)
    assert rc == ffi.AMSM_E_NO_DEVICE and not h.value
    from accumulation_amd import Context
    with pytest.raises(ffi.AmsmError) as e:
        Context(ffi.AMSM_PALLAS)
    assert e.value.status == ffi.AMSM_E_NO_DEVICE


def test_host_backend_computes_results_where_there_is_no_gpu(built_lib, cref):
    """SURVEY.md section 8(b) / BASELINE.json config 1: with n_dev == 0 the SAME entry points return results, not an error --
    an MSM and a hiding Pedersen commitment through the raw C ABI against the C restatement."""
    import numpy as np
    h = C.c_void_p()
    assert built_lib.amsm_ctx_create_multi(C.byref(h), ffi.AMSM_PALLAS, None, 0) == ffi.AMSM_OK and h.value
    assert built_lib.amsm_ctx_is_host(h) == 1
    n = 777
    ck = C.c_void_p()
    assert built_lib.amsm_bases_generate(h, 0x5EED1001, n + 1, ffi.AMSM_BASES_DEFAULT, C.byref(ck)) == ffi.AMSM_OK
    xy = np.zeros((n + 1, 8), dtype=np.uint64)
    assert built_lib.amsm_bases_read(h, ck, 0, n + 1, xy.ctypes.data_as(C.c_void_p), None) == ffi.AMSM_OK
    assert np.array_equal(xy, cref.rng_points(ffi.AMSM_PALLAS, 0x5EED1001, n + 1))
    sc = cref.rng_scalars(99, n)
    out, inf = np.zeros(8, dtype=np.uint64), C.c_uint8(9)
    assert built_lib.amsm_msm(h, ck, 0, sc.ctypes.data_as(C.c_void_p), n, 0, out.ctypes.data_as(C.c_void_p), C.byref(inf)) == ffi.AMSM_OK
    ref, ref_inf = cref.msm(ffi.AMSM_PALLAS, xy[:n], sc)
    assert inf.value == ref_inf and np.array_equal(out, ref)
    elems, r = cref.fr_to_mont(ffi.AMSM_PALLAS, sc), cref.fr_to_mont(ffi.AMSM_PALLAS, cref.rng_scalars(5, 1))
    hid = np.ascontiguousarray(xy[n])
    assert built_lib.amsm_pedersen_commit(h, ck, elems.ctypes.data_as(C.c_void_p), n, r.ctypes.data_as(C.c_void_p),
                                          hid.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.byref(inf)) == ffi.AMSM_OK
    both, both_inf = cref.msm(ffi.AMSM_PALLAS, xy, np.concatenate([sc, cref.rng_scalars(5, 1)]))
    assert inf.value == both_inf and np.array_equal(out, both)
    built_lib.amsm_bases_free(ck)
    built_lib.amsm_ctx_destroy(h)


def test_invalid_arguments_rejected(built_lib):
    h = C.c_void_p()
    assert built_lib.amsm_ctx_create(None, 0, 0, None) == ffi.AMSM_E_INVALID_ARG
    assert built_lib.amsm_ctx_create(C.byref(h), 7, 0, None) == ffi.AMSM_E_INVALID_ARG
    assert built_lib.amsm_bases_len(None) == 0
    assert built_lib.amsm_partial_bytes(None) == 0
    assert built_lib.amsm_ctx_set_window(None, 8) == ffi.AMSM_E_INVALID_ARG


def test_product_does_not_import_oracle():
    """The shipped package must never route through oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "accumulation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "libark_msm" not in txt and "ark_msm.c" not in txt, f


def test_every_entry_point_survives_null_arguments(built_lib):
    """Each of the ABI's entry points called with NULL handles / NULL buffers / zero sizes returns (an error status, or
    the neutral value of a query) instead of dereferencing: the boundary is what a foreign host binds, and a binding bug
    must come back as a status.  Runs in a child process so that a crash is a test failure, not the end of the session."""
    import subprocess
    import sys
    probe = r"""
import ctypes as C, sys
from accumulation_amd import ffi
lib = ffi.load()
for name, (restype, argtypes) in ffi.SIGNATURES.items():
    args = [t(0) if t in (C.c_int, C.c_uint, C.c_size_t, C.c_uint64) else None for t in argtypes]
    r = getattr(lib, name)(*args)
    if restype is C.c_int and name not in ("amsm_device_count", "amsm_stage_count", "amsm_bases_precomputed", "amsm_bases_num_shards",
                                           "amsm_bases_window_bits") and argtypes and argtypes[0] is C.c_void_p:
        assert r in (ffi.AMSM_E_INVALID_ARG, ffi.AMSM_OK), (name, r)
print("survived", len(ffi.SIGNATURES))
"""
    p = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
    assert "survived" in p.stdout


def test_header_is_plain_c(built_lib, tmp_path):
    """include/amsm.h is the FFI surface a Rust / Go / C host binds: it must compile as C99 (no C++-isms) and link against
    the library with nothing but the C runtime."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "amsm.h"\n#include <stdio.h>\n'
                   'int main(void) { printf("%s %d %d\\n", amsm_strerror(0), amsm_device_count() >= 0, amsm_stage_count()); return 0; }\n')
    exe = tmp_path / "abi_c"
    libdir = os.path.join(ROOT, "accumulation_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe), "-L", libdir, "-lamsm", "-Wl,-rpath," + libdir])
    out = subprocess.check_output([str(exe)], text=True)
    assert out.startswith("ok 1 ")
