"""The C++ driver include/amsm_r1cs_nark.hpp (R1CSNark::{index, prove, verify} restated from
src/r1cs_nark_as/r1cs_nark/mod.rs): compiles as plain C++17 (CPU check); on a GPU it passes the reference's
`test_simple_circuit` with and without zk and -- same sponge, same matrix hash, same rng -- produces byte-identical proofs
to the Python mirror accumulation_amd/r1cs_nark.py."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "r1cs_nark_check.cpp")
EXE = os.path.join(ROOT, "build", "r1cs_nark_check")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "accumulation_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
                           "-L", libdir, "-l:libamsm.so", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])


def test_cpp_r1cs_nark_compiles(built_lib):
    build()
    assert os.path.exists(EXE)


def _template_and_cross_check(device):
    """device 0: the HIP path; -1: the library's host backend (AMSM_DEVICE_HOST) -- same program, same mirror, same bytes"""
    from accumulation_amd import Context, ffi
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.scalar_field import Fr
    from accumulation_amd.sponge import Sha256Sponge
    from oracle import pyref as o
    from tests import helpers as h
    from tests.test_hp_as_scheme_gpu import SchemeRng
    from tests.test_r1cs_nark_gpu import dummy_circuit
    build()
    out = subprocess.run([EXE], capture_output=True, text=True, env=dict(os.environ, AMSM_CHECK_DEVICE=str(device)), timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln.split() for ln in out.stdout.splitlines()]
    assert ["done"] in lines and ["mode", "nozk", "ok"] in lines and ["mode", "zk", "ok"] in lines
    vals = {ln[0]: ln[1:] for ln in lines}
    c = o.PALLAS
    ctx = Context(ffi.AMSM_PALLAS, device=device)
    fr = Fr(ctx.curve)
    A, B, C_, _, _ = dummy_circuit(5, 100, 2, 3, c.r)
    ipk = nark.index(ctx, A, B, C_, 6, 8, key_seed=7)
    assert bytes(int(x, 16) for x in vals["matrices_hash"][1:]) == ipk.index_info.matrices_hash

    def same_point(name, pt):
        got = vals[name]
        assert int(got[0]) == int(bool(pt[1])), name
        assert [int(x, 16) for x in got[1:]] == [int(v) for v in np.asarray(pt[0]).reshape(-1)], name

    for make_zk, tag in ((False, "nozk"), (True, "zk")):
        rng = SchemeRng(9)
        for _ in range(3):
            a, b = rng.field() % c.r, rng.field() % c.r
            _, _, _, inst, w = dummy_circuit(5, 100, a, b, c.r)
            proof = nark.prove(ipk, inst, ctx.upload(h.fr_mont_np(c, w)), make_zk, Sha256Sponge(), rng if make_zk else None)
        f = proof.first_msg
        same_point(f"{tag}_comm_a", f.comm_a)
        same_point(f"{tag}_comm_b", f.comm_b)
        same_point(f"{tag}_comm_c", f.comm_c)
        if make_zk:
            same_point("zk_comm_r_a", f.randomness.comm_r_a)
            same_point("zk_comm_1", f.randomness.comm_1)
            same_point("zk_comm_2", f.randomness.comm_2)
            s = proof.second_msg.randomness
            assert [int(x, 16) for x in vals["zk_sigma_a"][1:]] == [int(v) for v in fr.to_limbs(s.sigma_a)]
            assert [int(x, 16) for x in vals["zk_sigma_o"][1:]] == [int(v) for v in fr.to_limbs(s.sigma_o)]
    ctx.close()
    return out.stdout


@pytest.mark.gpu
def test_cpp_r1cs_nark_simple_circuit_and_python_cross_check(built_lib):
    gpu = _template_and_cross_check(0)
    # ... and the host backend behind the same ABI prints the same accumulators, byte for byte
    host = subprocess.run([EXE], capture_output=True, text=True, timeout=900, env=dict(os.environ, AMSM_CHECK_DEVICE="-1"))
    assert host.returncode == 0, host.stdout + host.stderr
    assert host.stdout == gpu


def test_cpp_r1cs_nark_simple_circuit_and_python_cross_check_on_the_host_backend(built_lib):
    """no GPU needed (-m "not gpu"): BASELINE.json config 1 'plumbing, no GPU', SURVEY.md section 8(b)"""
    _template_and_cross_check(-1)
