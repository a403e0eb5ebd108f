"""The C++ scheme driver include/amsm_trivial_pc_as.hpp (ASForTrivialPC restated from src/trivial_pc_as/mod.rs): compiles
as plain C++17 (CPU check); on a GPU it passes the reference's six-scenario template and -- same sponge, same rng, same
inputs -- produces a byte-identical accumulator to the Python mirror accumulation_amd/trivial_pc_as.py."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "trivial_pc_as_check.cpp")
EXE = os.path.join(ROOT, "build", "trivial_pc_as_check")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "accumulation_amd")
    # (compiled beside the target and moved into place: pytest -n workers build and RUN the same program at the same time)
    tmp = EXE + f".{os.getpid()}"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", tmp,
                           "-L", libdir, "-l:libamsm.so", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])
    os.replace(tmp, EXE)


def test_cpp_trivial_pc_as_compiles(built_lib):
    build()
    assert os.path.exists(EXE)


def _template_and_cross_check(device):
    """device 0: the HIP path; -1: the library's host backend (AMSM_DEVICE_HOST) -- same program, same mirror, same bytes"""
    from accumulation_amd import Context, ffi
    from accumulation_amd.scalar_field import Fr
    from accumulation_amd.trivial_pc_as import ASForTrivialPC as AS, TrivialPC
    from tests.test_hp_as_scheme_gpu import SchemeRng
    from tests.test_trivial_pc_as_scheme_gpu import DEGREE, generate_inputs
    build()
    out = subprocess.run([EXE], capture_output=True, text=True, env=dict(os.environ, AMSM_CHECK_DEVICE=str(device)), timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln.split() for ln in out.stdout.splitlines()]
    assert ["done"] in lines and ["tampered_decide", "0"] in lines
    ok = {ln[1] for ln in lines if ln[0] == "scenario" and ln[2] == "ok"}
    assert ok == {"single_input_init", "multiple_inputs_init", "simple_accumulation", "multiple_inputs_accumulation",
                  "accumulators_only", "no_inputs_init"}
    vals = {ln[0]: ln[1:] for ln in lines if ln[0].startswith("acc_")}
    ctx = Context(ffi.AMSM_PALLAS, device=device)
    fr = Fr(ctx.curve)
    pp = TrivialPC.setup(ctx, DEGREE, seed=0x7121A1)
    ck, _ = TrivialPC.trim(pp, DEGREE)
    pk, vk, dk = AS.index(pp, DEGREE)
    inputs = generate_inputs((ctx, pp), ck, 7, SchemeRng(777))
    old, start = [], 0
    for k in (1, 1, 2, 3):
        acc, proof = AS.prove(pk, inputs[start:start + k], old, None, None)
        start += k
        old.append(acc)
    comm = acc.instance.commitment.elem
    assert int(vals["acc_comm"][0]) == int(bool(comm[1]))
    assert [int(x, 16) for x in vals["acc_comm"][1:]] == [int(v) for v in np.asarray(comm[0]).reshape(-1)]
    assert [int(x, 16) for x in vals["acc_point"][1:]] == [int(v) for v in fr.to_limbs(acc.instance.point)]
    assert [int(x, 16) for x in vals["acc_eval"][1:]] == [int(v) for v in fr.to_limbs(acc.instance.eval)]
    ctx.close()
    return out.stdout


@pytest.mark.gpu
def test_cpp_trivial_pc_as_template_and_python_cross_check(built_lib):
    gpu = _template_and_cross_check(0)
    # ... and the host backend behind the same ABI prints the same accumulators, byte for byte
    host = subprocess.run([EXE], capture_output=True, text=True, timeout=900, env=dict(os.environ, AMSM_CHECK_DEVICE="-1"))
    assert host.returncode == 0, host.stdout + host.stderr
    assert host.stdout == gpu
    # ... and so does a multi-device context of two and of five shards (include/amsm.hpp Context(curve, devices))
    for shards in ("2", "5"):
        many = subprocess.run([EXE], capture_output=True, text=True, timeout=900, env=dict(os.environ, AMSM_CHECK_SHARDS=shards))
        assert many.returncode == 0, many.stdout + many.stderr
        assert many.stdout == gpu, shards


def test_cpp_trivial_pc_as_template_and_python_cross_check_on_the_host_backend(built_lib):
    """no GPU needed (-m "not gpu"): BASELINE.json config 1 'plumbing, no GPU', SURVEY.md section 8(b)"""
    _template_and_cross_check(-1)
