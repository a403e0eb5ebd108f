"""The C++ scheme driver include/amsm_r1cs_nark_as.hpp (ASForR1CSNark restated from src/r1cs_nark_as/mod.rs): compiles as
plain C++17 (CPU check); on a GPU it passes the reference's six-scenario template with and without zk and -- same
sponges, same hashes, same rng -- produces byte-identical accumulators to the Python mirror."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "r1cs_nark_as_check.cpp")
EXE = os.path.join(ROOT, "build", "r1cs_nark_as_check")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "accumulation_amd")
    # (compiled beside the target and moved into place: pytest -n workers build and RUN the same program at the same time)
    tmp = EXE + f".{os.getpid()}"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", tmp,
                           "-L", libdir, "-l:libamsm.so", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])
    os.replace(tmp, EXE)


def test_cpp_r1cs_nark_as_compiles(built_lib):
    build()
    assert os.path.exists(EXE)


def _template_and_cross_check(device):
    """device 0: the HIP path; -1: the library's host backend (AMSM_DEVICE_HOST) -- same program, same mirror, same bytes"""
    from accumulation_amd import Context, ffi
    from accumulation_amd import r1cs_nark as nark
    from accumulation_amd.r1cs_nark_as import ASForR1CSNark as AS
    from accumulation_amd.scalar_field import MODULI
    from tests.test_hp_as_scheme_gpu import SchemeRng
    from tests.test_r1cs_nark_as_scheme_gpu import NUM_CONSTRAINTS, NUM_INPUTS, generate_inputs
    from tests.test_r1cs_nark_gpu import dummy_circuit
    build()
    out = subprocess.run([EXE], capture_output=True, text=True, env=dict(os.environ, AMSM_CHECK_DEVICE=str(device)), timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln.split() for ln in out.stdout.splitlines()]
    assert ["done"] in lines
    ok = {(ln[1], ln[2]) for ln in lines if ln[0] == "scenario" and ln[3] == "ok"}
    names = ["single_input_init", "multiple_inputs_init", "simple_accumulation", "multiple_inputs_accumulation",
             "accumulators_only", "no_inputs_init"]
    assert ok == {(n, z) for n in names for z in ("zk", "no_zk")}
    vals = {ln[0]: ln[1:] for ln in lines if ln[0].startswith(("zk_", "nozk_"))}
    ctx = Context(ffi.AMSM_PALLAS, device=device)
    r = MODULI[ctx.curve]
    A, B, C_, _, _ = dummy_circuit(NUM_INPUTS, NUM_CONSTRAINTS, 2, 3, r)
    ipk = nark.index(ctx, A, B, C_, NUM_INPUTS + 1, NUM_INPUTS + 3, key_seed=31337)
    env = (ctx, ipk, r)
    pk, vk, dk = AS.index(ipk)

    def same_point(name, pt):
        got = vals[name]
        assert int(got[0]) == int(bool(pt[1])), name
        assert [int(x, 16) for x in got[1:]] == [int(v) for v in np.asarray(pt[0]).reshape(-1)], name

    for make_zk, tag in ((False, "nozk"), (True, "zk")):
        rng = SchemeRng(2024)
        inputs = generate_inputs(env, 7, make_zk, rng)
        old, start = [], 0
        for k in (1, 1, 2, 3):
            acc, proof = AS.prove(pk, inputs[start:start + k], old, rng if make_zk else None, None)
            start += k
            old.append(acc)
        i = acc.instance
        same_point(f"{tag}_comm_a", i.comm_a)
        same_point(f"{tag}_comm_b", i.comm_b)
        same_point(f"{tag}_comm_c", i.comm_c)
        same_point(f"{tag}_hp_comm_3", i.hp_instance.comm_3)
        words = [int(x, 16) for x in vals[f"{tag}_r1cs_input"][1:]]
        got_inputs = [sum(words[4 * j + t] << (64 * t) for t in range(4)) for j in range(len(words) // 4)]
        assert got_inputs == [x % r for x in i.r1cs_input], tag
    ctx.close()
    return out.stdout


@pytest.mark.gpu
def test_cpp_r1cs_nark_as_template_and_python_cross_check(built_lib):
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


def test_cpp_r1cs_nark_as_template_and_python_cross_check_on_the_host_backend(built_lib):
    """no GPU needed (-m "not gpu"): BASELINE.json config 1 'plumbing, no GPU', SURVEY.md section 8(b)"""
    _template_and_cross_check(-1)
