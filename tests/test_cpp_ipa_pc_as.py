"""The C++ scheme driver include/amsm_ipa_pc_as.hpp (InnerProductArgPC + AtomicASForInnerProductArgPC restated from
src/ipa_pc_as/mod.rs and the ipa_pc interface it calls): compiles as plain C++17 (CPU check); on a GPU it passes the
reference's six-scenario template with and without zk and -- same sponge, same rng -- produces byte-identical accumulators
to the Python mirror."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "ipa_pc_as_check.cpp")
EXE = os.path.join(ROOT, "build", "ipa_pc_as_check")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "accumulation_amd")
    # (compiled beside the target and moved into place: pytest -n workers build and RUN the same program at the same time)
    tmp = EXE + f".{os.getpid()}"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", tmp,
                           "-L", libdir, "-l:libamsm.so", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])
    os.replace(tmp, EXE)


def test_cpp_ipa_pc_as_compiles(built_lib):
    build()
    assert os.path.exists(EXE)


def _template_and_cross_check(device):
    """device 0: the HIP path; -1: the library's host backend (AMSM_DEVICE_HOST) -- same program, same mirror, same bytes"""
    from accumulation_amd import Context, ffi
    from accumulation_amd.ipa_pc import InnerProductArgPC as IpaPC
    from accumulation_amd.ipa_pc_as import AtomicASForInnerProductArgPC as AS
    from accumulation_amd.scalar_field import MODULI
    from tests.test_hp_as_scheme_gpu import SchemeRng
    from tests.test_ipa_gpu import DEGREE, generate_inputs
    build()
    out = subprocess.run([EXE], capture_output=True, text=True, env=dict(os.environ, AMSM_CHECK_DEVICE=str(device)), timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln.split() for ln in out.stdout.splitlines()]
    assert ["done"] in lines
    assert ["ipa_pc", "no_zk", "1", "0", "0"] in lines and ["ipa_pc", "zk", "1", "0", "0"] in lines
    assert ["fold_invariance", "1"] in lines
    assert ["missing_rng", "raised"] in lines and ["malformed_input", "raised"] in lines
    ok = {(ln[1], ln[2]) for ln in lines if ln[0] == "scenario" and ln[3] == "ok"}
    names = ["single_input_init", "multiple_inputs_init", "simple_accumulation", "multiple_inputs_accumulation",
             "accumulators_only", "no_inputs_init"]
    assert ok == {(n, z) for n in names for z in ("zk", "no_zk")}
    vals = {ln[0]: ln[1:] for ln in lines if ln[0].startswith(("zk_", "nozk_"))}
    ctx = Context(ffi.AMSM_PALLAS, device=device)
    r = MODULI[ctx.curve]
    assert DEGREE == 11
    pp = IpaPC.setup(ctx, DEGREE, seed=0xABCDEF)
    pk, vk, dk = AS.index(pp, DEGREE)

    def same_point(name, pt):
        got = vals[name]
        assert int(got[0]) == int(bool(pt[1])), name
        assert [int(x, 16) for x in got[1:]] == [int(v) for v in np.asarray(pt[0]).reshape(-1)], name

    def same_scalar(name, v):
        words = [int(x, 16) for x in vals[name][1:]]
        assert sum(w << (64 * t) for t, w in enumerate(words)) == v % r, name

    for make_zk, tag in ((False, "nozk"), (True, "zk")):
        rng = SchemeRng(4096)
        inputs = generate_inputs((ctx, pp), pk, 7, make_zk, rng)
        old, start = [], 0
        for k in (1, 1, 2, 3):
            acc, proof = AS.prove(pk, inputs[start:start + k], [a.instance for a in old], rng if make_zk else None, None)
            start += k
            old.append(acc)
        i = acc.instance
        same_point(f"{tag}_comm", i.ipa_commitment.comm)
        same_point(f"{tag}_final_comm_key", i.ipa_proof.final_comm_key)
        same_point(f"{tag}_l_last", i.ipa_proof.l_vec[-1])
        same_point(f"{tag}_r_first", i.ipa_proof.r_vec[0])
        same_scalar(f"{tag}_point", i.point)
        same_scalar(f"{tag}_evaluation", i.evaluation)
        same_scalar(f"{tag}_c", i.ipa_proof.c)
    ctx.close()
    return out.stdout


@pytest.mark.gpu
def test_cpp_ipa_pc_as_template_and_python_cross_check(built_lib):
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


def test_cpp_ipa_pc_as_template_and_python_cross_check_on_the_host_backend(built_lib):
    """no GPU needed (-m "not gpu"): BASELINE.json config 1 'plumbing, no GPU', SURVEY.md section 8(b)"""
    _template_and_cross_check(-1)
