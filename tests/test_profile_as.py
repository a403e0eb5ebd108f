"""tools/profile_as.cpp -- the reference's benchmark harness (examples/scaling-as.rs:38-138) on the C++ drivers: compiles on a
CPU box; on a GPU runs all four schemes in the harness's shape (1 input + the same accumulator twice, zk) and in the n_all = 2
no-zk shape, with the SHA-256 stand-in and with the Poseidon sponge, checks verify / decide / the serialisation round trip
(deserialised accumulator still decides, re-serialises to the same bytes) and the `serialized_size()` figures against the
wire-format oracle's arithmetic."""
import json
import os
import subprocess

import pytest

from oracle import pyref as o
from oracle import pyref_ser as ser

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tools", "profile_as.cpp")
EXE = os.path.join(ROOT, "build", "profile_as")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "accumulation_amd")
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(libdir, "libamsm.so"))):
        return
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE + f".{os.getpid()}",
                           "-L", libdir, "-l:libamsm.so", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])
    os.replace(EXE + f".{os.getpid()}", EXE)  # (atomic: pytest -n workers may build the same program at once)


def test_profile_as_compiles(built_lib):
    build()
    assert os.path.exists(EXE)


def run(*args):
    build()
    out = subprocess.run([EXE, *args], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return [json.loads(line[5:]) for line in out.stdout.splitlines() if line.startswith("JSON ")], out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("sponge", ["sha256", "poseidon"])
def test_all_schemes_both_shapes(built_lib, sponge):
    rows, text = run("all", "6", "6", "--sponge", sponge, "--reps", "1")
    assert len(rows) == 10  # 4 accumulation schemes x 2 shapes + the NARK on its own (examples/scaling-nark.rs), zk off / on
    assert "Indexer:" in text and "Prover:" in text and "Accumulator witness size:" in text
    c = o.PALLAS
    P, F = ser.point_size(c), 32
    n = 64
    for r in rows:
        assert r["verified"] and r["decided"] and r["serialize_roundtrip_decides"], r
        assert r["sponge"] == sponge
        zk = r["zk"]
        if r["scheme"] == "hp_as":
            # instance: 3 points; witness: 2 vectors + Option<3 scalars>
            assert r["instance_bytes"] == 3 * P
            assert r["witness_bytes"] == 2 * (8 + F * n) + 1 + (3 * F if zk else 0)
            assert r["accumulator_bytes"] == r["instance_bytes"] + r["witness_bytes"]
        elif r["scheme"] == "r1cs_nark_as":
            # instance: Vec<F>(6) + 3 points + hp instance; witness: Vec<F>(2) + hp witness (2 vectors of n) + 2 Options
            assert r["instance_bytes"] == (8 + 6 * F) + 3 * P + 3 * P
            hp_w = 2 * (8 + F * n) + 1 + (3 * F if zk else 0)
            assert r["witness_bytes"] == (8 + 2 * F) + hp_w + 1 + (3 * F if zk else 0)
        elif r["scheme"] == "ipa_pc_as":
            # LabeledCommitment{label "", comm, shifted None, degree_bound None} + point + eval + Proof{l, r (6 each), key, c, 2 Options}
            proof = 2 * (8 + 6 * P) + P + F + 1 + (P if zk else 0) + 1 + (F if zk else 0)
            assert r["accumulator_bytes"] == (8 + P + 1 + 1) + 2 * F + proof
        elif r["scheme"] == "r1cs_nark":
            # Proof{first_msg: 3 points + Option<5 points>, second_msg: Vec<F>(num_constraints - 5 + 1) + Option<4 F>}
            n_wit = n - 5 + 1
            assert r["accumulator_bytes"] == 3 * P + 1 + (5 * P if zk else 0) + (8 + F * n_wit) + 1 + (4 * F if zk else 0)
        elif r["scheme"] == "trivial_pc_as":
            assert r["instance_bytes"] == (8 + P + 1) + 2 * F
            assert r["witness_bytes"] == 8 + (8 + F * n) + 1 + 1


@pytest.mark.gpu
def test_harness_shape_is_the_references(built_lib):
    """1 input + the same accumulator twice with zk: the hp_as proof then carries 2 (n_all - 1) = 4 product-polynomial
    commitments and the hiding commitments"""
    rows, _ = run("hp_as", "8", "8", "--shape", "harness", "--reps", "1")
    assert len(rows) == 1 and rows[0]["zk"] and rows[0]["shape"].startswith("harness")
    rows, _ = run("hp_as", "8", "8", "--shape", "n2", "--reps", "1", "--constant")
    assert len(rows) == 1 and not rows[0]["zk"] and rows[0]["decided"]
