"""The C++ scheme drivers -- the path bench.py times for `accumulations/sec` (tools/profile_as.cpp, the reference's harness
examples/scaling-as.rs:38-138) -- against the oracle-checked Python mirrors, byte for byte (VERDICT r4 "next" item 1):
`profile_as --dump` writes the serialised new accumulator and proof; tests/harness_mirror.py rebuilds the same accumulation on the
mirrors (same keys, vectors, rng stream, sponge) and tests/ser_mirror.py serialises it.  On the GPU box at the sizes BASELINE.json
names -- ipa_pc_as 2^16, r1cs_nark_as 2^18, hp_as 2^22, n2 and harness-zk shapes, Poseidon -- where the C++ call sequence reaches
the bucket-per-lane / bucket-split / two-valued pipelines and the physical folds; without a GPU at small sizes on the library's
host backend (both sponges, all four schemes)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests import harness_mirror

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "build", "profile_as")


def build():
    libdir = os.path.join(ROOT, "accumulation_amd")
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    src = os.path.join(ROOT, "tools", "profile_as.cpp")
    deps = [src] + [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), src, "-o", EXE + f".{os.getpid()}", "-L", libdir,
                               "-l:libamsm.so", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])
        os.replace(EXE + f".{os.getpid()}", EXE)  # (atomic: pytest -n workers may build the same program at once)


def cpp_dump(tmp_path, scheme, lg, shape, sponge, device, seed=0, extra=()):
    build()
    f = str(tmp_path / f"{scheme}_{lg}_{shape}_{sponge}.bin")
    cmd = [EXE, scheme, str(lg), str(lg), "--shape", shape, "--sponge", sponge, "--device", str(device), "--seed", str(seed),
           "--dump", f, *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert '"verified": true' in out.stdout and '"decided": true' in out.stdout
    raw = open(f, "rb").read()
    recs, at = [], 0
    while at < len(raw):
        (n,) = struct.unpack_from("<Q", raw, at)
        recs.append(raw[at + 8:at + 8 + n])
        at += 8 + n
    assert len(recs) == 2
    return recs, out.stdout


def first_difference(a, b):
    if len(a) != len(b):
        return f"lengths {len(a)} != {len(b)}"
    x, y = np.frombuffer(a, dtype=np.uint8), np.frombuffer(b, dtype=np.uint8)
    d = np.nonzero(x != y)[0]
    return None if d.size == 0 else f"{d.size} bytes differ, the first at offset {int(d[0])} of {len(a)}"


def compare(tmp_path, scheme, lg, shape, sponge, device, seed=0, curve=0):
    from accumulation_amd import Context
    (acc_cpp, proof_cpp), _ = cpp_dump(tmp_path, scheme, lg, shape, sponge, device, seed, extra=("--curve", str(curve)) if curve else ())
    ctx = Context(curve, device=device)
    try:
        acc_py, proof_py = harness_mirror.SCHEMES[scheme](ctx, lg, shape == "harness", sponge, seed)
    finally:
        ctx.close()
    assert first_difference(proof_cpp, proof_py) is None, ("proof", first_difference(proof_cpp, proof_py))
    assert first_difference(acc_cpp, acc_py) is None, ("accumulator", first_difference(acc_cpp, acc_py))
    return len(acc_cpp), len(proof_cpp)


# ---- no GPU: small sizes on the host backend ------------------------------------------------------------------------------------
@pytest.mark.parametrize("sponge", ["sha256", "poseidon"])
@pytest.mark.parametrize("shape", ["n2", "harness"])
@pytest.mark.parametrize("scheme,lg", [("hp_as", 7), ("r1cs_nark_as", 6), ("ipa_pc_as", 5), ("trivial_pc_as", 6)])
def test_cpp_driver_bytes_equal_the_mirror_on_the_host_backend(built_lib, tmp_path, scheme, lg, shape, sponge):
    compare(tmp_path, scheme, lg, shape, sponge, -1, seed=3)


@pytest.mark.parametrize("shape", ["n2", "harness"])
@pytest.mark.parametrize("scheme,lg", [("hp_as", 6), ("r1cs_nark_as", 5), ("ipa_pc_as", 4), ("trivial_pc_as", 5)])
def test_cpp_driver_bytes_equal_the_mirror_over_bls12_381(built_lib, tmp_path, scheme, lg, shape):
    """the drivers are not tied to Pallas (BASELINE config 3 runs ipa_pc_as on BLS12-381): 48-byte points, the 381-bit sponge field"""
    from accumulation_amd import ffi
    compare(tmp_path, scheme, lg, shape, "poseidon", -1, seed=4, curve=ffi.AMSM_BLS12_381_G1)


@pytest.mark.parametrize("curve", ["pallas", "bls12_381_g1"])
def test_ser_mirror_against_the_big_int_wire_format_oracle(built_lib, curve):
    """tests/ser_mirror.py itself (it leans on the library's primitives for points and device vectors): one hp_as accumulator
    and proof re-encoded with oracle/pyref_ser.py's big-int restatement of ark-serialize"""
    from accumulation_amd import Context, PedersenCommitment, ffi
    from accumulation_amd.hp_as import ASForHadamardProducts as AS
    from oracle import pyref as o
    from oracle import pyref_ser as ser
    from tests import helpers as h
    from tests.ser_mirror import Ser
    from tests.test_hp_as_scheme_gpu import SchemeRng, generate_inputs
    c = o.CURVES[curve]
    ctx = Context(c.curve_id, device=ffi.AMSM_DEVICE_HOST)
    ck = PedersenCommitment.setup(ctx, 11, seed=4242)
    pk, _, _ = AS.index(ck)
    inputs = generate_inputs(ctx, ck, 2, True)
    acc, proof = AS.prove(pk, inputs, [], SchemeRng(5), None)

    def pt(p):
        return ser.point_serialize(c, h.np_to_point(c, p[0], bool(p[1])))

    def vec(v):
        return ser.vec([ser.fr_serialize(c, x) for x in h.fr_from_mont_np(c, v.download())])

    i, w = acc.instance, acc.witness
    want_acc = pt(i.comm_1) + pt(i.comm_2) + pt(i.comm_3) + vec(w.a_vec) + vec(w.b_vec) + ser.option(
        b"".join(ser.fr_serialize(c, x) for x in (w.randomness.rand_1, w.randomness.rand_2, w.randomness.rand_3)))
    hc = proof.hiding_comms
    want_proof = ser.vec([pt(p) for p in proof.product_poly_comm.low]) + ser.vec([pt(p) for p in proof.product_poly_comm.high]) + \
        ser.option(pt(hc.comm_1) + pt(hc.comm_2) + pt(hc.comm_3))
    s = Ser(ctx)
    assert s.hp_accumulator(acc) == want_acc and s.hp_proof(proof) == want_proof
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scheme,lg", [("ipa_pc_as", 16), ("r1cs_nark_as", 12), ("hp_as", 14), ("trivial_pc_as", 10)])
def test_cpp_driver_bytes_equal_the_mirror_over_bls12_381_on_the_gpu(built_lib, tmp_path, scheme, lg):
    """over BLS12-381 with the reference's sponge over ITS base field (BASELINE config 3's curve; ipa_pc_as at d + 1 = 2^16 -- the
    2^20 opening itself is oracle-checked through the mirror in tests/test_ipa_open_vs_oracle_gpu.py -- and the other three drivers
    at sizes that reach the device kernels of the 384-bit field)"""
    from accumulation_amd import ffi
    compare(tmp_path, scheme, lg, "harness", "poseidon", 0, curve=ffi.AMSM_BLS12_381_G1)


# ---- GPU: the sizes of BASELINE.json's configs, the reference's sponge -------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["n2", "harness"])
@pytest.mark.parametrize("scheme,lg", [("ipa_pc_as", 16), ("r1cs_nark_as", 18), ("hp_as", 22), ("trivial_pc_as", 10)])
def test_cpp_driver_bytes_equal_the_mirror_at_config_size(built_lib, tmp_path, scheme, lg, shape):
    acc_bytes, _ = compare(tmp_path, scheme, lg, shape, "poseidon", 0)
    if scheme == "hp_as":
        assert acc_bytes > 2 * 32 * (1 << lg)  # the 2^22-element witness vectors are in the comparison


@pytest.mark.gpu
def test_cpp_driver_on_both_backends_same_bytes(built_lib, tmp_path):
    """the C++ harness at a mid size on device 0 and on the host backend: identical dumps"""
    for scheme, lg in (("hp_as", 12), ("r1cs_nark_as", 10), ("ipa_pc_as", 8)):
        g, _ = cpp_dump(tmp_path, scheme, lg, "harness", "poseidon", 0)
        hh, _ = cpp_dump(tmp_path, scheme, lg, "harness", "poseidon", -1)
        assert g == hh, scheme


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0,0,0", "0,0,0,0,0,0,0,0"], ids=["2_shards", "3_shards", "8_shards"])
def test_cpp_driver_over_sharded_keys_same_bytes(built_lib, tmp_path, devices):
    """`profile_as --devices a,b,..` (one multi-device context: sharded keys, one exchange per commit round / grouped MSM / IPA
    round) against the single-device run: identical accumulators and proofs for all four schemes -- ipa_pc_as at BASELINE config
    2's size, so that the 8-GPU form of every config has run through the C++ call sequence before the first real node does."""
    for scheme, lg in (("hp_as", 18), ("r1cs_nark_as", 14), ("ipa_pc_as", 16 if devices.count(",") == 7 else 11), ("trivial_pc_as", 10)):
        one, _ = cpp_dump(tmp_path, scheme, lg, "harness", "poseidon", 0)
        many, _ = cpp_dump(tmp_path, scheme, lg, "harness", "poseidon", 0, extra=("--devices", devices))
        assert one == many, (scheme, devices)


@pytest.mark.gpu
def test_cpp_driver_over_sharded_keys_same_bytes_over_bls12_381(built_lib, tmp_path):
    """the same over BLS12-381 (192-byte partial records between the shards), three shards, all four schemes at small sizes"""
    bls = ("--curve", "1")
    for scheme, lg in (("hp_as", 14), ("r1cs_nark_as", 12), ("ipa_pc_as", 11), ("trivial_pc_as", 10)):
        one, _ = cpp_dump(tmp_path, scheme, lg, "harness", "poseidon", 0, extra=bls)
        many, _ = cpp_dump(tmp_path, scheme, lg, "harness", "poseidon", 0, extra=bls + ("--devices", "0,0,0"))
        assert one == many, scheme


# ---- the uniform-witness circuit of `profile_as --uniform` (rows w_i * w_i = v_i: what bench.py reports beside the harness's two-valued lines)
@pytest.mark.parametrize("shape", ["n2", "harness"])
def test_uniform_witness_circuit_on_the_host_backend(built_lib, tmp_path, shape):
    (acc, proof), out = cpp_dump(tmp_path, "r1cs_nark_as", 7, shape, "poseidon", -1, extra=("--uniform",))
    assert "uniform witness" in out and '"serialize_roundtrip_decides": true' in out
    assert len(acc) > 2 * 32 * (1 << 7)  # the blinded witness (2 (n - 1) variables) and the hp vectors are in the accumulator


@pytest.mark.gpu
def test_uniform_witness_circuit_same_bytes_on_both_backends_and_over_shards(built_lib, tmp_path):
    g, out = cpp_dump(tmp_path, "r1cs_nark_as", 14, "harness", "poseidon", 0, extra=("--uniform",))
    assert '"serialize_roundtrip_decides": true' in out
    hh, _ = cpp_dump(tmp_path, "r1cs_nark_as", 14, "harness", "poseidon", -1, extra=("--uniform",))
    many, _ = cpp_dump(tmp_path, "r1cs_nark_as", 14, "harness", "poseidon", 0, extra=("--uniform", "--devices", "0,0,0"))
    assert g == hh and g == many


@pytest.mark.gpu
def test_cpp_driver_over_replicated_keys_same_bytes(built_lib, tmp_path):
    """`profile_as --devices 0 x 8 --replicate-below 18` (amsm.h AMSM_BASES_REPLICATE: every device holds the whole key, the
    independent MSMs of a commit round are dealt to the devices, no exchange) against the single-device run at BASELINE config 4's
    size: r1cs_nark_as at 2^18 constraints over the circuit with a uniform witness (windowed MSMs, not the harness's two-valued
    ones) and over the harness's own circuit; hp_as at 2^16; identical accumulators and proofs."""
    dev8 = ("--devices", "0,0,0,0,0,0,0,0", "--replicate-below", "18")
    for scheme, lg, extra in (("r1cs_nark_as", 18, ("--uniform",)), ("r1cs_nark_as", 18, ()), ("hp_as", 16, ())):
        for shape in ("harness", "n2"):
            one, _ = cpp_dump(tmp_path, scheme, lg, shape, "poseidon", 0, extra=extra)
            many, _ = cpp_dump(tmp_path, scheme, lg, shape, "poseidon", 0, extra=extra + dev8)
            assert one == many, (scheme, shape, extra)
    # ipa_pc_as at 2^14 / 2^16: the opening's rounds (grouped MSMs, the jump fold where the key qualifies) run over the PRIMARY's copy of a
    # replicated key
    for lg in (14, 16):
        one, _ = cpp_dump(tmp_path, "ipa_pc_as", lg, "harness", "poseidon", 0)
        many, _ = cpp_dump(tmp_path, "ipa_pc_as", lg, "harness", "poseidon", 0, extra=("--devices", "0,0,0", "--replicate-below", "18"))
        assert one == many, ("ipa_pc_as", lg)


# ---- the jump fold of the IPA opening (round 6): the last rounds on the host over amsm_ipa_jump_fold's generators -----------------------
def _cpp_dump_env(tmp_path, env, *args, **kw):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return cpp_dump(tmp_path, *args, **kw)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("shape", ["n2", "harness"])
def test_ipa_jump_fold_does_not_change_the_proof_on_the_host_backend(built_lib, tmp_path, shape):
    """AMSM_IPA_JUMP=0 (every round through amsm_ipa_round_fused, the final key from the check polynomial's MSM) against jumps at 4 and
    at 16 entries (the C++ driver's host rounds): the same accumulator and proof bytes -- and equal to the Python mirror's"""
    ref, _ = _cpp_dump_env(tmp_path, {"AMSM_IPA_JUMP": "0"}, "ipa_pc_as", 7, shape, "poseidon", -1)
    for m in ("4", "16"):
        got, _ = _cpp_dump_env(tmp_path, {"AMSM_IPA_JUMP": m}, "ipa_pc_as", 7, shape, "poseidon", -1)
        assert got == ref, m
    os.environ["AMSM_IPA_JUMP"] = "8"
    try:
        compare(tmp_path, "ipa_pc_as", 7, shape, "poseidon", -1)  # mirror (jumping at 8) == C++ (jumping at 8) == the above
    finally:
        del os.environ["AMSM_IPA_JUMP"]


@pytest.mark.gpu
def test_ipa_jump_fold_on_the_device_same_bytes(built_lib, tmp_path):
    """BASELINE config 2's size on the GPU: the jump at 64 entries after ten device rounds (the default) against no jump at all"""
    for shape in ("n2", "harness"):
        ref, _ = _cpp_dump_env(tmp_path, {"AMSM_IPA_JUMP": "0"}, "ipa_pc_as", 16, shape, "poseidon", 0)
        got, _ = _cpp_dump_env(tmp_path, {"AMSM_IPA_JUMP": "64"}, "ipa_pc_as", 16, shape, "poseidon", 0)
        big, _ = _cpp_dump_env(tmp_path, {"AMSM_IPA_JUMP": "256"}, "ipa_pc_as", 16, shape, "poseidon", 0)
        assert got == ref and big == ref, shape
