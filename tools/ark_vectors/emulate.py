#!/usr/bin/env python3
"""NOT A PIN.  Writes files in the layout of tools/ark_vectors (the Rust generator) from THIS REPO'S OWN oracle, into a scratch
directory, so that the consuming tests (tests/test_ark_vectors_cpu.py / _gpu.py) can be exercised where no Rust toolchain
exists:

    python tools/ark_vectors/emulate.py /tmp/ark_emulated && ARK_VECTORS_DIR=/tmp/ark_emulated python -m pytest tests/test_ark_vectors_cpu.py

It refuses to write into tests/golden/ (files there must come from the real arkworks stack) and marks its output in the
"generator" field.  ark_hp_as.json is not emulated (it needs the reference crate's prove)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref as o  # noqa: E402
from oracle import pyref_poseidon as pp  # noqa: E402
from oracle import pyref_ser as ser  # noqa: E402

GEN = "tools/ark_vectors/emulate.py -- THIS REPO'S OWN ORACLE, NOT arkworks: exercises the consumers, pins nothing"


def hp(P):
    return None if P is None else [hex(P[0]), hex(P[1])]


def msm_cases(c):
    cases = []
    for i, n in enumerate([1, 2, 31, 32, 33, 255]):
        sp, ss = 0x5EEDA000 + i, 0x5EEDB000 + i
        cases.append({"kind": "seeded", "n": n, "seed_points": sp, "seed_scalars": ss,
                      "expected": hp(o.msm_pippenger(c, o.rng_points(c, sp, n), o.rng_frs(c, ss, n)))})
    pts = o.rng_points(c, 0x5EEDC000, 8)
    edge = [("all_zero_scalars", pts, [0] * 8), ("all_one_scalars", pts, [1] * 8), ("all_r_minus_1", pts, [c.r - 1] * 8),
            ("duplicate_bases", [pts[0]] * 8, o.rng_scalars(0x5EEDC001, 8)),
            ("opposite_bases_cancel", [pts[1], o.neg(c, pts[1])], [7, 7]),
            ("identity_among_bases", [pts[2], None, pts[3]], o.rng_scalars(0x5EEDC002, 3)),
            ("more_bases_than_scalars", pts, o.rng_scalars(0x5EEDC003, 5)),
            ("powers_of_two", pts, [pow(2, 31 * k + 1, c.r) for k in range(8)])]
    for name, b, s in edge:
        k = min(len(b), len(s))
        cases.append({"kind": "explicit", "name": name, "points": [hp(P) for P in b], "scalars": [hex(x % c.r) for x in s],
                      "expected": hp(o.msm_naive(c, b[:k], [x % c.r for x in s[:k]]))})
    return cases


def serialize_cases(c):
    rows = []
    for v in (0, 1, c.r - 1, o.rng_scalar(0x5EEDD000, 0) % c.r):
        rows.append({"type": "fr", "value": hex(v), "bytes": ser.fr_serialize(c, v).hex()})
    g = o.generator(c)
    for P in [g, o.neg(c, g), None] + o.rng_points(c, 0x5EEDD001, 4):
        rows.append({"type": "point", "value": hp(P), "compressed": ser.point_serialize(c, P, True).hex(),
                     "uncompressed": ser.point_serialize(c, P, False).hex()})
    v = [x % c.r for x in o.rng_scalars(0x5EEDD002, 3)]
    rows.append({"type": "vec_fr", "values": [hex(x) for x in v], "bytes": ser.vec([ser.fr_serialize(c, x) for x in v]).hex()})
    rows.append({"type": "option_fr", "value": None, "bytes": ser.option(None).hex()})
    rows.append({"type": "option_fr", "value": hex(v[0]), "bytes": ser.option(ser.fr_serialize(c, v[0])).hex()})
    return rows


def poseidon_cases():
    c = o.PALLAS
    s = pp.PoseidonSponge(c.p)
    steps = [{"squeeze_fq": [hex(x) for x in s.squeeze(3)]}]
    s.absorb([1, 2, 3, 4, 5])
    steps.append({"absorb_fq": [hex(x) for x in (1, 2, 3, 4, 5)]})
    steps.append({"squeeze_fq": [hex(x) for x in s.squeeze(4)]})
    v = s.squeeze_bits_int(300)
    steps.append({"squeeze_bits": "".join("1" if (v >> i) & 1 else "0" for i in range(300))})
    out = [{"name": "native", "steps": steps}]
    s = pp.PoseidonSponge(c.p)
    b = bytes(range(77))
    s.absorb_bytes(b)
    s.absorb([11])
    P = o.rng_points(c, 0x5EEDE000, 1)[0]
    s.absorb_point(P)
    steps = [{"absorb_bytes": b.hex()}, {"absorb_usize": 11}, {"absorb_point": hp(P)},
             {"squeeze_nonnative_truncated_128": [hex(x) for x in s.squeeze_nonnative(128, 3)]},
             {"squeeze_nonnative_truncated_128": [hex(x) for x in s.squeeze_nonnative(128, 1)]}]
    out.append({"name": "encodings", "steps": steps})
    # one encoding per case (main.rs block 3)
    sq = lambda s: {"squeeze_fq": [hex(x) for x in s.squeeze(2)]}  # noqa: E731
    b = bytes(range(33))
    s = pp.PoseidonSponge(c.p)
    s.absorb_bytes(b)
    out.append({"name": "bytes_only", "steps": [{"absorb_bytes": b.hex()}, sq(s)]})
    s = pp.PoseidonSponge(c.p)
    s.absorb_point(None)
    out.append({"name": "identity_point", "steps": [{"absorb_point": hp(None)}, sq(s)]})
    s = pp.PoseidonSponge(c.p)
    s.absorb([1])
    s.absorb_bytes(b)
    out.append({"name": "option_some_bytes", "steps": [{"absorb_option_bytes": b.hex()}, sq(s)]})
    s = pp.PoseidonSponge(c.p)
    s.absorb([0])
    out.append({"name": "option_none", "steps": [{"absorb_option_bytes": None}, sq(s)]})
    s = pp.PoseidonSponge(c.p).fork(b"AS-FOR-HP-2020")
    out.append({"name": "fork", "steps": [{"fork": b"AS-FOR-HP-2020".hex()}, sq(s)]})
    return out


def pedersen_cases():
    c = o.PALLAS
    rows = []
    for n in (1, 8, 33):
        pts = o.rng_points(c, 0x5EEDF200 + n, n + 1)
        gens, H = pts[:n], pts[n]
        v = [x % c.r for x in o.rng_scalars(0x5EEDF000 + n, n)]
        r = o.rng_scalar(0x5EEDF100, 0) % c.r
        rows.append({"n": n, "generators": [hp(P) for P in gens], "hiding_generator": hp(H), "elems": [hex(x) for x in v],
                     "rand": hex(r), "commit": hp(o.pedersen_commit(c, gens, H, v, None)),
                     "commit_hiding": hp(o.pedersen_commit(c, gens, H, v, r))})
    return rows


def main():
    out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else "/tmp/ark_emulated")
    if out.startswith(os.path.join(ROOT, "tests", "golden")):
        sys.exit("refusing to write emulated vectors into tests/golden/: files there must come from the Rust generator")
    os.makedirs(out, exist_ok=True)
    files = {
        "ark_msm.json": {"generator": GEN, **{c.name: msm_cases(c) for c in (o.PALLAS, o.BLS12_381_G1)}},
        "ark_serialize.json": {"generator": GEN, **{c.name: serialize_cases(c) for c in (o.PALLAS, o.BLS12_381_G1)}},
        "ark_poseidon.json": {"generator": GEN, "cases": poseidon_cases()},
        "ark_pedersen.json": {"generator": GEN, "cases": pedersen_cases()},
    }
    for name, body in files.items():
        with open(os.path.join(out, name), "w") as f:
            json.dump(body, f, indent=1)
        print("wrote", os.path.join(out, name))


if __name__ == "__main__":
    main()
