"""The oracles (and the host-side parts of libamsm.so) against vectors produced by the REAL arkworks stack -- the pin the
round-3 verdict asked to ship although it cannot be generated in the build image.  Every test skips with "parity unpinned:
<file> absent" until tests/golden/ark_*.json exist (tools/ark_vectors; see tests/ark_vectors.py).
  ark_msm        ark-ec 0.2 VariableBaseMSM::multi_scalar_mul          -> oracle/ark_msm.c, oracle/pyref.py
  ark_serialize  ark-serialize 0.2 CanonicalSerialize                   -> oracle/pyref_ser.py, amsm_fr/points_serialize
  ark_poseidon   ark-sponge PoseidonSponge<Fq> (accumulation-experimental) -> oracle/pyref_poseidon.py, amsm_poseidon_*
  ark_pedersen   ark-poly-commit trivial_pc::PedersenCommitment::commit -> oracle/pyref.py pedersen_commit"""
import ctypes as C

import numpy as np
import pytest

from oracle import pyref as o
from oracle import pyref_poseidon as pp
from oracle import pyref_ser as ser
from tests import ark_vectors as av
from tests import helpers as h


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("curve", ["pallas", "bls12_381_g1"])
def test_msm_vectors_vs_oracles(cref, curve):
    c = o.CURVES[curve]
    for case in av.load("ark_msm.json")[curve]:
        if case["kind"] == "seeded":
            sp, ss, n = av.seeded_msm_inputs(c, case)
            xy = cref.rng_points(c.curve_id, sp, n)
            sc = cref.rng_frs(c.curve_id, ss, n)
            got, inf = cref.msm(c.curve_id, xy, sc, threads=8)
            assert h.np_to_point(c, got, inf) == av.pt(case["expected"]), ("seeded", n)
            if n <= 64:
                assert o.msm_naive(c, o.rng_points(c, sp, n), o.rng_frs(c, ss, n)) == av.pt(case["expected"])
        else:
            pts, sc = [av.pt(p) for p in case["points"]], av.ints(case["scalars"])
            assert o.msm_naive(c, pts, sc) == av.pt(case["expected"]), case["name"]
            k = min(len(pts), len(sc))
            xy, inf = h.points_to_np(c, pts[:k])
            got, ginf = cref.msm(c.curve_id, xy, h.scalars_to_np(sc[:k]), is_inf=inf)
            assert h.np_to_point(c, got, ginf) == av.pt(case["expected"]), case["name"]


@pytest.mark.parametrize("curve", ["pallas", "bls12_381_g1"])
def test_serialize_vectors(built_lib, curve):
    c = o.CURVES[curve]
    for row in av.load("ark_serialize.json")[curve]:
        if row["type"] == "fr":
            v, want = int(row["value"], 16), bytes.fromhex(row["bytes"])
            assert ser.fr_serialize(c, v) == want
            out = np.zeros(32, dtype=np.uint8)
            assert built_lib.amsm_fr_serialize(c.curve_id, _ptr(h.fr_mont_np(c, [v])), 1, _ptr(out)) == 0
            assert bytes(out) == want
        elif row["type"] == "point":
            P = av.pt(row["value"])
            for comp, key in ((True, "compressed"), (False, "uncompressed")):
                want = bytes.fromhex(row[key])
                assert ser.point_serialize(c, P, comp) == want and ser.point_deserialize(c, want, comp) == P
                xy, inf = h.points_to_np(c, [P])
                out = np.zeros(len(want), dtype=np.uint8)
                assert built_lib.amsm_points_serialize(c.curve_id, _ptr(xy), _ptr(inf), 1, int(comp), _ptr(out)) == 0
                assert bytes(out) == want
        elif row["type"] == "vec_fr":
            assert ser.vec([ser.fr_serialize(c, v) for v in av.ints(row["values"])]) == bytes.fromhex(row["bytes"])
        elif row["type"] == "option_fr":
            item = None if row["value"] is None else ser.fr_serialize(c, int(row["value"], 16))
            assert ser.option(item) == bytes.fromhex(row["bytes"])


def _replay(sponge_new, steps, c):
    """absorbs are replayed, squeezes compared; sponge_new() -> an object with the oracle sponge's method names"""
    s = sponge_new()
    for step in steps:
        (op, val), = step.items()
        if op == "absorb_fq":
            s.absorb(av.ints(val))
        elif op == "absorb_bytes":
            s.absorb_bytes(bytes.fromhex(val))
        elif op == "absorb_usize":
            s.absorb([int(val)])
        elif op == "absorb_point":
            s.absorb_point(av.pt(val))
        elif op == "absorb_option_bytes":  # Option<Vec<u8>>: the tag as one element, then the byte string by itself
            s.absorb([0 if val is None else 1])
            if val is not None:
                s.absorb_bytes(bytes.fromhex(val))
        elif op == "fork":
            s = s.fork(bytes.fromhex(val))
        elif op == "squeeze_fq":
            assert s.squeeze(len(val)) == av.ints(val), op
        elif op == "squeeze_bits":
            got = s.squeeze_bits_int(len(val))
            assert "".join("1" if (got >> i) & 1 else "0" for i in range(len(val))) == val, op
        elif op == "squeeze_nonnative_truncated_128":
            assert s.squeeze_nonnative(128, len(val)) == av.ints(val), op
        elif op == "squeeze_nonnative_full":
            pytest.skip("FieldElementSize::Full squeezes are not used by the hot path's schemes and not restated")
        else:
            raise AssertionError(f"unknown transcript step {op}")


def test_poseidon_transcripts_vs_oracle_and_library(built_lib):
    from tests.test_poseidon_cpu import LibSponge
    c = o.PALLAS

    class LibAdapter(LibSponge):
        def absorb(self, elems):
            if elems:
                super().absorb(elems)
    for case in av.load("ark_poseidon.json")["cases"]:
        steps = [s for s in case["steps"] if "squeeze_nonnative_full" not in s]
        _replay(lambda: pp.PoseidonSponge(c.p), steps, c)
        _replay(lambda: LibAdapter(built_lib, c), steps, c)


def test_pedersen_vectors_vs_oracle():
    c = o.PALLAS
    for case in av.load("ark_pedersen.json")["cases"]:
        gens, H = [av.pt(p) for p in case["generators"]], av.pt(case["hiding_generator"])
        v, r = av.ints(case["elems"]), int(case["rand"], 16)
        assert o.pedersen_commit(c, gens, H, v, None) == av.pt(case["commit"])
        assert o.pedersen_commit(c, gens, H, v, r) == av.pt(case["commit_hiding"])
