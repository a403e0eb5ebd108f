"""The C++ host-side mirror (include/amsm.hpp): compiles as plain C++17 against the C ABI (CPU check) and, on a
GPU, reproduces the oracle's results through the same reference-shaped interface
(VariableBaseMSM / PedersenCommitment / hp_as::{compute_hp, combine_vectors, compute_t_vecs,
compute_product_poly_comm})."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "mirror_check.cpp")
EXE = os.path.join(ROOT, "build", "mirror_check")


def build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    libdir = os.path.join(ROOT, "accumulation_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", EXE,
                           "-L", libdir, "-l:libamsm.so", f"-Wl,-rpath,{libdir}", "-Wl,--allow-shlib-undefined"])


def test_cpp_mirror_compiles(built_lib):
    build()
    assert os.path.exists(EXE)


def _mirror_matches_oracle(cref, device):
    build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300, env=dict(os.environ, AMSM_CHECK_DEVICE=str(device)))
    assert out.returncode == 0, out.stdout + out.stderr
    vals = {}
    for line in out.stdout.splitlines():
        parts = line.split()
        vals[parts[0]] = parts[1:]
    assert "done" in vals and vals["supported_num_elems"] == ["1000"]
    c = o.PALLAS
    n = 1000

    def pt(name):
        v = vals[name]
        return h.np_to_point(c, np.array([int(x, 16) for x in v[1:]], dtype=np.uint64), int(v[0]))

    xy = cref.rng_points(c.curve_id, 0x5EED1001, n + 1)
    H = h.np_to_point(c, xy[n], 0)
    a, b = cref.rng_frs(c.curve_id, 11, n), cref.rng_frs(c.curve_id, 12, n)  # (FrVector::random: the scalar stream)
    a2, b2 = cref.rng_frs(c.curve_id, 13, n), cref.rng_frs(c.curve_id, 14, n)

    def msm(sc):
        out_, inf = cref.msm(c.curve_id, xy[:n], h.scalars_to_np(sc), threads=4)
        return h.np_to_point(c, out_, inf)

    ai, bi, a2i, b2i = (h.np_to_ints(v) for v in (a, b, a2, b2))
    Pa, Pb = msm(ai), msm(bi)
    assert pt("commit_a") == Pa and pt("msm_a_bigint") == Pa and pt("commit_b") == Pb
    assert pt("oneshot_a") == Pa and pt("oneshot_a_without_7") == msm([0 if i == 7 else v for i, v in enumerate(ai)])
    assert pt("oneshot_a_first_300") == msm(ai[:300] + [0] * (n - 300))  # min(bases.len(), scalars.len()) pairs
    assert pt("commit_a_plus_3b") == o.add(c, Pa, o.mul(c, 3, Pb))
    assert pt("commit_a_had_b") == msm(o.compute_hp(c, ai, bi))
    assert pt("commit_a_hiding_3") == o.add(c, Pa, o.mul(c, 3, H))
    t = o.compute_t_vecs(c, [ai, a2i], [bi, b2i], [1, 3], n)
    assert vals["t_vecs"] == ["3", "middle_skipped", "1"]
    assert pt("ppc_low0") == msm(t[0]) and pt("ppc_high0") == msm(t[2])
    assert pt("batch_a") == Pa and pt("batch_b") == Pb
    g0 = [v if ((i >> 3) & 1) == 0 else 0 for i, v in enumerate(ai)]
    g1 = [v if ((i >> 3) & 1) == 1 else 0 for i, v in enumerate(ai)]
    assert pt("grouped_0") == msm(g0) and pt("grouped_1") == msm(g1)
    lo = msm(ai[:500] + [0] * 500)
    hi = msm([0] * 500 + ai[:500])
    assert pt("win_lo") == lo and pt("win_hi") == hi
    assert pt("fold_commit") == o.add(c, lo, o.mul(c, 3, hi))  # <a, key_l + 3 key_r> = <a, key_l> + 3 <a, key_r>
    # host slices, several per call (amsm_msm_batch / amsm_pedersen_commit_batch)
    assert pt("hostbatch_a") == Pa and pt("hostbatch_b") == Pb and pt("hostbatch_a2") == Pa
    Pb777 = msm(bi[:777] + [0] * (n - 777))
    assert pt("hostcommit_a") == Pa and pt("hostcommit_b777") == Pb777 and pt("commit_b777") == Pb777
    assert pt("hostcommit_a_hiding_3") == o.add(c, Pa, o.mul(c, 3, H))
    assert vals["error_check"] == ["-1"]


@pytest.mark.gpu
def test_cpp_mirror_matches_oracle(built_lib, cref):
    _mirror_matches_oracle(cref, 0)


def test_cpp_mirror_matches_oracle_on_the_host_backend(built_lib, cref):
    """the same program with its Context on AMSM_DEVICE_HOST: no GPU needed"""
    _mirror_matches_oracle(cref, -1)
