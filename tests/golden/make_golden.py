#!/usr/bin/env python3
"""Regenerate tests/golden/msm_golden.json from the pure-Python big-integer oracle (oracle/pyref.py).

These vectors are NOT outputs of the reference (it cannot be built here and holds no fixtures --
PARITY UNPINNED, see oracle/pyref.py); they pin the build's three implementations (Python oracle, C
restatement, HIP) to one another and to the mathematical definition, across rounds.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import pyref as o  # noqa: E402


def hexpt(P):
    return None if P is None else [hex(P[0]), hex(P[1])]


def case(c, name, pts, scalars, note=""):
    res = o.msm_naive(c, pts, scalars) if len(pts) <= 96 else o.msm_pippenger(c, pts, scalars, window=8)
    assert res == o.msm_pippenger(c, pts, scalars)
    xy, inf = o.point_to_mont_limbs(c, res)
    return {
        "name": name,
        "note": note,
        "n": min(len(pts), len(scalars)),
        "points": [hexpt(P) for P in pts],
        "scalars": [hex(s) for s in scalars],
        "expected_affine": hexpt(res),
        "expected_mont_limbs": [hex(v) for v in xy],
        "expected_is_inf": inf,
    }


def seeded_case(c, name, seed_pts, seed_sc, n):
    pts = o.rng_points(c, seed_pts, n)
    sc = o.rng_frs(c, seed_sc, n)  # the scalar stream of amsm_vec_random: uniform in [0, r) (round 6)
    res = o.msm_pippenger(c, pts, sc)
    xy, inf = o.point_to_mont_limbs(c, res)
    return {"name": name, "n": n, "seed_points": seed_pts, "seed_scalars": seed_sc,
            "expected_affine": hexpt(res), "expected_mont_limbs": [hex(v) for v in xy], "expected_is_inf": inf}


def main():
    out = {"generator": "tests/golden/make_golden.py (oracle/pyref.py)", "curves": {}}
    for c in (o.PALLAS, o.BLS12_381_G1):
        g = o.generator(c)
        kat = {
            "generator": hexpt(g),
            "2G": hexpt(o.mul(c, 2, g)),
            "3G": hexpt(o.mul(c, 3, g)),
            "(r-1)G": hexpt(o.mul(c, c.r - 1, g)),
            "rG_is_inf": o.mul(c, c.r, g) is None,
            "mont": {k: hex(v) for k, v in o.mont_constants(c.p, c.limbs).items()},
            "mont_fr": {k: hex(v) for k, v in o.mont_constants(c.r, 4).items()},
        }
        pts = o.rng_points(c, 0x5EED1001 + c.curve_id, 40)
        sc = o.rng_scalars(0x5EED0001, 40)
        cases = []
        cases.append(case(c, "empty", [], [], "n = 0 -> identity"))
        cases.append(case(c, "single", pts[:1], sc[:1]))
        cases.append(case(c, "two", pts[:2], sc[:2]))
        for n in (31, 32, 33):
            cases.append(case(c, f"uniform_{n}", pts[:n], sc[:n], "window-size switch of ark-ec at n = 32"))
        cases.append(case(c, "all_zero_scalars", pts[:16], [0] * 16, "default instances, SURVEY.md App. E.9"))
        cases.append(case(c, "all_one_scalars", pts[:16], [1] * 16, "scalar == 1 fast path"))
        cases.append(case(c, "r_minus_1", pts[:8], [c.r - 1] * 8, "top-bit-heavy scalars"))
        cases.append(case(c, "all_equal_scalar", pts[:33], [sc[0]] * 33, "vec![rand; len], SURVEY.md F8"))
        cases.append(case(c, "repeated_points", [pts[0]] * 9 + pts[1:4], sc[:12], "P + P needs the doubling branch"))
        cases.append(case(c, "p_and_minus_p", [pts[0], o.neg(c, pts[0]), pts[1]], [sc[0], sc[0], sc[1]],
                          "cancellation to the identity inside a bucket"))
        cases.append(case(c, "cancel_to_identity", [pts[0], o.neg(c, pts[0])], [5, 5], "result is the identity"))
        cases.append(case(c, "infinity_bases", [None, pts[0], None, pts[1]], sc[:4], "identity bases contribute nothing"))
        cases.append(case(c, "small_scalars", pts[:20], list(range(20)), "digits only in window 0"))
        cases.append(case(c, "power_of_two_scalars", pts[:16], [1 << (16 * i) for i in range(16)], "one digit per window"))
        cases.append(case(c, "half_window", pts[:8], [(1 << 15) + i for i in range(8)], "signed-digit boundary 2^(c-1)"))
        cases.append(case(c, "more_scalars_than_bases", pts[:5], sc[:9], "uses min(len)"))
        seeded = [seeded_case(c, "seeded_256", 0x5EED1001, 0x5EED0001, 256),
                  seeded_case(c, "seeded_1024", 0x5EED1001, 0x5EED0002, 1024)]
        out["curves"][c.name] = {"kat": kat, "cases": cases, "seeded": seeded}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "msm_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
