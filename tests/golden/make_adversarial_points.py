"""Generate tests/golden/adversarial_points.json: curve points whose coordinates, IN THE DEVICE'S INTERNAL MONTGOMERY RADIX
(2^261 for Pallas, 2^392 for BLS12-381: csrc/fpu.h), sit at the edges the lazy arithmetic of csrc/ec.h cares about -- tiny
values and values just below p.  The bug fixed in round 2 (doubling a negated point whose internal y was below 2^(B (L - 1)))
needed exactly such a point and random tests meet one in 2^17 (BLS12-381) / 2^22 (Pallas) points.

Pallas has cofactor 1, so points are CONSTRUCTED: pick the internal coordinate, solve the curve equation (cube root /
square root).  BLS12-381 G1 needs points of the prime-order subgroup, so a pool of multiples of the generator (the C oracle's
rng_points) is SEARCHED for the most extreme coordinates.  Data only; run from the repo root:  python tests/golden/make_adversarial_points.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cref, pyref as o  # noqa: E402
from oracle.pyref_ser import _sqrt  # noqa: E402

INTERNAL_BITS = {"pallas": 261, "bls12_381_g1": 392}


def cube_root(a, p):
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 3, p) != 1:
        return None
    s, t = 0, p - 1
    while t % 3 == 0:
        s, t = s + 1, t // 3
    e = pow(3, -1, t)
    x0 = pow(a, e, p)
    g = 2
    while pow(g, (p - 1) // 3, p) == 1:
        g += 1
    cgen = pow(g, t, p)  # generates the 3-Sylow subgroup (order 3^s, s <= 2 for both fields)
    w = 1
    for _ in range(3 ** s):
        if pow(x0 * w % p, 3, p) == a:
            return x0 * w % p
        w = w * cgen % p
    raise AssertionError("cube root not found")


def constructed(c, per_kind):
    p, rinv = c.p, pow(1 << INTERNAL_BITS[c.name], -1, c.p)
    out = {}
    for kind, coord, values in (("tiny_y", "y", range(1, 4000)), ("near_p_y", "y", range(p - 1, p - 4000, -1)),
                                ("tiny_x", "x", range(1, 4000)), ("near_p_x", "x", range(p - 1, p - 4000, -1))):
        pts = []
        for v_int in values:
            v = v_int * rinv % p
            if coord == "y":
                x = cube_root((v * v - c.b) % p, p)
                P = None if x is None else (x, v)
            else:
                y = _sqrt((v * v * v + c.b) % p, p)
                P = None if y is None else (v, y)
            if P is not None and o.is_on_curve(c, P):
                pts.append(P)
                if len(pts) == per_kind:
                    break
        assert len(pts) == per_kind, (c.name, kind)
        out[kind] = pts
    return out


def searched(c, per_kind, n_pool, seeds):
    p, shift = c.p, INTERNAL_BITS[c.name] - 64 * c.limbs
    best = {"tiny_y": [], "near_p_y": [], "tiny_x": [], "near_p_x": []}
    for seed in seeds:
        xy = cref.rng_points(c.curve_id, seed, n_pool)  # C-ABI Montgomery limbs
        L = c.limbs
        # top limb as a cheap filter: the internal value is abi_mont << shift mod p; compute exactly for all (Python ints)
        for i in range(n_pool):
            xm = sum(int(v) << (64 * k) for k, v in enumerate(xy[i][:L]))
            ym = sum(int(v) << (64 * k) for k, v in enumerate(xy[i][L:]))
            xi, yi = (xm << shift) % p, (ym << shift) % p
            for kind, key in (("tiny_y", yi), ("near_p_y", p - yi), ("tiny_x", xi), ("near_p_x", p - xi)):
                lst = best[kind]
                if len(lst) < per_kind or key < lst[-1][0]:
                    lst.append((key, seed, i))
                    lst.sort()
                    del lst[per_kind:]
    out = {}
    rabi = pow(1 << (64 * c.limbs), -1, p)
    for kind, lst in best.items():
        pts = []
        for key, seed, i in lst:
            xy = cref.rng_points(c.curve_id, seed, i + 1)[i]
            xm = sum(int(v) << (64 * k) for k, v in enumerate(xy[:c.limbs]))
            ym = sum(int(v) << (64 * k) for k, v in enumerate(xy[c.limbs:]))
            P = (xm * rabi % p, ym * rabi % p)
            assert o.is_on_curve(c, P)
            pts.append(P)
        out[kind] = pts
        print(c.name, kind, "distance from the edge: 2^%.1f .. 2^%.1f" % (np.log2(float(lst[0][0]) + 1), np.log2(float(lst[-1][0]) + 1)))
    return out


if __name__ == "__main__":
    doc = {"comment": "affine points (canonical integers, hex) with extreme coordinates in the device's internal Montgomery radix; "
                      "made by tests/golden/make_adversarial_points.py", "internal_radix_bits": INTERNAL_BITS, "curves": {}}
    doc["curves"]["pallas"] = {k: [[hex(P[0]), hex(P[1])] for P in v] for k, v in constructed(o.PALLAS, 6).items()}
    doc["curves"]["bls12_381_g1"] = {k: [[hex(P[0]), hex(P[1])] for P in v]
                                     for k, v in searched(o.BLS12_381_G1, 4, 1 << 18, [4101, 4102, 4103, 4104]).items()}
    json.dump(doc, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "adversarial_points.json"), "w"), indent=1)
    print("written")
