"""Shared helpers for the parity tests (numpy <-> oracle conversions)."""
import json
import os

import numpy as np

from oracle import pyref as o

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msm_golden.json")


def load_golden():
    with open(GOLDEN) as f:
        return json.load(f)


def pt_from_hex(h):
    return None if h is None else (int(h[0], 16), int(h[1], 16))


def points_to_np(c, pts):
    """list of affine points (None = infinity) -> ((n, 2L) uint64 Montgomery, (n,) uint8)."""
    n = len(pts)
    xy = np.zeros((n, 2 * c.limbs), dtype=np.uint64)
    inf = np.zeros((n,), dtype=np.uint8)
    for i, P in enumerate(pts):
        w, f = o.point_to_mont_limbs(c, P)
        xy[i] = np.array(w, dtype=np.uint64)
        inf[i] = f
    return xy, inf


def np_to_point(c, xy, is_inf):
    return o.point_from_mont_limbs(c, [int(v) for v in np.asarray(xy).reshape(-1)], int(bool(is_inf)))


def scalars_to_np(scalars):
    return np.array([o.int_to_limbs(int(s), 4) for s in scalars], dtype=np.uint64).reshape(-1, 4)


def np_to_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [o.limbs_to_int([int(x) for x in row]) for row in a]


def fr_mont_np(c, vals):
    """canonical ints -> (n,4) uint64 Montgomery (raw Vec<Fr> memory)."""
    return scalars_to_np([o.fr_to_mont(c, int(v) % c.r) for v in vals])


def fr_from_mont_np(c, a):
    return [o.fr_from_mont(c, v) for v in np_to_ints(a)]
