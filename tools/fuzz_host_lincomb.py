#!/usr/bin/env python3
"""Randomised check of the host-side group algebra (amsm_host_lincomb_batch: GLV halves, signed digits, terms shared over the host
pool, fixed-base tables) against the big-int oracle -- no GPU.  Every case is a batch of 1..6 jobs of 0..12 terms over a pool of
points that recur (so that fixed-base tables get built, hit and go stale), with scalars drawn from the shapes the schemes produce:
0, 1, small, 128-bit challenges, their products and inverses (full size), values next to the cube roots of unity and to r,
digit strings of 8s / 9s / Fs; points at infinity, repeated points, a point and its negative.

    python tools/fuzz_host_lincomb.py [--seconds 60] [--seed 1]
Test infrastructure (uses oracle/)."""
import argparse
import ctypes as C
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from accumulation_amd import ffi  # noqa: E402
from accumulation_amd.engine import _ptr  # noqa: E402
from oracle import pyref as o  # noqa: E402
from tests import helpers as h  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rnd = random.Random(a.seed)
    lib = ffi.load()
    t_end = time.time() + a.seconds
    cases = terms = 0
    while time.time() < t_end:
        c = rnd.choice([o.PALLAS, o.BLS12_381_G1])
        g = o.generator(c)
        lam = next(pow(x, (c.r - 1) // 3, c.r) for x in range(2, 50) if pow(x, (c.r - 1) // 3, c.r) != 1)
        pool = [o.mul(c, rnd.randrange(1, c.r), g) for _ in range(6)]
        pool += [o.neg(c, pool[0]), None]
        w = 2 * c.limbs
        for _ in range(12):  # several calls over the same pool: the fixed-base cache sees the points again
            x = rnd.getrandbits(128) | 1

            def scalar():
                k = rnd.randrange(12)
                if k == 0:
                    return rnd.choice([0, 1, 2, 15, 16, 17, c.r - 1, c.r - 2])
                if k <= 3:
                    return rnd.getrandbits(128)
                if k == 4:
                    return x * x % c.r
                if k == 5:
                    return pow(x, -1, c.r)
                if k == 6:
                    return (lam * rnd.getrandbits(rnd.choice([1, 8, 127, 128])) + rnd.getrandbits(rnd.choice([1, 64, 128]))) % c.r
                if k == 7:
                    return int(rnd.choice("89F7") * rnd.randrange(30, 64), 16) % c.r
                if k == 8:
                    return (c.r - rnd.getrandbits(rnd.choice([3, 64, 130]))) % c.r
                if k == 9:
                    return rnd.getrandbits(rnd.choice([130, 136, 137, 140, 200])) % c.r
                return rnd.randrange(c.r)

            jobs = []
            for _j in range(rnd.randrange(1, 7)):
                n = rnd.randrange(0, 13)
                jobs.append(([rnd.choice(pool) for _ in range(n)], [scalar() for _ in range(n)]))
            nj = len(jobs)
            n_terms = (C.c_size_t * nj)()
            xy_p, inf_p, sc_p = (C.c_void_p * nj)(), (C.c_void_p * nj)(), (C.c_void_p * nj)()
            keep = []
            for j, (P, S) in enumerate(jobs):
                n_terms[j] = len(P)
                xy, inf = h.points_to_np(c, P) if P else (np.zeros((1, w), dtype=np.uint64), np.zeros((1,), dtype=np.uint8))
                sc = h.fr_mont_np(c, S) if S else np.zeros((1, 4), dtype=np.uint64)
                keep.append((xy, inf, sc))
                xy_p[j], inf_p[j], sc_p[j] = xy.ctypes.data, inf.ctypes.data, sc.ctypes.data
            out = np.zeros((nj, w), dtype=np.uint64)
            oinf = np.zeros((nj,), dtype=np.uint8)
            ffi.check(lib.amsm_host_lincomb_batch(c.curve_id, nj, n_terms, xy_p, inf_p, sc_p, _ptr(out), _ptr(oinf)), "batch")
            for j, (P, S) in enumerate(jobs):
                want = None
                for p, s in zip(P, S):
                    want = o.add(c, want, o.mul(c, s, p))
                got = h.np_to_point(c, out[j], bool(oinf[j]))
                if got != want:
                    print(f"MISMATCH seed {a.seed} curve {c.name} job {j}: scalars {[hex(s) for s in S]}", flush=True)
                    return 1
                terms += len(P)
            cases += 1
    print(f"fuzz_host_lincomb ok: {cases} batches, {terms} terms against the big-int oracle in {a.seconds:.0f} s (seed {a.seed})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
