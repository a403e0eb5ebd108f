"""The C ABI's host scalar-field helpers (amsm_fr_mul / add / to_mont / from_mont) need no GPU: checked against
Python integers for both curves, including the values at the ends of the range."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyref as o
from tests import helpers as h


@pytest.mark.parametrize("c", [o.PALLAS, o.BLS12_381_G1], ids=lambda c: c.name)
def test_host_fr_helpers(built_lib, c):
    from accumulation_amd import ffi
    from accumulation_amd.engine import _ptr
    lib = ffi.load()
    R = 1 << 256
    vals_a = [0, 1, c.r - 1, c.r - 2, 2, (1 << 128) - 1] + [o.rng_scalar(3, i) % c.r for i in range(40)]
    vals_b = [c.r - 1, 0, c.r - 1, 3, c.r - 2, (1 << 128) + 5] + [o.rng_scalar(4, i) % c.r for i in range(40)]
    n = len(vals_a)
    a = h.scalars_to_np(vals_a)
    b = h.scalars_to_np(vals_b)
    am, bm = np.zeros_like(a), np.zeros_like(b)
    ffi.check(lib.amsm_fr_to_mont(c.curve_id, _ptr(a), n, _ptr(am)), "to_mont")
    ffi.check(lib.amsm_fr_to_mont(c.curve_id, _ptr(b), n, _ptr(bm)), "to_mont")
    assert h.np_to_ints(am) == [v * R % c.r for v in vals_a]
    back = np.zeros_like(a)
    ffi.check(lib.amsm_fr_from_mont(c.curve_id, _ptr(am), n, _ptr(back)), "from_mont")
    assert h.np_to_ints(back) == vals_a
    prod, summ = np.zeros_like(a), np.zeros_like(a)
    ffi.check(lib.amsm_fr_mul(c.curve_id, _ptr(am), _ptr(bm), n, _ptr(prod)), "mul")
    ffi.check(lib.amsm_fr_add(c.curve_id, _ptr(am), _ptr(bm), n, _ptr(summ)), "add")
    assert h.np_to_ints(prod) == [x * y * R % c.r for x, y in zip(vals_a, vals_b)]
    assert h.np_to_ints(summ) == [(x + y) * R % c.r for x, y in zip(vals_a, vals_b)]
    diff, inv = np.zeros_like(a), np.zeros_like(a)
    ffi.check(lib.amsm_fr_sub(c.curve_id, _ptr(am), _ptr(bm), n, _ptr(diff)), "sub")
    ffi.check(lib.amsm_fr_inv(c.curve_id, _ptr(am), n, _ptr(inv)), "inv")
    assert h.np_to_ints(diff) == [(x - y) * R % c.r for x, y in zip(vals_a, vals_b)]
    assert h.np_to_ints(inv) == [(pow(x, -1, c.r) if x else 0) * R % c.r for x in vals_a]
    assert lib.amsm_fr_mul(7, _ptr(am), _ptr(bm), n, _ptr(prod)) == ffi.AMSM_E_INVALID_ARG


@pytest.mark.parametrize("c", [o.PALLAS, o.BLS12_381_G1], ids=lambda c: c.name)
def test_host_lincomb_vs_oracle(built_lib, c):
    """amsm_host_lincomb (windowed, shared doublings, fixed-base tables for recurring bases) against the big-int oracle:
    scalar edge values, infinity inputs, P + P / P - P collisions, many terms, and one base reused often enough with
    full-size scalars to be served from a fixed-base table (built on its third use)."""
    import ctypes as C
    from accumulation_amd import ffi
    from accumulation_amd.engine import _ptr
    lib = ffi.load()
    g = o.generator(c)
    pts = [o.mul(c, 3 + 5 * i, g) for i in range(12)]

    def lincomb(points, scalars):
        xy, inf = h.points_to_np(c, points)
        sc = h.fr_mont_np(c, scalars)
        out = np.zeros((2 * c.limbs,), dtype=np.uint64)
        oinf = C.c_uint8(0)
        ffi.check(lib.amsm_host_lincomb(c.curve_id, _ptr(xy), _ptr(inf), _ptr(sc), len(points), _ptr(out), C.byref(oinf)),
                  "amsm_host_lincomb")
        return h.np_to_point(c, out, oinf.value)

    def expect(points, scalars):
        acc = None
        for P, s in zip(points, scalars):
            acc = o.add(c, acc, o.mul(c, s % c.r, P))
        return acc

    edge = [0, 1, 2, 15, 16, 17, 255, 256, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 128, (1 << 129) + 1, c.r - 1, c.r - 2,
            0xF0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0F0 % c.r]
    # (round 5: scalars above 136 bits go in as their two GLV halves with signed digits) the cube roots of unity and their
    # neighbours -- halves of 0, 1 and -1 --, digit strings of all 8s / 9s / Fs (the signed recoding's carry chain), r - lambda
    lam = next(pow(x, (c.r - 1) // 3, c.r) for x in range(2, 50) if pow(x, (c.r - 1) // 3, c.r) != 1)
    edge += [lam, lam - 1, lam + 1, c.r - lam, lam * lam % c.r, (lam * lam + 1) % c.r, (1 << 136) - 1, 1 << 136, (1 << 137) + 5,
             int("8" * 62, 16) % c.r, int("9" * 62, 16) % c.r, int("F" * 34, 16), int("F" * 32, 16), int("7" * 63, 16) % c.r,
             (lam * 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF + 0x88888888888888888888888888888888) % c.r]
    for s in edge:
        assert lincomb([pts[0]], [s]) == expect([pts[0]], [s]), hex(s)
    assert lincomb([], []) is None
    assert lincomb([None, pts[1]], [5, 0]) is None
    assert lincomb([pts[2], pts[2]], [7, c.r - 7]) is None                      # P - P
    assert lincomb([pts[2], pts[2]], [1, 1]) == o.mul(c, 2, pts[2])             # P + P through the "ones" path
    assert lincomb([pts[2], o.neg(c, pts[2])], [9, 9]) is None
    for trial in range(6):
        k = [1, 2, 3, 7, 12, 12][trial]
        sc = [o.rng_scalar(50 + trial, i) % c.r for i in range(k)]
        if trial == 5:
            sc = [s % (1 << 128) for s in sc]  # the schemes' 128-bit challenges
        assert lincomb(pts[:k], sc) == expect(pts[:k], sc), trial
    base = o.mul(c, 0xABCDEF, g)
    for use in range(8):  # third use onwards: fixed-base table
        s, t = o.rng_scalar(70, use) % c.r, o.rng_scalar(71, use) % c.r
        assert lincomb([pts[use], base], [t, s]) == expect([pts[use], base], [t, s]), use
    for s in (c.r - 1, (1 << 200) + 1, 1 << 252):
        assert lincomb([base], [s]) == o.mul(c, s, base)
    # more recurring bases than cache slots, several of them inside one call: eviction must never corrupt a result
    bases = [o.mul(c, 1000 + 17 * i, g) for i in range(7)]
    for rnd in range(12):
        sel = [bases[(rnd + j) % 7] for j in range(1 + rnd % 6)]
        sc = [o.rng_scalar(90 + rnd, j) % c.r for j in range(len(sel))]
        assert lincomb(sel, sc) == expect(sel, sc), rnd


@pytest.mark.parametrize("c", [o.PALLAS, o.BLS12_381_G1], ids=lambda c: c.name)
def test_host_fr_inverse_many(built_lib, c):
    """amsm_fr_inv (binary extended Euclid on the host) against Python's modular inverse: small values, powers of two and
    their neighbours (long runs of halvings), values next to the modulus, and random ones."""
    from accumulation_amd import ffi
    from accumulation_amd.engine import _ptr
    lib = ffi.load()
    R = 1 << 256
    vals = list(range(0, 40)) + [c.r - k for k in range(1, 40)]
    vals += [(1 << k) % c.r for k in range(1, 256, 5)] + [((1 << k) - 1) % c.r for k in range(2, 256, 7)]
    vals += [((1 << k) + 1) % c.r for k in range(2, 256, 9)] + [(c.r - 1) // 2, (c.r + 1) // 2, c.r // 3]
    vals += [o.rng_scalar(0x1A7, i) % c.r for i in range(600)]
    n = len(vals)
    a = h.scalars_to_np(vals)
    am, inv = np.zeros_like(a), np.zeros_like(a)
    ffi.check(lib.amsm_fr_to_mont(c.curve_id, _ptr(a), n, _ptr(am)), "to_mont")
    ffi.check(lib.amsm_fr_inv(c.curve_id, _ptr(am), n, _ptr(inv)), "inv")
    assert h.np_to_ints(inv) == [(pow(x, -1, c.r) if x else 0) * R % c.r for x in vals]


@pytest.mark.parametrize("c", [o.PALLAS, o.BLS12_381_G1], ids=lambda c: c.name)
def test_host_lincomb_batch_equals_single_calls_and_oracle(built_lib, c):
    """amsm_host_lincomb_batch (independent jobs on the host pool, one normalisation) against amsm_host_lincomb job by job
    and against the big-int oracle: empty jobs, a job that sums to infinity, points at infinity, 128-bit and full-size
    scalars, and one 40-point job (a single call of that size is itself split over the pool)."""
    from accumulation_amd import ffi
    from accumulation_amd.engine import _ptr
    lib = ffi.load()
    g = o.generator(c)
    pts = [o.mul(c, 11 + 3 * i, g) for i in range(40)]
    sc128 = [o.rng_scalar(0x51, i) % (1 << 128) for i in range(40)]
    sc255 = [o.rng_scalar(0x52, i) % c.r for i in range(40)]
    jobs = [
        (pts[:2], [1, sc128[0]]),
        (pts[2:5], [1, sc128[1], sc128[1] * sc128[1] % c.r]),
        ([], []),
        ([pts[5], pts[5]], [7, c.r - 7]),                      # cancels
        ([None, pts[6], None], [5, sc255[2], 9]),              # points at infinity
        (pts, sc128),                                          # 40 points, challenge-sized scalars
        (pts[:9], sc255[:9]),
        ([pts[7]], [0]),
    ]
    nj = len(jobs)
    w = 2 * c.limbs
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
        expect = None
        for p, s in zip(P, S):
            expect = o.add(c, expect, o.mul(c, s, p))
        assert h.np_to_point(c, out[j], bool(oinf[j])) == expect, j
        xy, inf, sc = keep[j]
        one = np.zeros((w,), dtype=np.uint64)
        one_inf = C.c_uint8(0)
        ffi.check(lib.amsm_host_lincomb(c.curve_id, _ptr(xy), _ptr(inf), _ptr(sc), len(P), _ptr(one), C.byref(one_inf)), "single")
        assert bool(one_inf.value) == bool(oinf[j]) and (bool(oinf[j]) or np.array_equal(one, out[j])), j
    assert lib.amsm_host_lincomb_batch(c.curve_id, nj, None, xy_p, inf_p, sc_p, _ptr(out), _ptr(oinf)) == ffi.AMSM_E_INVALID_ARG
    assert lib.amsm_host_lincomb_batch(c.curve_id, 0, None, None, None, None, None, None) == ffi.AMSM_OK


def test_host_pool_survives_fork(built_lib):
    """The host thread pool behind amsm_host_lincomb[_batch] must not hang a fork()ed child (Python multiprocessing's default
    start method): the child inherits the pool object but none of its threads.  Parent uses the pool, forks, the child
    runs a batch and a split combination and reports through its exit code."""
    import os
    from accumulation_amd import ffi
    from accumulation_amd.engine import _ptr
    lib = ffi.load()
    c = o.PALLAS
    g = o.generator(c)
    pts = [o.mul(c, 3 + i, g) for i in range(12)]
    sc = [o.rng_scalar(0x99, i) % (1 << 128) for i in range(12)]
    xy, inf = h.points_to_np(c, pts)
    scm = h.fr_mont_np(c, sc)
    want = None
    for P, s in zip(pts, sc):
        want = o.add(c, want, o.mul(c, s, P))

    def run():
        out = np.zeros((2 * c.limbs,), dtype=np.uint64)
        oinf = C.c_uint8(0)
        ffi.check(lib.amsm_host_lincomb(c.curve_id, _ptr(xy), _ptr(inf), _ptr(scm), 12, _ptr(out), C.byref(oinf)), "lincomb")
        return h.np_to_point(c, out, bool(oinf.value)) == want

    assert run()  # creates the pool in the parent (12 points: split over it)
    pid = os.fork()
    if pid == 0:
        ok = False
        try:
            ok = run() and run()
        finally:
            os._exit(0 if ok else 1)
    import signal
    import time
    t0 = time.time()
    while True:
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            break
        if time.time() - t0 > 60:
            os.kill(pid, signal.SIGKILL)
            os.waitpid(pid, 0)
            raise AssertionError("the forked child hung in the host pool")
        time.sleep(0.05)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
    assert run()  # the parent's pool still works


def test_host_helpers_from_concurrent_threads(built_lib):
    """include/amsm.h promises that the context-free host helpers may be called from any thread concurrently: four Python
    threads (ctypes releases the GIL) run single, split and batched combinations at once -- one of them gets the host pool,
    the others fall back to their own thread -- and every result equals the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    from accumulation_amd import ffi
    from accumulation_amd.engine import _ptr
    lib = ffi.load()
    c = o.PALLAS
    g = o.generator(c)
    pts = [o.mul(c, 7 + 2 * i, g) for i in range(16)]
    xy, inf = h.points_to_np(c, pts)

    def worker(seed):
        ok = True
        for it in range(12):
            n = [1, 3, 9, 16][it % 4]
            sc = [o.rng_scalar(seed, 16 * it + i) % (c.r if it % 3 == 0 else (1 << 128)) for i in range(n)]
            scm = h.fr_mont_np(c, sc)
            out = np.zeros((2 * c.limbs,), dtype=np.uint64)
            oinf = C.c_uint8(0)
            ffi.check(lib.amsm_host_lincomb(c.curve_id, _ptr(xy), _ptr(inf), _ptr(scm), n, _ptr(out), C.byref(oinf)), "lincomb")
            want = None
            for P, s in zip(pts, sc):
                want = o.add(c, want, o.mul(c, s, P))
            ok = ok and h.np_to_point(c, out, bool(oinf.value)) == want
            # a batch of three jobs over the same inputs
            nj = 3
            n_terms = (C.c_size_t * nj)(n, max(n - 1, 0), 1)
            xy_p, inf_p, sc_p = (C.c_void_p * nj)(), (C.c_void_p * nj)(), (C.c_void_p * nj)()
            for j in range(nj):
                xy_p[j], inf_p[j], sc_p[j] = xy.ctypes.data, inf.ctypes.data, scm.ctypes.data
            bout = np.zeros((nj, 2 * c.limbs), dtype=np.uint64)
            binf = np.zeros((nj,), dtype=np.uint8)
            ffi.check(lib.amsm_host_lincomb_batch(c.curve_id, nj, n_terms, xy_p, inf_p, sc_p, _ptr(bout), _ptr(binf)), "batch")
            ok = ok and h.np_to_point(c, bout[0], bool(binf[0])) == want
            ok = ok and h.np_to_point(c, bout[2], bool(binf[2])) == o.mul(c, sc[0], pts[0])
        return ok

    with ThreadPoolExecutor(max_workers=4) as ex:
        assert all(ex.map(worker, [101, 202, 303, 404]))
