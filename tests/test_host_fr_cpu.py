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
