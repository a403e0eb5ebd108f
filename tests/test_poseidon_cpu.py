"""Poseidon sponge behind the C ABI (amsm_poseidon_*, host only: runs without a GPU) against the independent big-int
restatement oracle/pyref_poseidon.py: round constants, the permutation, duplex absorb / squeeze sequences, the byte / point /
usize encodings, fork, and the batched truncated non-native squeeze the schemes use for their challenges."""
import ctypes as C
import random

import numpy as np
import pytest

from oracle import pyref as o
from oracle import pyref_poseidon as pp
from tests import helpers as h


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def test_chacha20_core_matches_rfc7539():
    """RFC 7539 section 2.3.2: the oracle's block function (the product has its own copy; both feed the same constants)"""
    key = list(np.frombuffer(bytes(range(32)), dtype="<u4"))
    out = pp.chacha20_block([1, 0x09000000, 0x4A000000, 0x00000000], [int(k) for k in key])
    assert out[:4] == [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3]
    assert out[12:] == [0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]


def fq_limbs(c):
    return c.limbs


def to_mont_words(c, vals):
    out = np.zeros((len(vals), c.limbs), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = np.array(o.int_to_limbs(o.fq_to_mont(c, v % c.p), c.limbs), dtype=np.uint64)
    return out


def from_mont_words(c, arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, c.limbs)
    return [o.fq_from_mont(c, o.limbs_to_int([int(x) for x in row])) for row in arr]


@pytest.mark.parametrize("curve", ["pallas", "bls12_381_g1"])
def test_round_constants_and_permutation(built_lib, curve):
    c = o.CURVES[curve]
    ref = pp.PoseidonSponge(c.p)
    rc = np.zeros((39 * 3, c.limbs), dtype=np.uint64)
    assert built_lib.amsm_poseidon_round_constants(c.curve_id, _ptr(rc)) == 0
    assert from_mont_words(c, rc) == [x for row in ref.ark for x in row]
    rnd = random.Random(5)
    for state in ([0, 0, 0], [1, 2, 3], [rnd.randrange(c.p) for _ in range(3)], [c.p - 1, 0, c.p - 2]):
        w = to_mont_words(c, state)
        assert built_lib.amsm_poseidon_permute(c.curve_id, _ptr(w)) == 0
        ref.state = list(state)
        ref.permute()
        assert from_mont_words(c, w) == ref.state


class LibSponge:
    def __init__(self, lib, c, handle=None):
        self.lib, self.c = lib, c
        if handle is None:
            handle = C.c_void_p()
            assert lib.amsm_poseidon_new(c.curve_id, C.byref(handle)) == 0
        self.h = handle

    def absorb(self, elems):
        w = to_mont_words(self.c, elems)
        assert self.lib.amsm_poseidon_absorb_native(self.h, _ptr(w), len(elems)) == 0

    def squeeze(self, n):
        out = np.zeros((n, self.c.limbs), dtype=np.uint64)
        assert self.lib.amsm_poseidon_squeeze_native(self.h, n, _ptr(out)) == 0
        return from_mont_words(self.c, out)

    def absorb_bytes(self, b):
        buf = np.frombuffer(b, dtype=np.uint8).copy() if b else np.zeros(1, dtype=np.uint8)
        assert self.lib.amsm_poseidon_absorb_bytes(self.h, _ptr(buf), len(b)) == 0

    def absorb_point(self, P):
        xy, inf = h.points_to_np(self.c, [P])
        assert self.lib.amsm_poseidon_absorb_points(self.h, _ptr(xy), _ptr(inf), 1) == 0

    def absorb_u64(self, v):
        assert self.lib.amsm_poseidon_absorb_u64(self.h, v) == 0

    def fork(self, domain):
        out = C.c_void_p()
        buf = np.frombuffer(domain, dtype=np.uint8).copy()
        assert self.lib.amsm_poseidon_fork(self.h, _ptr(buf), len(domain), C.byref(out)) == 0
        return LibSponge(self.lib, self.c, out)

    def squeeze_nonnative(self, n_bits, count):
        out = np.zeros((count, 4), dtype=np.uint64)
        assert self.lib.amsm_poseidon_squeeze_nonnative(self.h, n_bits, count, _ptr(out)) == 0
        return h.np_to_ints(out)

    def squeeze_bits_int(self, n_bits):
        out = np.zeros((n_bits + 7) // 8, dtype=np.uint8)
        assert self.lib.amsm_poseidon_squeeze_bits(self.h, n_bits, _ptr(out)) == 0
        return int.from_bytes(bytes(out), "little")

    def __del__(self):
        self.lib.amsm_poseidon_free(self.h)


@pytest.mark.parametrize("curve", ["pallas", "bls12_381_g1"])
def test_duplex_sequences(built_lib, curve):
    """random interleavings of absorb / squeeze of random lengths (rate boundaries, absorb-after-squeeze, empty absorbs)"""
    c = o.CURVES[curve]
    rnd = random.Random(11)
    for trial in range(6):
        a, b = LibSponge(built_lib, c), pp.PoseidonSponge(c.p)
        for _ in range(14):
            if rnd.random() < 0.55:
                els = [rnd.randrange(c.p) for _ in range(rnd.choice([0, 1, 1, 2, 3, 5]))]
                a.absorb(els)
                b.absorb(els)
            else:
                k = rnd.choice([1, 1, 2, 3, 4])
                assert a.squeeze(k) == b.squeeze(k)
        assert a.squeeze(2) == b.squeeze(2)


@pytest.mark.parametrize("curve", ["pallas", "bls12_381_g1"])
def test_encodings_fork_and_challenges(built_lib, curve):
    c = o.CURVES[curve]
    a, b = LibSponge(built_lib, c), pp.PoseidonSponge(c.p)
    g = o.generator(c)
    for P in (g, None, o.mul(c, 7, g)):
        a.absorb_point(P)
        b.absorb_point(P)
    a.absorb_u64(1 << 40)
    b.absorb([1 << 40])
    for blob in (b"", b"\x01", bytes(range(31)), bytes(range(32)), bytes(range(100)), b"\xff" * 62):
        a.absorb_bytes(blob)
        b.absorb_bytes(blob)
    fa, fb = a.fork(b"AS-FOR-HP-2020"), b.fork(b"AS-FOR-HP-2020")
    assert fa.squeeze_nonnative(128, 3) == fb.squeeze_nonnative(128, 3)   # the mu challenges: one squeeze of 384 bits
    assert fa.squeeze_nonnative(128, 1) == fb.squeeze_nonnative(128, 1)   # then nu
    assert fa.squeeze_nonnative(184, 1) == fb.squeeze_nonnative(184, 1)   # CHALLENGE_POINT_SIZE of the PC schemes
    assert a.squeeze_bits_int(300) == b.squeeze_bits_int(300)
    # the parent was not disturbed by its fork
    assert a.squeeze(3) == b.squeeze(3)
    # a batch is NOT the same as squeezing one challenge at a time (each call starts a fresh native element)
    x, y = pp.PoseidonSponge(c.p), pp.PoseidonSponge(c.p)
    x.absorb([5]); y.absorb([5])
    assert x.squeeze_nonnative(128, 2)[1] != [y.squeeze_nonnative(128, 1), y.squeeze_nonnative(128, 1)][1][0]


def test_chacha20_zero_key_keystream():
    """the well-known all-zero key / counter / nonce keystream block (with the RFC 7539 block above: two published vectors for
    the function that generates the Poseidon round constants)"""
    import struct
    ks = b"".join(struct.pack("<I", w) for w in pp.chacha20_block([0, 0, 0, 0], [0] * 8))
    assert ks.hex() == ("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                        "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
