"""Sponge interface used by the scheme mirrors for Fiat-Shamir challenges.

The reference is generic over `S: CryptographicSponge<ConstraintF<G>>` (src/hp_as/mod.rs:98-103) and its
tests instantiate a Poseidon sponge from ark-sponge (not in the reference tree, not pinned).  Hashing
O(#inputs) commitments is host-side work outside the accelerated path, so the mirrors take any object with
this interface.  `Sha256Sponge` is a simple stand-in for tests/benchmarks -- NOT the reference's Poseidon,
so challenges (and therefore accumulators) differ from a Rust run; the algebraic prove/verify/decide
relations the reference tests (src/lib.rs:334-395) hold for any sponge."""
from __future__ import annotations

import hashlib
from typing import List

import numpy as np


class CryptographicSponge:
    def absorb_bytes(self, b: bytes) -> None:
        raise NotImplementedError

    def squeeze_bits(self, n_bits: int) -> int:
        raise NotImplementedError

    # helpers shared by implementations
    def absorb_u64(self, x: int) -> None:
        self.absorb_bytes(int(x).to_bytes(8, "little"))

    def absorb_point(self, pt) -> None:
        xy, inf = pt
        self.absorb_bytes(np.asarray(xy, dtype=np.uint64).tobytes() + (b"\x01" if inf else b"\x00"))

    def absorb_len(self, n: int) -> None:
        """The length of a Vec about to be absorbed item by item: framing for hash-based sponges only -- the reference's
        field-element encoding of a Vec has none (PoseidonSponge ignores it)."""
        self.absorb_u64(n)

    def absorb_points(self, pts) -> None:
        self.absorb_len(len(pts))
        for p in pts:
            self.absorb_point(p)

    def squeeze_field_elements(self, n: int, n_bits: int = 128) -> List[int]:
        """`squeeze_nonnative_field_elements_with_sizes(Truncated(n_bits))`: n integers < 2^n_bits."""
        return [self.squeeze_bits(n_bits) for _ in range(n)]

    def fork(self, domain: bytes) -> "CryptographicSponge":
        raise NotImplementedError


class Sha256Sponge(CryptographicSponge):
    def __init__(self, state: bytes = b"amsm-sha256-sponge"):
        self._state = hashlib.sha256(state).digest()
        self._ctr = 0

    def absorb_bytes(self, b: bytes) -> None:
        self._state = hashlib.sha256(self._state + b"A" + len(b).to_bytes(8, "little") + b).digest()
        self._ctr = 0

    def squeeze_bits(self, n_bits: int) -> int:
        out = b""
        while len(out) * 8 < n_bits:
            out += hashlib.sha256(self._state + b"S" + self._ctr.to_bytes(8, "little")).digest()
            self._ctr += 1
        return int.from_bytes(out, "little") & ((1 << n_bits) - 1)

    def fork(self, domain: bytes) -> "Sha256Sponge":
        """Domain-separated child sponge; fork(b"") is a plain clone of the current state."""
        if not domain:
            c = Sha256Sponge()
            c._state, c._ctr = self._state, self._ctr
            return c
        return Sha256Sponge(self._state + b"F" + domain)


class PoseidonSponge(CryptographicSponge):
    """The reference's sponge: ark-sponge `PoseidonSponge<ConstraintF<G>>` over the curve's base field (ext; parameters and
    encodings as stated in accumulation_amd/csrc/host_poseidon.h -- PARITY UNPINNED), through the host-side C ABI
    (include/amsm.h: amsm_poseidon_*).  Absorb calls of the scheme mirrors map to the reference's `Absorbable` encodings:
    bytes -> 31-byte chunks, usize -> one element, point -> (x, y, infinity), Vec -> items without a length."""

    def __init__(self, curve: int = 0, _handle=None):
        import ctypes as C

        from . import ffi
        self._C, self._ffi, self._lib, self.curve = C, ffi, ffi.load(), curve
        if _handle is None:
            _handle = C.c_void_p()
            ffi.check(self._lib.amsm_poseidon_new(curve, C.byref(_handle)), "amsm_poseidon_new")
        self._h = _handle

    def __del__(self):
        try:
            self._lib.amsm_poseidon_free(self._h)
        except Exception:
            pass

    def absorb_bytes(self, b: bytes) -> None:
        buf = (self._C.c_uint8 * max(len(b), 1)).from_buffer_copy(b or b"\x00")
        self._ffi.check(self._lib.amsm_poseidon_absorb_bytes(self._h, buf, len(b)), "amsm_poseidon_absorb_bytes")

    def absorb_u64(self, x: int) -> None:
        self._ffi.check(self._lib.amsm_poseidon_absorb_u64(self._h, int(x)), "amsm_poseidon_absorb_u64")

    def absorb_len(self, n: int) -> None:
        pass  # a Vec is absorbed item by item, without its length

    def absorb_point(self, pt) -> None:
        xy, inf = pt
        a = np.ascontiguousarray(xy, dtype=np.uint64)
        f = np.array([1 if inf else 0], dtype=np.uint8)
        self._ffi.check(self._lib.amsm_poseidon_absorb_points(self._h, a.ctypes.data_as(self._C.c_void_p),
                                                              f.ctypes.data_as(self._C.c_void_p), 1), "amsm_poseidon_absorb_points")

    def squeeze_field_elements(self, n: int, n_bits: int = 128) -> List[int]:
        if n == 0:
            return []
        out = np.zeros((n, 4), dtype=np.uint64)
        self._ffi.check(self._lib.amsm_poseidon_squeeze_nonnative(self._h, n_bits, n, out.ctypes.data_as(self._C.c_void_p)),
                        "amsm_poseidon_squeeze_nonnative")
        return [sum(int(w) << (64 * i) for i, w in enumerate(row)) for row in out]

    def squeeze_bits(self, n_bits: int) -> int:
        return self.squeeze_field_elements(1, n_bits)[0]

    def fork(self, domain: bytes) -> "PoseidonSponge":
        C = self._C
        out = C.c_void_p()
        if not domain:  # fork(b"") of the mirrors = a plain clone
            self._ffi.check(self._lib.amsm_poseidon_clone(self._h, C.byref(out)), "amsm_poseidon_clone")
        else:
            buf = (C.c_uint8 * len(domain)).from_buffer_copy(domain)
            self._ffi.check(self._lib.amsm_poseidon_fork(self._h, buf, len(domain), C.byref(out)), "amsm_poseidon_fork")
        return PoseidonSponge(self.curve, out)
