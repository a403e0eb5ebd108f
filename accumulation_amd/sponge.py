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

    def absorb_points(self, pts) -> None:
        self.absorb_u64(len(pts))
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
