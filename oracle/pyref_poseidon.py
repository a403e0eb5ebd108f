"""TEST INFRASTRUCTURE ONLY -- big-integer restatement of the reference's sponge: ark-sponge `PoseidonSponge<F>` at git
branch `accumulation-experimental` (Cargo.toml:18; NOT in /root/reference, not pinned: PARITY UNPINNED -- parameters and
encodings AS RECALLED from that branch's poseidon/mod.rs and lib.rs; see accumulation_amd/csrc/host_poseidon.h for the list).
Independent of the product: Python ints, its own ChaCha20 (checked against the RFC 7539 block-function vector in the tests).
Only tests/ may import this."""
from __future__ import annotations

from typing import List, Optional, Sequence

MASK32 = 0xFFFFFFFF
MASK64 = (1 << 64) - 1


def chacha20_block(words12: Sequence[int], key_words: Sequence[int]) -> List[int]:
    """state = constants | key (8 words) | words 12..15 (counter / nonce); 20 rounds; returns 16 output words."""
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + list(words12)
    x = list(s)

    def rotl(v, n):
        return ((v << n) & MASK32) | (v >> (32 - n))

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & MASK32; x[d] = rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & MASK32; x[b] = rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & MASK32; x[d] = rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & MASK32; x[b] = rotl(x[b] ^ x[c], 7)
    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & MASK32 for a, b in zip(x, s)]


class ChaCha20Rng:
    """rand_chacha::ChaCha20Rng::seed_from_u64: PCG32-expanded key, 64-bit block counter from 0, stream 0."""

    def __init__(self, seed_u64: int):
        MUL, INC = 6364136223846793005, 11634580027462260723
        st = seed_u64
        self.key = []
        for _ in range(8):
            st = (st * MUL + INC) & MASK64
            xs = (((st >> 18) ^ st) >> 27) & MASK32
            rot = st >> 59
            self.key.append(((xs >> rot) | (xs << ((32 - rot) & 31))) & MASK32)
        self.counter = 0
        self.buf: List[int] = []

    def next_u32(self) -> int:
        if not self.buf:
            self.buf = chacha20_block([self.counter & MASK32, self.counter >> 32, 0, 0], self.key)
            self.counter += 1
        return self.buf.pop(0)

    def next_u64(self) -> int:
        lo = self.next_u32()
        return lo | (self.next_u32() << 32)


class PoseidonSponge:
    RATE, CAP, FULL, PARTIAL, ALPHA = 2, 1, 8, 31, 17
    _params = {}

    def __init__(self, p: int):
        self.p = p
        self.limbs = (p.bit_length() + 63) // 64
        self.R = 1 << (64 * self.limbs)
        self.state = [0, 0, 0]
        self.squeezing = False
        self.idx = 0
        if p not in PoseidonSponge._params:
            rng = ChaCha20Rng(123456789)
            shave = 64 * self.limbs - p.bit_length()
            Rinv = pow(self.R, -1, p)
            ark = []
            for _ in range(self.FULL + self.PARTIAL):
                row = []
                for _ in range(3):
                    while True:
                        words = [rng.next_u64() for _ in range(self.limbs)]
                        words[-1] &= MASK64 >> shave
                        v = sum(w << (64 * i) for i, w in enumerate(words))
                        if v < p:
                            break
                    row.append(v * Rinv % p)  # the sampled integer is the MONTGOMERY representation
                ark.append(row)
            PoseidonSponge._params[p] = ark
        self.ark = PoseidonSponge._params[p]

    def clone(self) -> "PoseidonSponge":
        c = PoseidonSponge(self.p)
        c.state, c.squeezing, c.idx = list(self.state), self.squeezing, self.idx
        return c

    def permute(self):
        p, s = self.p, self.state
        for r in range(self.FULL + self.PARTIAL):
            s = [(s[i] + self.ark[r][i]) % p for i in range(3)]
            if r < self.FULL // 2 or r >= self.FULL // 2 + self.PARTIAL:
                s = [pow(x, self.ALPHA, p) for x in s]
            else:
                s[0] = pow(s[0], self.ALPHA, p)
            s = [(s[0] + s[2]) % p, (s[0] + s[1]) % p, (s[1] + s[2]) % p]
        self.state = s

    def absorb(self, elems: Sequence[int]):
        elems = [e % self.p for e in elems]
        if not elems:
            return
        if self.squeezing:
            self.permute()
            start = 0
        else:
            start = self.idx
            if start == self.RATE:
                self.permute()
                start = 0
        self.squeezing = False
        while True:
            if start + len(elems) <= self.RATE:
                for i, e in enumerate(elems):
                    self.state[start + i] = (self.state[start + i] + e) % self.p
                self.idx = start + len(elems)
                return
            take = self.RATE - start
            for i in range(take):
                self.state[start + i] = (self.state[start + i] + elems[i]) % self.p
            elems = elems[take:]
            self.permute()
            start = 0

    def squeeze(self, n: int) -> List[int]:
        if n == 0:
            return []
        if not self.squeezing:
            self.permute()
            start = 0
        else:
            start = self.idx
            if start == self.RATE:
                self.permute()
                start = 0
        self.squeezing = True
        out: List[int] = []
        while True:
            if start + (n - len(out)) <= self.RATE:
                k = n - len(out)
                out += self.state[start:start + k]
                self.idx = start + k
                return out
            out += self.state[start:self.RATE]
            self.permute()
            start = 0

    # ---- Absorbable encodings ----
    @property
    def usable_bytes(self) -> int:
        return (self.p.bit_length() - 1) // 8

    def absorb_bytes(self, b: bytes):
        ub = self.usable_bytes
        self.absorb([int.from_bytes(b[i:i + ub], "little") for i in range(0, len(b), ub)])

    def absorb_point(self, P):
        # (the identity: ark-ec ^0.2.0's `GroupAffine::zero()` = new(zero, ONE, true), fields absorbed as they are)
        self.absorb([0, 1, 1] if P is None else [P[0], P[1], 0])

    def fork(self, domain: bytes) -> "PoseidonSponge":
        c = self.clone()
        c.absorb_bytes(len(domain).to_bytes(8, "little") + domain)
        return c

    def squeeze_bits_int(self, n_bits: int) -> int:
        """the n_bits-bit stream as one integer (bit i of the stream = bit i of the result)"""
        ub = self.usable_bytes
        n_el = (n_bits + 8 * ub - 1) // (8 * ub)
        v = 0
        for i, e in enumerate(self.squeeze(n_el)):
            v |= (e & ((1 << (8 * ub)) - 1)) << (8 * ub * i)
        return v & ((1 << n_bits) - 1)

    def squeeze_nonnative(self, n_bits: int, count: int) -> List[int]:
        v = self.squeeze_bits_int(n_bits * count)
        return [(v >> (n_bits * k)) & ((1 << n_bits) - 1) for k in range(count)]
