"""TEST INFRASTRUCTURE ONLY -- big-integer restatement of the ark-serialize 0.2 wire format (ext: ark-serialize / ark-ff /
ark-ec are not in /root/reference; PARITY UNPINNED, see oracle/pyref.py).  The reference derives `CanonicalSerialize` for
every instance / witness / proof type (e.g. src/hp_as/data_structures.rs:13,53,76,94) and prints `serialized_size()` at
examples/scaling-as.rs:123-131.

  Fp                  canonical integer, little-endian, ceil((modulus bits + flag bits) / 8) bytes, flags in the top bits of
                      the last byte                                       (ark-ff: impl CanonicalSerializeWithFlags for Fp)
  SW point compressed x with SWFlags: bit 7 = y > -y (as integers), bit 6 = infinity     (ark-ec: GroupAffine::serialize)
  SW point uncompressed  x | y with the infinity flag                         (GroupAffine::serialize_uncompressed)
  Vec<T>              u64 little-endian length, then the elements          (ark-serialize: impl for Vec<T>)
  Option<T>           one byte 0 / 1, then the value                       (impl for Option<T>)
  usize / u64         8 bytes little-endian;  bool: one byte
Only tests/ may import this."""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

from . import pyref as o

FLAG_POSITIVE_Y, FLAG_INFINITY = 1 << 7, 1 << 6


def fp_size(modulus: int, flag_bits: int = 0) -> int:
    return (modulus.bit_length() + flag_bits + 7) // 8


def fr_serialize(c, x: int) -> bytes:
    return (x % c.r).to_bytes(fp_size(c.r), "little")


def fr_deserialize(c, b: bytes) -> int:
    x = int.from_bytes(b[: fp_size(c.r)], "little")
    if x >= c.r:
        raise ValueError("non-canonical scalar")
    return x


def point_size(c, compressed: bool = True) -> int:
    return fp_size(c.p, 2) if compressed else fp_size(c.p) + fp_size(c.p, 2)


def point_serialize(c, P, compressed: bool = True) -> bytes:
    sf = fp_size(c.p, 2)
    if compressed:
        if P is None:
            out = bytearray(sf)
            out[-1] |= FLAG_INFINITY
            return bytes(out)
        x, y = P
        out = bytearray(x.to_bytes(sf, "little"))
        if y > (-y) % c.p:
            out[-1] |= FLAG_POSITIVE_Y
        return bytes(out)
    sx = fp_size(c.p)
    if P is None:
        out = bytearray(sx + sf)
        out[-1] |= FLAG_INFINITY
        return bytes(out)
    return P[0].to_bytes(sx, "little") + P[1].to_bytes(sf, "little")


def _sqrt(a: int, p: int) -> Optional[int]:
    """Tonelli-Shanks"""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    if p % 4 == 3:
        return pow(a, (p + 1) // 4, p)
    s, t = 0, p - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
    z = 2
    while pow(z, (p - 1) // 2, p) == 1:
        z += 1
    m, cc, x, b = s, pow(z, t, p), pow(a, (t + 1) // 2, p), pow(a, t, p)
    while b != 1:
        i, b2 = 0, b
        while b2 != 1:
            b2, i = b2 * b2 % p, i + 1
        e = pow(cc, 1 << (m - i - 1), p)
        x, cc = x * e % p, e * e % p
        b, m = b * cc % p, i
    return x


def point_deserialize(c, b: bytes, compressed: bool = True):
    sf, sx = fp_size(c.p, 2), fp_size(c.p)
    buf = bytearray(b[: (sf if compressed else sx + sf)])
    pos, inf = bool(buf[-1] & FLAG_POSITIVE_Y), bool(buf[-1] & FLAG_INFINITY)
    if pos and inf:
        raise ValueError("both flags")
    buf[-1] &= 0x3F
    if compressed:
        x = int.from_bytes(buf, "little")
        if x >= c.p:
            raise ValueError("non-canonical x")
        if inf:
            return None
        y = _sqrt(x * x * x + c.b, c.p)
        if y is None:
            raise ValueError("no point with this x")
        ny = (-y) % c.p
        if (y > ny) != pos:
            y = ny
        P = (x, y)
    else:
        x, y = int.from_bytes(buf[:sx], "little"), int.from_bytes(buf[sx:], "little")
        if x >= c.p or y >= c.p or pos:
            raise ValueError("bad encoding")
        if inf:
            return None
        P = (x, y)
        if not o.is_on_curve(c, P):
            raise ValueError("off curve")
    if o.mul(c, c.r, P) is not None:
        raise ValueError("outside the prime-order subgroup")
    return P


def u64(x: int) -> bytes:
    return int(x).to_bytes(8, "little")


def vec(items: Sequence[bytes]) -> bytes:
    return u64(len(items)) + b"".join(items)


def option(item: Optional[bytes]) -> bytes:
    return b"\x00" if item is None else b"\x01" + item
