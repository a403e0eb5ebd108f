"""Scalar-field (Fr) bookkeeping for the host-side scheme mirrors: Python-int arithmetic for the O(#inputs)
challenge algebra and conversions to/from the ABI's Montgomery limbs.  (Product code; independent of oracle/.)"""
from __future__ import annotations

import numpy as np

from . import ffi

MODULI = {
    ffi.AMSM_PALLAS: 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,
    ffi.AMSM_BLS12_381_G1: 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
}
_R = 1 << 256
_M64 = (1 << 64) - 1


class Fr:
    def __init__(self, curve: int):
        self.r = MODULI[curve]
        self._rinv = pow(_R, -1, self.r)

    def to_limbs(self, x: int) -> np.ndarray:
        """canonical int -> (4,) uint64 Montgomery limbs"""
        m = (x % self.r) * _R % self.r
        return np.array([(m >> (64 * i)) & _M64 for i in range(4)], dtype=np.uint64)

    def to_limbs_many(self, xs) -> np.ndarray:
        """canonical ints -> (n, 4) uint64 Montgomery limbs (one bytes join instead of n small arrays: the O(d) host
        loops of trivial_pc_as and the NARK's assignment uploads convert thousands of scalars per call)"""
        if not len(xs):
            return np.zeros((0, 4), dtype=np.uint64)
        r = self.r
        buf = b"".join(((int(x) % r) * _R % r).to_bytes(32, "little") for x in xs)
        return np.frombuffer(buf, dtype="<u8").reshape(-1, 4).copy()

    def from_limbs(self, limbs) -> int:
        m = 0
        for i, l in enumerate(np.asarray(limbs, dtype=np.uint64).reshape(4)):
            m |= int(l) << (64 * i)
        return m * self._rinv % self.r
