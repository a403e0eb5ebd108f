"""Wire format of field elements and points (amsm_fr_serialize / amsm_points_serialize, host only: runs without a GPU)
against the big-int restatement of ark-serialize 0.2 (oracle/pyref_ser.py): byte-for-byte, round trips, and the rejections
`CanonicalDeserialize` makes (non-canonical integers, x without a point, both flags, points outside the subgroup)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyref as o
from oracle import pyref_ser as ser
from tests import helpers as h


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def lib_points_serialize(lib, c, pts, compressed):
    xy, inf = h.points_to_np(c, pts)
    sz = lib.amsm_point_serialized_size(c.curve_id, int(compressed))
    out = np.zeros(len(pts) * sz, dtype=np.uint8)
    assert lib.amsm_points_serialize(c.curve_id, _ptr(xy), _ptr(inf), len(pts), int(compressed), _ptr(out)) == 0
    return [bytes(out[i * sz:(i + 1) * sz]) for i in range(len(pts))], sz


def lib_points_deserialize(lib, c, blobs, compressed):
    n = len(blobs)
    buf = np.frombuffer(b"".join(blobs), dtype=np.uint8).copy()
    xy = np.zeros((n, 2 * c.limbs), dtype=np.uint64)
    inf = np.zeros(n, dtype=np.uint8)
    rc = lib.amsm_points_deserialize(c.curve_id, _ptr(buf), n, int(compressed), _ptr(xy), _ptr(inf))
    return rc, [h.np_to_point(c, xy[i], inf[i]) for i in range(n)]


@pytest.mark.parametrize("curve", ["pallas", "bls12_381_g1"])
def test_sizes(built_lib, curve):
    c = o.CURVES[curve]
    assert built_lib.amsm_fr_serialized_size(c.curve_id) == 32 == ser.fp_size(c.r)
    assert built_lib.amsm_point_serialized_size(c.curve_id, 1) == ser.point_size(c, True) == {"pallas": 33, "bls12_381_g1": 48}[curve]
    assert built_lib.amsm_point_serialized_size(c.curve_id, 0) == ser.point_size(c, False) == {"pallas": 65, "bls12_381_g1": 96}[curve]


@pytest.mark.parametrize("curve", ["pallas", "bls12_381_g1"])
def test_scalars(built_lib, curve):
    c = o.CURVES[curve]
    vals = [0, 1, 2, c.r - 1, c.r // 2, (1 << 200) + 12345] + [s % c.r for s in o.rng_scalars(77, 20)]
    mont = h.fr_mont_np(c, vals)
    out = np.zeros(32 * len(vals), dtype=np.uint8)
    assert built_lib.amsm_fr_serialize(c.curve_id, _ptr(mont), len(vals), _ptr(out)) == 0
    assert bytes(out) == b"".join(ser.fr_serialize(c, v) for v in vals)
    back = np.zeros_like(mont)
    assert built_lib.amsm_fr_deserialize(c.curve_id, _ptr(out), len(vals), _ptr(back)) == 0
    assert np.array_equal(back, mont)
    bad = np.frombuffer(c.r.to_bytes(32, "little"), dtype=np.uint8).copy()  # r itself is not canonical
    assert built_lib.amsm_fr_deserialize(c.curve_id, _ptr(bad), 1, _ptr(back)) != 0


@pytest.mark.parametrize("curve", ["pallas", "bls12_381_g1"])
@pytest.mark.parametrize("compressed", [True, False], ids=["compressed", "uncompressed"])
def test_points(built_lib, curve, compressed):
    c = o.CURVES[curve]
    g = o.generator(c)
    pts = [None, g, o.neg(c, g), o.mul(c, 2, g), o.mul(c, 3, g)] + [o.mul(c, k, g) for k in o.rng_scalars(5, 12)]
    blobs, sz = lib_points_serialize(built_lib, c, pts, compressed)
    for P, b in zip(pts, blobs):
        assert b == ser.point_serialize(c, P, compressed)
        assert ser.point_deserialize(c, b, compressed) == P
    rc, back = lib_points_deserialize(built_lib, c, blobs, compressed)
    assert rc == 0 and back == pts
    # P and -P differ only in the sign bit (compressed)
    if compressed:
        a, b = blobs[1], blobs[2]
        assert a[:-1] == b[:-1] and (a[-1] ^ b[-1]) == ser.FLAG_POSITIVE_Y


@pytest.mark.parametrize("curve", ["pallas", "bls12_381_g1"])
def test_rejections(built_lib, curve):
    c = o.CURVES[curve]
    sf = ser.fp_size(c.p, 2)
    g = o.generator(c)

    def rc_of(blob, compressed=True):
        return lib_points_deserialize(built_lib, c, [blob], compressed)[0]
    good = ser.point_serialize(c, g)
    assert rc_of(good) == 0
    both = bytearray(good)
    both[-1] |= ser.FLAG_POSITIVE_Y | ser.FLAG_INFINITY
    assert rc_of(bytes(both)) != 0                                   # both flags
    assert rc_of(c.p.to_bytes(sf, "little")) != 0                    # x = p: not canonical
    x = 1
    while ser._sqrt(x * x * x + c.b, c.p) is not None:               # an x with no point on the curve
        x += 1
    assert rc_of(x.to_bytes(sf, "little")) != 0
    off = g[0].to_bytes(ser.fp_size(c.p), "little") + ((g[1] + 1) % c.p).to_bytes(sf, "little")
    assert rc_of(off, compressed=False) != 0                         # off the curve
    if curve == "bls12_381_g1":                                      # on the curve, outside G1 (cofactor != 1)
        x = 1
        while True:
            y = ser._sqrt(x * x * x + c.b, c.p)
            if y is not None and o.mul(c, c.r, (x, y)) is not None:
                break
            x += 1
        assert rc_of(ser.point_serialize(c, (x, y))) != 0
        with pytest.raises(ValueError):
            ser.point_deserialize(c, ser.point_serialize(c, (x, y)))


def test_bls12_381_generator_known_encoding(built_lib):
    """One known-answer vector that does not come from this repo's own oracle: the compressed arkworks (0.2 / 0.3) encoding
    of the BLS12-381 G1 generator as it circulates in arkworks-based projects -- x little-endian, no flag bit set (the
    generator's y is the smaller of the two roots, infinity flag clear).  It pins the byte order, the position of the flag
    bits and the sign convention of the point encoding; recalled from public material, /root/reference holds no vectors."""
    c = o.CURVES["bls12_381_g1"]
    known = bytes.fromhex("bbc622db0af03afbef1a7af93fe8556c58ac1b173f3a4ea105b974974f8c68c3"
                          "0faca94f8c63952694d79731a7d3f117")
    g = o.generator(c)
    assert ser.point_serialize(c, g, True) == known
    blobs, _ = lib_points_serialize(built_lib, c, [g, o.neg(c, g)], True)
    assert blobs[0] == known
    assert blobs[1] == known[:-1] + bytes([known[-1] | ser.FLAG_POSITIVE_Y])
    rc, back = lib_points_deserialize(built_lib, c, [known], True)
    assert rc == 0 and back == [g]
