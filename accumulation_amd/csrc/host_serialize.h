// Wire format of field elements and curve points: ark-serialize 0.2 `CanonicalSerialize` (ext: the crate is not in
// /root/reference; the reference derives it for every instance / witness / proof type, e.g.
// src/hp_as/data_structures.rs:13,53,76,94, and prints the sizes at examples/scaling-as.rs:123-131).  PARITY UNPINNED: the
// layout below is the published one of ark-serialize / ark-ff / ark-ec 0.2, restated; no vector from the reference pins it.
//
//   Fp (ark-ff `impl CanonicalSerializeWithFlags for Fp`): the CANONICAL integer (`into_repr()`), little-endian, in
//       ceil((MODULUS_BITS + flag bits) / 8) bytes; the flags occupy the top bits of the LAST byte.
//   SW affine point, compressed (`GroupAffine::serialize`): x with SWFlags (2 flag bits): bit 7 of the last byte =
//       "y is the larger of (y, -y)" (`SWFlags::from_y_sign(y > -y)`), bit 6 = point at infinity (x = 0).  Both set is
//       invalid.  Pallas: 255 + 2 bits -> 33 bytes; BLS12-381 G1: 381 + 2 -> 48 bytes.
//   SW affine point, uncompressed (`serialize_uncompressed`): x (no flags) then y with SWFlags (infinity bit only):
//       Pallas 32 + 33 = 65 bytes, BLS12-381 G1 48 + 48 = 96 bytes.
//   Deserialisation rejects non-canonical integers (>= modulus), x without a square root, points off the curve
//       (uncompressed) and points outside the prime-order subgroup (BLS12-381 G1; Pallas has cofactor 1).
#pragma once
#include "host_field.h"

namespace amsm {
namespace host {

template <class P>
constexpr int h_modulus_bits() {
  int bits = 0;
  for (int i = HFe<P>::N - 1; i >= 0; i--) {
    u64 w = hmod<P>(i);
    if (w) {
      int b = 0;
      while (w) {
        w >>= 1;
        b++;
      }
      return i * 64 + b;
    }
  }
  return bits;
}
template <class P>
constexpr size_t h_serialized_size(int flag_bits) {
  return (size_t)(h_modulus_bits<P>() + flag_bits + 7) / 8;
}

// a^e, e as little-endian u64 limbs
template <class P>
inline HFe<P> h_pow(const HFe<P>& a, const u64* e, int n_limbs) {
  HFe<P> r = h_one<P>();
  bool started = false;
  for (int i = n_limbs * 64 - 1; i >= 0; i--) {
    if (started) r = h_sqr<P>(r);
    if ((e[i >> 6] >> (i & 63)) & 1) {
      r = started ? h_mul<P>(r, a) : a;
      started = true;
    }
  }
  return r;
}

// canonical-integer comparison a > b (both Montgomery): what `Ord for Fp` compares
template <class P>
inline bool h_gt_canonical(const HFe<P>& a, const HFe<P>& b) {
  HFe<P> x = h_from_mont<P>(a), y = h_from_mont<P>(b);
  for (int i = HFe<P>::N - 1; i >= 0; i--) {
    if (x.v[i] != y.v[i]) return x.v[i] > y.v[i];
  }
  return false;
}

// Square root by Tonelli-Shanks (any odd prime; Pallas Fq has 2-adicity 32, BLS12-381 Fq 1).  Returns false when a is
// not a square.
template <class P>
inline bool h_sqrt(const HFe<P>& a, HFe<P>* out) {
  constexpr int N = HFe<P>::N;
  if (h_is_zero<P>(a)) {
    *out = a;
    return true;
  }
  // m - 1 = 2^s * t
  u64 t[N], half[N];  // t; (m - 1) / 2
  for (int i = 0; i < N; i++) t[i] = hmod<P>(i);
  t[0] -= 1;  // m is odd
  for (int i = 0; i < N; i++) half[i] = (t[i] >> 1) | (i + 1 < N ? t[i + 1] << 63 : 0);
  int s = 0;
  while (!(t[0] & 1)) {
    for (int i = 0; i < N; i++) t[i] = (t[i] >> 1) | (i + 1 < N ? t[i + 1] << 63 : 0);
    s++;
  }
  const HFe<P> one = h_one<P>();
  if (!h_eq<P>(h_pow<P>(a, half, N), one)) return false;  // Euler's criterion
  // a non-residue z (smallest small integer that is one), c = z^t
  static const HFe<P> c0 = [&] {
    HFe<P> z = h_one<P>();
    for (;;) {
      z = h_add<P>(z, h_one<P>());
      if (!h_eq<P>(h_pow<P>(z, half, N), h_one<P>())) break;
    }
    return h_pow<P>(z, t, N);
  }();
  u64 t1[N];  // (t + 1) / 2
  {
    u128 c = 1;
    u64 tp[N];
    for (int i = 0; i < N; i++) {
      c += t[i];
      tp[i] = (u64)c;
      c >>= 64;
    }
    for (int i = 0; i < N; i++) t1[i] = (tp[i] >> 1) | (i + 1 < N ? tp[i + 1] << 63 : 0);
  }
  HFe<P> c = c0, x = h_pow<P>(a, t1, N), b = h_pow<P>(a, t, N);
  int m = s;
  while (!h_eq<P>(b, one)) {
    int i = 0;
    HFe<P> b2 = b;
    while (!h_eq<P>(b2, one)) {
      b2 = h_sqr<P>(b2);
      i++;
      if (i >= m) return false;
    }
    HFe<P> e = c;
    for (int k = 0; k < m - i - 1; k++) e = h_sqr<P>(e);
    x = h_mul<P>(x, e);
    c = h_sqr<P>(e);
    b = h_mul<P>(b, c);
    m = i;
  }
  *out = x;
  return true;
}

// canonical little-endian bytes of a Montgomery element into `size` bytes (size >= 8 N is padded with zeros)
template <class P>
inline void h_write_le(const HFe<P>& a_mont, uint8_t* out, size_t size) {
  HFe<P> c = h_from_mont<P>(a_mont);
  memset(out, 0, size);
  memcpy(out, c.v, std::min(size, sizeof(c.v)));  // little-endian host
}
// -> Montgomery; false when the integer is not canonical (>= modulus) or does not fit
template <class P>
inline bool h_read_le(const uint8_t* in, size_t size, HFe<P>* out_mont) {
  HFe<P> c = h_zero<P>();
  memcpy(c.v, in, std::min(size, sizeof(c.v)));
  for (size_t i = sizeof(c.v); i < size; i++)
    if (in[i]) return false;
  if (h_geq_mod<P>(c)) return false;
  *out_mont = h_to_mont<P>(c);
  return true;
}

constexpr uint8_t SW_FLAG_POSITIVE_Y = 1u << 7, SW_FLAG_INFINITY = 1u << 6;

template <class Fq>
inline HFe<Fq> curve_b_mont(int b_small) {
  HFe<Fq> b = h_zero<Fq>();
  b.v[0] = (u64)b_small;
  return h_to_mont<Fq>(b);
}

// y^2 = x^3 + b ?
template <class Fq>
inline bool on_curve(const HFe<Fq>& x, const HFe<Fq>& y, int b_small) {
  HFe<Fq> rhs = h_add<Fq>(h_mul<Fq>(h_sqr<Fq>(x), x), curve_b_mont<Fq>(b_small));
  return h_eq<Fq>(h_sqr<Fq>(y), rhs);
}

template <class Fq>
inline size_t point_serialized_size(bool compressed) {
  return compressed ? h_serialized_size<Fq>(2) : h_serialized_size<Fq>(0) + h_serialized_size<Fq>(2);
}

template <class Fq>
inline void point_serialize(const u64* xy_mont, bool is_inf, bool compressed, uint8_t* out) {
  constexpr int N = HFe<Fq>::N;
  const size_t sx = h_serialized_size<Fq>(0), sf = h_serialized_size<Fq>(2);
  HFe<Fq> x, y;
  memcpy(x.v, xy_mont, 8 * N);
  memcpy(y.v, xy_mont + N, 8 * N);
  if (is_inf) {
    x = h_zero<Fq>();
    y = h_zero<Fq>();
  }
  if (compressed) {
    h_write_le<Fq>(x, out, sf);
    if (is_inf) out[sf - 1] |= SW_FLAG_INFINITY;
    else if (h_gt_canonical<Fq>(y, h_neg<Fq>(y))) out[sf - 1] |= SW_FLAG_POSITIVE_Y;
  } else {
    h_write_le<Fq>(x, out, sx);
    h_write_le<Fq>(y, out + sx, sf);
    if (is_inf) out[sx + sf - 1] |= SW_FLAG_INFINITY;
  }
}

// r_limbs: the group order (subgroup check when cofactor != 1); returns false on an invalid encoding
template <class Fq>
inline bool point_deserialize(const uint8_t* in, bool compressed, int b_small, bool check_subgroup, const u64 r_limbs[4],
                              u64* xy_mont, uint8_t* is_inf) {
  constexpr int N = HFe<Fq>::N;
  const size_t sx = h_serialized_size<Fq>(0), sf = h_serialized_size<Fq>(2);
  std::vector<uint8_t> buf(in, in + (compressed ? sf : sx + sf));
  uint8_t& last = buf.back();
  const bool positive = (last & SW_FLAG_POSITIVE_Y) != 0, infinity = (last & SW_FLAG_INFINITY) != 0;
  if (positive && infinity) return false;  // `SWFlags::from_u8`: only one way to write the point at infinity
  last &= (uint8_t)~(SW_FLAG_POSITIVE_Y | SW_FLAG_INFINITY);
  HFe<Fq> x, y;
  if (compressed) {
    if (!h_read_le<Fq>(buf.data(), sf, &x)) return false;
    if (infinity) {
      memset(xy_mont, 0, 16 * N);
      *is_inf = 1;
      return true;  // ark-ec: `Self::zero()` whatever x was
    }
    HFe<Fq> rhs = h_add<Fq>(h_mul<Fq>(h_sqr<Fq>(x), x), curve_b_mont<Fq>(b_small));
    if (!h_sqrt<Fq>(rhs, &y)) return false;
    HFe<Fq> ny = h_neg<Fq>(y);
    // `get_point_from_x(x, greatest)`: the larger root when the flag is set
    const bool y_is_larger = h_gt_canonical<Fq>(y, ny);
    if (y_is_larger != positive) y = ny;
  } else {
    if (!h_read_le<Fq>(buf.data(), sx, &x) || !h_read_le<Fq>(buf.data() + sx, sf, &y)) return false;
    if (positive) return false;  // the uncompressed form never sets the sign bit
    if (infinity) {
      memset(xy_mont, 0, 16 * N);
      *is_inf = 1;
      return true;
    }
    if (!on_curve<Fq>(x, y, b_small)) return false;
  }
  memcpy(xy_mont, x.v, 8 * N);
  memcpy(xy_mont + N, y.v, 8 * N);
  *is_inf = 0;
  if (check_subgroup) {
    HXYZZ<Fq> p = hx_from_affine<Fq>(xy_mont, false);
    if (!hx_is_inf<Fq>(hx_mul<Fq>(p, r_limbs))) return false;
  }
  return true;
}

}  // namespace host
}  // namespace amsm
