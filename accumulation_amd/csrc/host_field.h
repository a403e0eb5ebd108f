// Host-side (x86-64, u64 limbs, unsigned __int128) field + curve arithmetic used by the C ABI for the
// O(1)-sized work that stays on the host by design: folding the handful of per-block / per-window
// partial sums the GPU hands back, the final projective -> affine normalisation (what the reference
// does with `.into_affine()` / `batch_normalization_into_affine`, src/hp_as/mod.rs:468) and the single
// scalar-mul of a hiding term ("stay on host").
// This is product code, independent of oracle/ (which is test infrastructure).
#pragma once
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "fp.h"

namespace amsm {
namespace host {

typedef unsigned __int128 u128;

template <class P>
struct HFe {
  static constexpr int N = P::L / 2;
  u64 v[N];
};

template <class P>
constexpr u64 hmod(int i) {
  return (u64)P::mod(2 * i) | ((u64)P::mod(2 * i + 1) << 32);
}
template <class P>
constexpr u64 hone(int i) {
  return (u64)P::one(2 * i) | ((u64)P::one(2 * i + 1) << 32);
}
template <class P>
constexpr u64 hr2(int i) {
  return (u64)P::r2(2 * i) | ((u64)P::r2(2 * i + 1) << 32);
}
template <class P>
constexpr u64 hinv64() {  // -m^-1 mod 2^64 by Newton iteration
  u64 m = hmod<P>(0), x = 1;
  for (int i = 0; i < 6; i++) x *= 2 - m * x;
  return (u64)0 - x;
}

template <class P>
inline HFe<P> h_zero() {
  HFe<P> r;
  for (int i = 0; i < HFe<P>::N; i++) r.v[i] = 0;
  return r;
}
template <class P>
inline HFe<P> h_one() {
  HFe<P> r;
  for (int i = 0; i < HFe<P>::N; i++) r.v[i] = hone<P>(i);
  return r;
}
template <class P>
inline bool h_is_zero(const HFe<P>& a) {
  u64 o = 0;
  for (int i = 0; i < HFe<P>::N; i++) o |= a.v[i];
  return o == 0;
}
template <class P>
inline bool h_eq(const HFe<P>& a, const HFe<P>& b) {
  u64 o = 0;
  for (int i = 0; i < HFe<P>::N; i++) o |= a.v[i] ^ b.v[i];
  return o == 0;
}
template <class P>
inline bool h_geq_mod(const HFe<P>& a) {
  for (int i = HFe<P>::N - 1; i >= 0; i--) {
    if (a.v[i] > hmod<P>(i)) return true;
    if (a.v[i] < hmod<P>(i)) return false;
  }
  return true;
}
template <class P>
inline void h_sub_mod(HFe<P>& a) {
  u64 br = 0;
  for (int i = 0; i < HFe<P>::N; i++) {
    u128 x = (u128)a.v[i] - hmod<P>(i) - br;
    a.v[i] = (u64)x;
    br = (u64)(x >> 64) & 1;
  }
}
template <class P>
inline HFe<P> h_add(const HFe<P>& a, const HFe<P>& b) {
  HFe<P> r;
  u128 c = 0;
  for (int i = 0; i < HFe<P>::N; i++) {
    c += (u128)a.v[i] + b.v[i];
    r.v[i] = (u64)c;
    c >>= 64;
  }
  if (c || h_geq_mod<P>(r)) h_sub_mod<P>(r);
  return r;
}
template <class P>
inline HFe<P> h_sub(const HFe<P>& a, const HFe<P>& b) {
  HFe<P> r;
  u64 br = 0;
  for (int i = 0; i < HFe<P>::N; i++) {
    u128 x = (u128)a.v[i] - b.v[i] - br;
    r.v[i] = (u64)x;
    br = (u64)(x >> 64) & 1;
  }
  if (br) {
    u128 c = 0;
    for (int i = 0; i < HFe<P>::N; i++) {
      c += (u128)r.v[i] + hmod<P>(i);
      r.v[i] = (u64)c;
      c >>= 64;
    }
  }
  return r;
}
template <class P>
inline HFe<P> h_neg(const HFe<P>& a) {
  if (h_is_zero<P>(a)) return a;
  return h_sub<P>(h_zero<P>(), a);
}
// Montgomery product, general form (CIOS with an overflow word): any a, b < 2^(64 N)
template <class P>
inline HFe<P> h_mul_general(const HFe<P>& a, const HFe<P>& b) {
  constexpr int N = HFe<P>::N;
  constexpr u64 inv = hinv64<P>();
  u64 t[N + 2];
  for (int i = 0; i < N + 2; i++) t[i] = 0;
  for (int i = 0; i < N; i++) {
    u128 c = 0;
    for (int j = 0; j < N; j++) {
      c += (u128)a.v[j] * b.v[i] + t[j];
      t[j] = (u64)c;
      c >>= 64;
    }
    c += t[N];
    t[N] = (u64)c;
    t[N + 1] = (u64)(c >> 64);
    u64 m = t[0] * inv;
    c = ((u128)m * hmod<P>(0) + t[0]) >> 64;
    for (int j = 1; j < N; j++) {
      c += (u128)m * hmod<P>(j) + t[j];
      t[j - 1] = (u64)c;
      c >>= 64;
    }
    c += t[N];
    t[N - 1] = (u64)c;
    t[N] = t[N + 1] + (u64)(c >> 64);
  }
  HFe<P> r;
  for (int i = 0; i < N; i++) r.v[i] = t[i];
  if (t[N] || h_geq_mod<P>(r)) h_sub_mod<P>(r);
  return r;
}
// Montgomery product.  Every modulus of the library leaves the top bit of its top word clear (255 / 381 bits in 256 / 384): with
// operands below 2^(64 N - 1) -- every reduced element, and what the C ABI documents -- the two carry chains of a round never
// overflow a word ("no-carry" CIOS): one fully unrolled pass per word of b, 28 against 33 ns (Pallas) and 53 against 62 ns
// (BLS12-381) per dependent product on the build box.  Same integer as the general form, which wider operands still take.
template <class P>
inline HFe<P> h_mul(const HFe<P>& a, const HFe<P>& b) {
  constexpr int N = HFe<P>::N;
  constexpr u64 inv = hinv64<P>();
  static_assert((hmod<P>(N - 1) >> 63) == 0, "the no-carry form needs a spare top bit");
  if (__builtin_expect(((a.v[N - 1] | b.v[N - 1]) >> 63) != 0, 0)) return h_mul_general<P>(a, b);
  u64 t[N];
  for (int i = 0; i < N; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    u128 c = (u128)a.v[0] * b.v[i] + t[0];
    const u64 lo = (u64)c;
    u64 A = (u64)(c >> 64);
    const u64 m = lo * inv;
    u128 d = (u128)m * hmod<P>(0) + lo;
    u64 B = (u64)(d >> 64);
#pragma unroll
    for (int j = 1; j < N; j++) {
      c = (u128)a.v[j] * b.v[i] + t[j] + A;
      A = (u64)(c >> 64);
      d = (u128)m * hmod<P>(j) + (u64)c + B;
      B = (u64)(d >> 64);
      t[j - 1] = (u64)d;
    }
    t[N - 1] = A + B;
  }
  HFe<P> r;
  for (int i = 0; i < N; i++) r.v[i] = t[i];
  if (h_geq_mod<P>(r)) h_sub_mod<P>(r);
  return r;
}
template <class P>
inline HFe<P> h_sqr(const HFe<P>& a) {
  return h_mul<P>(a, a);
}
template <class P>
inline HFe<P> h_inv_fermat(const HFe<P>& a) {  // a^(m-2): the reference formulation h_inv is checked against
  constexpr int N = HFe<P>::N;
  u64 e[N];
  u64 br = 2;
  for (int i = 0; i < N; i++) {
    u128 x = (u128)hmod<P>(i) - br;
    e[i] = (u64)x;
    br = (u64)(x >> 64) & 1;
  }
  HFe<P> r = h_one<P>();
  for (int i = N * 64 - 1; i >= 0; i--) {
    r = h_sqr<P>(r);
    if ((e[i >> 6] >> (i & 63)) & 1) r = h_mul<P>(r, a);
  }
  return r;
}
// 1 / a (Montgomery in, Montgomery out; 0 -> 0) by the binary extended Euclid on the plain integers: ~2 * bits shift /
// subtract steps on N limbs (3 us for a 255-bit field against 12 us for the 384 multiplications of Fermat's a^(m-2), which
// sat twice in every IPA round's host path).  With a_mont = a R the loop yields (a R)^-1; two multiplications by R^2 lift
// it to a^-1 R.
template <class P>
inline HFe<P> h_inv(const HFe<P>& a) {
  constexpr int N = HFe<P>::N;
  if (h_is_zero<P>(a)) return a;
  u64 u[N], v[N], x1[N], x2[N], m[N];
  for (int i = 0; i < N; i++) {
    u[i] = a.v[i];
    v[i] = m[i] = hmod<P>(i);
    x1[i] = x2[i] = 0;
  }
  x1[0] = 1;
  auto is_one = [](const u64* x) {
    u64 o = x[0] ^ 1ull;
    for (int i = 1; i < N; i++) o |= x[i];
    return o == 0;
  };
  auto shr1 = [](u64* x, u64 top) {  // x = (top : x) >> 1
    for (int i = 0; i < N - 1; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 63);
    x[N - 1] = (x[N - 1] >> 1) | (top << 63);
  };
  auto halve_mod = [&](u64* x) {  // x / 2 mod m (m odd)
    u64 top = 0;
    if (x[0] & 1) {
      u128 c = 0;
      for (int i = 0; i < N; i++) {
        c += (u128)x[i] + m[i];
        x[i] = (u64)c;
        c >>= 64;
      }
      top = (u64)c;
    }
    shr1(x, top);
  };
  auto geq = [](const u64* x, const u64* y) {
    for (int i = N - 1; i >= 0; i--)
      if (x[i] != y[i]) return x[i] > y[i];
    return true;
  };
  auto sub = [](u64* x, const u64* y) {  // x -= y (x >= y)
    u64 br = 0;
    for (int i = 0; i < N; i++) {
      u128 d = (u128)x[i] - y[i] - br;
      x[i] = (u64)d;
      br = (u64)(d >> 64) & 1;
    }
  };
  auto sub_mod = [&](u64* x, const u64* y) {  // x = x - y mod m (x, y < m)
    u64 br = 0;
    for (int i = 0; i < N; i++) {
      u128 d = (u128)x[i] - y[i] - br;
      x[i] = (u64)d;
      br = (u64)(d >> 64) & 1;
    }
    if (br) {
      u128 c = 0;
      for (int i = 0; i < N; i++) {
        c += (u128)x[i] + m[i];
        x[i] = (u64)c;
        c >>= 64;
      }
    }
  };
  while (!is_one(u) && !is_one(v)) {
    while (!(u[0] & 1)) {
      shr1(u, 0);
      halve_mod(x1);
    }
    while (!(v[0] & 1)) {
      shr1(v, 0);
      halve_mod(x2);
    }
    if (geq(u, v)) {
      sub(u, v);
      sub_mod(x1, x2);
    } else {
      sub(v, u);
      sub_mod(x2, x1);
    }
  }
  HFe<P> r;
  const u64* x = is_one(u) ? x1 : x2;
  for (int i = 0; i < N; i++) r.v[i] = x[i];
  HFe<P> r2;
  for (int i = 0; i < N; i++) r2.v[i] = hr2<P>(i);
  return h_mul<P>(h_mul<P>(r, r2), r2);
}
template <class P>
inline HFe<P> h_from_mont(const HFe<P>& a) {
  HFe<P> o = h_zero<P>();
  o.v[0] = 1;
  return h_mul<P>(a, o);
}
template <class P>
inline HFe<P> h_to_mont(const HFe<P>& a) {
  HFe<P> r2;
  for (int i = 0; i < HFe<P>::N; i++) r2.v[i] = hr2<P>(i);
  return h_mul<P>(a, r2);
}

// ---- curve (a = 0), XYZZ ---------------------------------------------------------------------
template <class P>
struct HXYZZ {
  HFe<P> x, y, zz, zzz;
};
template <class P>
inline HXYZZ<P> hx_inf() {
  HXYZZ<P> r;
  r.x = r.y = r.zz = r.zzz = h_zero<P>();
  return r;
}
template <class P>
inline bool hx_is_inf(const HXYZZ<P>& p) {
  return h_is_zero<P>(p.zz);
}
template <class P>
inline HXYZZ<P> hx_dbl(const HXYZZ<P>& p) {
  if (hx_is_inf<P>(p)) return p;
  HFe<P> u = h_add<P>(p.y, p.y);
  HFe<P> v = h_sqr<P>(u);
  HFe<P> w = h_mul<P>(u, v);
  HFe<P> s = h_mul<P>(p.x, v);
  HFe<P> xx = h_sqr<P>(p.x);
  HFe<P> m = h_add<P>(h_add<P>(xx, xx), xx);
  HXYZZ<P> r;
  r.x = h_sub<P>(h_sub<P>(h_sqr<P>(m), s), s);
  r.y = h_sub<P>(h_mul<P>(m, h_sub<P>(s, r.x)), h_mul<P>(w, p.y));
  r.zz = h_mul<P>(v, p.zz);
  r.zzz = h_mul<P>(w, p.zzz);
  return r;
}
template <class P>
inline HXYZZ<P> hx_add(const HXYZZ<P>& a, const HXYZZ<P>& b) {
  if (hx_is_inf<P>(b)) return a;
  if (hx_is_inf<P>(a)) return b;
  HFe<P> u1 = h_mul<P>(a.x, b.zz), u2 = h_mul<P>(b.x, a.zz);
  HFe<P> s1 = h_mul<P>(a.y, b.zzz), s2 = h_mul<P>(b.y, a.zzz);
  HFe<P> p = h_sub<P>(u2, u1), r = h_sub<P>(s2, s1);
  if (h_is_zero<P>(p)) {
    if (h_is_zero<P>(r)) return hx_dbl<P>(a);
    return hx_inf<P>();
  }
  HFe<P> pp = h_sqr<P>(p), ppp = h_mul<P>(p, pp), q = h_mul<P>(u1, pp);
  HXYZZ<P> o;
  o.x = h_sub<P>(h_sub<P>(h_sub<P>(h_sqr<P>(r), ppp), q), q);
  o.y = h_sub<P>(h_mul<P>(r, h_sub<P>(q, o.x)), h_mul<P>(s1, ppp));
  o.zz = h_mul<P>(h_mul<P>(a.zz, b.zz), pp);
  o.zzz = h_mul<P>(h_mul<P>(a.zzz, b.zzz), ppp);
  return o;
}
// a += (x2, y2), an affine point that is not the identity (madd-2008-s; the host MSM's bucket accumulation)
template <class P>
inline void hx_madd(HXYZZ<P>& a, const HFe<P>& x2, const HFe<P>& y2) {
  if (hx_is_inf<P>(a)) {
    a.x = x2;
    a.y = y2;
    a.zz = a.zzz = h_one<P>();
    return;
  }
  HFe<P> p = h_sub<P>(h_mul<P>(x2, a.zz), a.x), r = h_sub<P>(h_mul<P>(y2, a.zzz), a.y);
  if (h_is_zero<P>(p)) {
    if (!h_is_zero<P>(r)) {
      a = hx_inf<P>();
      return;
    }
    HXYZZ<P> q;
    q.x = x2;
    q.y = y2;
    q.zz = q.zzz = h_one<P>();
    a = hx_dbl<P>(q);
    return;
  }
  HFe<P> pp = h_sqr<P>(p), ppp = h_mul<P>(p, pp), q = h_mul<P>(a.x, pp);
  HFe<P> x3 = h_sub<P>(h_sub<P>(h_sub<P>(h_sqr<P>(r), ppp), q), q);
  a.y = h_sub<P>(h_mul<P>(r, h_sub<P>(q, x3)), h_mul<P>(a.y, ppp));
  a.x = x3;
  a.zz = h_mul<P>(a.zz, pp);
  a.zzz = h_mul<P>(a.zzz, ppp);
}
// affine (Montgomery x|y) -> XYZZ; (0,0) or is_inf -> infinity
template <class P>
inline HXYZZ<P> hx_from_affine(const u64* xy, bool is_inf) {
  constexpr int N = HFe<P>::N;
  HXYZZ<P> r;
  memcpy(r.x.v, xy, 8 * N);
  memcpy(r.y.v, xy + N, 8 * N);
  if (is_inf || (h_is_zero<P>(r.x) && h_is_zero<P>(r.y))) return hx_inf<P>();
  r.zz = h_one<P>();
  r.zzz = h_one<P>();
  return r;
}
template <class P>
inline void hx_to_affine(const HXYZZ<P>& p, u64* xy, uint8_t* is_inf) {
  constexpr int N = HFe<P>::N;
  if (hx_is_inf<P>(p)) {
    memset(xy, 0, 16 * N);
    *is_inf = 1;
    return;
  }
  HFe<P> inv = h_inv<P>(h_mul<P>(p.zz, p.zzz));
  HFe<P> x = h_mul<P>(p.x, h_mul<P>(inv, p.zzz));
  HFe<P> y = h_mul<P>(p.y, h_mul<P>(inv, p.zz));
  memcpy(xy, x.v, 8 * N);
  memcpy(xy + N, y.v, 8 * N);
  *is_inf = 0;
}
// ---- scalar multiplication on the host (the O(#inputs) point algebra of the verifiers) -------------
inline int hx_scalar_bits(const u64 k[4]) {
  for (int i = 3; i >= 0; i--)
    if (k[i]) return 64 * i + 64 - __builtin_clzll(k[i]);
  return 0;
}
inline unsigned hx_nibble(const u64 k[4], int w) { return (unsigned)(k[w >> 4] >> ((w & 15) * 4)) & 15u; }

// d * P for d = 0..15
template <class P>
inline void hx_small_multiples(const HXYZZ<P>& p, HXYZZ<P> tab[16]) {
  tab[0] = hx_inf<P>();
  tab[1] = p;
  tab[2] = hx_dbl<P>(p);
  for (int d = 3; d < 16; d++) tab[d] = hx_add<P>(tab[d - 1], p);
}

// k * P, k = canonical 256-bit integer (4 u64): 4-bit windows from the top non-zero one
template <class P>
inline HXYZZ<P> hx_mul(const HXYZZ<P>& p, const u64 k[4]) {
  int bits = hx_scalar_bits(k);
  if (bits == 0 || hx_is_inf<P>(p)) return hx_inf<P>();
  if (bits == 1) return p;
  if (bits <= 8) {  // double-and-add: not worth a table
    HXYZZ<P> acc = p;
    for (int i = bits - 2; i >= 0; i--) {
      acc = hx_dbl<P>(acc);
      if ((k[0] >> i) & 1) acc = hx_add<P>(acc, p);
    }
    return acc;
  }
  HXYZZ<P> tab[16];
  hx_small_multiples<P>(p, tab);
  int w = (bits + 3) / 4 - 1;
  HXYZZ<P> acc = tab[hx_nibble(k, w)];
  for (w--; w >= 0; w--) {
    for (int t = 0; t < 4; t++) acc = hx_dbl<P>(acc);
    unsigned d = hx_nibble(k, w);
    if (d) acc = hx_add<P>(acc, tab[d]);
  }
  return acc;
}

// Fixed-base table: d * 16^w * P for w < 64, d = 1..15 -> k * P is at most 64 additions, no doublings.  Building one
// costs ~3.5 windowed multiplications; host_lincomb keeps a few for bases that recur with full-size scalars (the
// h' = xi_0 * h of an IPA opening is multiplied 2 log2(d+1) times, ipa_pc ext under src/ipa_pc_as/mod.rs:454).
template <class P>
struct HFixedBase {
  static constexpr int N = HFe<P>::N;
  u64 key[2 * N];
  std::vector<HXYZZ<P>> tab;  // [w * 15 + d - 1]; empty until the base has been seen often enough
  unsigned uses = 0;
  u64 stamp = 0;
  void build(const HXYZZ<P>& p) {
    tab.resize(64 * 15);
    HXYZZ<P> base = p;
    for (int w = 0; w < 64; w++) {
      tab[w * 15] = base;
      tab[w * 15 + 1] = hx_dbl<P>(base);
      for (int d = 3; d < 16; d++) tab[w * 15 + d - 1] = hx_add<P>(tab[w * 15 + d - 2], base);
      if (w < 63) base = hx_dbl<P>(tab[w * 15 + 7]);  // 16 * base = 2 * (8 * base)
    }
  }
  HXYZZ<P> mul(const u64 k[4]) const {
    HXYZZ<P> acc = hx_inf<P>();
    for (int w = 0; w < 64; w++) {
      unsigned d = hx_nibble(k, w);
      if (d) acc = hx_add<P>(acc, tab[w * 15 + d - 1]);
    }
    return acc;
  }
};

// sum_i k_i * P_i with shared doublings (Straus, 4-bit windows); k_i canonical.  `fixed` (may be null): per-point
// fixed-base tables that replace the point's windows.
template <class P>
inline HXYZZ<P> hx_lincomb(const HXYZZ<P>* pts, const u64 (*ks)[4], size_t n, const HFixedBase<P>* const* fixed) {
  HXYZZ<P> ones = hx_inf<P>();
  std::vector<size_t> idx;
  int top = -1;
  for (size_t i = 0; i < n; i++) {
    int bits = hx_scalar_bits(ks[i]);
    if (bits == 0 || hx_is_inf<P>(pts[i])) continue;
    if (bits == 1) {
      ones = hx_add<P>(ones, pts[i]);
    } else if (fixed && fixed[i]) {
      ones = hx_add<P>(ones, fixed[i]->mul(ks[i]));
    } else {
      idx.push_back(i);
      top = std::max(top, (bits + 3) / 4 - 1);
    }
  }
  if (idx.empty()) return ones;
  if (idx.size() == 1) return hx_add<P>(ones, hx_mul<P>(pts[idx[0]], ks[idx[0]]));
  std::vector<HXYZZ<P>> tabs(idx.size() * 16);
  for (size_t j = 0; j < idx.size(); j++) hx_small_multiples<P>(pts[idx[j]], &tabs[j * 16]);
  HXYZZ<P> acc = hx_inf<P>();
  for (int w = top; w >= 0; w--) {
    for (int t = 0; t < 4; t++) acc = hx_dbl<P>(acc);
    for (size_t j = 0; j < idx.size(); j++) {
      unsigned d = hx_nibble(ks[idx[j]], w);
      if (d) acc = hx_add<P>(acc, tabs[j * 16 + d]);
    }
  }
  return hx_add<P>(acc, ones);
}
// device XYZZ record (4*L u32, little-endian) -> host
template <class P>
inline HXYZZ<P> hx_from_device(const u32* rec) {
  constexpr int N = HFe<P>::N;
  HXYZZ<P> r;
  memcpy(r.x.v, rec, 8 * N);
  memcpy(r.y.v, rec + P::L, 8 * N);
  memcpy(r.zz.v, rec + 2 * P::L, 8 * N);
  memcpy(r.zzz.v, rec + 3 * P::L, 8 * N);
  return r;
}

}  // namespace host

// generator coordinates in Montgomery form (computed from canonical constants at first use)
template <class Fq>
std::vector<u32> generator_mont(int curve) {
  using H = host::HFe<Fq>;
  H gx = host::h_zero<Fq>(), gy = host::h_zero<Fq>();
  if (curve == 0 /* AMSM_PALLAS */) {
    // (-1, 2)
    H one = host::h_zero<Fq>();
    one.v[0] = 1;
    H two = host::h_zero<Fq>();
    two.v[0] = 2;
    gx = host::h_neg<Fq>(host::h_to_mont<Fq>(one));
    gy = host::h_to_mont<Fq>(two);
  } else {
    static const u64 X[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull,
                             0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};
    static const u64 Y[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull,
                             0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};
    for (int i = 0; i < H::N && i < 6; i++) {
      gx.v[i] = X[i];
      gy.v[i] = Y[i];
    }
    gx = host::h_to_mont<Fq>(gx);
    gy = host::h_to_mont<Fq>(gy);
  }
  std::vector<u32> g(2 * Fq::L);
  memcpy(g.data(), gx.v, 4 * Fq::L);
  memcpy(g.data() + Fq::L, gy.v, 4 * Fq::L);
  return g;
}

}  // namespace amsm
