// Host side of the GLV split for scalar multiplications by ONE scalar shared by many points (the commitment-key folds of the
// IPA opening, `key_l += key_r * xi` of ark_poly_commit::ipa_pc ext under src/ipa_pc_as/mod.rs:454 -> k_points_fold).
//
// Both curves (y^2 = x^3 + b: j-invariant 0) carry the endomorphism phi(x, y) = (beta x, y) = [lambda](x, y) with beta a
// primitive cube root of unity in Fq and lambda one in Fr.  A scalar k is written k = k1 + k2 lambda (mod r) with |k1|, |k2| ~
// sqrt(r), so k P = k1 P + k2 phi(P) takes HALF the doublings of a plain double-and-add (Gallant-Lambert-Vanstone, CRYPTO 2001).
//
// Nothing here is a memorised constant: lambda and beta are computed as g^((m-1)/3), paired by checking [lambda] G = (beta Gx, Gy)
// on the curve's generator with the host group law, the lattice basis comes from the extended Euclidean algorithm on (r, lambda)
// (Guide to Elliptic Curve Cryptography, Alg. 3.74), and every decomposition is verified (k1 + k2 lambda = k in Fr) before it
// is used -- a failed check makes the caller fall back to the plain double-and-add, never to a wrong result.
// Product code (the reference reaches the same group elements through ark-ec's plain `mul`; only canonical affine results are
// compared).
#pragma once
#include "host_field.h"

namespace amsm {
namespace host {

// ---- a small signed big integer (sign + 640-bit magnitude): set-up code, not a hot path ----
struct Big {
  static constexpr int N = 20;
  u32 w[N];
  bool neg;
};
inline Big big_zero() {
  Big r;
  memset(r.w, 0, sizeof(r.w));
  r.neg = false;
  return r;
}
inline Big big_from_u64(const u64* v, int n) {
  Big r = big_zero();
  for (int i = 0; i < n && 2 * i + 1 < Big::N; i++) {
    r.w[2 * i] = (u32)v[i];
    r.w[2 * i + 1] = (u32)(v[i] >> 32);
  }
  return r;
}
inline Big big_small(u32 v) {
  Big r = big_zero();
  r.w[0] = v;
  return r;
}
inline bool big_is_zero(const Big& a) {
  for (int i = 0; i < Big::N; i++)
    if (a.w[i]) return false;
  return true;
}
inline int big_cmp_mag(const Big& a, const Big& b) {
  for (int i = Big::N - 1; i >= 0; i--)
    if (a.w[i] != b.w[i]) return a.w[i] > b.w[i] ? 1 : -1;
  return 0;
}
inline int big_bits(const Big& a) {
  for (int i = Big::N - 1; i >= 0; i--)
    if (a.w[i]) return 32 * i + 32 - __builtin_clz(a.w[i]);
  return 0;
}
inline void big_add_mag(Big& r, const Big& a, const Big& b) {
  u64 c = 0;
  for (int i = 0; i < Big::N; i++) {
    c += (u64)a.w[i] + b.w[i];
    r.w[i] = (u32)c;
    c >>= 32;
  }
}
inline void big_sub_mag(Big& r, const Big& a, const Big& b) {  // |a| >= |b|
  int64_t c = 0;
  for (int i = 0; i < Big::N; i++) {
    c += (int64_t)a.w[i] - b.w[i];
    r.w[i] = (u32)c;
    c >>= 32;
  }
}
inline Big big_add(const Big& a, const Big& b) {
  Big r = big_zero();
  if (a.neg == b.neg) {
    big_add_mag(r, a, b);
    r.neg = a.neg;
  } else if (big_cmp_mag(a, b) >= 0) {
    big_sub_mag(r, a, b);
    r.neg = a.neg;
  } else {
    big_sub_mag(r, b, a);
    r.neg = b.neg;
  }
  if (big_is_zero(r)) r.neg = false;
  return r;
}
inline Big big_negate(const Big& a) {
  Big r = a;
  r.neg = !a.neg && !big_is_zero(a);
  return r;
}
inline Big big_sub(const Big& a, const Big& b) { return big_add(a, big_negate(b)); }
inline Big big_mul(const Big& a, const Big& b) {  // truncated to 640 bits (callers stay far below)
  Big r = big_zero();
  for (int i = 0; i < Big::N; i++) {
    if (!a.w[i]) continue;
    u64 c = 0;
    for (int j = 0; i + j < Big::N; j++) {
      c += (u64)a.w[i] * b.w[j] + r.w[i + j];
      r.w[i + j] = (u32)c;
      c >>= 32;
    }
  }
  r.neg = (a.neg != b.neg) && !big_is_zero(r);
  return r;
}
inline Big big_shl1(const Big& a) {
  Big r = a;
  for (int i = Big::N - 1; i > 0; i--) r.w[i] = (a.w[i] << 1) | (a.w[i - 1] >> 31);
  r.w[0] = a.w[0] << 1;
  return r;
}
inline Big big_shr(const Big& a, int s) {
  Big r = big_zero();
  r.neg = a.neg;
  const int ws = s >> 5, bs = s & 31;
  for (int i = 0; i + ws < Big::N; i++) {
    u64 lo = a.w[i + ws], hi = i + ws + 1 < Big::N ? a.w[i + ws + 1] : 0;
    r.w[i] = bs ? (u32)((lo >> bs) | (hi << (32 - bs))) : (u32)lo;
  }
  if (big_is_zero(r)) r.neg = false;
  return r;
}
// magnitudes: q = floor(|a| / |b|), rem = |a| - q |b|   (shift-subtract; b != 0)
inline void big_divmod_mag(const Big& a, const Big& b, Big& q, Big& rem) {
  q = big_zero();
  rem = big_zero();
  for (int i = big_bits(a) - 1; i >= 0; i--) {
    rem = big_shl1(rem);
    rem.w[0] |= (a.w[i >> 5] >> (i & 31)) & 1u;
    if (big_cmp_mag(rem, b) >= 0) {
      big_sub_mag(rem, rem, b);
      q.w[i >> 5] |= 1u << (i & 31);
    }
  }
}
inline Big big_div_round(const Big& a, const Big& b) {  // round(a / b), b > 0, sign of a kept
  Big q, rem;
  big_divmod_mag(a, b, q, rem);
  Big twice = big_shl1(rem);
  twice.neg = false;
  Big bb = b;
  bb.neg = false;
  if (big_cmp_mag(twice, bb) >= 0) {
    Big one = big_small(1);
    big_add_mag(q, q, one);
  }
  q.neg = a.neg && !big_is_zero(q);
  return q;
}

template <class P>
inline Big big_modulus() {
  u64 m[HFe<P>::N];
  for (int i = 0; i < HFe<P>::N; i++) m[i] = hmod<P>(i);
  return big_from_u64(m, HFe<P>::N);
}
template <class P>
inline HFe<P> h_from_big_mag(const Big& a) {  // |a| < modulus -> Montgomery form
  HFe<P> r;
  for (int i = 0; i < HFe<P>::N; i++) r.v[i] = (u64)a.w[2 * i] | ((u64)a.w[2 * i + 1] << 32);
  return h_to_mont<P>(r);
}
template <class P>
inline HFe<P> h_from_big(const Big& a) {  // signed, |a| < modulus
  HFe<P> m = h_from_big_mag<P>(a);
  return a.neg ? h_neg<P>(m) : m;
}
template <class P>
inline HFe<P> h_pow_big(const HFe<P>& a, const Big& e) {
  HFe<P> r = h_one<P>();
  for (int i = big_bits(e) - 1; i >= 0; i--) {
    r = h_sqr<P>(r);
    if ((e.w[i >> 5] >> (i & 31)) & 1u) r = h_mul<P>(r, a);
  }
  return r;
}
// a primitive cube root of unity of the field (3 | modulus - 1 on both fields of both curves); zero if there is none
template <class P>
inline HFe<P> h_cube_root_of_unity() {
  Big m1 = big_sub(big_modulus<P>(), big_small(1)), q, rem;
  big_divmod_mag(m1, big_small(3), q, rem);
  if (!big_is_zero(rem)) return h_zero<P>();
  HFe<P> one = h_one<P>();
  for (u32 g = 2; g < 64; g++) {
    HFe<P> w = h_pow_big<P>(h_from_big_mag<P>(big_small(g)), q);
    if (!h_eq<P>(w, one)) return w;
  }
  return h_zero<P>();
}

// What a fold by x needs: the non-adjacent forms of k1 and k2 (signs folded into the digit masks) and beta.
struct GlvDigits {
  u32 pos1[5], neg1[5], pos2[5], neg2[5];  // up to 160 digits each (|k_i| < 2^131 in practice)
  u32 nd;                                  // digit positions to run (max over the two)
};

template <class Fq, class Fr>
struct Glv {
  bool ok = false;
  HFe<Fq> beta;    // Montgomery form (C-ABI radix)
  HFe<Fr> lambda;  // Montgomery form
  Big a1, b1, a2, b2, r;
  Big g1, g2;  // round(2^384 b2 / r), round(2^384 (-b1) / r): `decompose` multiplies and shifts instead of dividing

  // gen_xy: the curve's generator, affine, Montgomery form (N words x, N words y)
  void setup(const u64* gen_xy) {
    ok = false;
    constexpr int NQ = HFe<Fq>::N;
    r = big_modulus<Fr>();
    lambda = h_cube_root_of_unity<Fr>();
    beta = h_cube_root_of_unity<Fq>();
    if (h_is_zero<Fr>(lambda) || h_is_zero<Fq>(beta)) return;
    // pair them: [lambda] G == (beta Gx, Gy); otherwise the other root beta^2
    HFe<Fr> lc = h_from_mont<Fr>(lambda);
    u64 k[4] = {0, 0, 0, 0};
    for (int i = 0; i < HFe<Fr>::N && i < 4; i++) k[i] = lc.v[i];
    HXYZZ<Fq> g = hx_from_affine<Fq>(gen_xy, false);
    u64 out[2 * NQ];
    uint8_t inf = 0;
    hx_to_affine<Fq>(hx_mul<Fq>(g, k), out, &inf);
    if (inf) return;
    HFe<Fq> gx, ox, oy, gy;
    memcpy(gx.v, gen_xy, 8 * NQ);
    memcpy(gy.v, gen_xy + NQ, 8 * NQ);
    memcpy(ox.v, out, 8 * NQ);
    memcpy(oy.v, out + NQ, 8 * NQ);
    if (!h_eq<Fq>(oy, gy)) return;
    if (!h_eq<Fq>(h_mul<Fq>(beta, gx), ox)) {
      beta = h_sqr<Fq>(beta);
      if (!h_eq<Fq>(h_mul<Fq>(beta, gx), ox)) return;
    }
    // lattice basis of {(a, b): a + b lambda = 0 mod r}: extended Euclid on (r, lambda), remainders r_i = s_i r + t_i lambda
    Big lam = big_zero();
    for (int i = 0; i < HFe<Fr>::N; i++) {
      lam.w[2 * i] = (u32)lc.v[i];
      lam.w[2 * i + 1] = (u32)(lc.v[i] >> 32);
    }
    Big r0 = r, r1 = lam, t0 = big_zero(), t1 = big_small(1);
    const int half = (big_bits(r) + 1) / 2;
    // run until r1 < sqrt(r) for the first time: then (r0, t0) is index l, (r1, t1) index l + 1
    while (big_bits(r1) > half) {
      Big q, rem;
      big_divmod_mag(r0, r1, q, rem);
      Big t2 = big_sub(t0, big_mul(q, t1));
      r0 = r1;
      t0 = t1;
      r1 = rem;
      t1 = t2;
      if (big_is_zero(r1)) return;
    }
    Big q, r2;
    big_divmod_mag(r0, r1, q, r2);
    Big t2 = big_sub(t0, big_mul(q, t1));
    a1 = r1;
    b1 = big_negate(t1);
    // the shorter of (r_l, -t_l) and (r_{l+2}, -t_{l+2})
    auto norm = [](const Big& a, const Big& b) { return big_add(big_mul(a, a), big_mul(b, b)); };
    if (big_cmp_mag(norm(r0, t0), norm(r2, t2)) <= 0) {
      a2 = r0;
      b2 = big_negate(t0);
    } else {
      a2 = r2;
      b2 = big_negate(t2);
    }
    // orientation: a1 b2 - a2 b1 = +r (the rounding formulas of `decompose` assume it); -r -> take -v2
    {
      Big det = big_sub(big_mul(a1, b2), big_mul(a2, b1));
      if (big_cmp_mag(det, r) != 0) return;
      if (det.neg) {
        a2 = big_negate(a2);
        b2 = big_negate(b2);
      }
    }
    // both vectors must lie in the lattice, and be short enough for 160-digit masks
    for (int v = 0; v < 2; v++) {
      const Big& a = v ? a2 : a1;
      const Big& b = v ? b2 : b1;
      if (big_bits(a) > 136 || big_bits(b) > 136) return;
      HFe<Fr> s = h_add<Fr>(h_from_big<Fr>(a), h_mul<Fr>(h_from_big<Fr>(b), lambda));
      if (!h_is_zero<Fr>(s)) return;
    }
    auto shl384 = [](const Big& a) {
      Big t = big_zero();
      for (int i = 0; i + 12 < Big::N; i++) t.w[i + 12] = a.w[i];
      t.neg = a.neg;
      return t;
    };
    g1 = big_div_round(shl384(b2), r);
    g2 = big_div_round(shl384(big_negate(b1)), r);
    ok = true;
  }

  // k canonical (4 u64, < r).  false: no decomposition (caller uses the plain form)
  bool decompose(const u64 k[4], Big& k1, Big& k2) const {
    if (!ok) return false;
    Big kk = big_from_u64(k, 4);
    // c_i = round(k g_i / 2^384): within one of round(b2 k / r), round(-b1 k / r) -- the halves stay below 2^131 either way and the
    // identity k1 + k2 lambda = k is checked below (a bit-serial division here cost ~10 us per scalar: the host combinations of a
    // scheme split dozens of them per call)
    auto mul_shift = [](const Big& g, const Big& k) {
      Big p = big_mul(g, k);
      const bool neg = p.neg;
      p.neg = false;
      Big half = big_zero();
      half.w[11] = 0x80000000u;  // 2^383
      big_add_mag(p, p, half);
      Big q = big_zero();
      for (int i = 0; i + 12 < Big::N; i++) q.w[i] = p.w[i + 12];
      q.neg = neg && !big_is_zero(q);
      return q;
    };
    Big c1 = mul_shift(g1, kk);
    Big c2 = mul_shift(g2, kk);
    k1 = big_sub(big_sub(kk, big_mul(c1, a1)), big_mul(c2, a2));
    k2 = big_sub(big_negate(big_mul(c1, b1)), big_mul(c2, b2));
    if (big_bits(k1) > 150 || big_bits(k2) > 150) return false;
    HFe<Fr> chk = h_add<Fr>(h_from_big<Fr>(k1), h_mul<Fr>(h_from_big<Fr>(k2), lambda));
    HFe<Fr> want;
    for (int i = 0; i < HFe<Fr>::N; i++) want.v[i] = i < 4 ? k[i] : 0;
    return h_eq<Fr>(chk, h_to_mont<Fr>(want));
  }
};

// non-adjacent form of |k| into (pos, neg) masks, swapped when k is negative; returns the number of digit positions
inline u32 big_naf(const Big& k, u32 pos[5], u32 neg[5]) {
  Big m = k;
  m.neg = false;
  u32 nd = 0;
  for (int i = 0; i < 5; i++) pos[i] = neg[i] = 0;
  u32* plus = k.neg ? neg : pos;
  u32* minus = k.neg ? pos : neg;
  const Big one = big_small(1);
  for (u32 d = 0; d < 160 && !big_is_zero(m); d++) {
    if (m.w[0] & 1u) {
      if ((m.w[0] & 3u) == 1u) {
        plus[d >> 5] |= 1u << (d & 31);
        m.w[0] &= ~1u;
      } else {
        minus[d >> 5] |= 1u << (d & 31);
        big_add_mag(m, m, one);
      }
      nd = d + 1;
    }
    m = big_shr(m, 1);
  }
  return big_is_zero(m) ? nd : 0xffffffffu;
}

template <class Fq, class Fr>
inline bool glv_digits(const Glv<Fq, Fr>& g, const u64 k[4], GlvDigits& out) {
  Big k1, k2;
  if (!g.decompose(k, k1, k2)) return false;
  u32 n1 = big_naf(k1, out.pos1, out.neg1), n2 = big_naf(k2, out.pos2, out.neg2);
  if (n1 == 0xffffffffu || n2 == 0xffffffffu) return false;
  out.nd = std::max(n1, n2);
  return true;
}


// ---- linear combinations on the host (the schemes' O(#inputs) group algebra): GLV halves + signed 4-bit windows ----------------------
// One term of a combination: |k| (up to 4 words) times p, the sign already folded into p.
template <class P>
struct HTerm {
  HXYZZ<P> p;
  u64 k[4];
};
template <class P>
inline HXYZZ<P> hx_neg(const HXYZZ<P>& p) {
  HXYZZ<P> r = p;
  r.y = h_neg<P>(p.y);
  return r;
}
// k * p (k canonical, p not the identity) as terms: scalars of more than 136 bits are split into the two ~128-bit GLV halves
// (k = k1 + k2 lambda, phi(x, y) = (beta x, y)): the Straus loop below then runs 128 doublings instead of 255
template <class Fq, class Fr>
inline void hx_push_terms(const Glv<Fq, Fr>* glv, const HXYZZ<Fq>& p, const u64 k[4], std::vector<HTerm<Fq>>& out) {
  Big k1, k2;
  if (glv && hx_scalar_bits(k) > 136 && glv->decompose(k, k1, k2) && big_bits(k1) <= 192 && big_bits(k2) <= 192) {
    for (int h = 0; h < 2; h++) {
      const Big& kk = h ? k2 : k1;
      if (big_is_zero(kk)) continue;
      HTerm<Fq> t;
      t.p = p;
      if (h) t.p.x = h_mul<Fq>(p.x, glv->beta);
      if (kk.neg) t.p = hx_neg<Fq>(t.p);
      for (int i = 0; i < 4; i++) t.k[i] = (u64)kk.w[2 * i] | ((u64)kk.w[2 * i + 1] << 32);
      out.push_back(t);
    }
    return;
  }
  HTerm<Fq> t;
  t.p = p;
  memcpy(t.k, k, 32);
  out.push_back(t);
}
// sum of the terms: shared doublings, signed 4-bit digits in [-8, 8] (a table of 8 multiples per term: 1 doubling + 6 additions,
// mixed while the point is affine), one addition per non-zero digit
template <class P>
inline HXYZZ<P> hx_straus_signed(const HTerm<P>* terms, size_t n) {
  if (n == 0) return hx_inf<P>();
  std::vector<HXYZZ<P>> tab(n * 8);
  std::vector<int8_t> dig(n * 66, 0);
  int top = -1;
  for (size_t j = 0; j < n; j++) {
    const HXYZZ<P>& p = terms[j].p;
    HXYZZ<P>* t = &tab[j * 8];
    t[0] = p;
    t[1] = hx_dbl<P>(p);
    const bool affine = h_eq<P>(p.zz, h_one<P>()) && h_eq<P>(p.zzz, h_one<P>());
    for (int d = 2; d < 8; d++) {
      t[d] = t[d - 1];
      if (affine) hx_madd<P>(t[d], p.x, p.y);
      else t[d] = hx_add<P>(t[d], p);
    }
    int carry = 0;
    int8_t* dg = &dig[j * 66];
    for (int w = 0; w < 65; w++) {
      int v = (w < 64 ? (int)hx_nibble(terms[j].k, w) : 0) + carry;
      carry = v > 8 ? 1 : 0;
      if (carry) v -= 16;
      dg[w] = (int8_t)v;
      if (v) top = std::max(top, w);
    }
  }
  HXYZZ<P> acc = hx_inf<P>();
  for (int w = top; w >= 0; w--) {
    if (!hx_is_inf<P>(acc))
      for (int t = 0; t < 4; t++) acc = hx_dbl<P>(acc);
    for (size_t j = 0; j < n; j++) {
      const int d = dig[j * 66 + w];
      if (d > 0) acc = hx_add<P>(acc, tab[j * 8 + d - 1]);
      else if (d < 0) acc = hx_add<P>(acc, hx_neg<P>(tab[j * 8 - d - 1]));
    }
  }
  return acc;
}

}  // namespace host
}  // namespace amsm
