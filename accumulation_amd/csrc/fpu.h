// Unsaturated-limb Montgomery arithmetic for the base fields of the MSM hot loop (gfx950): Pallas Fq on 9 x 29 bits,
// BLS12-381 Fq on 14 x 28 bits.
//
// gfx950 has no carry-in on v_mad_u64_u32 and a VALU carry write costs wait states, so the saturated 8 x 32-bit
// schedule (fp_mul_gfx950.h) pays one v_addc per product.  Here an element is L limbs of B bits (Pallas Fq: 9 x 29,
// R' = 2^261): every column sum of a product fits a 64-bit accumulator, so a multiplication is one v_mad_u64_u32
// per limb product and NO carry handling (measured 17-21 % faster, tools/fp29_proto.hip), squarings need 45
// instead of 81 products, and additions / subtractions are carry-free limb-wise adds followed by one
// carry-propagation pass.
//
// Representation rules (every function states what it needs and gives):
//   * "tight"  : limbs 0..L-2 < 2^B, value < 2^(B*L).  Everything held in registers between operations is tight.
//   * "lazy"   : limbs < 2^(B+1), only allowed as ONE operand of a multiplication.
//   * values are only bounded, not reduced: a multiplication gives  out < p + A*B / 2^(B*L)  and the group-law
//     formulas in ec.h carry the bound of every intermediate in comments (cap = 2^261 ~ 128 p for Pallas).
//   * memory holds canonical values (< p), packed into W 32-bit words, in the INTERNAL Montgomery radix R'.  The
//     C ABI's radix is R = 2^(32 W); fe_import / fe_export convert (one multiplication by a constant) at the
//     edges (key load / read, the final fold of an MSM, amsm_points_fold), see DESIGN.md.
//
// Included from fp.h (after Fe<>), not on its own.
#pragma once

namespace amsm {

struct PallasFq;  // the saturated pack (fp.h) this field shadows

struct PallasFqU {
  using Sat = PallasFq;
  static constexpr int L = 9;  // register limbs
  static constexpr int W = 8;  // 32-bit words in memory
  static constexpr int B = 29;
  static constexpr bool UNSAT = true;
  static constexpr u32 NINV = 0x1fffffffu;  // -p^-1 mod 2^B
  static constexpr bool CHAIN = true;       // u_opaque after every column (measured, tools/fp_bench.hip: mixed addition -1 %)
  // p, R' mod p, R'^2 / R mod p (ABI -> internal), R mod p (internal -> ABI); radix 2^29, checked in tests/test_oracle.py
  AMSM_TABLE(mod, 9, 0x00000001u, 0x09698768u, 0x133e46e6u, 0x0d31f812u, 0x00000224u, 0x00000000u, 0x00000000u, 0x00000000u, 0x00400000u)
  AMSM_TABLE(one, 9, 0x1fffff81u, 0x14a5d367u, 0x141ad3c0u, 0x1435eec5u, 0x1ffeefefu, 0x1fffffffu, 0x1fffffffu, 0x1fffffffu, 0x003fffffu)
  AMSM_TABLE(k_import, 9, 0x1ffff001u, 0x10f30767u, 0x0ecfe231u, 0x0db0ce73u, 0x1fddbb8bu, 0x1fffffffu, 0x1fffffffu, 0x1fffffffu, 0x003fffffu)
  AMSM_TABLE(k_export, 9, 0x1ffffffdu, 0x03c369c7u, 0x06452b4du, 0x186a17c8u, 0x1ffff992u, 0x1fffffffu, 0x1fffffffu, 0x1fffffffu, 0x003fffffu)
};

struct Bls12381Fq;
struct Bls12381FqU {  // 14 x 28 bits, R' = 2^392 ~ 2520 p: measured against the saturated schedule in tools/fp_bench.hip
  using Sat = Bls12381Fq;
  static constexpr int L = 14;
  static constexpr int W = 12;
  static constexpr int B = 28;
  static constexpr bool UNSAT = true;
  static constexpr u32 NINV = 0x0ffcfffdu;
  static constexpr bool CHAIN = false;  // measured: the opaque carry costs the 14-limb mixed addition 4 % (register pressure)
  AMSM_TABLE(mod, 14, 0x0fffaaabu, 0x0fefffffu, 0x03ffffb9u, 0x0fffeb15u, 0x06241eabu, 0x0a0f6b0fu, 0x0f6730d2u, 0x0f38512bu, 0x04774b84u, 0x04bacd76u, 0x0ba7b643u, 0x0e69a4b1u, 0x01ea397fu, 0x0001a011u)
  AMSM_TABLE(one, 14, 0x0347fcb8u, 0x0d800000u, 0x0002b119u, 0x00cde6d2u, 0x0c7212e0u, 0x083a2090u, 0x0037669fu, 0x0da0f73eu, 0x09b09b42u, 0x01297bb0u, 0x0515d98fu, 0x0012ca7cu, 0x0659fcfau, 0x0000577au)
  AMSM_TABLE(k_import, 14, 0x080e6299u, 0x03500034u, 0x0eb12856u, 0x0deb2699u, 0x0c988670u, 0x04ef6697u, 0x070983e8u, 0x0a4e6fe9u, 0x03e8a053u, 0x0ecf271eu, 0x0c20d323u, 0x06eb6385u, 0x047f1286u, 0x000156dau)
  AMSM_TABLE(k_export, 14, 0x0002fffdu, 0x00900000u, 0x0c000276u, 0x0000bc40u, 0x08baebf4u, 0x05753c75u, 0x055f4898u, 0x07052574u, 0x07ce5853u, 0x056ec6d7u, 0x071a97a2u, 0x0e4935c0u, 0x0ec3fa80u, 0x00015f65u)
};

template <class P>
AMSM_HD constexpr u32 u_mask() {
  return (1u << P::B) - 1u;
}

// ---- the reduction step's two tricks (round 3; the A/B against round 2's schedule is in DESIGN.md 4.1) ----
// The compiler's own schedule of a 9 x 29 product (round 2) kept one 64-bit accumulator chain PER COLUMN, started from zero,
// and merged it with the carry of the column below by a 64-bit add (v_lshl_add_u64: half rate); the reduction step added
// m * p_0 = m as a zero-extended 64-bit add (v_mov + v_lshl_add_u64) and m * 2^22 as a 64-bit shift + 64-bit add: 99 non-MAD
// instructions per multiplication, 63 of them half rate, next to 117 MADs.  Round 3:
//   chain:  an opaque (empty-asm) use of the accumulator after every column shift, so that the next column's MADs chain
//                from the carry instead of from zero (no merge add);
//   const:  p_0 = 1 (Pallas: p = 1 mod 2^29): (acc + m) >> B == (acc + 2^B - 1) >> B, one 64-bit add of a constant and no
//                m in the low column; a power-of-two limb of p (2^22 at limb 8) as a MAD by a constant held in an SGPR the
//                compiler cannot see through (one half-rate instruction instead of shift + add).
// Measured (MI355X, cycles per operation per SIMD at 2 / 4 waves per SIMD): Pallas multiplication 899 / 862 -> 877 / 822, squaring
// 747 / 709 -> 715 / 685, mixed addition 8 640 / 8 288 -> 8 208 / 7 963 (-5 % / -4 %); 126 MADs + 81 full-rate + 40 half-rate
// instructions per multiplication instead of 117 + 89 + 59.  BLS12-381 (p_0 != 1, no power-of-two limb) is unchanged.  A strictly
// linear chain (an opaque use after EVERY product) removes the remaining merge adds but spills: 12 300.
template <class P>
AMSM_DEV void u_opaque(u64& acc) {
  if constexpr (P::CHAIN) asm("" : "+v"(acc));
}
template <u32 V>
AMSM_DEV u32 u_sgpr_const() {  // V in an SGPR, opaque to the optimiser (hoisted out of loops like any pure expression)
  u32 r;
  asm("s_mov_b32 %0, %1" : "=s"(r) : "i"(V));
  return r;
}
AMSM_HD constexpr bool u_is_pow2(u32 v) { return v != 0 && (v & (v - 1)) == 0; }

// acc += m_i * p_j for the limbs j >= 1 of p that are not zero (column k of a product: i + j = k)
template <class P, int I, int K>
AMSM_DEV void u_red_term(u64& acc, const u32* m) {
  constexpr int J = K - I;
  if constexpr (I < K && J >= 1 && J < P::L) {
    if constexpr (P::mod(J) != 0) {
      if constexpr (u_is_pow2(P::mod(J))) acc += (u64)m[I] * u_sgpr_const<P::mod(J)>();
      else acc += (u64)m[I] * P::mod(J);
    }
  }
}
template <class P, int K, int I = 0>
AMSM_DEV void u_red_terms(u64& acc, const u32* m) {
  if constexpr (I < P::L) {
    u_red_term<P, I, K>(acc, m);
    u_red_terms<P, K, I + 1>(acc, m);
  }
}
// low column k < L: m_k = -acc * p^-1 mod 2^B, acc = (acc + m_k p_0) >> B
template <class P>
AMSM_DEV u32 u_red_low(u64& acc) {
  constexpr u32 M = u_mask<P>();
  const u32 lo = (u32)acc & M;
  u32 mk;
  if constexpr (P::NINV == M) {  // p_0 = 1
    mk = (0u - lo) & M;
    acc = (acc + (u64)M) >> P::B;  // == (acc + mk) >> B: lo + mk is 0 or 2^B
  } else {
    mk = (lo * P::NINV) & M;
    acc += (u64)mk * P::mod(0);
    acc >>= P::B;
  }
  u_opaque<P>(acc);
  return mk;
}
// high column k >= L: emit limb k - L
template <class P>
AMSM_DEV u32 u_red_high(u64& acc, bool top) {
  const u32 r = top ? (u32)acc : ((u32)acc & u_mask<P>());
  acc >>= P::B;
  if (!top) u_opaque<P>(acc);
  return r;
}

// K*p as tight limbs (compile-time)
template <class P, u32 K>
struct UKp {
  u32 v[P::L];
  AMSM_HD constexpr UKp() : v{} {
    u64 carry = 0;
    for (int j = 0; j < P::L; j++) {
      u64 t = (u64)K * P::mod(j) + carry;
      if (j < P::L - 1) {
        v[j] = (u32)t & u_mask<P>();
        carry = t >> P::B;
      } else {
        v[j] = (u32)t;
      }
    }
  }
};

// K*p in a redundant form whose limbs 0..L-2 are all >= S*(2^B - 1): a + UKpBp - (sum of S tight values) never
// borrows below the top limb (compile-time)
template <class P, u32 K, u32 S>
struct UKpBp {
  u32 v[P::L];
  AMSM_HD constexpr UKpBp() : v{} {
    UKp<P, K> t;
    for (int j = 0; j < P::L; j++) {
      if (j == 0) v[j] = t.v[j] + (S << P::B);
      else if (j < P::L - 1) v[j] = t.v[j] + (S << P::B) - S;
      else v[j] = t.v[j] - S;
    }
  }
};

// carry propagation: limbs < 2^32 (true value in [0, 2^(B*L))) -> tight.  The top limb may have wrapped mod 2^32
// during limb-wise arithmetic; it is exact again after the carries arrive.
template <class P>
AMSM_DEV void u_carry(Fe<P>& a) {
#pragma unroll
  for (int i = 0; i < P::L - 1; i++) {
    a.v[i + 1] += a.v[i] >> P::B;
    a.v[i] &= u_mask<P>();
  }
}

// Montgomery product a*b / 2^(B*L) mod p.  Needs: limbs of a < 2^(B+1), limbs of b < 2^B (or the reverse).
// Gives: tight, value < p + a*b / 2^(B*L).
// With ADD: returns a*b / 2^(B*L) + add, the addend's limbs (any u32, e.g. the unnormalised K*p - x) going straight
// into the upper columns of the product, so "product minus value" costs L 64-bit adds instead of a subtraction and
// a carry pass, and the result is tight.
template <class P, int K, class F>
AMSM_DEV void u_column(u64& acc, u32* m, Fe<P>& r, const Fe<P>* add, bool has_add, F&& products) {
  products(acc);
  u_red_terms<P, K>(acc, m);
  if constexpr (K < P::L) {
    m[K] = u_red_low<P>(acc);
  } else {
    if (has_add) acc += add->v[K - P::L];
    r.v[K - P::L] = u_red_high<P>(acc, K == 2 * P::L - 1);
  }
}
template <class P, int K, class F>
AMSM_DEV void u_columns(u64& acc, u32* m, Fe<P>& r, const Fe<P>* add, bool has_add, F&& products) {
  if constexpr (K < 2 * P::L) {
    u_column<P, K>(acc, m, r, add, has_add, [&](u64& a) { products(a, K); });
    u_columns<P, K + 1>(acc, m, r, add, has_add, products);
  }
}

template <class P, bool ADD = false>
AMSM_DEV Fe<P> u_mul(const Fe<P>& a, const Fe<P>& b, const Fe<P>* add = nullptr) {
  constexpr int L = P::L;
  u64 acc = 0;
  u32 m[L];
  Fe<P> r;
  u_columns<P, 0>(acc, m, r, add, ADD, [&](u64& ac, int k) {
#pragma unroll
    for (int i = 0; i < L; i++) {
      int j = k - i;
      if (j >= 0 && j < L) {
        ac += (u64)a.v[i] * b.v[j];
      }
    }
  });
  return r;
}

// (a*b + c*d) / 2^(B*L) mod p with ONE reduction: both products are accumulated column by column before the
// Montgomery step (saves a whole reduction, ~40 % of a multiplication, wherever the group law subtracts two products).
// Needs: limbs of a, c < 2^(B+1), limbs of b, d < 2^B.  Gives: tight, value < p + (a*b + c*d) / 2^(B*L).
template <class P>
AMSM_DEV Fe<P> u_mul_add_mul(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c, const Fe<P>& d) {
  constexpr int L = P::L;
  u64 acc = 0;
  u32 m[L];
  Fe<P> r;
  u_columns<P, 0>(acc, m, r, nullptr, false, [&](u64& ac, int k) {
#pragma unroll
    for (int i = 0; i < L; i++) {
      int j = k - i;
      if (j >= 0 && j < L) {
        ac += (u64)a.v[i] * b.v[j];
        ac += (u64)c.v[i] * d.v[j];
      }
    }
  });
  return r;
}

// sum_{t < K} a[t] b[t] / 2^(B*L) mod p with ONE reduction (K <= 4 products of tight operands: 4 * 9 * 2^58 plus the reduction
// terms stay below 2^64 per column).  Gives: tight, value < p + sum a[t] b[t] / 2^(B*L).
template <class P, int K>
AMSM_DEV Fe<P> u_dot(const Fe<P>* a, const Fe<P>* b) {
  static_assert(K >= 1 && K <= 4, "column accumulators hold four products");
  constexpr int L = P::L;
  u64 acc = 0;
  u32 m[L];
  Fe<P> r;
  u_columns<P, 0>(acc, m, r, nullptr, false, [&](u64& ac, int k) {
#pragma unroll
    for (int t = 0; t < K; t++) {
#pragma unroll
      for (int i = 0; i < L; i++) {
        int j = k - i;
        if (j >= 0 && j < L) ac += (u64)a[t].v[i] * b[t].v[j];
      }
    }
  });
  return r;
}

// limb-wise a + b (both tight, sum < 2^(B*L)): tight
template <class P>
AMSM_DEV Fe<P> u_add(const Fe<P>& a, const Fe<P>& b) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = a.v[i] + b.v[i];
  u_carry<P>(r);
  return r;
}

// K*p - y with lazy limbs (< 2^(B+1)): multiplication operand only.  Needs y tight and y < (K - 1)*p, so that the
// (unnormalised) top limb cannot go negative (exactly: y < K*p - 2^(B (L - 1)); there is no carry pass to repair it).
template <class P, u32 K>
AMSM_DEV Fe<P> u_kp_minus_lazy(const Fe<P>& y) {
  constexpr UKpBp<P, K, 1> c{};
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = c.v[i] - y.v[i];
  return r;
}

// a^2 / 2^(B*L): the cross products are taken once against doubled limbs (45 instead of 81 products for L = 9).
// Needs: a tight.  Gives: tight, value < p + a^2 / 2^(B*L).
template <class P, bool ADD = false>
AMSM_DEV Fe<P> u_sqr(const Fe<P>& a, const Fe<P>* add = nullptr) {
  constexpr int L = P::L;
  u32 a2[L];
#pragma unroll
  for (int i = 0; i < L; i++) a2[i] = a.v[i] << 1;
  u64 acc = 0;
  u32 m[L];
  Fe<P> r;
  u_columns<P, 0>(acc, m, r, add, ADD, [&](u64& ac, int k) {
#pragma unroll
    for (int i = 0; i < L; i++) {
      int j = k - i;
      if (j >= 0 && j < L) {
        if (i < j) ac += (u64)a2[i] * a.v[j];
        else if (i == j) ac += (u64)a.v[i] * a.v[i];

      }
    }
  });
  return r;
}

// a + K*p - b.  Needs: a, b tight, b < K*p.  Gives: tight, value = a - b + K*p.
template <class P, u32 K>
AMSM_DEV Fe<P> u_sub_k(const Fe<P>& a, const Fe<P>& b) {
  constexpr UKpBp<P, K, 1> c{};
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = a.v[i] + c.v[i] - b.v[i];
  u_carry<P>(r);
  return r;
}

// a + K*p - b - 2c.  Needs: a, b, c tight, b + 2c < K*p.  Gives: tight, value = a - b - 2c + K*p.
template <class P, u32 K>
AMSM_DEV Fe<P> u_sub_bcc_k(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c) {
  constexpr UKpBp<P, K, 3> kp{};
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = a.v[i] + kp.v[i] - b.v[i] - 2u * c.v[i];
  u_carry<P>(r);
  return r;
}

// K*p - b - 2c, unnormalised (limbs < 2^32): addend of u_mul<ADD> / u_sqr<ADD> only.  Needs b, c tight, b + 2c < K*p.
template <class P, u32 K>
AMSM_DEV Fe<P> u_kp_minus_bcc_raw(const Fe<P>& b, const Fe<P>& c) {
  constexpr UKpBp<P, K, 3> kp{};
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = kp.v[i] - b.v[i] - 2u * c.v[i];
  return r;
}

// small multiples.  Needs: a tight.  Gives: tight, value = MULT*a (MULT <= 4).
template <class P, u32 MULT>
AMSM_DEV Fe<P> u_times(const Fe<P>& a) {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = MULT * a.v[i];
  u_carry<P>(r);
  return r;
}

// 2p - y for a canonical y (< p).  Gives: LAZY limbs (< 2^(B+1)), value in (p, 2p]: multiplication operand only.
template <class P>
AMSM_DEV Fe<P> u_neg_lazy(const Fe<P>& y) {
  constexpr UKpBp<P, 2, 1> c{};
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = c.v[i] - y.v[i];
  return r;
}

// conditional subtraction of K*p, K*p/2, ..., p: a tight with value < 2*K*p  ->  canonical (< p) for K a power of 2
template <class P, u32 K>
AMSM_DEV void u_canon_steps(Fe<P>& a) {
  constexpr UKp<P, K> kp{};
  constexpr u32 M = u_mask<P>();
  u32 d[P::L];
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < P::L - 1; i++) {
    u32 x = a.v[i] - kp.v[i] - br;
    br = x >> 31;
    d[i] = x & M;
  }
  u32 x = a.v[P::L - 1] - kp.v[P::L - 1] - br;
  bool lt = (int)x < 0;
  d[P::L - 1] = x;
#pragma unroll
  for (int i = 0; i < P::L; i++) a.v[i] = lt ? a.v[i] : d[i];
  if constexpr (K > 1) u_canon_steps<P, K / 2>(a);
}

// a tight, value < KMAX*p (KMAX a power of two) -> canonical
template <class P, u32 KMAX>
AMSM_DEV void u_canon(Fe<P>& a) {
  static_assert((KMAX & (KMAX - 1)) == 0 && KMAX >= 2, "KMAX must be a power of two");
  u_canon_steps<P, KMAX / 2>(a);
}

// is a == 0 (mod p)?  a tight, value < KMAX*p.  The low limb of k*p filters all but ~KMAX/2^B of the non-zero
// values; the exact test behind it canonicalises.
template <class P, u32 KMAX>
AMSM_DEV bool u_is_zero_mod(const Fe<P>& a) {
  constexpr u32 M = u_mask<P>();
  bool maybe = false;
  if constexpr (P::mod(0) == 1) {
    maybe = a.v[0] < KMAX;
  } else {
#pragma unroll
    for (u32 k = 0; k < KMAX; k++) maybe |= a.v[0] == ((k * P::mod(0)) & M);
  }
  if (!maybe) return false;
  Fe<P> c = a;
  u_canon<P, KMAX>(c);
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) o |= c.v[i];
  return o == 0;
}

// memory (W packed 32-bit words, value < 2^(32 W)) <-> register limbs
template <class P>
AMSM_DEV Fe<P> u_unpack(const u32* w) {
  constexpr u32 M = u_mask<P>();
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) {
    int o = i * P::B, j = o >> 5, sh = o & 31;
    u32 lo = j < P::W ? w[j] : 0u;
    u32 hi = (j + 1) < P::W ? w[j + 1] : 0u;
    u32 x = sh ? (u32)((((u64)hi << 32) | lo) >> sh) : lo;
    r.v[i] = (i < P::L - 1) ? (x & M) : x;
  }
  return r;
}

template <class P>
AMSM_DEV void u_pack(const Fe<P>& a, u32* w) {  // a tight, value < 2^(32 W)
#pragma unroll
  for (int j = 0; j < P::W; j++) {
    // word j = bits [32j, 32j+32): limbs i0 .. i0+2 overlap it
    int i0 = (32 * j) / P::B;
    u32 x = 0;
#pragma unroll
    for (int i = i0; i < i0 + 3 && i < P::L; i++) {
      int s = i * P::B - 32 * j;  // bit position of limb i inside the word (may be negative)
      if (s <= -32 || s >= 32) continue;
      x |= s >= 0 ? (a.v[i] << s) : (a.v[i] >> (-s));
    }
    w[j] = x;
  }
}

}  // namespace amsm
