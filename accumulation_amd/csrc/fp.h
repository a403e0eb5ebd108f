// Prime-field Montgomery arithmetic for gfx950 (CDNA4), 32-bit limbs, little-endian.
//
// Replaces (device side) ark-ff ^0.2.0 `Fp256`/`Fp384` Montgomery arithmetic, the L0 layer under
// every MSM / Pedersen commit in the reference (Cargo.toml:15-16 -- the source is
// not in /root/reference).  Memory format is identical to ark-ff's: an element is x*R mod m stored as
// little-endian u64 limbs, R = 2^(64*limbs); a u64 limb is two consecutive u32 limbs here.
//
// gfx950 notes (measured with tools/ubench_valu.hip): v_mad_u64_u32 issues at ~4-5
// cycles per wave64 per SIMD (half rate, same as v_fma_f64), v_add_co/v_addc at ~2; there is no 64-bit
// multiplier and no carry-in on the MAD, and a VALU write of VCC/SGPR needs 2 wait states before a VALU
// reads it as carry-in.  All loops are fully unrolled over compile-time limb counts and compile-time
// moduli so zero / one limbs of the modulus fold away (Pallas: 3 real MADs per reduction round).
//
// Two representations live behind one set of primitives: the saturated 32-bit-limb fields of this file (both scalar
// fields, the BLS12-381 base field) and the unsaturated 9 x 29-bit Pallas base field of fpu.h, which the MSM kernels
// compute in (DevField below).  The group law (ec.h) only uses the bound-aware primitives near the end of this file
// (fe_mul, fe_sqr, fe_sub_k, fe_mul_sub_k, fe_mul_sub_mul_k, ...), which are the plain modular operations here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amsm {

typedef uint32_t u32;
typedef uint64_t u64;

#define AMSM_DEV __device__ __forceinline__
#define AMSM_HD __host__ __device__ __forceinline__

// ------------------------------------------------------------------------------------------------
// Field parameter packs.  mod(i)/one(i)/r2(i) are constexpr functions (not arrays) so device code
// never ODR-uses a host object.  Constants verified in tests/test_oracle.py against oracle/pyref.py.
// ------------------------------------------------------------------------------------------------
#define AMSM_TABLE(name, n, ...)                                   \
  AMSM_HD static constexpr u32 name(int i) {                       \
    constexpr u32 t[n] = {__VA_ARGS__};                            \
    return t[i];                                                   \
  }

struct PallasFq {  // base field of Pallas (coordinates); 255 bits
  static constexpr int L = 8;  // register limbs
  static constexpr int W = 8;  // 32-bit words in memory
  static constexpr bool UNSAT = false;
  static constexpr u32 INV = 0xffffffffu;  // -p^-1 mod 2^32
  AMSM_TABLE(mod, 8, 0x00000001u, 0x992d30edu, 0x094cf91bu, 0x224698fcu, 0x00000000u, 0x00000000u, 0x00000000u, 0x40000000u)
  AMSM_TABLE(one, 8, 0xfffffffdu, 0x34786d38u, 0xe41914adu, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu)
  AMSM_TABLE(r2, 8, 0x0000000fu, 0x8c78ecb3u, 0x8b0de0e7u, 0xd7d30dbdu, 0xc3c95d18u, 0x7797a99bu, 0x7b9cb714u, 0x096d41afu)
};

struct PallasFr {  // scalar field of Pallas; 255 bits
  static constexpr int L = 8;  // register limbs
  static constexpr int W = 8;  // 32-bit words in memory
  static constexpr bool UNSAT = false;
  static constexpr u32 INV = 0xffffffffu;
  AMSM_TABLE(mod, 8, 0x00000001u, 0x8c46eb21u, 0x0994a8ddu, 0x224698fcu, 0x00000000u, 0x00000000u, 0x00000000u, 0x40000000u)
  AMSM_TABLE(one, 8, 0xfffffffdu, 0x5b2b3e9cu, 0xe3420567u, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu)
  AMSM_TABLE(r2, 8, 0x0000000fu, 0xfc9678ffu, 0x891a16e3u, 0x67bb433du, 0x04ccf590u, 0x7fae2310u, 0x7ccfdaa9u, 0x096d41afu)
};

struct Bls12381Fq {  // base field of BLS12-381; 381 bits
  static constexpr int L = 12;  // register limbs
  static constexpr int W = 12;  // 32-bit words in memory
  static constexpr bool UNSAT = false;
  static constexpr u32 INV = 0xfffcfffdu;
  AMSM_TABLE(mod, 12, 0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
             0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau)
  AMSM_TABLE(one, 12, 0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
             0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u)
  AMSM_TABLE(r2, 12, 0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u,
             0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u)
};

struct Bls12381Fr {  // scalar field of BLS12-381; 255 bits
  static constexpr int L = 8;  // register limbs
  static constexpr int W = 8;  // 32-bit words in memory
  static constexpr bool UNSAT = false;
  static constexpr u32 INV = 0xffffffffu;
  AMSM_TABLE(mod, 8, 0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u)
  AMSM_TABLE(one, 8, 0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u)
  AMSM_TABLE(r2, 8, 0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu, 0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u)
};

// ------------------------------------------------------------------------------------------------
// Element type and helpers
// ------------------------------------------------------------------------------------------------
template <class P>
struct Fe {
  u32 v[P::L];
};

}  // namespace amsm
#include "fpu.h"  // unsaturated-limb packs (PallasFqU) and their u_* primitives
namespace amsm {

// Field the DEVICE kernels compute in for a given ABI field: Pallas Fq runs on 9 x 29-bit unsaturated limbs
// (internal Montgomery radix 2^261, fpu.h), BLS12-381 Fq on 14 x 28-bit limbs (radix 2^392).  (The saturated 32-bit-limb
// schedules serve both scalar fields; for the base fields they measured 21 % / 23 % slower: DESIGN.md 4.1.)
template <class Fq>
struct DevField {
  using type = Fq;
};
template <>
struct DevField<PallasFq> {
  using type = PallasFqU;
};
template <>
struct DevField<Bls12381Fq> {  // 14 x 28-bit limbs, internal radix 2^392: mixed addition 22.6 k vs 29.4 k cycles
  using type = Bls12381FqU;
};

template <class P>
AMSM_DEV Fe<P> fe_zero() {
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = 0;
  return r;
}

template <class P>
AMSM_DEV Fe<P> fe_one() {  // Montgomery form of 1
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = P::one(i);
  return r;
}

template <class P>
AMSM_DEV bool fe_is_zero(const Fe<P>& a) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) o |= a.v[i];
  return o == 0;
}

template <class P>
AMSM_DEV bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
  u32 o = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) o |= a.v[i] ^ b.v[i];
  return o == 0;
}

// Carry chains are written with __builtin_addc/__builtin_subc so hipcc emits v_add_co/v_addc_co (and
// pads the gfx950 VCC-write -> carry-read hazard itself) instead of emulating carries in 64-bit adds.
// r = a - m if a >= m else a   (a < 2m, possibly with a carry word `hi` in {0,1})
template <class P>
AMSM_DEV void fe_cond_sub(Fe<P>& a, u32 hi = 0) {
  static_assert(!P::UNSAT, "saturated fields only");
  u32 d[P::L];
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) d[i] = __builtin_subc(a.v[i], P::mod(i), br, &br);
  bool ge = (hi != 0) || (br == 0);
#pragma unroll
  for (int i = 0; i < P::L; i++) a.v[i] = ge ? d[i] : a.v[i];
}

template <class P>
AMSM_DEV Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
  static_assert(!P::UNSAT, "unsaturated fields use the bound-aware fe_*_k primitives");
  Fe<P> r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
  fe_cond_sub<P>(r, c);
  return r;
}

template <class P>
AMSM_DEV Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
  static_assert(!P::UNSAT, "unsaturated fields use the bound-aware fe_*_k primitives");
  Fe<P> r;
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = __builtin_subc(a.v[i], b.v[i], br, &br);
  // add the modulus back when the subtraction borrowed
  u32 mask = (u32)0 - br;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = __builtin_addc(r.v[i], P::mod(i) & mask, c, &c);
  return r;
}

template <class P>
AMSM_DEV Fe<P> fe_neg(const Fe<P>& a) {
  static_assert(!P::UNSAT, "unsaturated fields use fe_neg_lazy");
  if (fe_is_zero<P>(a)) return a;
  Fe<P> r;
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = __builtin_subc(P::mod(i), a.v[i], br, &br);
  return r;
}


// ------------------------------------------------------------------------------------------------
// Montgomery multiplication, CIOS (coarsely integrated operand scanning), r = a*b/R mod m.
// Every step computes x = a_j*b_i + t_j + c <= 2^64-1, which maps to one v_mad_u64_u32 plus a
// 64-bit add.  The reduction rows multiply by the compile-time modulus limbs: zero limbs vanish and
// the limb equal to 1 becomes an add (Pallas Fq/Fr: mod = {1,p1,p2,p3,0,0,0,2^30}).
// ------------------------------------------------------------------------------------------------
template <class P>
AMSM_DEV Fe<P> fe_mul_ref(const Fe<P>& a, const Fe<P>& b) {
  constexpr int L = P::L;
  u32 t[L + 2];
#pragma unroll
  for (int i = 0; i < L + 2; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < L; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < L; j++) {
      u64 x = (u64)a.v[j] * b.v[i] + t[j] + c;
      t[j] = (u32)x;
      c = x >> 32;
    }
    u64 x = (u64)t[L] + c;
    t[L] = (u32)x;
    t[L + 1] = (u32)(x >> 32);
    u32 m = t[0] * P::INV;
    c = ((u64)m * P::mod(0) + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < L; j++) {
      u64 y = (u64)m * P::mod(j) + t[j] + c;
      t[j - 1] = (u32)y;
      c = y >> 32;
    }
    x = (u64)t[L] + c;
    t[L - 1] = (u32)x;
    t[L] = t[L + 1] + (u32)(x >> 32);
  }
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < L; i++) r.v[i] = t[i];
  fe_cond_sub<P>(r, t[L]);
  return r;
}

// fe_mul: the portable loop above is the definition; for gfx950 every field gets a generated
// column-wise v_mad_u64_u32 / v_addc_co_u32 schedule (fp_mul_gfx950.h, tools/gen_fp_asm.py) which is
// bit-identical and ~2x fewer VALU instructions.  -DAMSM_NO_ASM_MUL keeps the portable loop (A/B, debug).
template <class P>
AMSM_DEV Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
  if constexpr (P::UNSAT) return u_mul<P>(a, b);
  else return fe_mul_ref<P>(a, b);
}

// a0 b0 + a1 b1 (+ a2 b2): the definition; the scalar fields get generated schedules that accumulate the products of a column
// before ONE Montgomery reduction (fp_mul_gfx950.h; canonical operands in, canonical result out: bit-identical)
template <class P>
AMSM_DEV Fe<P> fe_dot2(const Fe<P>& a0, const Fe<P>& b0, const Fe<P>& a1, const Fe<P>& b1) {
  return fe_add<P>(fe_mul<P>(a0, b0), fe_mul<P>(a1, b1));
}
template <class P>
AMSM_DEV Fe<P> fe_dot3(const Fe<P>& a0, const Fe<P>& b0, const Fe<P>& a1, const Fe<P>& b1, const Fe<P>& a2, const Fe<P>& b2) {
  return fe_add<P>(fe_add<P>(fe_mul<P>(a0, b0), fe_mul<P>(a1, b1)), fe_mul<P>(a2, b2));
}

}  // namespace amsm
#if defined(__HIP_DEVICE_COMPILE__)
#include "fp_mul_gfx950.h"
#endif
namespace amsm {

template <class P>
AMSM_DEV Fe<P> fe_sqr(const Fe<P>& a) {
  if constexpr (P::UNSAT) return u_sqr<P>(a);
  else return fe_mul<P>(a, a);
}

// ------------------------------------------------------------------------------------------------
// Bound-aware primitives the group law (ec.h) is written in.  On a saturated field they are the plain modular
// operations (every value canonical); on an unsaturated field (fpu.h) values are only bounded and the K's say how
// many multiples of p keep a difference positive -- the callers' comments carry the bounds.
// ------------------------------------------------------------------------------------------------
template <class P, u32 K>
AMSM_DEV Fe<P> fe_sub_k(const Fe<P>& a, const Fe<P>& b) {  // a - b   (unsat: + K p, needs b < K p)
  if constexpr (P::UNSAT) return u_sub_k<P, K>(a, b);
  else return fe_sub<P>(a, b);
}
template <class P, u32 K>
AMSM_DEV Fe<P> fe_sub_bcc_k(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c) {  // a - b - 2c   (unsat: + K p)
  if constexpr (P::UNSAT) return u_sub_bcc_k<P, K>(a, b, c);
  else return fe_sub<P>(fe_sub<P>(fe_sub<P>(a, b), c), c);
}
// a*b - c   (unsat: + K p, needs c < (K - 1) p; the subtraction rides in the product's upper columns, result tight)
template <class P, u32 K>
AMSM_DEV Fe<P> fe_mul_sub_k(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c) {
  if constexpr (P::UNSAT) {
    Fe<P> add = u_kp_minus_lazy<P, K>(c);
    return u_mul<P, true>(a, b, &add);
  } else {
    return fe_sub<P>(fe_mul<P>(a, b), c);
  }
}
// a^2 - b - 2c   (unsat: + K p, needs b + 2c < K p)
template <class P, u32 K>
AMSM_DEV Fe<P> fe_sqr_sub_bcc_k(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c) {
  if constexpr (P::UNSAT) {
    Fe<P> add = u_kp_minus_bcc_raw<P, K>(b, c);
    return u_sqr<P, true>(a, &add);
  } else {
    return fe_sub<P>(fe_sub<P>(fe_sub<P>(fe_sqr<P>(a), b), c), c);
  }
}
// a*b - c*d   (unsat: + (K p * d) / R', needs c < K p; ONE Montgomery reduction for both products)
template <class P, u32 K>
AMSM_DEV Fe<P> fe_mul_sub_mul_k(const Fe<P>& a, const Fe<P>& b, const Fe<P>& c, const Fe<P>& d) {
  if constexpr (P::UNSAT) return u_mul_add_mul<P>(a, b, u_kp_minus_lazy<P, K>(c), d);
  else return fe_sub<P>(fe_mul<P>(a, b), fe_mul<P>(c, d));
}
template <class P>
AMSM_DEV Fe<P> fe_dbl(const Fe<P>& a) {
  if constexpr (P::UNSAT) return u_times<P, 2>(a);
  else return fe_add<P>(a, a);
}
template <class P>
AMSM_DEV Fe<P> fe_triple(const Fe<P>& a) {
  if constexpr (P::UNSAT) return u_times<P, 3>(a);
  else return fe_add<P>(fe_add<P>(a, a), a);
}
// -y for a canonical y.  unsat: 2p - y with lazy limbs -- only valid as a multiplication operand or through fe_tight
template <class P>
AMSM_DEV Fe<P> fe_neg_lazy(const Fe<P>& y) {
  if constexpr (P::UNSAT) return u_neg_lazy<P>(y);
  else return fe_neg<P>(y);
}
template <class P>
AMSM_DEV Fe<P> fe_tight(const Fe<P>& a) {  // lazy -> tight (no-op on saturated fields)
  Fe<P> r = a;
  if constexpr (P::UNSAT) u_carry<P>(r);
  return r;
}
// a == 0 mod p, for a value < KMAX p (KMAX a power of two)
template <class P, u32 KMAX>
AMSM_DEV bool fe_is_zero_mod(const Fe<P>& a) {
  if constexpr (P::UNSAT) return u_is_zero_mod<P, KMAX>(a);
  else return fe_is_zero<P>(a);
}
// C-ABI Montgomery radix (2^(32 W)) <-> the device's internal radix (identity on saturated fields).
// fe_import: any value < 2^(32 W) in, < 2p out.  fe_export: value < 8p in, canonical out.
template <class P>
AMSM_DEV Fe<P> fe_import(const Fe<P>& a) {
  if constexpr (P::UNSAT) {
    Fe<P> k;
#pragma unroll
    for (int i = 0; i < P::L; i++) k.v[i] = P::k_import(i);
    return u_mul<P>(a, k);
  } else {
    return a;
  }
}
template <class P>
AMSM_DEV Fe<P> fe_export(const Fe<P>& a) {
  if constexpr (P::UNSAT) {
    Fe<P> k;
#pragma unroll
    for (int i = 0; i < P::L; i++) k.v[i] = P::k_export(i);
    Fe<P> r = u_mul<P>(a, k);
    u_canon<P, 2>(r);
    return r;
  } else {
    return a;
  }
}

// Montgomery -> canonical integer (ark-ff `into_repr`, visible at
// src/r1cs_nark_as/r1cs_nark/mod.rs:59): one Montgomery reduction = multiply by the integer 1.
template <class P>
AMSM_DEV Fe<P> fe_from_mont(const Fe<P>& a) {
  Fe<P> o = fe_zero<P>();
  o.v[0] = 1;
  return fe_mul<P>(a, o);
}

template <class P>
AMSM_DEV Fe<P> fe_to_mont(const Fe<P>& a) {
  Fe<P> r2;
#pragma unroll
  for (int i = 0; i < P::L; i++) r2.v[i] = P::r2(i);
  return fe_mul<P>(a, r2);
}

template <class P, bool U = P::UNSAT>
struct SatOf {
  using type = P;
};
template <class P>
struct SatOf<P, true> {
  using type = typename P::Sat;
};

// a^(m-2): Fermat inversion (only used off the hot path: key precomputation, tests).
template <class P>
AMSM_DEV Fe<P> fe_inv(const Fe<P>& a) {
  // exponent m-2 (32-bit words of the modulus), scanned MSB->LSB
  using E = typename SatOf<P>::type;
  u32 e[E::L];
  u64 br = 2;
#pragma unroll
  for (int i = 0; i < E::L; i++) {
    u64 x = (u64)E::mod(i) - br;
    e[i] = (u32)x;
    br = (x >> 32) & 1;
  }
  Fe<P> r = fe_one<P>();
  for (int i = E::L * 32 - 1; i >= 0; i--) {
    r = fe_sqr<P>(r);
    if ((e[i >> 5] >> (i & 31)) & 1) r = fe_mul<P>(r, a);
  }
  return r;
}

// 16-byte vector load/store of an element (coalesced dwordx4 per lane; Guideline 13).  Memory is W packed words;
// an unsaturated field unpacks to / canonicalises and packs from its register limbs here.
template <class P>
AMSM_DEV Fe<P> fe_from_words(const u32* w) {  // W memory words (already in registers) -> element
  if constexpr (P::UNSAT) {
    return u_unpack<P>(w);
  } else {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::L; i++) r.v[i] = w[i];
    return r;
  }
}

template <class P>
AMSM_DEV Fe<P> fe_load(const u32* __restrict__ p) {
  u32 w[P::W];
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < P::W / 4; i++) {
    uint4 x = q[i];
    w[4 * i + 0] = x.x;
    w[4 * i + 1] = x.y;
    w[4 * i + 2] = x.z;
    w[4 * i + 3] = x.w;
  }
  return fe_from_words<P>(w);
}

template <class P>
AMSM_DEV void fe_store(u32* __restrict__ p, const Fe<P>& a) {  // unsat: a tight, value < 8p; stored canonical
  u32 w[P::W];
  if constexpr (P::UNSAT) {
    Fe<P> c = a;
    u_canon<P, 8>(c);
    u_pack<P>(c, w);
  } else {
#pragma unroll
    for (int i = 0; i < P::L; i++) w[i] = a.v[i];
  }
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < P::W / 4; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

}  // namespace amsm
