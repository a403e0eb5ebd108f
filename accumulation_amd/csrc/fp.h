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
  static constexpr int L = 8;
  static constexpr u32 INV = 0xffffffffu;  // -p^-1 mod 2^32
  AMSM_TABLE(mod, 8, 0x00000001u, 0x992d30edu, 0x094cf91bu, 0x224698fcu, 0x00000000u, 0x00000000u, 0x00000000u, 0x40000000u)
  AMSM_TABLE(one, 8, 0xfffffffdu, 0x34786d38u, 0xe41914adu, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu)
  AMSM_TABLE(r2, 8, 0x0000000fu, 0x8c78ecb3u, 0x8b0de0e7u, 0xd7d30dbdu, 0xc3c95d18u, 0x7797a99bu, 0x7b9cb714u, 0x096d41afu)
};

struct PallasFr {  // scalar field of Pallas; 255 bits
  static constexpr int L = 8;
  static constexpr u32 INV = 0xffffffffu;
  AMSM_TABLE(mod, 8, 0x00000001u, 0x8c46eb21u, 0x0994a8ddu, 0x224698fcu, 0x00000000u, 0x00000000u, 0x00000000u, 0x40000000u)
  AMSM_TABLE(one, 8, 0xfffffffdu, 0x5b2b3e9cu, 0xe3420567u, 0x992c350bu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu)
  AMSM_TABLE(r2, 8, 0x0000000fu, 0xfc9678ffu, 0x891a16e3u, 0x67bb433du, 0x04ccf590u, 0x7fae2310u, 0x7ccfdaa9u, 0x096d41afu)
};

struct Bls12381Fq {  // base field of BLS12-381; 381 bits
  static constexpr int L = 12;
  static constexpr u32 INV = 0xfffcfffdu;
  AMSM_TABLE(mod, 12, 0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
             0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau)
  AMSM_TABLE(one, 12, 0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu, 0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
             0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u)
  AMSM_TABLE(r2, 12, 0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u, 0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u,
             0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u)
};

struct Bls12381Fr {  // scalar field of BLS12-381; 255 bits
  static constexpr int L = 8;
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
  Fe<P> r;
  u32 c = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = __builtin_addc(a.v[i], b.v[i], c, &c);
  fe_cond_sub<P>(r, c);
  return r;
}

template <class P>
AMSM_DEV Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
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
  if (fe_is_zero<P>(a)) return a;
  Fe<P> r;
  u32 br = 0;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = __builtin_subc(P::mod(i), a.v[i], br, &br);
  return r;
}

template <class P>
AMSM_DEV Fe<P> fe_dbl(const Fe<P>& a) {
  return fe_add<P>(a, a);
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
  return fe_mul_ref<P>(a, b);
}

}  // namespace amsm
#if !defined(AMSM_NO_ASM_MUL) && defined(__HIP_DEVICE_COMPILE__)
#include "fp_mul_gfx950.h"
#endif
namespace amsm {

template <class P>
AMSM_DEV Fe<P> fe_sqr(const Fe<P>& a) {
  return fe_mul<P>(a, a);
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

// a^(m-2): Fermat inversion (only used off the hot path: key precomputation, tests).
template <class P>
AMSM_DEV Fe<P> fe_inv(const Fe<P>& a) {
  // exponent m-2, scanned MSB->LSB
  u32 e[P::L];
  u64 br = 2;
#pragma unroll
  for (int i = 0; i < P::L; i++) {
    u64 x = (u64)P::mod(i) - br;
    e[i] = (u32)x;
    br = (x >> 32) & 1;
  }
  Fe<P> r = fe_one<P>();
  for (int i = P::L * 32 - 1; i >= 0; i--) {
    r = fe_sqr<P>(r);
    if ((e[i >> 5] >> (i & 31)) & 1) r = fe_mul<P>(r, a);
  }
  return r;
}

// 16-byte vector load/store of an element (coalesced dwordx4 per lane; Guideline 13).
template <class P>
AMSM_DEV Fe<P> fe_load(const u32* __restrict__ p) {
  Fe<P> r;
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < P::L / 4; i++) {
    uint4 x = q[i];
    r.v[4 * i + 0] = x.x;
    r.v[4 * i + 1] = x.y;
    r.v[4 * i + 2] = x.z;
    r.v[4 * i + 3] = x.w;
  }
  return r;
}

template <class P>
AMSM_DEV void fe_store(u32* __restrict__ p, const Fe<P>& a) {
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < P::L / 4; i++) q[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
}

}  // namespace amsm
