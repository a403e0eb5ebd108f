// Short-Weierstrass (a = 0) group law for gfx950 in extended-Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ = 0).
//
// Replaces (device side) ark-ec ^0.2.0 `GroupProjective::{add_assign_mixed, add_assign,
// double_in_place}` used under `VariableBaseMSM::multi_scalar_mul` (source not in
// /root/reference, Cargo.toml:15).  ark-ec uses Jacobian (madd-2007-bl, 7M+4S); XYZZ is chosen here
// because the bucket-accumulation hot loop is multiplier-bound on gfx950 and XYZZ mixed addition is
// 8M+2S.  Intermediate projective values are never compared with the reference -- only the
// canonical affine result is.
//
// The formulas are written once, in the bound-aware primitives of fp.h.  On a saturated field every value is
// canonical and the K's are ignored.  On an unsaturated field (fpu.h; Pallas Fq, cap = 2^261 ~ 128 p) values are
// only bounded; the invariant of a point held in registers or memory is
//        X < 8p      Y < 3p      ZZ, ZZZ < 2p      (affine / loaded coordinates: canonical, < p)
// and the comments `[< k p]` give the bound of each intermediate: a product of A < a p and B < b p is
// < (1 + a b / 128) p, a difference `x - y (+K p)` needs y < K p and is < (x + K) p; the fused
// `a b - c d` (ONE reduction for both products, fe_mul_sub_mul_k<K>) needs c < (K - 1) p and is
// < (1 + (a b + K d) / 128) p.
#pragma once
#include "fp.h"

namespace amsm {

template <class P>
struct Affine {  // Montgomery-form coordinates; (0, 0) encodes the point at infinity
  Fe<P> x, y;
};

template <class P>
struct XYZZ {
  Fe<P> x, y, zz, zzz;
};

template <class P>
AMSM_DEV XYZZ<P> xyzz_inf() {
  XYZZ<P> r;
  r.x = fe_zero<P>();
  r.y = fe_zero<P>();
  r.zz = fe_zero<P>();
  r.zzz = fe_zero<P>();
  return r;
}

template <class P>
AMSM_DEV bool xyzz_is_inf(const XYZZ<P>& p) {
  return fe_is_zero<P>(p.zz);
}

template <class P>
AMSM_DEV bool affine_is_inf(const Affine<P>& p) {
  return fe_is_zero<P>(p.x) && fe_is_zero<P>(p.y);
}

template <class P>
AMSM_DEV XYZZ<P> xyzz_from_affine(const Affine<P>& p) {
  XYZZ<P> r;
  if (affine_is_inf<P>(p)) return xyzz_inf<P>();
  r.x = p.x;
  r.y = p.y;
  r.zz = fe_one<P>();
  r.zzz = fe_one<P>();
  return r;
}

// dbl-2008-s-1 with a = 0: 6M + 3S
template <class P>
AMSM_DEV XYZZ<P> xyzz_dbl(const XYZZ<P>& p) {
  if (xyzz_is_inf<P>(p)) return p;
  Fe<P> u = fe_dbl<P>(p.y);                                  // [< 16p]
  Fe<P> v = fe_sqr<P>(u);                                    // [< 3p]
  Fe<P> w = fe_mul<P>(u, v);                                 // [< 1.4p]
  Fe<P> s = fe_mul<P>(p.x, v);                               // [< 1.2p]
  Fe<P> xx = fe_sqr<P>(p.x);                                 // [< 1.5p]
  Fe<P> m = fe_triple<P>(xx);                                // [< 4.5p]
  Fe<P> zero = fe_zero<P>();
  XYZZ<P> r;
  r.x = fe_sub_bcc_k<P, 4>(fe_sqr<P>(m), zero, s);           // m^2 [< 1.2p] - 2s (+4p)  [< 5.2p]
  Fe<P> t = fe_sub_k<P, 8>(s, r.x);                          // [< 9.2p]
  r.y = fe_mul_sub_mul_k<P, 4>(m, t, p.y, w);                // (m t + (4p - y) w) / R': (42 + 6) / 128  [< 1.4p]
  r.zz = fe_mul<P>(v, p.zz);                                 // [< 1.1p]
  r.zzz = fe_mul<P>(w, p.zzz);                               // [< 1.1p]
  return r;
}

// mdbl-2008-s-1 with a = 0 (affine input): 3M + 3S
template <class P>
AMSM_DEV XYZZ<P> xyzz_dbl_affine(const Affine<P>& p) {  // p.x < 2p, p.y <= 2p (a lazily negated y through fe_tight), tight
  Fe<P> u = fe_dbl<P>(p.y);                                  // [< 4p]
  Fe<P> v = fe_sqr<P>(u);                                    // [< 1.2p]
  Fe<P> w = fe_mul<P>(u, v);                                 // [< 1.1p]
  Fe<P> s = fe_mul<P>(p.x, v);                               // [< 1.1p]
  Fe<P> xx = fe_sqr<P>(p.x);                                 // [< 1.1p]
  Fe<P> m = fe_triple<P>(xx);                                // [< 3.3p]
  Fe<P> zero = fe_zero<P>();
  XYZZ<P> r;
  r.x = fe_sub_bcc_k<P, 4>(fe_sqr<P>(m), zero, s);           // [< 5.2p]
  Fe<P> t = fe_sub_k<P, 8>(s, r.x);                          // [< 9.1p]
  // K = 4, not 2: y reaches 2p when xyzz_madd doubles a NEGATED point (q.y = 2p - y, y small), and K p - y is formed limb-wise
  // without a carry pass -- its top limb must not go negative, i.e. y < K p - 2^(B (L - 1)).  With K = 2 a point whose y lies
  // (in the internal Montgomery radix) below 2^(B (L - 1)) (one in 2^17 for BLS12-381, one in 2^22 for Pallas) doubled to garbage on that path: found by
  // tools/fuzz_msm.py as ONE wrong point in a 124 124-point key fold by x = r - 2 (tests/golden/bls12_381_negated_doubling.json).
  r.y = fe_mul_sub_mul_k<P, 4>(m, t, p.y, w);                // (m t + (4p - y) w) / R': (30 + 4.4) / 128  [< 1.3p]
  r.zz = v;
  r.zzz = w;
  return r;
}

// madd-2008-s: acc += q (q affine, not infinity unless encoded (0,0)); 8M + 2S on the generic path.
// Exceptional cases (acc = inf, q = inf, q = +-acc) are handled exactly: duplicate bases and
// bases summing to zero do occur in the reference's degenerate inputs.
// q = (x, y or -y): x canonical; `qy` is y or fe_neg_lazy(y) (lazy limbs on an unsaturated field, value <= 2p), which
// is why the negation is taken by the caller through affine_neg_if and q.y is only used as a multiplication
// operand (or through fe_tight).
template <class P>
AMSM_DEV void xyzz_madd(XYZZ<P>& acc, const Affine<P>& q) {
  if (affine_is_inf<P>(q)) return;
  if (xyzz_is_inf<P>(acc)) {
    acc.x = q.x;
    acc.y = fe_tight<P>(q.y);
    acc.zz = fe_one<P>();
    acc.zzz = fe_one<P>();
    return;
  }
  Fe<P> p = fe_mul_sub_k<P, 9>(q.x, acc.zz, acc.x);   // u2 - X1 (+9p)  [< 10.1p]
  Fe<P> r = fe_mul_sub_k<P, 4>(q.y, acc.zzz, acc.y);  // s2 - Y1 (+4p)  [< 5.1p]
  if (fe_is_zero_mod<P, 16>(p)) {
    if (fe_is_zero_mod<P, 16>(r)) {
      Affine<P> qt;
      qt.x = q.x;
      qt.y = fe_tight<P>(q.y);
      acc = xyzz_dbl_affine<P>(qt);
    } else {
      acc = xyzz_inf<P>();
    }
    return;
  }
  Fe<P> pp = fe_sqr<P>(p);                                // [< 1.8p]
  Fe<P> ppp = fe_mul<P>(p, pp);                           // [< 1.2p]
  Fe<P> qq = fe_mul<P>(acc.x, pp);                        // [< 1.2p]
  Fe<P> x3 = fe_sqr_sub_bcc_k<P, 4>(r, ppp, qq);          // r^2 [< 1.3p] - ppp - 2qq [< 3.6p] (+4p)  [< 5.3p]
  Fe<P> t = fe_sub_k<P, 8>(qq, x3);                       // [< 9.2p]
  Fe<P> y3 = fe_mul_sub_mul_k<P, 4>(r, t, acc.y, ppp);   // (r t + (4p - y1) ppp) / R': (84 + 5) / 128  [< 1.7p]
  acc.x = x3;
  acc.y = y3;
  acc.zz = fe_mul<P>(acc.zz, pp);     // [< 1.1p]
  acc.zzz = fe_mul<P>(acc.zzz, ppp);  // [< 1.1p]
}

// add-2008-s: acc += q (both XYZZ); 12M + 2S on the generic path.
template <class P>
AMSM_DEV void xyzz_add(XYZZ<P>& acc, const XYZZ<P>& q) {
  if (xyzz_is_inf<P>(q)) return;
  if (xyzz_is_inf<P>(acc)) {
    acc = q;
    return;
  }
  Fe<P> u1 = fe_mul<P>(acc.x, q.zz);   // [< 1.2p]
  Fe<P> u2 = fe_mul<P>(q.x, acc.zz);   // [< 1.2p]
  Fe<P> s1 = fe_mul<P>(acc.y, q.zzz);  // [< 1.2p]
  Fe<P> s2 = fe_mul<P>(q.y, acc.zzz);  // [< 1.2p]
  Fe<P> p = fe_sub_k<P, 2>(u2, u1);    // [< 3.2p]
  Fe<P> r = fe_sub_k<P, 2>(s2, s1);    // [< 3.2p]
  if (fe_is_zero_mod<P, 4>(p)) {
    if (fe_is_zero_mod<P, 4>(r)) {
      acc = xyzz_dbl<P>(acc);
    } else {
      acc = xyzz_inf<P>();
    }
    return;
  }
  Fe<P> pp = fe_sqr<P>(p);                               // [< 1.1p]
  Fe<P> ppp = fe_mul<P>(p, pp);                          // [< 1.1p]
  Fe<P> qq = fe_mul<P>(u1, pp);                          // [< 1.1p]
  Fe<P> x3 = fe_sub_bcc_k<P, 4>(fe_sqr<P>(r), ppp, qq);  // [< 1.1p] - [< 3.3p] (+4p)  [< 5.1p]
  Fe<P> t = fe_sub_k<P, 8>(qq, x3);                      // [< 9.1p]
  Fe<P> y3 = fe_mul_sub_mul_k<P, 2>(r, t, s1, ppp);      // (r t + (2p - s1) ppp) / R': (29 + 3) / 128  [< 1.3p]
  acc.x = x3;
  acc.y = y3;
  acc.zz = fe_mul<P>(fe_mul<P>(acc.zz, q.zz), pp);       // [< 1.1p]
  acc.zzz = fe_mul<P>(fe_mul<P>(acc.zzz, q.zzz), ppp);   // [< 1.1p]
}

template <class P>
AMSM_DEV Affine<P> affine_neg_if(const Affine<P>& p, bool negate) {
  Affine<P> r;
  r.x = p.x;
  Fe<P> ny = fe_neg_lazy<P>(p.y);
  negate = negate && !fe_is_zero<P>(p.y);  // keeps the (0, 0) encoding of infinity (the lazy -0 is 2p, not 0)
#pragma unroll
  for (int i = 0; i < P::L; i++) r.y.v[i] = negate ? ny.v[i] : p.y.v[i];
  return r;
}

// ---------------------------------------------------------------------------------------------
// Jacobian coordinates (x = X / Z^2, y = Y / Z^3; infinity <=> Z = 0) for the LADDERS of the key folds (k_points_fold: 128
// doublings and ~43 mixed additions per point): the doubling is 3M + 4S against XYZZ's 6M + 3S, the mixed addition 8M + 3S
// against 8M + 2S -- 1/6 fewer multiplier passes over a fold.  (The bucket accumulation adds, it does not double: XYZZ stays
// there.)  ark-ec's own formulas for these curves are Jacobian as well (ec.h header); only canonical affine results are compared.
// Invariant of a point held in registers (unsaturated fields, cap = 2^261 ~ 128 p for Pallas, far more for BLS12-381):
//        X < 12p      Y < 13p      Z < 3p        all tight
// with the same bound notation as above.
// ---------------------------------------------------------------------------------------------
template <class P>
struct Jac {
  Fe<P> x, y, z;
};
template <class P>
AMSM_DEV Jac<P> jac_inf() {
  Jac<P> r;
  r.x = fe_zero<P>();
  r.y = fe_zero<P>();
  r.z = fe_zero<P>();
  return r;
}
template <class P>
AMSM_DEV bool jac_is_inf(const Jac<P>& p) {
  return fe_is_zero<P>(p.z);
}
// dbl-2007-bl shape with a = 0 (S = X Y^2 as a product): 3M + 4S
template <class P>
AMSM_DEV Jac<P> jac_dbl(const Jac<P>& p) {
  if (jac_is_inf<P>(p)) return p;
  const Fe<P> zero = fe_zero<P>();
  Fe<P> xx = fe_sqr<P>(p.x);                                  // [< 2.2p]
  Fe<P> yy = fe_sqr<P>(p.y);                                  // [< 2.4p]
  Fe<P> yyyy = fe_sqr<P>(yy);                                 // [< 1.1p]
  Fe<P> s4 = fe_dbl<P>(fe_dbl<P>(fe_mul<P>(p.x, yy)));        // 4 X Y^2: product [< 1.3p] -> [< 5p]
  Fe<P> m = fe_triple<P>(xx);                                 // [< 6.5p]
  Jac<P> r;
  r.x = fe_sqr_sub_bcc_k<P, 10>(m, zero, s4);                 // m^2 [< 1.4p] - 2 s4 [< 10p] (+10p)  [< 11.4p]
  Fe<P> t = fe_sub_k<P, 12>(s4, r.x);                         // [< 17p]
  Fe<P> y8 = fe_dbl<P>(fe_dbl<P>(fe_dbl<P>(yyyy)));           // 8 Y^4  [< 8.5p]
  r.y = fe_mul_sub_k<P, 10>(m, t, y8);                        // m t [< 1.9p] - y8 (+10p)  [< 11.9p]
  r.z = fe_dbl<P>(fe_mul<P>(p.y, p.z));                       // 2 Y Z: product [< 1.4p] -> [< 2.7p]
  return r;
}
// acc += q (q affine; same conventions as xyzz_madd: (0, 0) is infinity, q.y may be a lazily negated y): 8M + 3S
template <class P>
AMSM_DEV void jac_madd(Jac<P>& acc, const Affine<P>& q) {
  if (affine_is_inf<P>(q)) return;
  if (jac_is_inf<P>(acc)) {
    acc.x = q.x;
    acc.y = fe_tight<P>(q.y);
    acc.z = fe_one<P>();
    return;
  }
  Fe<P> zz = fe_sqr<P>(acc.z);                                // [< 1.1p]
  Fe<P> zzz = fe_mul<P>(acc.z, zz);                           // [< 1.1p]
  Fe<P> h = fe_mul_sub_k<P, 13>(q.x, zz, acc.x);              // u2 - X1 (+13p)  [< 14.1p]
  Fe<P> r = fe_mul_sub_k<P, 14>(q.y, zzz, acc.y);             // s2 - Y1 (+14p)  [< 15.1p]
  if (fe_is_zero_mod<P, 16>(h)) {
    if (fe_is_zero_mod<P, 16>(r)) {
      Jac<P> d;
      d.x = q.x;
      d.y = fe_tight<P>(q.y);
      d.z = fe_one<P>();
      acc = jac_dbl<P>(d);
    } else {
      acc = jac_inf<P>();
    }
    return;
  }
  Fe<P> hh = fe_sqr<P>(h);                                    // [< 2.6p]
  Fe<P> hhh = fe_mul<P>(h, hh);                               // [< 1.3p]
  Fe<P> v = fe_mul<P>(acc.x, hh);                             // [< 1.3p]
  Fe<P> x3 = fe_sqr_sub_bcc_k<P, 4>(r, hhh, v);               // r^2 [< 2.8p] - hhh - 2v [< 3.9p] (+4p)  [< 6.8p]
  Fe<P> t = fe_sub_k<P, 8>(v, x3);                            // [< 9.3p]
  Fe<P> y3 = fe_mul_sub_mul_k<P, 16>(r, t, acc.y, hhh);       // (r t + (16p - Y1) hhh) / R': (141 + 21) / 128  [< 2.3p]
  acc.z = fe_mul<P>(acc.z, h);                                // [< 1.4p]
  acc.x = x3;
  acc.y = y3;
}
// -> XYZZ (ZZ = Z^2, ZZZ = Z^3), X and Y brought back under the XYZZ invariant by one multiplication by one each
template <class P>
AMSM_DEV XYZZ<P> xyzz_from_jac(const Jac<P>& p) {
  if (jac_is_inf<P>(p)) return xyzz_inf<P>();
  XYZZ<P> r;
  const Fe<P> one = fe_one<P>();
  r.zz = fe_sqr<P>(p.z);              // [< 1.1p]
  r.zzz = fe_mul<P>(p.z, r.zz);       // [< 1.1p]
  r.x = fe_mul<P>(p.x, one);          // [< 1.1p]
  r.y = fe_mul<P>(p.y, one);          // [< 1.2p]
  return r;
}

// ---------------------------------------------------------------------------------------------
// Quad-cooperative group law for the latency-bound tail kernels (bucket reduce, fold): the four lanes of an aligned quad
// hold the SAME operands (replicated state); the independent field multiplications of one level of the formula run on
// different lanes of the quad and the products are broadcast back, so a full addition is 4 multiplication-times deep
// instead of 14 and a doubling 3 instead of 9 (a squaring rides as a product on a level that has a free lane: one
// level less than squaring first).  The operations and their operand roles are those of xyzz_add / xyzz_dbl above, so
// the same bounds hold; only the schedule differs.  A lone
// product on a lane of a fused level is written as the two-product form with a zero second product.
// All four lanes of the quad must be active and hold equal operands.
template <class P, int K>
AMSM_DEV Fe<P> quad_bcast(const Fe<P>& v) {  // lane K of the quad -> all four lanes: one v_mov_b32 quad_perm per limb
  Fe<P> r;
  constexpr int ctrl = K | (K << 2) | (K << 4) | (K << 6);  // DPP quad_perm:[K,K,K,K]
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = (u32)__builtin_amdgcn_update_dpp(0, (int)v.v[i], ctrl, 0xf, 0xf, false);
  return r;
}
template <class P>
AMSM_DEV Fe<P> fe_sel(bool c, const Fe<P>& a, const Fe<P>& b) {  // c ? a : b
  Fe<P> r;
#pragma unroll
  for (int i = 0; i < P::L; i++) r.v[i] = c ? a.v[i] : b.v[i];
  return r;
}
template <class P>
AMSM_DEV Fe<P> fe_sel4(u32 k, const Fe<P>& a0, const Fe<P>& a1, const Fe<P>& a2, const Fe<P>& a3) {
  return fe_sel<P>(k < 2u, fe_sel<P>(k == 0u, a0, a1), fe_sel<P>(k == 2u, a2, a3));
}

template <class P>
AMSM_DEV XYZZ<P> xyzz_dbl_quad(const XYZZ<P>& p) {
  if (xyzz_is_inf<P>(p)) return p;
  const u32 k = __lane_id() & 3u;
  const Fe<P> zero = fe_zero<P>();
  Fe<P> u = fe_dbl<P>(p.y);
  Fe<P> sq = fe_sqr<P>(fe_sel<P>(k == 1u, p.x, u));  // lane 0: v = u^2, lane 1: xx = x^2
  Fe<P> v = quad_bcast<P, 0>(sq), xx = quad_bcast<P, 1>(sq);
  Fe<P> m = fe_triple<P>(xx);
  // lane 0: w = u v, lane 1: s = x v, lane 2: zz3 = v zz, lane 3: mm = m^2 (as a product: one level instead of two)
  Fe<P> m2 = fe_mul<P>(fe_sel4<P>(k, u, p.x, v, m), fe_sel4<P>(k, v, v, p.zz, m));
  Fe<P> w = quad_bcast<P, 0>(m2), s = quad_bcast<P, 1>(m2), mm = quad_bcast<P, 3>(m2);
  XYZZ<P> r;
  r.zz = quad_bcast<P, 2>(m2);
  r.x = fe_sub_bcc_k<P, 4>(mm, zero, s);
  Fe<P> t = fe_sub_k<P, 8>(s, r.x);
  // lane 0: y3 = m t + (4p - y) w, lane 1: zzz3 = w zzz (+ 0)
  Fe<P> m3 = fe_mul_sub_mul_k<P, 4>(fe_sel<P>(k == 0u, m, w), fe_sel<P>(k == 0u, t, p.zzz), fe_sel<P>(k == 0u, p.y, zero),
                                    fe_sel<P>(k == 0u, w, zero));
  r.y = quad_bcast<P, 0>(m3);
  r.zzz = quad_bcast<P, 1>(m3);
  return r;
}

template <class P>
AMSM_DEV void xyzz_add_quad(XYZZ<P>& acc, const XYZZ<P>& q) {
  if (xyzz_is_inf<P>(q)) return;
  if (xyzz_is_inf<P>(acc)) {
    acc = q;
    return;
  }
  const u32 k = __lane_id() & 3u;
  const Fe<P> zero = fe_zero<P>();
  // lane 0: u1 = X1 ZZ2, lane 1: u2 = X2 ZZ1, lane 2: s1 = Y1 ZZZ2, lane 3: s2 = Y2 ZZZ1
  Fe<P> m1 = fe_mul<P>(fe_sel4<P>(k, acc.x, q.x, acc.y, q.y), fe_sel4<P>(k, q.zz, acc.zz, q.zzz, acc.zzz));
  Fe<P> u1 = quad_bcast<P, 0>(m1), u2 = quad_bcast<P, 1>(m1), s1 = quad_bcast<P, 2>(m1), s2 = quad_bcast<P, 3>(m1);
  Fe<P> p = fe_sub_k<P, 2>(u2, u1);
  Fe<P> r = fe_sub_k<P, 2>(s2, s1);
  if (fe_is_zero_mod<P, 4>(p)) {  // uniform over the quad: replicated operands
    if (fe_is_zero_mod<P, 4>(r)) acc = xyzz_dbl_quad<P>(acc);
    else acc = xyzz_inf<P>();
    return;
  }
  // lane 0: pp = p^2, lane 1: rr = r^2 (as products: ONE level for all four), lane 2: ZZ1 ZZ2, lane 3: ZZZ1 ZZZ2
  Fe<P> m2 = fe_mul<P>(fe_sel4<P>(k, p, r, acc.zz, acc.zzz), fe_sel4<P>(k, p, r, q.zz, q.zzz));
  Fe<P> pp = quad_bcast<P, 0>(m2), rr = quad_bcast<P, 1>(m2);
  Fe<P> zz12 = quad_bcast<P, 2>(m2), zzz12 = quad_bcast<P, 3>(m2);
  // lane 0: ppp = p pp, lane 1: qq = u1 pp, lane 2: zz3 = zz12 pp
  Fe<P> m3 = fe_mul<P>(fe_sel4<P>(k, p, u1, zz12, zz12), pp);
  Fe<P> ppp = quad_bcast<P, 0>(m3), qq = quad_bcast<P, 1>(m3);
  acc.zz = quad_bcast<P, 2>(m3);
  Fe<P> x3 = fe_sub_bcc_k<P, 4>(rr, ppp, qq);
  Fe<P> t = fe_sub_k<P, 8>(qq, x3);
  // lane 0: y3 = r t + (2p - s1) ppp, lane 1: zzz3 = zzz12 ppp (+ 0)
  Fe<P> m4 = fe_mul_sub_mul_k<P, 2>(fe_sel<P>(k == 0u, r, zzz12), fe_sel<P>(k == 0u, t, ppp), fe_sel<P>(k == 0u, s1, zero),
                                    fe_sel<P>(k == 0u, ppp, zero));
  acc.x = x3;
  acc.y = quad_bcast<P, 0>(m4);
  acc.zzz = quad_bcast<P, 1>(m4);
}

// XYZZ -> affine with one field inversion (off the hot path: key precomputation only).
template <class P>
AMSM_DEV Affine<P> xyzz_to_affine(const XYZZ<P>& p) {
  Affine<P> r;
  if (xyzz_is_inf<P>(p)) {
    r.x = fe_zero<P>();
    r.y = fe_zero<P>();
    return r;
  }
  // 1/ZZZ, then 1/ZZ = ZZ^2 / ZZZ^2 ... cheaper: inv = 1/(ZZ*ZZZ); 1/ZZ = inv*ZZZ; 1/ZZZ = inv*ZZ
  Fe<P> inv = fe_inv<P>(fe_mul<P>(p.zz, p.zzz));
  r.x = fe_mul<P>(p.x, fe_mul<P>(inv, p.zzz));
  r.y = fe_mul<P>(p.y, fe_mul<P>(inv, p.zz));
  return r;
}

// Loads/stores.  Affine = 2*W u32 contiguous (x|y); XYZZ = 4*W u32 contiguous (canonical values).
template <class P>
AMSM_DEV Affine<P> affine_import(const Affine<P>& a) {  // C-ABI Montgomery radix -> internal
  Affine<P> r;
  r.x = fe_import<P>(a.x);
  r.y = fe_import<P>(a.y);
  return r;
}
template <class P>
AMSM_DEV Affine<P> affine_export(const Affine<P>& a) {  // internal -> C-ABI radix, canonical
  Affine<P> r;
  r.x = fe_export<P>(a.x);
  r.y = fe_export<P>(a.y);
  return r;
}

template <class P>
AMSM_DEV Affine<P> affine_load(const u32* __restrict__ base, size_t idx) {
  const u32* p = base + idx * (2 * P::W);
  Affine<P> r;
  r.x = fe_load<P>(p);
  r.y = fe_load<P>(p + P::W);
  return r;
}

template <class P>
AMSM_DEV void affine_store(u32* __restrict__ base, size_t idx, const Affine<P>& a) {
  u32* p = base + idx * (2 * P::W);
  fe_store<P>(p, a.x);
  fe_store<P>(p + P::W, a.y);
}

template <class P>
AMSM_DEV XYZZ<P> xyzz_load(const u32* __restrict__ base, size_t idx) {
  const u32* p = base + idx * (4 * P::W);
  XYZZ<P> r;
  r.x = fe_load<P>(p);
  r.y = fe_load<P>(p + P::W);
  r.zz = fe_load<P>(p + 2 * P::W);
  r.zzz = fe_load<P>(p + 3 * P::W);
  return r;
}

template <class P>
AMSM_DEV void xyzz_store(u32* __restrict__ base, size_t idx, const XYZZ<P>& a) {
  u32* p = base + idx * (4 * P::W);
  fe_store<P>(p, a.x);
  fe_store<P>(p + P::W, a.y);
  fe_store<P>(p + 2 * P::W, a.zz);
  fe_store<P>(p + 3 * P::W, a.zzz);
}

}  // namespace amsm
