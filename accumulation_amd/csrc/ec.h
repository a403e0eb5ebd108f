// Short-Weierstrass (a = 0) group law for gfx950 in extended-Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> ZZ = 0).
//
// Replaces (device side) ark-ec ^0.2.0 `GroupProjective::{add_assign_mixed, add_assign,
// double_in_place}` used under `VariableBaseMSM::multi_scalar_mul` (source not in
// /root/reference, Cargo.toml:15).  ark-ec uses Jacobian (madd-2007-bl, 7M+4S); XYZZ is chosen here
// because the bucket-accumulation hot loop is multiplier-bound on gfx950 and XYZZ mixed addition is
// 8M+2S.  Intermediate projective values are never compared with the reference -- only the
// canonical affine result is.
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
  Fe<P> u = fe_dbl<P>(p.y);
  Fe<P> v = fe_sqr<P>(u);
  Fe<P> w = fe_mul<P>(u, v);
  Fe<P> s = fe_mul<P>(p.x, v);
  Fe<P> xx = fe_sqr<P>(p.x);
  Fe<P> m = fe_add<P>(fe_dbl<P>(xx), xx);
  XYZZ<P> r;
  r.x = fe_sub<P>(fe_sub<P>(fe_sqr<P>(m), s), s);
  r.y = fe_sub<P>(fe_mul<P>(m, fe_sub<P>(s, r.x)), fe_mul<P>(w, p.y));
  r.zz = fe_mul<P>(v, p.zz);
  r.zzz = fe_mul<P>(w, p.zzz);
  return r;
}

// mdbl-2008-s-1 with a = 0 (affine input): 3M + 3S
template <class P>
AMSM_DEV XYZZ<P> xyzz_dbl_affine(const Affine<P>& p) {
  Fe<P> u = fe_dbl<P>(p.y);
  Fe<P> v = fe_sqr<P>(u);
  Fe<P> w = fe_mul<P>(u, v);
  Fe<P> s = fe_mul<P>(p.x, v);
  Fe<P> xx = fe_sqr<P>(p.x);
  Fe<P> m = fe_add<P>(fe_dbl<P>(xx), xx);
  XYZZ<P> r;
  r.x = fe_sub<P>(fe_sub<P>(fe_sqr<P>(m), s), s);
  r.y = fe_sub<P>(fe_mul<P>(m, fe_sub<P>(s, r.x)), fe_mul<P>(w, p.y));
  r.zz = v;
  r.zzz = w;
  return r;
}

// madd-2008-s: acc += q (q affine, not infinity unless encoded (0,0)); 8M + 2S on the generic path.
// Exceptional cases (acc = inf, q = inf, q = +-acc) are handled exactly: duplicate bases and
// bases summing to zero do occur in the reference's degenerate inputs.
template <class P>
AMSM_DEV void xyzz_madd(XYZZ<P>& acc, const Affine<P>& q) {
  if (affine_is_inf<P>(q)) return;
  if (xyzz_is_inf<P>(acc)) {
    acc.x = q.x;
    acc.y = q.y;
    acc.zz = fe_one<P>();
    acc.zzz = fe_one<P>();
    return;
  }
  Fe<P> u2 = fe_mul<P>(q.x, acc.zz);
  Fe<P> s2 = fe_mul<P>(q.y, acc.zzz);
  Fe<P> p = fe_sub<P>(u2, acc.x);
  Fe<P> r = fe_sub<P>(s2, acc.y);
  if (fe_is_zero<P>(p)) {
    if (fe_is_zero<P>(r)) {
      acc = xyzz_dbl_affine<P>(q);
    } else {
      acc = xyzz_inf<P>();
    }
    return;
  }
  Fe<P> pp = fe_sqr<P>(p);
  Fe<P> ppp = fe_mul<P>(p, pp);
  Fe<P> qq = fe_mul<P>(acc.x, pp);
  Fe<P> x3 = fe_sub<P>(fe_sub<P>(fe_sub<P>(fe_sqr<P>(r), ppp), qq), qq);
  Fe<P> y3 = fe_sub<P>(fe_mul<P>(r, fe_sub<P>(qq, x3)), fe_mul<P>(acc.y, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = fe_mul<P>(acc.zz, pp);
  acc.zzz = fe_mul<P>(acc.zzz, ppp);
}

// add-2008-s: acc += q (both XYZZ); 12M + 2S on the generic path.
template <class P>
AMSM_DEV void xyzz_add(XYZZ<P>& acc, const XYZZ<P>& q) {
  if (xyzz_is_inf<P>(q)) return;
  if (xyzz_is_inf<P>(acc)) {
    acc = q;
    return;
  }
  Fe<P> u1 = fe_mul<P>(acc.x, q.zz);
  Fe<P> u2 = fe_mul<P>(q.x, acc.zz);
  Fe<P> s1 = fe_mul<P>(acc.y, q.zzz);
  Fe<P> s2 = fe_mul<P>(q.y, acc.zzz);
  Fe<P> p = fe_sub<P>(u2, u1);
  Fe<P> r = fe_sub<P>(s2, s1);
  if (fe_is_zero<P>(p)) {
    if (fe_is_zero<P>(r)) {
      acc = xyzz_dbl<P>(acc);
    } else {
      acc = xyzz_inf<P>();
    }
    return;
  }
  Fe<P> pp = fe_sqr<P>(p);
  Fe<P> ppp = fe_mul<P>(p, pp);
  Fe<P> qq = fe_mul<P>(u1, pp);
  Fe<P> x3 = fe_sub<P>(fe_sub<P>(fe_sub<P>(fe_sqr<P>(r), ppp), qq), qq);
  Fe<P> y3 = fe_sub<P>(fe_mul<P>(r, fe_sub<P>(qq, x3)), fe_mul<P>(s1, ppp));
  acc.x = x3;
  acc.y = y3;
  acc.zz = fe_mul<P>(fe_mul<P>(acc.zz, q.zz), pp);
  acc.zzz = fe_mul<P>(fe_mul<P>(acc.zzz, q.zzz), ppp);
}

template <class P>
AMSM_DEV Affine<P> affine_neg_if(const Affine<P>& p, bool negate) {
  Affine<P> r;
  r.x = p.x;
  Fe<P> ny = fe_neg<P>(p.y);
#pragma unroll
  for (int i = 0; i < P::L; i++) r.y.v[i] = negate ? ny.v[i] : p.y.v[i];
  return r;
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

// Loads/stores.  Affine = 2*L u32 contiguous (x|y); XYZZ = 4*L u32 contiguous.
template <class P>
AMSM_DEV Affine<P> affine_load(const u32* __restrict__ base, size_t idx) {
  const u32* p = base + idx * (2 * P::L);
  Affine<P> r;
  r.x = fe_load<P>(p);
  r.y = fe_load<P>(p + P::L);
  return r;
}

template <class P>
AMSM_DEV void affine_store(u32* __restrict__ base, size_t idx, const Affine<P>& a) {
  u32* p = base + idx * (2 * P::L);
  fe_store<P>(p, a.x);
  fe_store<P>(p + P::L, a.y);
}

template <class P>
AMSM_DEV XYZZ<P> xyzz_load(const u32* __restrict__ base, size_t idx) {
  const u32* p = base + idx * (4 * P::L);
  XYZZ<P> r;
  r.x = fe_load<P>(p);
  r.y = fe_load<P>(p + P::L);
  r.zz = fe_load<P>(p + 2 * P::L);
  r.zzz = fe_load<P>(p + 3 * P::L);
  return r;
}

template <class P>
AMSM_DEV void xyzz_store(u32* __restrict__ base, size_t idx, const XYZZ<P>& a) {
  u32* p = base + idx * (4 * P::L);
  fe_store<P>(p, a.x);
  fe_store<P>(p + P::L, a.y);
  fe_store<P>(p + 2 * P::L, a.zz);
  fe_store<P>(p + 3 * P::L, a.zzz);
}

}  // namespace amsm
