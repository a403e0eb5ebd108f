// Scalar-field (Fr) vector kernels: the data-parallel loops of the reference that PRODUCE the MSM
// scalars, so that those vectors are born in HBM and feed `amsm_msm_device` without a PCIe round trip.
// All operands are Montgomery-form 32-byte elements (raw `Vec<Fr>` memory); one lane per element,
// 2 x dwordx4 loads/stores per operand (coalesced 1 KiB per wave-instruction pair).
//
//   k_vec_hadamard : `compute_hp`                   src/hp_as/mod.rs:278-285   (K4, 96 B/elem)
//   k_vec_combine  : `combine_vectors`/`scale_vector` src/hp_as/mod.rs:482-512 (K6, 32(n+1) B/elem)
//   k_hp_t_vecs    : `compute_t_vecs`               src/hp_as/mod.rs:288-349   (K5, fused: reads 2n
//                    vectors once, writes the 2n-2 committed coefficient vectors)
#pragma once
#include "fp.h"
#include "msm_types.h"
#include "rng.h"

namespace amsm {

// One window of the signed-digit walk: takes the low bits of s, returns the bucket digit d (0: none) and its sign, leaves the
// recoding's carry for the next window and shifts s.  Window w is c bits wide, or c - 1 (MsmGeom::n_narrow: the top windows,
// digit doubled); the legacy short top window of a top_shift key is spread by its shift.  A scalar that does not fit the
// windows (non-canonical: >= 2^255) leaves carry = 1 behind the last window -- the callers' `rest`.
// MsmGeom::skip_ones: a scalar that IS the unit AS STORED (canonical 1, or the Montgomery form of 1) becomes 0 -- exactly the test
// k_tv_sum applies when it sums those scalars' generators apart (a non-reduced representation of 1 fails both and stays in the windows)
template <class Fr>
AMSM_DEV void fe_drop_stored_one(Fe<Fr>& s, int mont) {
  u32 rest = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) rest |= s.v[k] ^ (mont ? Fr::one(k) : (k == 0 ? 1u : 0u));
  if (rest == 0u) {
#pragma unroll
    for (int k = 0; k < 8; k++) s.v[k] = 0u;
  }
}
template <class Fr>
AMSM_DEV u32 digit_step(Fe<Fr>& s, const DigitWalk& dw, u32 w, u32& carry, u32& neg) {
  const u32 c = dw.c, W = dw.W;
  const u32 narrow = (w + dw.n_narrow >= W) ? 1u : 0u;  // uniform over the grid
  const u32 cw = c - narrow;
  const u32 raw = (s.v[0] & ((1u << cw) - 1u)) + carry;
#pragma unroll
  for (int k = 0; k < 7; k++) s.v[k] = (s.v[k] >> cw) | (s.v[k + 1] << (32 - cw));
  s.v[7] >>= cw;
  neg = 0;
  carry = 0;
  u32 d = raw;
  if (dw.top_shift && w == W - 1u) {
    d = raw << (narrow + dw.top_shift);  // MsmGeom::top_shift: the top window (never negative), spread over the bucket range
    if (d > (1u << (c - 1))) {            // only a non-canonical scalar gets here: reported, no entry
      carry = 1;
      d = 0;
    }
    return d;
  } else if (raw > (1u << (cw - 1))) {
    d = (1u << cw) - raw;
    neg = 1;
    carry = 1;
  }
  return d << narrow;
}


// The three streaming kernels below run as a grid-stride loop over a grid sized to the resident wave slots, with the NEXT
// element's operands loaded before the current element's multiplications (register double buffering): with one element per
// lane and one launch-sized grid the waves of a CU move in lock step -- all loading, then all multiplying, then all storing --
// and memory time and multiplier time ADD (round 2: two-vector combination 102 us = 46 us of multiplications + 56 us of
// memory where the Hadamard kernel, one multiplication per 96 bytes, hid its arithmetic completely).
// Outputs are written once and read next by another kernel's first pass over 128 MiB: non-temporal stores.
template <class P>
AMSM_DEV void fe_store_nt(u32* __restrict__ p, const Fe<P>& a) {
  typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t* q = reinterpret_cast<u32x4_t*>(p);
#pragma unroll
  for (int i = 0; i < P::W / 4; i++) {
    u32x4_t v = {a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]};
    __builtin_nontemporal_store(v, q + i);  // global_store_dwordx4 ... nt
  }
}

template <class Fr>
__global__ void __launch_bounds__(256)
    k_vec_hadamard(const u32* __restrict__ a, const u32* __restrict__ b, u32* __restrict__ out, u32 n) {
  const u32 stride = gridDim.x * blockDim.x;
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<Fr> x = fe_load<Fr>(a + (size_t)i * 8), y = fe_load<Fr>(b + (size_t)i * 8);
  for (;;) {
    const u32 nx = i + stride;
    const bool more = nx < n;
    Fe<Fr> x2 = x, y2 = y;
    if (more) {
      x2 = fe_load<Fr>(a + (size_t)nx * 8);
      y2 = fe_load<Fr>(b + (size_t)nx * 8);
    }
    fe_store_nt<Fr>(out + (size_t)i * 8, fe_mul<Fr>(x, y));
    if (!more) break;
    x = x2;
    y = y2;
    i = nx;
  }
}


// sum_{t < NP} pa[t] pb[t] on the 8 x 32 schedule, up to three products per Montgomery reduction (fe_dot2 / fe_dot3, fp.h)
template <class Fr, int NP, int G = 0>
AMSM_DEV Fe<Fr> fe_dot_n(const Fe<Fr>* pa, const Fe<Fr>* pb) {
  constexpr int K = NP - G;
  static_assert(K >= 1, "at least one product");
  if constexpr (K == 1) return fe_mul<Fr>(pa[G], pb[G]);
  else if constexpr (K == 2) return fe_dot2<Fr>(pa[G], pb[G], pa[G + 1], pb[G + 1]);
  else if constexpr (K == 3) return fe_dot3<Fr>(pa[G], pb[G], pa[G + 1], pb[G + 1], pa[G + 2], pb[G + 2]);
  else if constexpr (K == 4) return fe_add<Fr>(fe_dot2<Fr>(pa[G], pb[G], pa[G + 1], pb[G + 1]), fe_dot_n<Fr, NP, G + 2>(pa, pb));
  else return fe_add<Fr>(fe_dot3<Fr>(pa[G], pb[G], pa[G + 1], pb[G + 1], pa[G + 2], pb[G + 2]), fe_dot_n<Fr, NP, G + 3>(pa, pb));
}

// is the kernel-argument coefficient the field's one (Montgomery form)?  Uniform over the grid: the first challenge of every
// linear combination of the schemes is 1 (mu_0, nu^0, beta_0: src/hp_as/mod.rs:241,266, src/r1cs_nark_as/mod.rs:444), and
// skipping its multiplication takes a third off the arithmetic of the common two-vector combination.
template <class Fr>
AMSM_DEV bool coeff_is_one(const u32 c[8]) {
  const Fe<Fr> one = fe_one<Fr>();
  u32 d = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) d |= c[k] ^ one.v[k];
  return d == 0;
}

// NV = number of vectors, compile time: the loads of all operands are issued before the first multiplication (with a
// run-time vector count the loop serialised load -> multiply -> load: 4.2 TB/s for two vectors where the Hadamard kernel,
// with the same 96 bytes per element, reached 5.9).
template <class Fr, int NV>
__global__ void __launch_bounds__(256) k_vec_combine(CombineArgs a, u32* __restrict__ out) {
  const u32 stride = gridDim.x * blockDim.x;
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  auto load = [&](u32 e, Fe<Fr>* x, Fe<Fr>& h) {
#pragma unroll
    for (int j = 0; j < NV; j++) x[j] = e < a.len[j] ? fe_load<Fr>(a.vec[j] + (size_t)e * 8) : fe_zero<Fr>();
    h = (a.hiding && e < a.hiding_len) ? fe_load<Fr>(a.hiding + (size_t)e * 8) : fe_zero<Fr>();
  };
  Fe<Fr> x[NV], acc;
  load(i, x, acc);
  for (;;) {
    const u32 nx = i + stride;
    const bool more = nx < a.n;
    Fe<Fr> x2[NV], h2;
    if (more) load(nx, x2, h2);
    // a unit FIRST coefficient (the schemes' combinations: see coeff_is_one) is an addition; the other products share
    // Montgomery reductions three at a time
    if (coeff_is_one<Fr>(a.coeff[0])) {
      acc = fe_add<Fr>(acc, x[0]);
      if constexpr (NV > 1) {
        Fe<Fr> cf[NV - 1];
#pragma unroll
        for (int j = 1; j < NV; j++)
#pragma unroll
          for (int k = 0; k < 8; k++) cf[j - 1].v[k] = a.coeff[j][k];
        acc = fe_add<Fr>(acc, fe_dot_n<Fr, NV - 1>(cf, x + 1));
      }
    } else {
      Fe<Fr> cf[NV];
#pragma unroll
      for (int j = 0; j < NV; j++)
#pragma unroll
        for (int k = 0; k < 8; k++) cf[j].v[k] = a.coeff[j][k];
      acc = fe_add<Fr>(acc, fe_dot_n<Fr, NV>(cf, x));
    }
    // in-place chunked combination (more than VEC_MAX vectors) re-reads `out` as the hiding addend: plain store there
    if (a.hiding == out) fe_store<Fr>(out + (size_t)i * 8, acc);
    else fe_store_nt<Fr>(out + (size_t)i * 8, acc);
    if (!more) break;
#pragma unroll
    for (int j = 0; j < NV; j++) x[j] = x2[j];
    acc = h2;
    i = nx;
  }
}


// N = number of inputs (compile time so the N x N product stays in registers)
template <class Fr, int N>
__global__ void __launch_bounds__(256) k_hp_t_vecs(TVecArgs a) {
  const u32 stride = gridDim.x * blockDim.x;
  u32 li = blockIdx.x * blockDim.x + threadIdx.x;
  if (li >= a.len) return;
  // raw operands of element e: a_j, b_j and the two hiding vectors (zero where a vector is shorter: `.get(li)`, :306-318)
  auto load = [&](u32 e, Fe<Fr>* ra, Fe<Fr>* rb, Fe<Fr>& ha, Fe<Fr>& hb) {
#pragma unroll
    for (int j = 0; j < N; j++) {
      ra[j] = e < a.a_len[j] ? fe_load<Fr>(a.a[j] + (size_t)e * 8) : fe_zero<Fr>();
      rb[j] = e < a.b_len[j] ? fe_load<Fr>(a.b[j] + (size_t)e * 8) : fe_zero<Fr>();
    }
    ha = (a.hiding_a && e < a.hiding_a_len) ? fe_load<Fr>(a.hiding_a + (size_t)e * 8) : fe_zero<Fr>();
    hb = (a.hiding_b && e < a.hiding_b_len) ? fe_load<Fr>(a.hiding_b + (size_t)e * 8) : fe_zero<Fr>();
  };
  Fe<Fr> ra[N], rb[N], ha, hb;
  load(li, ra, rb, ha, hb);
  for (;;) {
    const u32 nx = li + stride;
    const bool more = nx < a.len;
    Fe<Fr> ra2[N], rb2[N], ha2, hb2;
    if (more) load(nx, ra2, rb2, ha2, hb2);
    Fe<Fr> ac[N], bc[N];
#pragma unroll
    for (int j = 0; j < N; j++) {
      ac[j] = ra[j];
      if (!coeff_is_one<Fr>(a.mu[j])) {  // mu_0 = 1 (src/hp_as/mod.rs:241): uniform branch
        Fe<Fr> mu;
#pragma unroll
        for (int k = 0; k < 8; k++) mu.v[k] = a.mu[j][k];
        ac[j] = fe_mul<Fr>(mu, ac[j]);
      }
      // b coefficients are reversed: B(X) = sum_j b_{N-1-j} X^j   (src/hp_as/mod.rs:320)
      bc[N - 1 - j] = rb[j];
    }
    if (a.hiding_a) {  // a_coeffs[0] += hiding_a[li] * mu[N]   (:322-325)
      Fe<Fr> mu;
#pragma unroll
      for (int k = 0; k < 8; k++) mu.v[k] = a.mu[N][k];
      ac[0] = fe_add<Fr>(ac[0], fe_mul<Fr>(ha, mu));
    }
    if (a.hiding_b) {  // b_coeffs[0] += hiding_b[li] * mu[1]   (:327-329)
      Fe<Fr> mu;
#pragma unroll
      for (int k = 0; k < 8; k++) mu.v[k] = a.mu[1][k];
      bc[0] = fe_add<Fr>(bc[0], fe_mul<Fr>(hb, mu));
    }
#pragma unroll
    for (int k = 0; k < 2 * N - 1; k++) {
      if (a.t[k] == nullptr) continue;  // uniform branch: coefficient N-1 is never committed
      // coefficient k = sum_{i + j = k} ac[i] bc[j]: i from lo, CNT terms, ONE reduction per three products
      const int lo = k < N ? 0 : k - (N - 1), cnt = (k < N ? k : N - 1) - lo + 1;
      Fe<Fr> pa[N], pb[N];
#pragma unroll
      for (int t = 0; t < N; t++) {
        pa[t] = ac[t < cnt ? lo + t : 0];
        pb[t] = bc[t < cnt ? k - lo - t : 0];
      }
      Fe<Fr> s = fe_zero<Fr>();
#pragma unroll
      for (int g = 0; g < N; g += 3) {
        if (g >= cnt) continue;
        const int rem = cnt - g, g1 = g + 1 < N ? g + 1 : N - 1, g2 = g + 2 < N ? g + 2 : N - 1;
        Fe<Fr> part;
        if (rem >= 3) part = fe_dot3<Fr>(pa[g], pb[g], pa[g1], pb[g1], pa[g2], pb[g2]);
        else if (rem == 2) part = fe_dot2<Fr>(pa[g], pb[g], pa[g1], pb[g1]);
        else part = fe_mul<Fr>(pa[g], pb[g]);
        s = g == 0 ? part : fe_add<Fr>(s, part);
      }
      fe_store_nt<Fr>(a.t[k] + (size_t)li * 8, s);
    }
    if (!more) break;
#pragma unroll
    for (int j = 0; j < N; j++) {
      ra[j] = ra2[j];
      rb[j] = rb2[j];
    }
    ha = ha2;
    hb = hb2;
    li = nx;
  }
}

// Row-sparse R1CS matrix times (input || witness): `matrix_vec_mul` / `inner_prod`,
// src/r1cs_nark_as/r1cs_nark/mod.rs:443-462 (K7).  CSR in HBM: row_ptr (rows+1), col (nnz), val (nnz x 8 u32,
// Montgomery).  One lane per row; the reference's "skip the multiplication when the coefficient is one"
// (:459) is arithmetic-neutral and kept (it saves a Montgomery multiplication for the common +-1 entries).
template <class Fr>
__global__ void __launch_bounds__(256)
    k_spmv(const u32* __restrict__ row_ptr, const u32* __restrict__ col, const u32* __restrict__ val,
           const u32* __restrict__ input, u32 n_input, const u32* __restrict__ witness, u32 n_witness,
           u32* __restrict__ out, u32 n_rows) {
  u32 r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  Fe<Fr> acc = fe_zero<Fr>();
  Fe<Fr> one = fe_one<Fr>();
  for (u32 k = row_ptr[r]; k < row_ptr[r + 1]; k++) {
    u32 i = col[k];
    Fe<Fr> x;
    if (i < n_input) x = fe_load<Fr>(input + (size_t)i * 8);
    else if (i - n_input < n_witness) x = fe_load<Fr>(witness + (size_t)(i - n_input) * 8);
    else x = fe_zero<Fr>();
    Fe<Fr> cf = fe_load<Fr>(val + (size_t)k * 8);
    acc = fe_add<Fr>(acc, fe_eq<Fr>(cf, one) ? x : fe_mul<Fr>(x, cf));
  }
  fe_store<Fr>(out + (size_t)r * 8, acc);
}

// ---- kernels for the inner-product-argument opening (ark_poly_commit::ipa_pc, ext; reached from
// src/ipa_pc_as/mod.rs:400,418,454,836 -- SURVEY.md section 2.1 K8) --------------------------------------
// out[i] = point^i (Montgomery), i < n: the evaluation vector z of `open`; square-and-multiply over the bits of i
template <class Fr>
__global__ void __launch_bounds__(256) k_vec_powers(CombineArgs a, u32* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  Fe<Fr> base, acc = fe_one<Fr>();
#pragma unroll
  for (int k = 0; k < 8; k++) base.v[k] = a.coeff[0][k];
  for (u32 e = i; e != 0; e >>= 1) {
    if (e & 1u) acc = fe_mul<Fr>(acc, base);
    base = fe_sqr<Fr>(base);
  }
  fe_store<Fr>(out + (size_t)i * 8, acc);
}

// block partial sums of sum_i a[i]*b[i]; one Montgomery element per workgroup, folded on the host
template <class Fr>
__global__ void __launch_bounds__(256)
    k_vec_inner_product(const u32* __restrict__ a, const u32* __restrict__ b, u32 n, u32* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) u32 lds[256 * 8];
  Fe<Fr> acc = fe_zero<Fr>();
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    acc = fe_add<Fr>(acc, fe_mul<Fr>(fe_load<Fr>(a + (size_t)i * 8), fe_load<Fr>(b + (size_t)i * 8)));
  fe_store<Fr>(lds + threadIdx.x * 8, acc);
  __syncthreads();
  for (u32 s = 128; s >= 1; s >>= 1) {
    if (threadIdx.x < s) {
      Fe<Fr> x = fe_load<Fr>(lds + threadIdx.x * 8), y = fe_load<Fr>(lds + (threadIdx.x + s) * 8);
      fe_store<Fr>(lds + threadIdx.x * 8, fe_add<Fr>(x, y));
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) fe_store<Fr>(out + (size_t)blockIdx.x * 8, fe_load<Fr>(lds));
}

// two inner products in one launch (an IPA round's <c_r, z_l> and <c_l, z_r>): blockIdx.y picks the pair;
// out[y * gridDim.x + blockIdx.x] = this workgroup's partial sum
template <class Fr>
__global__ void __launch_bounds__(256)
    k_vec_inner_product_pair(const u32* __restrict__ a0, const u32* __restrict__ b0, const u32* __restrict__ a1,
                             const u32* __restrict__ b1, u32 n, u32* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) u32 lds[256 * 8];
  const u32* a = blockIdx.y ? a1 : a0;
  const u32* b = blockIdx.y ? b1 : b0;
  Fe<Fr> acc = fe_zero<Fr>();
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    acc = fe_add<Fr>(acc, fe_mul<Fr>(fe_load<Fr>(a + (size_t)i * 8), fe_load<Fr>(b + (size_t)i * 8)));
  fe_store<Fr>(lds + threadIdx.x * 8, acc);
  __syncthreads();
  for (u32 s = 128; s >= 1; s >>= 1) {
    if (threadIdx.x < s) {
      Fe<Fr> x = fe_load<Fr>(lds + threadIdx.x * 8), y = fe_load<Fr>(lds + (threadIdx.x + s) * 8);
      fe_store<Fr>(lds + threadIdx.x * 8, fe_add<Fr>(x, y));
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) fe_store<Fr>(out + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8, fe_load<Fr>(lds));
}

// An IPA round's vector work in ONE launch (three dispatches otherwise, each a launch latency on an idle GPU): the previous
// round's fold, in place, c[i] += x^-1 c[cur + i], z[i] += x z[cur + i] (i < cur = 2 half), and on the folded vectors the
// two inner products <c_r, z_l>, <c_l, z_r> (halves of length `half`).  Lane i owns elements i and half + i of both
// vectors: it reads and writes nothing another lane touches.  out: as k_vec_inner_product_pair.
struct IpaFoldArgs {
  u32 x[8], xinv[8];
};
template <class Fr>
__global__ void __launch_bounds__(256) k_ipa_fold_ip(u32* __restrict__ c, u32* __restrict__ z, u32 half, IpaFoldArgs a,
                                                     u32* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) u32 lds[256 * 8];
  Fe<Fr> x, xinv;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    x.v[k] = a.x[k];
    xinv.v[k] = a.xinv[k];
  }
  const size_t cur = (size_t)2 * half;
  Fe<Fr> acc0 = fe_zero<Fr>(), acc1 = fe_zero<Fr>();
  for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < half; i += gridDim.x * blockDim.x) {
    const size_t lo = (size_t)i * 8, hi = ((size_t)half + i) * 8;
    Fe<Fr> cl = fe_add<Fr>(fe_load<Fr>(c + lo), fe_mul<Fr>(xinv, fe_load<Fr>(c + cur * 8 + lo)));
    Fe<Fr> cr = fe_add<Fr>(fe_load<Fr>(c + hi), fe_mul<Fr>(xinv, fe_load<Fr>(c + cur * 8 + hi)));
    Fe<Fr> zl = fe_add<Fr>(fe_load<Fr>(z + lo), fe_mul<Fr>(x, fe_load<Fr>(z + cur * 8 + lo)));
    Fe<Fr> zr = fe_add<Fr>(fe_load<Fr>(z + hi), fe_mul<Fr>(x, fe_load<Fr>(z + cur * 8 + hi)));
    fe_store<Fr>(c + lo, cl);
    fe_store<Fr>(c + hi, cr);
    fe_store<Fr>(z + lo, zl);
    fe_store<Fr>(z + hi, zr);
    acc0 = fe_add<Fr>(acc0, fe_mul<Fr>(cr, zl));
    acc1 = fe_add<Fr>(acc1, fe_mul<Fr>(cl, zr));
  }
#pragma unroll 1
  for (int which = 0; which < 2; which++) {
    fe_store<Fr>(lds + threadIdx.x * 8, which ? acc1 : acc0);
    __syncthreads();
    for (u32 s = 128; s >= 1; s >>= 1) {
      if (threadIdx.x < s) {
        Fe<Fr> p = fe_load<Fr>(lds + threadIdx.x * 8), q = fe_load<Fr>(lds + (threadIdx.x + s) * 8);
        fe_store<Fr>(lds + threadIdx.x * 8, fe_add<Fr>(p, q));
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) fe_store<Fr>(out + ((size_t)which * gridDim.x + blockIdx.x) * 8, fe_load<Fr>(lds));
    __syncthreads();
  }
}

// coefficients of the succinct-check polynomial h(X) = prod_{i=1..k} (1 + xi_i X^(2^(k-i)))
// (`SuccinctCheckPolynomial::compute_coeffs`, ext; called at src/ipa_pc_as/mod.rs:400): coefficient p is the
// product of the challenges xi_i whose bit (k-i) is set in p.  challenges in a.coeff[0..k) (k <= VEC_MAX*4).
struct CheckPolyArgs {
  u32 xi[32][8];
  u32 k;
};
template <class Fr>
__global__ void __launch_bounds__(256) k_check_poly_coeffs(CheckPolyArgs a, u32* __restrict__ out) {
  u32 p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (1u << a.k)) return;
  Fe<Fr> acc = fe_one<Fr>();
  for (u32 i = 1; i <= a.k; i++) {
    if ((p >> (a.k - i)) & 1u) {
      Fe<Fr> x;
#pragma unroll
      for (int q = 0; q < 8; q++) x.v[q] = a.xi[i - 1][q];
      acc = fe_mul<Fr>(acc, x);
    }
  }
  fe_store<Fr>(out + (size_t)p * 8, acc);
}

// Scalars of the two cross commitments of IPA round j over the ORIGINAL key (no key folding between rounds).
// The key of round j is G^(j)[i] = sum over the top-j-bit patterns b of (prod_t x_t^{b_t}) * G[i + sum_t b_t n/2^(t+1)],
// so with s_j(k) = product of the challenges x_t (t < j) whose bit (log n - 1 - t) is set in k, and h = n / 2^(j+1):
//   L_j = <c_r, G_l^(j)> = sum over k with bit (log n - 1 - j) clear of  c[h + (k mod h)] * s_j(k) * G[k]
//   R_j = <c_l, G_r^(j)> = sum over k with that bit set             of  c[(k mod h)]     * s_j(k) * G[k]
// (c = the current, folded coefficient vector of length 2h).  Each output has n/2 zeros, which the MSM skips.
template <class Fr>
__global__ void __launch_bounds__(256)
    k_ipa_round_scalars(CheckPolyArgs a, u32 j, u32 log_n, const u32* __restrict__ c, u32* __restrict__ out_l,
                        u32* __restrict__ out_r) {
  u32 k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (1u << log_n)) return;
  Fe<Fr> s = fe_one<Fr>();
  for (u32 t = 0; t < j; t++) {
    if ((k >> (log_n - 1 - t)) & 1u) {
      Fe<Fr> x;
#pragma unroll
      for (int q = 0; q < 8; q++) x.v[q] = a.xi[t][q];
      s = fe_mul<Fr>(s, x);
    }
  }
  const u32 h = 1u << (log_n - 1 - j);
  const u32 idx = k & (h - 1u);
  const bool right = (k >> (log_n - 1 - j)) & 1u;
  Fe<Fr> v = fe_mul<Fr>(fe_load<Fr>(c + (size_t)(right ? idx : h + idx) * 8), s);
  if (out_r == nullptr) {  // one vector for amsm_msm_grouped_device (group = that bit of k)
    fe_store<Fr>(out_l + (size_t)k * 8, v);
    return;
  }
  Fe<Fr> z = fe_zero<Fr>();
  fe_store<Fr>(out_l + (size_t)k * 8, right ? z : v);
  fe_store<Fr>(out_r + (size_t)k * 8, right ? v : z);
}

// out[i] = value  (the reference's `vec![F::rand(rng); len]` hiding vectors, src/hp_as/mod.rs:189-190)
__global__ void __launch_bounds__(256) k_vec_fill(u32* __restrict__ out, uint4 lo, uint4 hi, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint4* q = reinterpret_cast<uint4*>(out + (size_t)i * 8);
  q[0] = lo;
  q[1] = hi;
}

template <class Fr>
__global__ void __launch_bounds__(256) k_vec_random(u32* __restrict__ out, u64 seed, u32 n, int mont) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Fe<Fr> s;
  rng_scalar_fr<Fr>(seed, i, s.v);
  if (mont) s = fe_to_mont<Fr>(s);
  fe_store<Fr>(out + (size_t)i * 8, s);
}

// ---------------------------------------------------------------------------------------------
// Skew probe (round 3): would this scalar vector overflow the bucket-per-lane prep?  1024 evenly spaced samples, signed
// c-bit digits as the prep computes them; per window a 2048-bin histogram of the non-zero digits (hashed) in LDS.  Uniform
// digits put at most ~6 samples into a bin; a constant vector (`vec![x; len]`), a vector with a large share of
// equal or of few distinct values puts hundreds.  *flag = 1 when any bin reaches PROBE_LIMIT.  An optimisation only: the
// host then sends the vector straight to the chunked pipeline instead of finding out from the prep's overflow flag (which
// stays the safety net for what 1024 samples miss).  One 1024-lane workgroup.
// ---------------------------------------------------------------------------------------------
constexpr u32 PROBE_SAMPLES = 1024, PROBE_BINS = 2048, PROBE_LIMIT = 24;
template <class Fr>
// tv (may be null): the vector's k_tv_probe words, written earlier on the same stream -- a two-valued vector never reaches the
// pipelines this probe protects, so there is nothing to look at
__global__ void __launch_bounds__(1024) k_skew_probe(const u32* __restrict__ scalars, int mont, u32 n, DigitWalk dw,
                                                      u32* __restrict__ flag, const u32* __restrict__ tv) {
  __shared__ u32 bins[PROBE_BINS];
  __shared__ u32 worst;
  if (tv && tv[0] == 0u && (tv[1] == 1u || tv[2] == 1u)) return;
  const u32 t = threadIdx.x;
  if (t == 0) worst = 0;
  const u32 i = (u32)(((u64)t * n) / PROBE_SAMPLES);
  Fe<Fr> s = fe_load<Fr>(scalars + (size_t)i * 8);
  if (dw.skip_ones) fe_drop_stored_one<Fr>(s, mont);
  if (mont) s = fe_from_mont<Fr>(s);
  u32 carry = 0;
  dw.top_shift = 0;  // (the legacy short top window is looked at unshifted: the histogram only asks how many samples share a digit)
  for (u32 w = 0; w < dw.W; w++) {
    for (u32 k = t; k < PROBE_BINS; k += blockDim.x) bins[k] = 0;
    __syncthreads();
    u32 neg;
    const u32 d = digit_step<Fr>(s, dw, w, carry, neg);
    u32 seen = 0;
    if (d != 0 && t < n) seen = atomicAdd(&bins[(d * 2654435761u) >> 21], 1u) + 1u;
    if (seen >= PROBE_LIMIT) atomicMax(&worst, seen);
    __syncthreads();
    if (worst) break;  // (uniform: read behind the barrier) a constant vector's 1024 samples serialise on ONE bin counter --
                       // 72 us for the 16 windows of a 2^16-pair key; the first window says all there is to say
  }
  if (t == 0 && worst) *flag = 1u;
}

// ---------------------------------------------------------------------------------------------
// Two-valued vectors: is every scalar of the vector 0 or ONE other value v?  (ark-ec's multi_scalar_mul adds the bases of unit
// scalars directly and skips zero ones -- SURVEY.md App. C; the reference's own harnesses go further: hp_as inputs are
// `vec![rand; n]` (src/hp_as/mod.rs:189-190) and the DummyCircuit's A z, B z, C z are one value per row plus a zero row
// (src/r1cs_nark_as/mod.rs:1159-1188), boolean witnesses are {0, 1}.)  Then sum s_i G_i = v * sum_{s_i = v} G_i: n mixed
// additions and one scalar multiplication instead of n windows' worth (k_tv_sum, msm_kernels.h).
// Up to TV_EXC_MAX scalars may be something else (the accumulated vectors of the DummyCircuit end in ONE other value where the
// zero row picked up a blinding term): their indices are listed and the host adds s_j G_j for each.
// EXACT test, not a sample: v = the majority of the first three non-zero scalars among the first TV_HEAD; every lane compares its
// scalars with 0 and v; a wave that meets more than TV_EXC_MAX others at once -- any wave of a uniform vector, on its first load --
// raises out[0] and the kernel ends.
// out (TV_WORDS words, zeroed by the caller): [0] not of this form, [1] v found, [2] the head is zero (with [0] == 0: the whole
// vector is), [3] exceptions, [8..15] v as stored, [16..23] the exceptions' indices.
// ---------------------------------------------------------------------------------------------
constexpr u32 TV_HEAD = 1024, TV_EXC_MAX = 8, TV_WORDS = 32;
AMSM_DEV bool tv_words_equal(const u32* a, const u32* b) {
  bool eq = true;
#pragma unroll
  for (int k = 0; k < 8; k++) eq = eq && a[k] == b[k];
  return eq;
}
// out[4] (round 5): how many of TV_ONES_SAMPLES evenly spaced scalars equal ONE as stored (`one`: the unit in the vector's
// representation, canonical or Montgomery) -- a vector with a share of unit scalars (boolean wires of an R1CS witness) is not
// two-valued, but its ones are summed apart (MsmGeom::skip_ones).
constexpr u32 TV_ONES_SAMPLES = 1024;
struct TvOne {
  u32 w[8];
};
__global__ void __launch_bounds__(256) k_tv_probe(const u32* __restrict__ scalars, u32 n, u32* __restrict__ out, TvOne one) {
  __shared__ u32 cand[3];
  __shared__ u32 minv;
  __shared__ u32 vw[8];
  const u32 t = threadIdx.x;
  const uint4* s4 = (const uint4*)scalars;
  if (blockIdx.x == 0) {  // (before any early exit below)
    u32 cnt = 0;
    for (u32 k = 0; k < TV_ONES_SAMPLES / 256u; k++) {
      const u32 i = (u32)(((unsigned long long)(t + 256u * k) * n) / TV_ONES_SAMPLES);
      const uint4 a = s4[2 * (size_t)i], b = s4[2 * (size_t)i + 1];
      if (a.x == one.w[0] && a.y == one.w[1] && a.z == one.w[2] && a.w == one.w[3] && b.x == one.w[4] && b.y == one.w[5] &&
          b.z == one.w[6] && b.w == one.w[7])
        cnt++;
    }
    if (cnt) atomicAdd(&out[4], cnt);
  }
  // which of this lane's (at most four) head scalars are non-zero
  u32 nz = 0;
  for (u32 k = 0; k < TV_HEAD / 256u; k++) {
    const u32 i = t + 256u * k;
    if (i < n && i < TV_HEAD) {
      const uint4 a = s4[2 * (size_t)i], b = s4[2 * (size_t)i + 1];
      if ((a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) != 0u) nz |= 1u << k;
    }
  }
  // the three smallest non-zero head indices
  u32 prev = 0;
  for (u32 pass = 0; pass < 3u; pass++) {
    if (t == 0) minv = 0xffffffffu;
    __syncthreads();
    for (u32 k = 0; k < TV_HEAD / 256u; k++) {
      const u32 i = t + 256u * k;
      if (((nz >> k) & 1u) && (pass == 0u || i > prev)) {
        atomicMin(&minv, i);
        break;
      }
    }
    __syncthreads();
    if (t == 0) cand[pass] = minv;
    prev = minv;
    __syncthreads();
  }
  const u32 f = cand[0];
  if (f == 0xffffffffu) {  // head all zero: of this form only if EVERYTHING is zero (out[2]); a non-zero scalar further on -> out[0]
    const u32 stride0 = gridDim.x * 256u;
    volatile u32* bad0 = out;
    for (u32 i = blockIdx.x * 256u + t; i < n; i += stride0) {
      const uint4 a = s4[2 * (size_t)i], b = s4[2 * (size_t)i + 1];
      if ((a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) != 0u) {
        out[0] = 1u;
        return;
      }
      if (*bad0) return;
    }
    if (blockIdx.x == 0 && t == 0) out[2] = 1u;
    return;
  }
  if (t == 0) {  // majority of the (up to) three candidates: an odd value FIRST must not be taken for v
    u32 pick = cand[0];
    if (cand[1] != 0xffffffffu && cand[2] != 0xffffffffu) {
      const u32* s0 = scalars + (size_t)cand[0] * 8;
      const u32* s1 = scalars + (size_t)cand[1] * 8;
      const u32* s2 = scalars + (size_t)cand[2] * 8;
      if (!tv_words_equal(s0, s1) && !tv_words_equal(s0, s2) && tv_words_equal(s1, s2)) pick = cand[1];
    }
    for (int k = 0; k < 8; k++) vw[k] = scalars[(size_t)pick * 8 + k];
  }
  __syncthreads();
  const uint4 va = make_uint4(vw[0], vw[1], vw[2], vw[3]), vb = make_uint4(vw[4], vw[5], vw[6], vw[7]);
  if (blockIdx.x == 0 && t < 8) out[8 + t] = vw[t];
  if (blockIdx.x == 0 && t == 0) out[1] = 1u;
  const u32 stride = gridDim.x * 256u, lane = t & 63u;
  volatile u32* bad = out;
  for (u32 i = blockIdx.x * 256u + t; i < n; i += stride) {
    const uint4 a = s4[2 * (size_t)i], b = s4[2 * (size_t)i + 1];
    const bool zero = (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w) == 0u;
    const bool same = a.x == va.x && a.y == va.y && a.z == va.z && a.w == va.w && b.x == vb.x && b.y == vb.y && b.z == vb.z &&
                      b.w == vb.w;
    const bool exc = !zero && !same;
    const unsigned long long mask = __ballot(exc);  // over the lanes still in the loop
    if (mask) {
      const u32 cnt = (u32)__popcll(mask);
      if (cnt > TV_EXC_MAX) {
        out[0] = 1u;
        return;
      }
      const u32 leader = (u32)__ffsll((long long)mask) - 1u;
      u32 base = 0;
      if (lane == leader) base = atomicAdd(&out[3], cnt);
      base = (u32)__shfl((int)base, (int)leader, 64);
      if (base + cnt > TV_EXC_MAX) {
        out[0] = 1u;
        return;
      }
      if (exc) out[16u + base + (u32)__popcll(mask & ((1ull << lane) - 1ull))] = i;
    }
    if (*bad) return;  // some wave gave up: nothing left to learn
  }
}

// ---------------------------------------------------------------------------------------------
// digits: scalar -> W signed digits -> (key, value) entries, window-major (entry w*n + i).
// Reads are 2 x dwordx4 per lane, coalesced; writes are coalesced per window.
// ---------------------------------------------------------------------------------------------
// KeyT = u16 when every bucket id (and the "digit 0" key B) fits 16 bits: 25 % less sort traffic.
template <class Fr, class KeyT>
__global__ void __launch_bounds__(256)
    k_digits(const u32* __restrict__ scalars, int mont, MsmGeom g, KeyT* __restrict__ keys, u32* __restrict__ vals,
             u32* __restrict__ err) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.n) return;
  Fe<Fr> s = fe_load<Fr>(scalars + (size_t)i * 8);
  if (g.skip_ones) fe_drop_stored_one<Fr>(s, mont);
  if (mont) s = fe_from_mont<Fr>(s);
  // first bucket set of this scalar's group (grouped MSM: two sums over index classes in one pass)
  const u32 set0 = ((g.groups > 1u) ? ((i >> g.group_shift) & 1u) : 0u) * sets_per_group(g);
  u32 carry = 0;
  for (u32 w = 0; w < g.W; w++) {
    u32 neg;
    const u32 d = digit_step<Fr>(s, digit_walk_of(g), w, carry, neg);
    u32 set = set0 + window_set(g, w, i);
    u32 idx = g.base_off + i + (g.precomp ? w * g.table_stride : 0u);
    keys[(size_t)w * g.n + i] = (KeyT)(d == 0 ? g.B : set * g.nb + (d - 1));
    vals[(size_t)w * g.n + i] = idx | (neg << 31);
  }
  u32 rest = carry;
#pragma unroll
  for (int k = 0; k < 8; k++) rest |= s.v[k];
  if (rest) atomicOr(err, 1u);
}

// ---------------------------------------------------------------------------------------------
// bounds: start[b] = first sorted entry with key >= b (b = 0..B); items[b] = number of K0-sized chunks
// of the sorted entry list that bucket b's run [start[b], start[b+1]) touches = number of partials
// k_accum_l0 will write for it.  Also tags the last entry of every bucket (bit 30 of its value word).
// ---------------------------------------------------------------------------------------------
template <class KeyT>
AMSM_DEV u32 lower_bound_u32(const KeyT* __restrict__ a, u32 n, u32 x) {
  u32 lo = 0, hi = n;
  while (lo < hi) {
    u32 mid = (lo + hi) >> 1;
    if (a[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo;
}

template <class KeyT>
__global__ void __launch_bounds__(256)
    k_bounds(const KeyT* __restrict__ keys_sorted, u32* __restrict__ vals_sorted, MsmGeom g, u32* __restrict__ start,
             u32* __restrict__ items) {
  u32 b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= g.B) return;
  u32 lo = lower_bound_u32(keys_sorted, g.E, b);
  u32 hi = lower_bound_u32(keys_sorted, g.E, b + 1);
  start[b] = lo;
  if (hi > lo) vals_sorted[hi - 1] |= 0x40000000u;  // ENTRY_LAST: accumulate L0 no longer needs the keys
  items[b] = hi > lo ? chunk_of(g, hi - 1) - chunk_of(g, lo) + 1 : 0u;
  if (b == g.B - 1) {
    start[g.B] = hi;
    items[g.B] = 0;
    // accumulate L0 reads entries in groups of 4 and prefetches the point of every entry it reads: the up to 3
    // entries past E must hold a valid table index (the sort only wrote E of them)
    for (u32 k = g.E; k < ((g.E + 3u) & ~3u); k++) vals_sorted[k] = 0;
  }
}

}  // namespace amsm
