// Pippenger windowed-bucket MSM kernels for gfx950 (wave64).
//
// Replaces ark-ec ^0.2.0 `VariableBaseMSM::multi_scalar_mul` (ext; SURVEY.md Appendix C recalls its
// structure: unsigned c-bit windows, 2^c-1 Jacobian buckets per window, running-sum reduction, serial
// Horner).  The GPU pipeline is NOT a translation of that loop nest:
//
//   digits   : one lane per scalar; signed c-bit digits (halves the buckets), one (key,value) entry per
//              window: key = bucket id, value = sign | index into the (pre-multiplied) generator table.
//   sort     : device radix sort of the entries by bucket id (rocPRIM) -- turns the scatter into runs.
//   bounds   : per bucket, binary search of its run [start,end) + number of K0-sized work items.
//   accum L0 : one lane per work item: <= K0 mixed additions (XYZZ, 8M+2S) gathered from the table.
//              Work items are equal sized, so wave64 lanes stay converged whatever the digit
//              distribution is (the reference's harness feeds all-equal scalars).
//   accum L1 : one lane per bucket folds its L0 partials; buckets with many partials go to
//   accum L2 : one 256-lane workgroup per heavy bucket (strided sums + wave-shuffle / LDS tree).
//   reduce   : sum_j j*B_j per bucket set: per-lane running sums over s buckets, a small scalar
//              multiple, then a wave-shuffle + LDS tree per workgroup; one partial per workgroup.
//   fold     : one wave folds the workgroup partials with __shfl_xor (single XYZZ result).
//
// With a precomputed key (table[w][i] = 2^(c*w) * G_i, resident in HBM -- 288 GB makes W copies of the
// key affordable) all windows share ONE bucket set, so there is no serial Horner chain of c*W doublings
// and the reduce stage shrinks W-fold.
#pragma once
#include "ec.h"
#include "msm_types.h"
#include "rng.h"

namespace amsm {



// ---------------------------------------------------------------------------------------------
// accumulate L0: lane c owns the fixed-size chunk [c*K0, (c+1)*K0) of the SORTED entry list, whatever
// buckets it spans: every lane does exactly K0 gathered mixed additions, so all SIMDs stay at full
// occupancy until the end of the kernel and wave64 lanes stay converged for ANY digit distribution
// (uniform, or the all-equal vectors of SURVEY.md F8).  When the bucket id changes inside a chunk the
// lane flushes its running sum as one partial of the finished bucket.  The partials of bucket b are the
// consecutive records item_off[b] + (chunk - first_chunk(b)); k_bounds counted them (items[b]).
// ---------------------------------------------------------------------------------------------
template <class Fq>
__global__ void __launch_bounds__(256)
    k_accum_l0(const u32* __restrict__ table, const u32* __restrict__ keys_sorted, const u32* __restrict__ vals_sorted,
               const u32* __restrict__ start, const u32* __restrict__ item_off, MsmGeom g, u32* __restrict__ partials) {
  u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  u32 e_valid = start[g.B];  // entries with a non-zero digit (key B = digit 0 sorts last)
  u32 s = c * g.K0;
  if (s >= e_valid) return;
  u32 e = min(s + g.K0, e_valid);
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  // software pipeline: entry (key, value) two ahead, point one ahead (vals -> point loads are dependent);
  // issued unconditionally (the tail re-reads its last entry) so hipcc keeps them in flight across the
  // long mixed addition instead of branching around them
  u32 v = vals_sorted[s];
  u32 b_cur = keys_sorted[s];
  u32 i1 = min(s + 1, e - 1);
  u32 v1 = vals_sorted[i1], k1 = keys_sorted[i1];
  Affine<Fq> pt = affine_load<Fq>(table, v & 0x7fffffffu);
  for (u32 k = s; k < e; k++) {
    u32 i2 = min(k + 2, e - 1);
    u32 v2 = vals_sorted[i2], k2 = keys_sorted[i2];
    Affine<Fq> ptn = affine_load<Fq>(table, v1 & 0x7fffffffu);
    xyzz_madd<Fq>(acc, affine_neg_if<Fq>(pt, (v >> 31) != 0));
    if (k1 != b_cur || k + 1 == e) {  // bucket finished inside this chunk (or chunk finished): flush
      u32 slot = item_off[b_cur] + (c - start[b_cur] / g.K0);
      xyzz_store<Fq>(partials, slot, acc);
      acc = xyzz_inf<Fq>();
      b_cur = k1;
    }
    v = v1;
    v1 = v2;
    k1 = k2;
    pt = ptn;
  }
}

// Butterfly reduction of one XYZZ per lane over aligned groups of WIDTH lanes (WIDTH = 64: whole wave) with
// __shfl_xor; every lane of a group ends with the group's sum.  All 64 lanes must be active.
template <class Fq, int WIDTH>
AMSM_DEV void group_reduce_xyzz(XYZZ<Fq>& acc) {
#pragma unroll 1
  for (int m = WIDTH / 2; m >= 1; m >>= 1) {
    XYZZ<Fq> o;
#pragma unroll
    for (int i = 0; i < Fq::L; i++) {
      o.x.v[i] = __shfl_xor(acc.x.v[i], m, 64);
      o.y.v[i] = __shfl_xor(acc.y.v[i], m, 64);
      o.zz.v[i] = __shfl_xor(acc.zz.v[i], m, 64);
      o.zzz.v[i] = __shfl_xor(acc.zzz.v[i], m, 64);
    }
    // lanes l and l^m hold (a,b) and (b,a): add in a canonical order so both compute the same sum
    bool low = (threadIdx.x & m) == 0;
    XYZZ<Fq> a = low ? acc : o;
    XYZZ<Fq> b2 = low ? o : acc;
    xyzz_add<Fq>(a, b2);
    acc = a;
  }
}

template <class Fq>
AMSM_DEV void wave_reduce_xyzz(XYZZ<Fq>& acc) {
  group_reduce_xyzz<Fq, 64>(acc);
}

// ---------------------------------------------------------------------------------------------
// accumulate L1: LPB lanes cooperate on bucket b: lane j sums partials j, j+LPB, ... then a shuffle
// butterfly over the LPB lanes (latency log2(LPB) + n/LPB additions instead of n).  Buckets with more
// than K1 partials are recorded as heavy for L2.
// ---------------------------------------------------------------------------------------------
template <class Fq, int LPB>
__global__ void __launch_bounds__(256)
    k_accum_l1(const u32* __restrict__ partials, const u32* __restrict__ items, const u32* __restrict__ item_off,
               MsmGeom g, u32* __restrict__ buckets, u32* __restrict__ heavy_count, u32* __restrict__ heavy_list) {
  u32 gid = blockIdx.x * blockDim.x + threadIdx.x;
  u32 b = gid / LPB, j = gid % LPB;
  bool valid = b < g.B;
  u32 n = valid ? items[b] : 0u;
  bool heavy = n > g.K1;
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  if (valid && !heavy) {
    u32 off = item_off[b];
    for (u32 k = j; k < n; k += LPB) {
      XYZZ<Fq> p = xyzz_load<Fq>(partials, off + k);
      xyzz_add<Fq>(acc, p);
    }
  }
  if (LPB > 1) group_reduce_xyzz<Fq, LPB>(acc);
  if (valid && j == 0) {
    if (heavy) {
      u32 slot = atomicAdd(heavy_count, 1u);
      heavy_list[slot] = b;
    } else {
      xyzz_store<Fq>(buckets, b, acc);
    }
  }
}

// Workgroup (256 lanes = 4 waves) reduction: wave shuffles, then 4 records through LDS.
// lds must hold 4 XYZZ records (4 * 4*L u32).  Result valid in lane 0 of the workgroup.
template <class Fq>
AMSM_DEV void block_reduce_xyzz(XYZZ<Fq>& acc, u32* lds) {
  wave_reduce_xyzz<Fq>(acc);
  u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) xyzz_store<Fq>(lds, wave, acc);
  __syncthreads();
  if (threadIdx.x == 0) {
    for (u32 w = 1; w < (blockDim.x >> 6); w++) {
      XYZZ<Fq> o = xyzz_load<Fq>(lds, w);
      xyzz_add<Fq>(acc, o);
    }
  }
}

// accumulate L2: one workgroup per heavy bucket (grid-stride over the heavy list).
template <class Fq>
__global__ void __launch_bounds__(256)
    k_accum_l2(const u32* __restrict__ partials, const u32* __restrict__ items, const u32* __restrict__ item_off,
               const u32* __restrict__ heavy_count, const u32* __restrict__ heavy_list, u32* __restrict__ buckets) {
  __shared__ __attribute__((aligned(16))) u32 lds[4 * 4 * Fq::L];
  u32 nh = *heavy_count;
  for (u32 h = blockIdx.x; h < nh; h += gridDim.x) {
    u32 b = heavy_list[h];
    u32 n = items[b], off = item_off[b];
    XYZZ<Fq> acc = xyzz_inf<Fq>();
    for (u32 k = threadIdx.x; k < n; k += blockDim.x) {
      XYZZ<Fq> p = xyzz_load<Fq>(partials, off + k);
      xyzz_add<Fq>(acc, p);
    }
    block_reduce_xyzz<Fq>(acc, lds);
    if (threadIdx.x == 0) xyzz_store<Fq>(buckets, b, acc);
    __syncthreads();
  }
}

// k * p for a small non-negative integer k (double-and-add, MSB first)
template <class Fq>
AMSM_DEV XYZZ<Fq> xyzz_mul_small(const XYZZ<Fq>& p, u32 k) {
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  if (k == 0) return acc;
  int top = 31 - __clz(k);
  for (int i = top; i >= 0; i--) {
    acc = xyzz_dbl<Fq>(acc);
    if ((k >> i) & 1) xyzz_add<Fq>(acc, p);
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// reduce: per set, sum_{j=1..nb} j * bucket[j-1].  Lane t owns buckets [t*s, (t+1)*s).
// grid = (red_threads/256 or 1, n_sets).  One partial per workgroup: out[set*gridDim.x + blockIdx.x].
// ---------------------------------------------------------------------------------------------
template <class Fq>
__global__ void __launch_bounds__(256)
    k_bucket_reduce(const u32* __restrict__ buckets, MsmGeom g, u32* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) u32 lds[4 * 4 * Fq::L];
  u32 set = blockIdx.y;
  u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  XYZZ<Fq> total = xyzz_inf<Fq>();
  if (t < g.red_threads) {
    u32 lo = t * g.red_s;
    XYZZ<Fq> run = xyzz_inf<Fq>(), sum = xyzz_inf<Fq>();
    for (int k = (int)g.red_s - 1; k >= 0; k--) {
      XYZZ<Fq> bk = xyzz_load<Fq>(buckets, (size_t)set * g.nb + lo + k);
      xyzz_add<Fq>(run, bk);
      xyzz_add<Fq>(sum, run);
    }
    total = xyzz_mul_small<Fq>(run, lo);
    xyzz_add<Fq>(total, sum);
  }
  block_reduce_xyzz<Fq>(total, lds);
  if (threadIdx.x == 0) xyzz_store<Fq>(out, (size_t)set * gridDim.x + blockIdx.x, total);
}

// fold: wave b sums records [b*n, (b+1)*n) (lane-strided + shuffle butterfly) and writes out[b].
template <class Fq>
__global__ void __launch_bounds__(64) k_fold(const u32* __restrict__ in, u32 n, u32* __restrict__ out) {
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  for (u32 k = threadIdx.x; k < n; k += 64) {
    XYZZ<Fq> p = xyzz_load<Fq>(in, (size_t)blockIdx.x * n + k);
    xyzz_add<Fq>(acc, p);
  }
  wave_reduce_xyzz<Fq>(acc);
  if (threadIdx.x == 0) xyzz_store<Fq>(out, blockIdx.x, acc);
}

// ---------------------------------------------------------------------------------------------
// Key preparation
// ---------------------------------------------------------------------------------------------
// table[(level)*stride + i] = 2^c * table[(level-1)*stride + i]  (affine in, affine out)
template <class Fq>
__global__ void __launch_bounds__(256) k_precompute_level(u32* __restrict__ table, u32 stride, u32 level, u32 c) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= stride) return;
  Affine<Fq> p = affine_load<Fq>(table, (size_t)(level - 1) * stride + i);
  Affine<Fq> r;
  if (affine_is_inf<Fq>(p)) {
    r = p;
  } else {
    XYZZ<Fq> a = xyzz_dbl_affine<Fq>(p);
    for (u32 k = 1; k < c; k++) a = xyzz_dbl<Fq>(a);
    r = xyzz_to_affine<Fq>(a);
  }
  affine_store<Fq>(table, (size_t)level * stride + i, r);
}

// is_inf bytes -> (0,0) encoding on device
template <class Fq>
__global__ void __launch_bounds__(256) k_apply_inf(u32* __restrict__ table, const uint8_t* __restrict__ is_inf, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (is_inf[i]) {
    Affine<Fq> z;
    z.x = fe_zero<Fq>();
    z.y = fe_zero<Fq>();
    affine_store<Fq>(table, i, z);
  }
}

// G_i = k_i * G, k_i = rng_scalar(seed, i); generator (gx, gy) passed in Montgomery form.
template <class Fq>
__global__ void __launch_bounds__(256)
    k_generate_bases(u32* __restrict__ table, u64 seed, u32 n, Affine<Fq> gen) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 k[8];
  rng_scalar(seed, i, k);
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  for (int bit = 253; bit >= 0; bit--) {
    acc = xyzz_dbl<Fq>(acc);
    if ((k[bit >> 5] >> (bit & 31)) & 1) xyzz_madd<Fq>(acc, gen);
  }
  affine_store<Fq>(table, i, xyzz_to_affine<Fq>(acc));
}

}  // namespace amsm
