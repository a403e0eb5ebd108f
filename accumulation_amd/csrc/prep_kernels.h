// MSM "prep": scalars -> the bucket-sorted entry list accumulate L0 consumes, in SEVEN dependent dispatches (five that
// do the work + two that only act on heavily skewed inputs).
//
// The first version of this stage was k_digits + rocPRIM radix_sort_pairs + k_bounds + rocPRIM exclusive_scan: 14
// dependent dispatches.  Stand-alone that costs ~0.35 ms at 2^20 pairs, but in a batch it runs BESIDE the previous
// MSM's accumulate L0, which holds every wave slot and releases them only when a generation of its workgroups
// retires (every ~0.4 ms): each dependent dispatch waits for such a boundary, the chain outlives L0 and the next
// L0 starts late (measured with rocprofv3).  The chain here is short enough to finish inside L0:
//
//   k_prep_hist     per workgroup, LDS histogram of its entries over P coarse partitions (partition = bucket id
//                   >> SH, i.e. 2^SH consecutive buckets), then one global atomicAdd per non-empty partition.
//   k_prep_scan     one workgroup: part_start = exclusive scan of the P partition sizes.
//   k_prep_scatter  per workgroup: recompute the digits, reserve a contiguous run in every partition it feeds (one
//                   global atomicAdd per partition), regroup the entries by partition in LDS and write the runs.
//   k_prep_local    one workgroup per partition: counting sort of the partition by the low SH bits of the bucket id
//                   with LDS counters, entries go straight to their final position; emits start[], items[] and the
//                   last-of-bucket flag.  The order of entries INSIDE a bucket is arbitrary (LDS atomics): the bucket
//                   sum does not depend on it and only canonical affine results are ever compared.
//   k_prep_offsets  item_off[b] += partials of the partitions before b's; start[B]; zeroed pad entries.
//
// Entries with digit 0 are never emitted (the sort used to carry them to the end of the list).  Skewed inputs
// (all-equal scalars put every entry of a window into ONE bucket) only make one partition large: its
// workgroup would loop over it; k_prep_heavy_count / k_prep_heavy_place split such partitions over 64 workgroups.
#pragma once
#include "fp.h"
#include "msm_types.h"
#include "vec_kernels.h"  // digit_step

namespace amsm {

struct PrepGeom {
  u32 SH;   // log2(buckets per partition)
  u32 P;    // partitions = ceil(B / 2^SH)
  u32 SPB;  // scalars per workgroup of k_prep_hist / k_prep_scatter (256 lanes x 2)
  u32 IB;   // bits of a table index (entry word: negate | bucket id low bits << IB | index)
  u32 CAP;  // entries k_prep_local can assemble in LDS (larger partitions scatter straight to memory)
  u32 HEAVY;  // partitions with more entries than this are split over PREP_HEAVY_SLICES workgroups (k_prep_heavy_*)
  // FIX != 0 (bucket-per-lane prep, round 3): partition p owns entries [p * FIX, (p + 1) * FIX) of the interchange buffer --
  // no histogram, no scan: uniform digits fill a partition to E / P +- a few hundred entries, and FIX = CAP, what
  // k_prep_local_t can stage anyway; a partition that would overflow drops the excess and its count (the cursor) trips the
  // overflow flag there, i.e. the skew fallback that a partition above CAP takes in any case
  u32 FIX;
};

// A skewed digit distribution (the all-equal vectors the reference's harness commits to put a whole
// window into ONE bucket) makes a few partitions hold millions of entries.  One workgroup streams at one CU's share of
// the memory system (~30 GB/s: 0.45 ms per million entries for the two passes), so such partitions are listed by
// k_prep_scan and processed by PREP_HEAVY_SLICES workgroups each: k_prep_heavy_count (per-bucket totals), k_prep_local
// (offsets only) and k_prep_heavy_place (slices reserve runs in their buckets with one global atomicAdd per bucket).
constexpr u32 PREP_HEAVY_SLICES = 64, PREP_HEAVY_GRID = 1024, PREP_MAX_HEAVY = 4096;
struct PrepHeavy {
  u32* n;      // number of heavy partitions (k_prep_scan)
  u32* ids;    // their ids
  u32* cnt;    // B words, zeroed by the launcher: entries per bucket (heavy partitions only)
  u32* cur;    // B words: running cursor of every bucket of a heavy partition (offset inside the partition)
  u32* end;    // B words: end offset of the bucket inside the partition
};

constexpr u32 PREP_ENTRY_LAST = 0x40000000u;  // == ENTRY_LAST of msm_kernels.h

// ctr[idx]++ on an LDS counter, returning the old value.  When every active lane of the wave hits the SAME counter
// (skewed digit distributions -- the all-equal vectors of SURVEY.md F8 -- make that the common case) one lane adds
// the wave's population instead of 64 serialized same-address atomics.  Used by k_prep_local on oversized partitions
// only: there one workgroup walks millions of entries; everywhere else plain LDS atomics are as fast.
// (k_prep_local is bound by its 16.8 M scattered 4-byte stores -- L2 write transactions, ~62 us at 2^20 -- not by
// loads or atomics: keeping the partition in registers between the two passes did not change its 75 us.)
AMSM_DEV u32 lds_count(u32* ctr, u32 idx) {
  const u64 active = __ballot(1);
  const u32 first = __builtin_amdgcn_readfirstlane(idx);
  const u64 same = __ballot(idx == first);
  if (same == active) {
    const u32 lane = __lane_id();
    const u32 below = (u32)__popcll(same & ((1ull << lane) - 1ull));
    u32 base = 0;
    if (below == 0) base = atomicAdd(&ctr[first], (u32)__popcll(same));
    base = __builtin_amdgcn_readfirstlane(base);  // the first active lane is the one that did the add
    return base + below;
  }
  return atomicAdd(&ctr[idx], 1u);
}

// Calls f(key, value) for every non-zero signed digit of scalar i; returns non-zero when the scalar does not
// fit W windows (caller reports AMSM_E_SCALAR_RANGE).  key = bucket id, value = sign | index into the key table.
// (An unsigned top window -- W = ceil(255 / c) windows, the digit excess over the bucket count as a second entry --
// was built and measured: c = 17 / W = 15 is no faster than c = 16 / W = 16 at any size, and the extra branch in the
// unrolled walk cost k_prep_scatter 30 us; dropped.)
template <class Fr, class F>
AMSM_DEV u32 scalar_entries(const u32* __restrict__ scalars, int mont, const MsmGeom& g, u32 i, F&& f) {
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
    u32 idx = g.idx_rel_bits ? ((w << g.idx_rel_bits) | i) : g.base_off + i + (g.precomp ? w * g.table_stride : 0u);
    if (d != 0) f(set * g.nb + (d - 1), idx | (neg << 31));
  }
  u32 rest = carry;
#pragma unroll
  for (int k = 0; k < 8; k++) rest |= s.v[k];
  return rest;
}

// The same walk over a scalar already in registers (canonical form), the window loop unrolled to MAXW (>= g.W): f(w, key,
// value) sees w as an unrolled index, so callers can keep per-window state in registers.  k_prep_scatter walks every scalar
// twice and keeps it (8 registers) instead of loading -- and converting from Montgomery form -- a second time.
// Returns non-zero when the scalar does not fit W windows, like scalar_entries (callers that walk twice ignore it once).
template <class Fr, int MAXW, class F>
AMSM_DEV u32 scalar_entries_unrolled_reg(Fe<Fr> s, const MsmGeom& g, u32 i, F&& f) {
  const DigitWalk dw = digit_walk_of(g);
  const u32 set0 = ((g.groups > 1u) ? ((i >> g.group_shift) & 1u) : 0u) * sets_per_group(g);
  u32 carry = 0;
#pragma unroll
  for (int w = 0; w < MAXW; w++) {
    if ((u32)w < g.W) {
      u32 neg;
      const u32 d = digit_step<Fr>(s, dw, (u32)w, carry, neg);
      u32 set = set0 + window_set(g, (u32)w, i);
      u32 idx = g.idx_rel_bits ? (((u32)w << g.idx_rel_bits) | i) : g.base_off + i + (g.precomp ? (u32)w * g.table_stride : 0u);
      if (d != 0) f(w, set * g.nb + (d - 1), idx | (neg << 31));
    }
  }
  u32 rest = carry;
#pragma unroll
  for (int k = 0; k < 8; k++) rest |= s.v[k];
  return rest;
}

// dynamic LDS: P counters
template <class Fr>
__global__ void __launch_bounds__(1024)
    k_prep_hist(const u32* __restrict__ scalars, int mont, MsmGeom g, PrepGeom pg, u32* __restrict__ part_total,
                u32* __restrict__ err) {
  extern __shared__ u32 prep_lds[];
  u32* cnt = prep_lds;
  for (u32 p = threadIdx.x; p < pg.P; p += blockDim.x) cnt[p] = 0;
  __syncthreads();
  u32 bad = 0;
  for (u32 r = 0; r < pg.SPB; r += blockDim.x) {
    u32 i = blockIdx.x * pg.SPB + r + threadIdx.x;
    if (i < g.n) bad |= scalar_entries<Fr>(scalars, mont, g, i, [&](u32 key, u32) { atomicAdd(&cnt[key >> pg.SH], 1u); });
  }
  if (bad) atomicOr(err, 1u);
  __syncthreads();
  for (u32 p = threadIdx.x; p < pg.P; p += blockDim.x) {
    u32 v = cnt[p];
    if (v) atomicAdd(&part_total[p], v);
  }
}

// Exclusive scan of n values by ONE workgroup of 1024 lanes (n is at most a few thousand partitions / a few hundred
// thousand buckets): each lane sums a contiguous slice, the slice sums are scanned through LDS, then the slices are
// re-walked.  out[n] = total when `with_total`.
AMSM_DEV void block_exclusive_scan(const u32* __restrict__ in, u32* __restrict__ out, u32 n, bool with_total, u32* lds) {
  const u32 T = blockDim.x, t = threadIdx.x;
  u32 per = (n + T - 1) / T;
  u32 lo = min(t * per, n), hi = min(lo + per, n);
  u32 sum = 0;
  for (u32 k = lo; k < hi; k++) sum += in[k];
  lds[t] = sum;
  __syncthreads();
  // Hillis-Steele over T slice sums
  for (u32 d = 1; d < T; d <<= 1) {
    u32 v = t >= d ? lds[t - d] : 0u;
    __syncthreads();
    lds[t] += v;
    __syncthreads();
  }
  u32 run = lds[t] - sum;  // exclusive prefix of this slice
  for (u32 k = lo; k < hi; k++) {
    u32 v = in[k];
    out[k] = run;
    run += v;
  }
  if (with_total && t == T - 1) out[n] = lds[T - 1];
}

__global__ void __launch_bounds__(1024)
    k_prep_scan(const u32* __restrict__ part_total, u32* __restrict__ part_start, u32 P, u32 heavy_min, PrepHeavy hv) {
  __shared__ u32 lds[1024];
  block_exclusive_scan(part_total, part_start, P, true, lds);
  for (u32 p = threadIdx.x; p < P; p += blockDim.x)  // *hv.n was zeroed by the launcher
    if (part_total[p] > heavy_min) hv.ids[atomicAdd(hv.n, 1u)] = p;
}

// Entry word inside the partition buffers: bit 31 = negate, bits [IB, IB + SH) = bucket id low bits, bits [0, IB) =
// table index (IB = PrepGeom::IB; the host checks IB + SH <= 31).
//
// k_prep_scatter: a workgroup takes SPB = 512 scalars (one per lane), ranks every
// entry inside its partition with an LDS counter, reserves one contiguous run per partition in global memory (one
// global atomicAdd each), regroups the entries by partition in LDS and copies them out so that consecutive lanes
// write consecutive words (runs of SPB * W / P entries: 64 B at P = 512).  Scattering word by word instead cost
// 360 us at 2^20 pairs (33 M partial-line write transactions); staged it is bandwidth bound.
// dynamic LDS: 3 * P words + SPB * W words (staged entries) + SPB * W half-words (their partition).
// SPT = scalars per lane (SPB = blockDim * SPT), MAXW >= S (entry slots per scalar); SPB * S <= 8192.
// (Tried: 1024 partitions -> 32-byte runs, 162 us instead of 82; 1024 scalars per workgroup with 108 KiB of LDS to get
// the 64-byte runs back, 110 us and a slower batch.  512 partitions x 512 scalars it is.)
// WIDE (the bucket-per-lane pipeline, 20-bit windows): 64-bit interchange entries -- low word = negate | table index, high word
// = the bucket id's low bits -- because 1 + 10 + 24 bits do not fit one word; `part` then holds E 8-byte entries (P even).
template <class Fr, int MAXW, int SPT, bool WIDE = false>
__global__ void __launch_bounds__(512)
    k_prep_scatter(const u32* __restrict__ scalars, int mont, MsmGeom g, PrepGeom pg, const u32* __restrict__ part_start,
                   u32* __restrict__ part_cursor, u32* __restrict__ part, u32* __restrict__ err = nullptr) {
  extern __shared__ u32 prep_lds[];
  const u32 cap = pg.SPB * g.S;  // staged entries (upper bound)
  u32* cnt = prep_lds;           // entries per partition (rank counters in step 1)
  u32* loff = prep_lds + pg.P;   // block-local exclusive prefix
  u32* gbase = prep_lds + 2 * pg.P;
  u32* staged = prep_lds + 3 * pg.P;
  uint16_t* staged_p = reinterpret_cast<uint16_t*>(staged + (WIDE ? 2u : 1u) * cap);
  const u32 t = threadIdx.x, T = blockDim.x;
  for (u32 p = t; p < pg.P; p += T) cnt[p] = 0;
  __syncthreads();
  // 1. rank every entry inside its partition; the ranks (< SPB * W <= 8192) stay in registers, two per word, and so does
  //    the scalar (canonical form) for the second walk in step 3
  u32 rk[SPT][MAXW / 2];
  Fe<Fr> sreg[SPT];
#pragma unroll
  for (int r = 0; r < SPT; r++) {
#pragma unroll
    for (int k = 0; k < MAXW / 2; k++) rk[r][k] = 0;
    u32 i = blockIdx.x * pg.SPB + r * T + t;
    if (i < g.n) {
      sreg[r] = fe_load<Fr>(scalars + (size_t)i * 8);
      if (g.skip_ones) fe_drop_stored_one<Fr>(sreg[r], mont);
      if (mont) sreg[r] = fe_from_mont<Fr>(sreg[r]);
      const u32 bad = scalar_entries_unrolled_reg<Fr, MAXW>(sreg[r], g, i, [&](int w, u32 key, u32) {
        u32 rank = atomicAdd(&cnt[key >> pg.SH], 1u);
        rk[r][w >> 1] |= rank << ((w & 1) * 16);
      });
      if (pg.FIX && bad && err) atomicOr(err, 1u);  // (with a histogram pass, that pass reports it)
    }
  }
  __syncthreads();
  // 2. block-local exclusive prefix over partitions (slice per lane + Hillis-Steele over the slice sums) and the
  //    global run reservation.  The reservation is one returning atomicAdd per (workgroup, partition) on 512 addresses that
  //    all 2048 workgroups hit: its round trip (measured: 30 of the kernel's 78 us at 2^20) is only NEEDED by the copy-out of
  //    step 4, so the first RSV_MAX of a lane's partitions keep the returned offset in a register and step 3 -- pure ALU
  //    and LDS now that the scalar stays in registers -- runs while the atomics are in flight.
  constexpr u32 RSV_MAX = 4;
  u32 rsv[RSV_MAX], rsv_base[RSV_MAX];
  u32 per = (pg.P + T - 1) / T;
  u32 lo = min(t * per, pg.P), hi = min(lo + per, pg.P);
  {
    u32* sl = reinterpret_cast<u32*>(staged);  // scratch: T words, before staging starts
    u32 sum = 0;
    for (u32 p = lo; p < hi; p++) sum += cnt[p];
    sl[t] = sum;
    __syncthreads();
    for (u32 d = 1; d < T; d <<= 1) {
      u32 v = t >= d ? sl[t - d] : 0u;
      __syncthreads();
      sl[t] += v;
      __syncthreads();
    }
    u32 run = sl[t] - sum;
#pragma unroll
    for (u32 q = 0; q < RSV_MAX; q++) {
      u32 p = lo + q;
      rsv[q] = 0;
      rsv_base[q] = 0;
      if (p < hi) {
        u32 v = cnt[p];
        loff[p] = run;
        if (v) {
          rsv_base[q] = pg.FIX ? p * pg.FIX : part_start[p];
          rsv[q] = atomicAdd(&part_cursor[p], v);
        }
        run += v;
      }
    }
    for (u32 p = lo + RSV_MAX; p < hi; p++) {  // more partitions per lane than registers kept for them
      u32 v = cnt[p];
      loff[p] = run;
      gbase[p] = v ? (pg.FIX ? p * pg.FIX : part_start[p]) + atomicAdd(&part_cursor[p], v) : 0u;
      run += v;
    }
  }
  __syncthreads();
  // 3. regroup by partition in LDS (the digits are recomputed from the registers: cheaper than keeping 2 W entry words)
  const u32 low = (1u << pg.SH) - 1u;
#pragma unroll
  for (int r = 0; r < SPT; r++) {
    u32 i = blockIdx.x * pg.SPB + r * T + t;
    if (i < g.n)
      scalar_entries_unrolled_reg<Fr, MAXW>(sreg[r], g, i, [&](int w, u32 key, u32 val) {
        u32 p = key >> pg.SH;
        u32 rank = (rk[r][w >> 1] >> ((w & 1) * 16)) & 0xffffu;
        u32 slot = loff[p] + rank;
        if (WIDE) reinterpret_cast<u64*>(staged)[slot] = ((u64)(key & low) << 32) | val;
        else staged[slot] = (val & 0x80000000u) | ((key & low) << pg.IB) | (val & 0x7fffffffu);
        staged_p[slot] = (uint16_t)p;
      });
  }
#pragma unroll
  for (u32 q = 0; q < RSV_MAX; q++)
    if (lo + q < hi) gbase[lo + q] = rsv_base[q] + rsv[q];
  __syncthreads();
  // 4. copy out: consecutive lanes -> consecutive slots -> (mostly) consecutive global words
  u32 total = loff[pg.P - 1] + cnt[pg.P - 1];
  for (u32 j = t; j < total; j += T) {
    u32 p = staged_p[j];
    const u32 dst = gbase[p] + (j - loff[p]);
    if (pg.FIX && dst - p * pg.FIX >= pg.FIX) continue;  // the partition is full: skewed digits (the count flags it)
    if (WIDE) reinterpret_cast<u64*>(part)[dst] = reinterpret_cast<const u64*>(staged)[j];
    else part[dst] = staged[j];
  }
}

// One workgroup (1024 lanes: a partition is ~32 k entries and there are only ~512 of them) per partition: counting
// sort by the bucket id's low bits with LDS counters; entries go straight to their final position.
// dynamic LDS: 2 * 2^SH words (per-bucket count -> end, offset -> cursor) + blockDim scan words + CAP staged entries.
__global__ void __launch_bounds__(1024)
    k_prep_local(const u32* __restrict__ part_start, const u32* __restrict__ part, MsmGeom g, PrepGeom pg,
                 u32* __restrict__ vals_sorted, u32* __restrict__ start, u32* __restrict__ items, u32* __restrict__ item_off,
                 u32* __restrict__ part_items, PrepHeavy hv) {
  extern __shared__ u32 prep_lds[];
  const u32 NB = 1u << pg.SH;
  u32* cnt = prep_lds;       // entries per bucket of this partition, later the bucket's end offset
  u32* off = prep_lds + NB;  // exclusive prefix, later the running cursor
  u32* sl = prep_lds + 2 * NB;
  const u32 p = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  const u32 ps = part_start[p], pe = part_start[p + 1];
  const u32 low = NB - 1u, idx_mask = (1u << pg.IB) - 1u;
  const bool heavy = (pe - ps) > pg.HEAVY;  // uniform per workgroup; counted by k_prep_heavy_count
  for (u32 k = t; k < NB; k += T) cnt[k] = heavy ? hv.cnt[(p << pg.SH) + k] : 0u;
  __syncthreads();
  // entries are read four at a time once the partition is long enough to care (skewed inputs make partitions of
  // millions of entries; one workgroup still owns each): head up to 16-byte alignment, uint4 body, tail
  const u32 body_lo = min((ps + 3u) & ~3u, pe), body_hi = max(body_lo, pe & ~3u);
  // a partition several times the expected size means a skewed digit distribution: whole waves then hit one counter,
  // and the wave-aggregated increment is worth its extra instructions (uniform per workgroup)
  const bool skew = (pe - ps) > 4u * (g.E / pg.P + 1u);
  auto count = [&](u32 e) {
    u32 k = (e >> pg.IB) & low;
    if (skew) lds_count(cnt, k);
    else atomicAdd(&cnt[k], 1u);
  };
  if (!heavy) {
    for (u32 j = ps + t; j < body_lo; j += T) count(part[j]);
    for (u32 j = body_lo + 4u * t; j < body_hi; j += 4u * T) {
      uint4 e4 = *reinterpret_cast<const uint4*>(part + j);
      count(e4.x);
      count(e4.y);
      count(e4.z);
      count(e4.w);
    }
    for (u32 j = body_hi + t; j < pe; j += T) count(part[j]);
  }
  __syncthreads();
  const u32 b0 = p << pg.SH;
  // exclusive prefix of the bucket sizes (entries) and of the bucket chunk counts (partials), lane t owns a slice
  u32 per = (NB + T - 1) / T;
  u32 lo_k = min(t * per, NB), hi_k = min(lo_k + per, NB);
  u32 sum = 0;
  for (u32 k = lo_k; k < hi_k; k++) sum += cnt[k];
  sl[t] = sum;
  __syncthreads();
  for (u32 d = 1; d < T; d <<= 1) {
    u32 v = t >= d ? sl[t - d] : 0u;
    __syncthreads();
    sl[t] += v;
    __syncthreads();
  }
  u32 run = sl[t] - sum;
  u32 my_items = 0;
  for (u32 k = lo_k; k < hi_k; k++) {
    u32 c = cnt[k];
    off[k] = run;
    u32 lo = ps + run, hi = lo + c;
    u32 it = c ? chunk_of(g, hi - 1) - chunk_of(g, lo) + 1 : 0u;
    if (b0 + k < g.B) {
      start[b0 + k] = lo;  // also for empty buckets: where the bucket would start
      items[b0 + k] = it;
    }
    cnt[k] = run + c;  // end offset of the bucket inside the partition
    if (heavy) {
      hv.cur[b0 + k] = run;
      hv.end[b0 + k] = run + c;
    }
    my_items += it;
    run += c;
  }
  __syncthreads();
  // exclusive prefix of the chunk counts inside the partition (k_prep_offsets adds the partitions before it)
  sl[t] = my_items;
  __syncthreads();
  for (u32 d = 1; d < T; d <<= 1) {
    u32 v = t >= d ? sl[t - d] : 0u;
    __syncthreads();
    sl[t] += v;
    __syncthreads();
  }
  u32 irun = sl[t] - my_items;
  for (u32 k = lo_k; k < hi_k; k++) {
    if (b0 + k < g.B) {
      item_off[b0 + k] = irun;
      irun += items[b0 + k];  // own write, same lane
    }
  }
  if (t == T - 1) part_items[p] = sl[T - 1];
  if (heavy) return;  // placed by k_prep_heavy_place
  __syncthreads();
  // final placement; the entry that lands on the last position of its bucket carries the flag
  // Partitions of the usual size are assembled in LDS and written out as one coalesced stream: 16.8 M scattered
  // 4-byte stores cost ~62 us of L2 write transactions at 2^20 pairs, the staged copy is bandwidth bound.  Oversized
  // (skewed) partitions scatter straight to memory.
  u32* stage = prep_lds + 2 * NB + T;
  const bool staged = (pe - ps) <= pg.CAP;
  auto place = [&](u32 e) {
    u32 k = (e >> pg.IB) & low;
    u32 pos = skew ? lds_count(off, k) : atomicAdd(&off[k], 1u);
    u32 v = (e & 0x80000000u) | entry_abs_index(g, e & idx_mask);
    if (pos + 1 == cnt[k]) v |= PREP_ENTRY_LAST;
    if (staged) stage[pos] = v;
    else vals_sorted[ps + pos] = v;
  };
  for (u32 j = ps + t; j < body_lo; j += T) place(part[j]);
  for (u32 j = body_lo + 4u * t; j < body_hi; j += 4u * T) {
    uint4 e4 = *reinterpret_cast<const uint4*>(part + j);
    place(e4.x);
    place(e4.y);
    place(e4.z);
    place(e4.w);
  }
  for (u32 j = body_hi + t; j < pe; j += T) place(part[j]);
  if (staged) {
    __syncthreads();
    const u32 n_p = pe - ps;
    // 16-byte aligned body of the OUTPUT range, scalar head / tail
    const u32 o_lo = min((ps + 3u) & ~3u, pe), o_hi = max(o_lo, pe & ~3u);
    for (u32 j = ps + t; j < o_lo; j += T) vals_sorted[j] = stage[j - ps];
    for (u32 j = o_lo + 4u * t; j < o_hi; j += 4u * T) {
      u32 q = j - ps;
      *reinterpret_cast<uint4*>(vals_sorted + j) = make_uint4(stage[q], stage[q + 1], stage[q + 2], stage[q + 3]);
    }
    for (u32 j = o_hi + t; j < pe; j += T) vals_sorted[j] = stage[j - ps];
    (void)n_p;
  }
}

// Slice `s` of heavy partition hv.ids[h]: a contiguous run of its entries, one (partition, slice) pair per loop trip.
// Both kernels run with a fixed small grid and return at once when no partition is heavy (the usual case).
// dynamic LDS: 2 * 2^SH words.
AMSM_DEV void prep_heavy_slice(u32 ps, u32 pe, u32 s, u32& lo, u32& hi) {
  u32 per = ((pe - ps) + PREP_HEAVY_SLICES - 1u) / PREP_HEAVY_SLICES;
  lo = min(ps + s * per, pe);
  hi = min(lo + per, pe);
}
__global__ void __launch_bounds__(256)
    k_prep_heavy_count(const u32* __restrict__ part_start, const u32* __restrict__ part, PrepGeom pg, PrepHeavy hv) {
  extern __shared__ u32 prep_lds[];
  const u32 NB = 1u << pg.SH, low = NB - 1u, t = threadIdx.x, T = blockDim.x;
  u32* cnt = prep_lds;
  const u32 work = *hv.n * PREP_HEAVY_SLICES;
  for (u32 w = blockIdx.x; w < work; w += gridDim.x) {
    const u32 p = hv.ids[w / PREP_HEAVY_SLICES];
    u32 lo, hi;
    prep_heavy_slice(part_start[p], part_start[p + 1], w % PREP_HEAVY_SLICES, lo, hi);
    for (u32 k = t; k < NB; k += T) cnt[k] = 0;
    __syncthreads();
    for (u32 j = lo + t; j < hi; j += T) lds_count(cnt, (part[j] >> pg.IB) & low);
    __syncthreads();
    for (u32 k = t; k < NB; k += T)
      if (cnt[k]) atomicAdd(&hv.cnt[(p << pg.SH) + k], cnt[k]);
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256)
    k_prep_heavy_place(const u32* __restrict__ part_start, const u32* __restrict__ part, MsmGeom g, PrepGeom pg, PrepHeavy hv,
                       u32* __restrict__ vals_sorted) {
  extern __shared__ u32 prep_lds[];
  const u32 NB = 1u << pg.SH, low = NB - 1u, idx_mask = (1u << pg.IB) - 1u, t = threadIdx.x, T = blockDim.x;
  u32* cnt = prep_lds;        // entries of this slice per bucket, then the slice's cursor inside its run
  u32* base = prep_lds + NB;  // start of the run this slice reserved in the bucket
  const u32 work = *hv.n * PREP_HEAVY_SLICES;
  for (u32 w = blockIdx.x; w < work; w += gridDim.x) {
    const u32 p = hv.ids[w / PREP_HEAVY_SLICES], b0 = p << pg.SH;
    const u32 ps = part_start[p];
    u32 lo, hi;
    prep_heavy_slice(ps, part_start[p + 1], w % PREP_HEAVY_SLICES, lo, hi);
    for (u32 k = t; k < NB; k += T) cnt[k] = 0;
    __syncthreads();
    for (u32 j = lo + t; j < hi; j += T) lds_count(cnt, (part[j] >> pg.IB) & low);
    __syncthreads();
    for (u32 k = t; k < NB; k += T) {
      u32 c = cnt[k];
      base[k] = c ? atomicAdd(&hv.cur[b0 + k], c) : 0u;
      cnt[k] = 0;
    }
    __syncthreads();
    for (u32 j = lo + t; j < hi; j += T) {
      u32 e = part[j];
      u32 k = (e >> pg.IB) & low;
      u32 pos = base[k] + lds_count(cnt, k);
      u32 v = (e & 0x80000000u) | entry_abs_index(g, e & idx_mask);
      if (pos + 1 == hv.end[b0 + k]) v |= PREP_ENTRY_LAST;
      vals_sorted[ps + pos] = v;
    }
    __syncthreads();
  }
}

// item_off[b] += partials of the partitions before b's; closes start[] / items[] / item_off[] at index B and zeroes
// the entries accumulate L0 may prefetch past the end of the list.  dynamic LDS: P + 256 words.
__global__ void __launch_bounds__(256)
    k_prep_offsets(const u32* __restrict__ part_start, const u32* __restrict__ part_items, PrepGeom pg, MsmGeom g,
                   u32* __restrict__ start, u32* __restrict__ items, u32* __restrict__ item_off, u32* __restrict__ vals_sorted) {
  extern __shared__ u32 prep_lds[];
  u32* pref = prep_lds;  // exclusive prefix of part_items
  u32* sl = prep_lds + pg.P;
  const u32 t = threadIdx.x, T = blockDim.x;
  u32 per = (pg.P + T - 1) / T;
  u32 lo = min(t * per, pg.P), hi = min(lo + per, pg.P);
  u32 sum = 0;
  for (u32 p = lo; p < hi; p++) sum += part_items[p];
  sl[t] = sum;
  __syncthreads();
  for (u32 d = 1; d < T; d <<= 1) {
    u32 v = t >= d ? sl[t - d] : 0u;
    __syncthreads();
    sl[t] += v;
    __syncthreads();
  }
  u32 run = sl[t] - sum;
  for (u32 p = lo; p < hi; p++) {
    pref[p] = run;
    run += part_items[p];
  }
  __syncthreads();
  u32 b = blockIdx.x * T + t;
  if (b < g.B) item_off[b] += pref[b >> pg.SH];
  if (blockIdx.x == 0) {
    const u32 e_valid = part_start[pg.P];
    if (t == 0) {
      start[g.B] = e_valid;  // entries with a non-zero digit
      items[g.B] = 0;
      item_off[g.B] = sl[T - 1];  // total number of partials
    }
    // accumulate L0 reads entries in groups of 4 and prefetches the point of every entry it reads; its last active
    // workgroup starts at or before e_valid and covers 256 chunks (plus two groups of look-ahead), clamped
    // to the group holding entry E - 1: everything in that range past the real entries must be a valid table index
    const u32 span = 256u * max(g.K0, g.K0b);
    unsigned long long reach = (unsigned long long)e_valid + span + 16ull;
    const u32 pad_end = (u32)min(reach, (unsigned long long)((g.E + 3u) & ~3u));
    for (u32 k = e_valid + t; k < pad_end; k += T) vals_sorted[k] = 0;
  }
}

// ---------------------------------------------------------------------------------------------
// Bucket-per-lane pipeline (round 3; 2^20-pair MSMs over a key precomputed for 20-bit windows: 13 entries per scalar
// instead of 15, 2^19 buckets of ~26 entries).  One workgroup per partition (1024 buckets, ~26 k entries, assembled in
// LDS) sorts its entries by bucket, orders its BUCKETS by size, and writes the entries of every group of 64 consecutive
// buckets of that order TRANSPOSED: entry k of the group's lane l at  base + 64 k + l,  padded to the group's largest
// bucket (sizes inside a group differ by a few entries once sorted: ~6 % padding).  k_accum_bpl then gives each lane ONE
// WHOLE bucket: a wave reads its entries as 256-byte coalesced rows, runs max-size iterations with all lanes converged,
// and stores finished bucket sums -- no partial records, no accumulate L1 / L2, no per-flush bucket search.
// Skewed digit distributions (a partition above its LDS capacity, or a padded layout above the fixed per-partition
// stride) raise the overflow flag: every later kernel of the chain returns at once and the host reruns the MSM over the
// key's 17-bit-window table through the chunked pipeline above, which is built for exactly those inputs.
// ---------------------------------------------------------------------------------------------
constexpr u32 BPL_GROUP = 64;            // lanes per accumulate wave
constexpr u32 BPL_ENTRY_PAD = 0x40000000u;  // padding entry (bit 30; the last-of-bucket flag is not used in this pipeline)
constexpr u32 BPL_BINS = 256;            // size classes of the bucket ordering (sizes >= 255 share the last one)
// The 64 LARGEST buckets of a partition are split in two halves, one lane each, whose sums the accumulate kernel adds with one
// lane exchange.  On uniform scalars the largest are the buckets the spread top window feeds (MsmGeom::top_shift: every 32nd
// bucket holds ~88 entries against ~24; Pallas has 32 of them per partition, BLS12-381 64): unsplit, the group holding them
// ran 108 rows with half its lanes idle after 35 (rows per MSM 242 k -> 225 k, 88 % -> 94 % of the lane-iterations useful).
// Lane slots of a partition: [0, 64) halves of the buckets at positions 0..31 of the size order | [64, 1024) the buckets at
// positions 64..1023, one lane each | [1024, 1088) halves of positions 32..63 -- 17 groups of 64 lanes, rows nearly descending.
constexpr u32 BPL_SPLIT = 64;
constexpr u32 BPL_NB = 1024;                                   // buckets per partition
constexpr u32 BPL_LANES = BPL_NB + BPL_SPLIT;                  // 1088 lane slots per partition
constexpr u32 BPL_GROUPS = BPL_LANES / BPL_GROUP;              // 17
struct BplGroup {
  u32 base;  // first entry of the group's transposed block in ents_t
  u32 m;     // rows = entries of its largest lane
};
// lane slot -> position in the size order and which half (0: whole bucket, 1: first half, 2: second half)
AMSM_DEV void bpl_slot(u32 s, u32& pos, u32& half) {
  if (s < BPL_SPLIT) {
    pos = s >> 1;
    half = 1u + (s & 1u);
  } else if (s >= BPL_NB) {
    pos = BPL_SPLIT / 2u + ((s - BPL_NB) >> 1);
    half = 1u + (s & 1u);
  } else {
    pos = s;
    half = 0u;
  }
}

// dynamic LDS: (3 * NB + 2 * BPL_BINS + 2 * BPL_GROUPS + 2 + CAP) words, NB = 2^SH = 1024 (kern_fr.hip: BPL_LOCAL_FIXED_WORDS).
// Round 4: the 1024 scan words of the bucket-size prefix live in the (still empty) entry stage and a bucket's first staged
// entry is its cursor's final value minus its count -- 1792 words more for entries.
__global__ void __launch_bounds__(1024)
    k_prep_local_t(const u32* __restrict__ part_start, const u64* __restrict__ part, MsmGeom g, PrepGeom pg, u32 stride,
                   u32* __restrict__ ents_t, BplGroup* __restrict__ grp, u32* __restrict__ order, u32* __restrict__ err) {
  extern __shared__ u32 prep_lds[];
  const u32 NB = BPL_NB, NG = BPL_GROUPS;
  u32* cnt = prep_lds;            // entries per bucket
  u32* cur = cnt + NB;            // placement cursor: the bucket's first staged entry, after the placement its end
  u32* ord = cur + NB;            // bucket at position i of the size order
  u32* bins = ord + NB;           // BPL_BINS size classes: count, then first position
  u32* bscan = bins + BPL_BINS;   // BPL_BINS scan words of the size-class prefix
  u32* gm = bscan + BPL_BINS;     // NG: rows per group
  u32* gb = gm + NG;              // NG + 1: first entry of the group inside the partition's block
  u32* stage = gb + NG + 1;       // CAP sorted entries
  u32* sl = stage;                // 1024 scan words of the bucket-size prefix: done before the first entry is staged
  const u32 p = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  // fixed partitions (PrepGeom::FIX): `part_start` is the scatter's CURSOR array -- entries reserved, possibly more than fit
  const u32 ps = pg.FIX ? p * pg.FIX : part_start[p];
  const u32 n_p = pg.FIX ? part_start[p] : part_start[p + 1] - ps;
  const u32 pe = ps + (pg.FIX ? min(n_p, pg.FIX) : n_p);
  const u32 low = NB - 1u, idx_mask = (1u << pg.IB) - 1u, b0 = p << pg.SH;
  const bool fits = n_p <= pg.CAP && (!pg.FIX || n_p <= pg.FIX);  // (beyond FIX the scatter dropped entries)
  for (u32 k = t; k < NB; k += T) cnt[k] = 0;
  for (u32 k = t; k < BPL_BINS; k += T) bins[k] = 0;
  for (u32 k = t; k < NG; k += T) gm[k] = 0;
  __syncthreads();
  if (fits)
    for (u32 j = ps + t; j < pe; j += T) atomicAdd(&cnt[(u32)(part[j] >> 32) & low], 1u);
  __syncthreads();
  // exclusive prefix of the bucket sizes: lane t owns bucket t (NB == T)
  const u32 c_t = t < NB ? cnt[t] : 0u;
  sl[t] = c_t;
  __syncthreads();
  for (u32 d = 1; d < T; d <<= 1) {
    u32 v = t >= d ? sl[t - d] : 0u;
    __syncthreads();
    sl[t] += v;
    __syncthreads();
  }
  if (t < NB) cur[t] = sl[t] - c_t;
  // size class of my bucket and my rank inside it
  const u32 bin = min(c_t, BPL_BINS - 1u);
  u32 rank = 0;
  if (t < NB) rank = atomicAdd(&bins[bin], 1u);
  __syncthreads();  // (every lane has read its sl[t]: the stage may be written)
  // entries to their bucket's run (order inside a bucket is arbitrary: the sum does not depend on it)
  if (fits)
    for (u32 j = ps + t; j < pe; j += T) {
      const u64 e = part[j];
      const u32 k = (u32)(e >> 32) & low, lo = (u32)e;
      const u32 pos = atomicAdd(&cur[k], 1u);
      stage[pos] = (lo & 0x80000000u) | entry_abs_index(g, lo & idx_mask);
    }
  // descending size order: first position of class b = buckets in larger classes
  if (t < BPL_BINS) bscan[t] = bins[BPL_BINS - 1u - t];  // reversed, so that an inclusive scan counts the larger classes
  __syncthreads();
  for (u32 d = 1; d < BPL_BINS; d <<= 1) {
    u32 v = (t < BPL_BINS && t >= d) ? bscan[t - d] : 0u;
    __syncthreads();
    if (t < BPL_BINS) bscan[t] += v;
    __syncthreads();
  }
  if (t < BPL_BINS) bins[BPL_BINS - 1u - t] = bscan[t] - bins[BPL_BINS - 1u - t];  // exclusive
  __syncthreads();
  u32 my_pos = 0;
  if (t < NB) {
    my_pos = bins[bin] + rank;
    ord[my_pos] = t;
    // rows of the group(s) my bucket's lane(s) sit in (the last class is not sorted inside: maxima, not first elements)
    const u32 h0 = (c_t + 1u) >> 1;
    if (my_pos < BPL_SPLIT / 2u) atomicMax(&gm[0], h0);
    else if (my_pos < BPL_SPLIT) atomicMax(&gm[NG - 1u], h0);
    else atomicMax(&gm[my_pos / BPL_GROUP], c_t);
  }
  __syncthreads();
  if (t == 0) {
    u32 run = 0;
    for (u32 q = 0; q < NG; q++) {
      gb[q] = run;
      run += gm[q] * BPL_GROUP;
    }
    gb[NG] = run;
  }
  __syncthreads();
  const u32 total = gb[NG];
  const bool ok = fits && total <= stride;
  if (!ok && t == 0) atomicOr(err + 1, 1u);  // overflow: the host falls back to the chunked pipeline
  if (t < NG) {
    BplGroup h;
    h.base = p * stride + gb[t];
    h.m = ok ? gm[t] : 0u;
    grp[p * NG + t] = h;
  }
  if (t < NB) {  // which bucket each lane slot sums (both halves of a split bucket name it; the odd lane does not store)
    u32* o = order + (size_t)p * BPL_LANES;
    if (my_pos < BPL_SPLIT / 2u) o[2u * my_pos] = o[2u * my_pos + 1u] = b0 + t;
    else if (my_pos < BPL_SPLIT) o[BPL_NB + 2u * (my_pos - BPL_SPLIT / 2u)] = o[BPL_NB + 2u * (my_pos - BPL_SPLIT / 2u) + 1u] = b0 + t;
    else o[my_pos] = b0 + t;
  }
  if (!ok) return;
  for (u32 j = t; j < total; j += T) {
    u32 q = 0;
    while (q + 1u < NG && gb[q + 1u] <= j) q++;  // the group holding row element j (NG = 17 bases, ascending)
    const u32 k = (j - gb[q]) / BPL_GROUP, l = j % BPL_GROUP;
    u32 pos, half;
    bpl_slot(q * BPL_GROUP + l, pos, half);
    const u32 kb = ord[pos], c = cnt[kb], h0 = (c + 1u) >> 1;
    const u32 off = half == 2u ? h0 : 0u, sz = half == 0u ? c : (half == 1u ? h0 : c - h0);
    ents_t[(size_t)p * stride + j] = k < sz ? stage[cur[kb] - c + off + k] : BPL_ENTRY_PAD;  // (cur = the bucket's end now)
  }
}

// ---------------------------------------------------------------------------------------------
// Bucket-split layout for SMALL and MEDIUM MSMs (round 3; up to 2^19 pairs over a precomputed key, one or two bucket sets):
// the chip has ~131 k lanes to fill and such an MSM only 2^12 .. 2^16 buckets of 30 .. 300 entries, so every bucket is cut
// into L = 2^k parts (part r takes its entries r, r + L, ...: sizes differ by at most one) on L ADJACENT lanes of one wave;
// k_accum_bps gives each lane its part -- a handful of mixed additions read as coalesced rows like k_accum_bpl's -- and adds
// the L partial sums with log2 L lane exchanges.  Against the chunked pipeline this removes the partial records, accumulate
// L1 / L2 and four dispatches from a chain that is latency-bound at these sizes (a 2^16-pair call: 0.40 ms, of which
// accumulate L0 + L1 + L2 were 0.18).  A partition is 1024 / L consecutive buckets = 1024 lanes = 16 groups of 64; rows of a
// group = the largest part in it; no size ordering (parts of one bucket are equal, buckets of one group similar on uniform
// digits).  A skewed input makes one group long: if the padded block outgrows the partition's stride (or the partition its
// LDS stage) the overflow flag sends the MSM to the chunked pipeline over the SAME key.
// 32-bit interchange entries as in k_prep_local (negate | bucket low bits << IB | index).
// dynamic LDS: see k_prep_local_s; NBP = 2^SH = 1024 / L buckets per partition.
// ---------------------------------------------------------------------------------------------
constexpr u32 BPS_LANES = 1024, BPS_GROUPS = BPS_LANES / BPL_GROUP;  // lane slots / 64-lane groups per partition
// Round 3, late: the partition's buckets are ORDERED BY SIZE (descending, counting sort over BPL_BINS size classes, as in
// k_prep_local_t) before they are laid on the lanes: a group's rows are its largest part, and 32 neighbouring buckets of a
// uniform 2^16-pair MSM spread over 8 +- 2 entries per part -- 12.4 rows for 8.3 of work; sorted, a group's parts differ by
// one.  `order[p * NBP + pos]` = the bucket at position pos (k_accum_bps stores there).
// dynamic LDS: (4 * NBP + 1024 + BPL_BINS + 2 * 16 + 2 + CAP) words.
__global__ void __launch_bounds__(1024)
    k_prep_local_s(const u32* __restrict__ part_start, const u32* __restrict__ part, MsmGeom g, PrepGeom pg, u32 stride,
                   u32 log2_l, u32* __restrict__ ents_t, BplGroup* __restrict__ grp, u32* __restrict__ order,
                   u32* __restrict__ err) {
  extern __shared__ u32 prep_lds[];
  const u32 NBP = 1u << pg.SH, L = 1u << log2_l, NG = BPS_GROUPS;
  u32* cnt = prep_lds;        // entries per bucket
  u32* cur = cnt + NBP;       // placement cursor
  u32* beg = cur + NBP;       // first staged entry of the bucket
  u32* ord = beg + NBP;       // bucket at position i of the size order
  u32* sl = ord + NBP;        // 1024 scan words
  u32* bins = sl + 1024;      // BPL_BINS size classes: count, then first position
  u32* gm = bins + BPL_BINS;  // NG: rows per group
  u32* gb = gm + NG;          // NG + 1
  u32* stage = gb + NG + 2;   // CAP entries sorted by bucket
  const u32 p = blockIdx.x, t = threadIdx.x, T = blockDim.x;
  // fixed partitions (PrepGeom::FIX): `part_start` is the scatter's CURSOR array, as in k_prep_local_t
  const u32 ps = pg.FIX ? p * pg.FIX : part_start[p];
  const u32 n_p = pg.FIX ? part_start[p] : part_start[p + 1] - ps;
  const u32 pe = ps + (pg.FIX ? min(n_p, pg.FIX) : n_p);
  const u32 low = NBP - 1u, idx_mask = (1u << pg.IB) - 1u;
  const bool fits = n_p <= pg.CAP && (!pg.FIX || n_p <= pg.FIX);
  for (u32 k = t; k < NBP; k += T) cnt[k] = 0;
  for (u32 k = t; k < BPL_BINS; k += T) bins[k] = 0;
  for (u32 k = t; k < NG; k += T) gm[k] = 0;
  __syncthreads();
  if (fits)
    for (u32 j = ps + t; j < pe; j += T) atomicAdd(&cnt[(part[j] >> pg.IB) & low], 1u);
  __syncthreads();
  const u32 c_t = t < NBP ? cnt[t] : 0u;
  sl[t] = c_t;
  __syncthreads();
  for (u32 d = 1; d < T; d <<= 1) {
    u32 v = t >= d ? sl[t - d] : 0u;
    __syncthreads();
    sl[t] += v;
    __syncthreads();
  }
  if (t < NBP) {
    beg[t] = sl[t] - c_t;
    cur[t] = sl[t] - c_t;
  }
  // size class of my bucket and my rank inside it
  const u32 bin = min(c_t, BPL_BINS - 1u);
  u32 rank = 0;
  if (t < NBP) rank = atomicAdd(&bins[bin], 1u);
  __syncthreads();
  if (fits)
    for (u32 j = ps + t; j < pe; j += T) {
      const u32 e = part[j];
      const u32 pos = atomicAdd(&cur[(e >> pg.IB) & low], 1u);
      stage[pos] = (e & 0x80000000u) | entry_abs_index(g, e & idx_mask);
    }
  // descending size order: first position of class b = buckets in larger classes
  if (t < BPL_BINS) sl[t] = bins[BPL_BINS - 1u - t];
  __syncthreads();
  for (u32 d = 1; d < BPL_BINS; d <<= 1) {
    u32 v = (t < BPL_BINS && t >= d) ? sl[t - d] : 0u;
    __syncthreads();
    if (t < BPL_BINS) sl[t] += v;
    __syncthreads();
  }
  if (t < BPL_BINS) bins[BPL_BINS - 1u - t] = sl[t] - bins[BPL_BINS - 1u - t];  // exclusive
  __syncthreads();
  if (t < NBP) {
    const u32 my_pos = bins[bin] + rank;
    ord[my_pos] = t;
    order[(size_t)p * NBP + my_pos] = p * NBP + t;
    // the bucket at position pos sits on lanes [pos L, (pos + 1) L): all in one group (L <= 64); its largest part has
    // ceil(c / L) entries
    atomicMax(&gm[(my_pos << log2_l) / BPL_GROUP], (c_t + L - 1u) >> log2_l);
  }
  __syncthreads();
  if (t == 0) {
    u32 run = 0;
    for (u32 q = 0; q < NG; q++) {
      gb[q] = run;
      run += gm[q] * BPL_GROUP;
    }
    gb[NG] = run;
  }
  __syncthreads();
  const u32 total = gb[NG];
  const bool ok = fits && total <= stride;
  if (!ok && t == 0) atomicOr(err + 1, 1u);
  if (t < NG) {
    BplGroup h;
    h.base = p * stride + gb[t];
    h.m = ok ? gm[t] : 0u;
    grp[p * NG + t] = h;
  }
  if (!ok) return;
  for (u32 j = t; j < total; j += T) {
    u32 q = 0;
    while (q + 1u < NG && gb[q + 1u] <= j) q++;
    const u32 k = (j - gb[q]) / BPL_GROUP, l = j % BPL_GROUP;
    const u32 lane = q * BPL_GROUP + l, kb = ord[lane >> log2_l], r = lane & (L - 1u);
    const u32 i = (k << log2_l) + r;  // entry i of bucket kb belongs to part i mod L
    ents_t[(size_t)p * stride + j] = i < cnt[kb] ? stage[beg[kb] + i] : BPL_ENTRY_PAD;
  }
}

}  // namespace amsm
