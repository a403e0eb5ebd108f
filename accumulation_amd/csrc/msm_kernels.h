// Pippenger windowed-bucket MSM kernels for gfx950 (wave64).
//
// Replaces ark-ec ^0.2.0 `VariableBaseMSM::multi_scalar_mul` (ext; SURVEY.md Appendix C recalls its
// structure: unsigned c-bit windows, 2^c-1 Jacobian buckets per window, running-sum reduction, serial
// Horner).  The GPU pipeline is NOT a translation of that loop nest:
//
//   prep     : (prep_kernels.h) scalars -> signed c-bit digits (halves the buckets) -> one entry per non-zero digit
//              (value = sign | index into the pre-multiplied generator table) -> the entries grouped by bucket:
//              partition histogram, scatter into ~512 partitions, per-partition counting sort, bucket table
//              (start[], number of K0-sized work items, last-of-bucket flags).  Five dispatches (+2 for skewed inputs); the first version
//              (k_digits + rocPRIM radix sort + k_bounds + rocPRIM scan, 14 dispatches) is kept as a fallback.
//   accum L0 : one lane per work item: <= K0 mixed additions (XYZZ, 8M+2S) gathered from the table.
//              Work items are equal sized, so wave64 lanes stay converged whatever the digit
//              distribution is (the reference's harness feeds all-equal scalars).
//   accum L1 : 1 / 4 / 16 lanes per bucket fold its L0 partials; buckets with more than K1 partials go to
//   accum L2 : 32 workgroups per heavy bucket (strided sums + wave-shuffle / LDS tree), then one folding wave.
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
// accumulate L0: lane c owns the fixed-size chunk [chunk_start(c), chunk_start(c+1)) of the SORTED entry list, whatever
// buckets it spans: every lane of a workgroup does exactly K0 (K0b) gathered mixed additions, so all SIMDs stay at full
// occupancy until the end of the kernel and wave64 lanes stay converged for ANY digit distribution
// (uniform, or the all-equal vectors of SURVEY.md F8).  When the bucket id changes inside a chunk the
// lane flushes its running sum as one partial of the finished bucket.  The partials of bucket b are the
// consecutive records item_off[b] + (chunk - first_chunk(b)); k_bounds counted them (items[b]).
// ---------------------------------------------------------------------------------------------
// Cooperative gather: the 64 points a wave needs for one iteration are fetched with LDS-DMA
// (`global_load_lds_dwordx4`), 16 bytes per lane, lanes arranged so that a point's 64/96 bytes are read by
// ADJACENT lanes of ONE wave-instruction: one memory request per point instead of one per 16-byte piece
// (4-6x fewer L2/HBM requests than each lane loading its own point with dwordx4s; measured in DESIGN.md).
// Two regions per wave alternate, so a gather has two mixed additions (~8 us) to land.
// The LDS image is lane-linear (DMA writes base + lane*16): byte o of the wave's region belongs to point
// o / PB, so lane l then reads its own point back with ds_read_b128s.  One region per wave, no barrier.
template <class Fq>
struct GatherLds {
  static constexpr u32 PB = 2 * Fq::W * 4;           // bytes per affine point
  static constexpr u32 WAVE_BYTES = 64 * PB;          // 4 KiB (Pallas) / 6 KiB (BLS12-381)
  static constexpr u32 N_INSTR = WAVE_BYTES / 1024;   // DMA wave-instructions per gather
};

template <class Fq>
AMSM_DEV void gather_issue(const u32* __restrict__ table, u32 idx_own, u32* lds_wave, u32 lane) {
  using G = GatherLds<Fq>;
#pragma unroll
  for (u32 j = 0; j < G::N_INSTR; j++) {
    u32 o = j * 1024u + lane * 16u;          // byte offset of this lane's 16 B inside the wave's LDS region
    u32 pt = o / G::PB, piece = o % G::PB;   // which point of the wave, which 16-byte piece of it
    u32 idx = __shfl(idx_own, pt, 64);       // that point's table index lives in lane `pt`
    const char* src = reinterpret_cast<const char*>(table) + (size_t)idx * G::PB + piece;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(lds_wave) + j * 1024u),
                                     16, 0, 0);
  }
}

template <class Fq>
AMSM_DEV Affine<Fq> gather_read(const u32* lds_wave, u32 lane) {
  const u32* p = lds_wave + lane * (2 * Fq::W);
  Affine<Fq> r;
  r.x = fe_load<Fq>(p);
  r.y = fe_load<Fq>(p + Fq::W);
  return r;
}

// Entry word layout (u32): bit 31 = negate the point, bit 30 = last entry of its bucket (set by k_bounds
// after the sort), bits 0..29 = index into the generator table.
constexpr u32 ENTRY_NEG = 0x80000000u, ENTRY_LAST = 0x40000000u, ENTRY_IDX = 0x3fffffffu;

template <class Fq>
__global__ void __launch_bounds__(256)
    k_accum_l0(const u32* __restrict__ table, const u32* __restrict__ vals_sorted, const u32* __restrict__ start,
               const u32* __restrict__ item_off, MsmGeom g, u32* __restrict__ partials) {
  // two gather regions per wave: the points of mixed addition i+2 are fetched while i and i+1 are computed
  __shared__ __attribute__((aligned(16))) u32 lds[2 * 4 * GatherLds<Fq>::WAVE_BYTES / 4];
  const u32 lane = threadIdx.x & 63u;
  u32* lds_wave0 = lds + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) * (2 * GatherLds<Fq>::WAVE_BYTES / 4);
  u32* lds_wave1 = lds_wave0 + GatherLds<Fq>::WAVE_BYTES / 4;
  // (A persistent grid drawing 64-chunk units from a global counter -- a fixed share of the wave slots for accumulate L0, the
  // rest left to the other MSMs' prep / tail kernels -- was built and measured in round 2: 2 resident workgroups per CU run
  // a lone launch as fast as 3 (the VALU saturates with two waves per SIMD), but a batch gets no faster: the concurrent
  // kernels cost their stand-alone time whichever slots they run in.  profiles/r02_pipeline_experiments.md.)
  const u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 e_valid = start[g.B];  // entries with a non-zero digit
  // chunk size of this workgroup (msm_types.h: two phases, nA is a multiple of the workgroup size)
  const bool ph_a = blockIdx.x * blockDim.x < g.nA;
  const u32 K = ph_a ? g.K0 : g.K0b;
  if (chunk_start(g, blockIdx.x * blockDim.x) >= e_valid) return;  // the grid is sized for n*W entries; zero digits emit none
  u32 s = ph_a ? c * g.K0 : g.T0 + (c - g.nA) * g.K0b;  // K0, K0b multiples of 4: a chunk is a 16-byte aligned run of entries
  u32 e = min(s + K, e_valid);
  // every lane runs all K0 iterations (the gather is cooperative); lanes past their range replay a valid
  // group of entries and skip the arithmetic.  Entries are read 4 at a time (one dwordx4 per 4 mixed
  // additions): 8x fewer requests than word-by-word reads of a stride-128-byte access pattern.
  const u32 last_grp = ((g.E - 1) / 4) * 4;  // the arrays are padded to a multiple of 4 entries
  const uint4* v4 = reinterpret_cast<const uint4*>(vals_sorted);
  uint4 cur = v4[min(s, last_grp) / 4];
  uint4 nxt = v4[min(s + 4, last_grp) / 4];
  // bucket of the first entry: largest b with start[b] <= s.  Everything a flush needs is loaded when a bucket is ENTERED
  // (its end = start[b + 1], its first partial slot) so that closing a bucket costs ONE dependent load (the next bucket's end,
  // for the empty-bucket test): a wave pays a flush whenever any of its lanes closes a bucket -- every iteration on small MSMs.
  u32 b_cur = 0, s_next = 0, slot0 = 0;  // slot0 + c = the partial slot of (bucket, this chunk)
  if (s < e) {
    u32 lo = 0, hi = g.B;
    while (lo < hi) {
      u32 mid = (lo + hi + 1) >> 1;
      if (start[mid] <= s) lo = mid; else hi = mid - 1;
    }
    b_cur = lo;
    s_next = start[b_cur + 1];
    slot0 = item_off[b_cur] - chunk_of(g, start[b_cur]);  // first partial of the bucket, minus the chunk holding its first entry
  }
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  gather_issue<Fq>(table, cur.x & ENTRY_IDX, lds_wave0, lane);
  gather_issue<Fq>(table, cur.y & ENTRY_IDX, lds_wave1, lane);
  for (u32 grp = 0; grp < K; grp += 4) {
    uint4 nn = v4[min(s + grp + 8, last_grp) / 4];
    u32 w[6] = {cur.x, cur.y, cur.z, cur.w, nxt.x, nxt.y};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      u32 k = s + grp + i;
      u32* region = (i & 1) ? lds_wave1 : lds_wave0;  // grp is a multiple of 4: entry parity = i parity
      // The DMA of this iteration's points was issued two mixed additions ago; hipcc does not track it, so wait
      // explicitly: memory operations retire in order, so "at most the N_INSTR newest outstanding" (the gather of
      // the next iteration) means this one has landed.  Then pull the own point into registers and refill the region.
      if (GatherLds<Fq>::N_INSTR == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      Affine<Fq> pt = gather_read<Fq>(region, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      gather_issue<Fq>(table, w[i + 2] & ENTRY_IDX, region, lane);
      if (k < e) {
        xyzz_madd<Fq>(acc, affine_neg_if<Fq>(pt, (w[i] & ENTRY_NEG) != 0));
        if (k + 1 == s_next || k + 1 == e) {  // bucket (or chunk) finished: flush one partial
          xyzz_store<Fq>(partials, slot0 + c, acc);
          acc = xyzz_inf<Fq>();
          if (k + 1 < e) {  // entry k + 1 opens the next NON-EMPTY bucket: the largest b with start[b] <= k + 1
            u32 b = b_cur + 1;  // almost always the very next one (an empty bucket needs a digit value nobody drew)
            u32 sn = start[b + 1];  // (prefetching this one too costs the two registers that drop the occupancy to 2 waves)
            if (sn <= k + 1) {  // bucket b is empty: probe a few more, then binary search (skewed inputs leave long gaps)
              u32 probes = 0;
              do {
                b++;
                sn = start[b + 1];  // start[B] = e_valid > k + 1 ends the probe at the last bucket
                probes++;
              } while (probes < 4u && sn <= k + 1);
              if (sn <= k + 1) {
                u32 lo = b + 1, hi = g.B;
                while (lo < hi) {
                  u32 mid = (lo + hi + 1) >> 1;
                  if (start[mid] <= k + 1) lo = mid; else hi = mid - 1;
                }
                b = lo;
                sn = start[b + 1];
              }
            }
            b_cur = b;
            s_next = sn;
            slot0 = item_off[b] - chunk_of(g, k + 1);  // start[b] == k + 1: the buckets are contiguous
          }
        }
      }
    }
    cur = nxt;
    nxt = nn;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the last (unused) DMA before the LDS is released
}

// ---------------------------------------------------------------------------------------------
// accumulate, bucket per lane (round 3; the layout is written by k_prep_local_t, prep_kernels.h): wave gw owns the group of
// 64 buckets order[64 gw ..], lane l sums the WHOLE bucket order[64 gw + l]: its entry k sits at ents_t[base + 64 k + l]
// (one coalesced 256-byte row per iteration), rows past a bucket's size hold BPL_PAD.  m = the group's largest bucket, so
// all lanes run the same m iterations; buckets are ordered by size, so m exceeds the mean by a few entries only.  The
// points arrive by the same cooperative LDS-DMA gather as in k_accum_l0, two iterations ahead.  Every lane stores its
// finished bucket (empty buckets as the identity): the bucket table needs no clearing and there are no partial records.
// ---------------------------------------------------------------------------------------------
constexpr u32 BPL_PAD = 0x40000000u;  // == BPL_ENTRY_PAD of prep_kernels.h
template <class Fq, int WIDTH>
AMSM_DEV void group_reduce_xyzz(XYZZ<Fq>& acc);
struct BplGroupHdr {
  u32 base, m;
};
// ACC (round 5): the bucket table already holds the sums of the MSM's earlier ranges (an MSM longer than the key's 2^c-pair
// window runs as ranges of 2^c pairs over ONE bucket set: api_pipeline.inc Share) -- a lane starts from its bucket's sum instead
// of the identity; the odd lane of a split bucket still starts empty (the exchange below adds the halves).
template <class Fq, bool ACC>
__global__ void __launch_bounds__(256)
    k_accum_bpl(const u32* __restrict__ table, const u32* __restrict__ ents_t, const BplGroupHdr* __restrict__ grp,
                const u32* __restrict__ order, u32 n_groups, u32 groups_per_part, const u32* __restrict__ flags,
                u32* __restrict__ buckets) {
  __shared__ __attribute__((aligned(16))) u32 lds[2 * 4 * GatherLds<Fq>::WAVE_BYTES / 4];
  if (flags[1]) return;  // the prep overflowed (skewed digits): the host reruns this MSM through the chunked pipeline
  const u32 lane = threadIdx.x & 63u;
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u32* lds_wave0 = lds + wave * (2 * GatherLds<Fq>::WAVE_BYTES / 4);
  u32* lds_wave1 = lds_wave0 + GatherLds<Fq>::WAVE_BYTES / 4;
  // Which group a wave takes.  Group q of every partition is its q-th size class (q = 0: the largest buckets, several times
  // the rows of the last).  Two things were measured with groups in memory order (workgroup w = classes 4 (w mod 4) .. + 3):
  // workgroups go round-robin to the 8 XCDs, so XCDs 0 and 4 got ALL the largest classes (1.5x the balanced time), and with the
  // classes interleaved the long workgroups that start late leave a tail (1.36 ms against 0.96 ms for equal-length waves).
  // So: the four waves of a workgroup take the SAME class q of four different partitions (equal lengths: the workgroup's wave
  // slots and LDS are released together), and the workgroups run through the classes in DESCENDING order -- all of class 0
  // first, round-robin over the XCDs, the short classes last to fill the gaps (longest-processing-time-first).
  u32 gw = blockIdx.x * 4u + wave;
  const u32 n_parts = n_groups / groups_per_part;
  if ((n_parts & 3u) == 0u) {
    const u32 per_class = n_parts >> 2, q = blockIdx.x / per_class, j = blockIdx.x % per_class;
    gw = (j * 4u + wave) * groups_per_part + q;
  }
  if (gw >= n_groups) return;
  const u32 base = __builtin_amdgcn_readfirstlane(grp[gw].base), m = __builtin_amdgcn_readfirstlane(grp[gw].m);
  const u32 b = order[gw * 64u + lane];
  const u32* row = ents_t + base + lane;
  // the first and the last group of a partition hold the two halves of its 64 largest buckets on adjacent lanes
  // (prep_kernels.h: BPL_SPLIT): one exchange after the rows, the even lane stores the bucket
  const u32 q = gw % groups_per_part;
  const bool pairs = q == 0u || q + 1u == groups_per_part;  // uniform per wave
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  if (ACC) {  // (issued before every other memory operation of the wave: the oldest to retire, the in-order waits below hold)
    if (!pairs || !(lane & 1u)) acc = xyzz_load<Fq>(buckets, b);
  }
  if (m) {
    u32 e0 = row[0];
    u32 e1 = m > 1u ? row[64] : BPL_PAD;
    u32 e2 = m > 2u ? row[128] : BPL_PAD;
    gather_issue<Fq>(table, e0 & ENTRY_IDX & ~BPL_PAD, lds_wave0, lane);
    gather_issue<Fq>(table, e1 & ENTRY_IDX & ~BPL_PAD, lds_wave1, lane);
#pragma unroll 2
    for (u32 k = 0; k < m; k++) {
      // one row load per iteration, ALWAYS issued (the clamped row is discarded): the s_waitcnt below counts it
      const u32 e3_ld = row[(size_t)min(k + 3u, m - 1u) * 64u];
      const u32 e3 = k + 3u < m ? e3_ld : BPL_PAD;
      u32* region = (k & 1u) ? lds_wave1 : lds_wave0;
      // outstanding, oldest first: gather(k) | gather(k + 1) | the row load just issued.  Memory operations retire in
      // order, so "at most N_INSTR + 1 outstanding" means gather(k) -- and the row loaded one iteration ago -- have landed
      if (GatherLds<Fq>::N_INSTR == 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      Affine<Fq> pt = gather_read<Fq>(region, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      gather_issue<Fq>(table, e2 & ENTRY_IDX & ~BPL_PAD, region, lane);  // the points of iteration k + 2
      if (!(e0 & BPL_PAD)) xyzz_madd<Fq>(acc, affine_neg_if<Fq>(pt, (e0 & ENTRY_NEG) != 0));
      e0 = e1;
      e1 = e2;
      e2 = e3;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the last (unused) DMA before the LDS is released
  }
  if (pairs) group_reduce_xyzz<Fq, 2>(acc);
  if (!pairs || !(lane & 1u)) xyzz_store<Fq>(buckets, b, acc);
}

// accumulate, bucket-split (k_prep_local_s): lane l of group gw sums part (l mod L) of the bucket at position (64 gw + l) / L; after the
// rows, log2 L exchanges add the parts and the first lane of every bucket stores it.  Same loop as k_accum_bpl.
template <class Fq>
__global__ void __launch_bounds__(256)
    k_accum_bps(const u32* __restrict__ table, const u32* __restrict__ ents_t, const BplGroupHdr* __restrict__ grp,
                const u32* __restrict__ order, u32 n_groups, u32 log2_l, const u32* __restrict__ flags, u32* __restrict__ buckets) {
  __shared__ __attribute__((aligned(16))) u32 lds[2 * 4 * GatherLds<Fq>::WAVE_BYTES / 4];
  if (flags[1]) return;  // the prep overflowed: the host reruns this MSM through the chunked pipeline
  const u32 lane = threadIdx.x & 63u;
  const u32 wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  u32* lds_wave0 = lds + wave * (2 * GatherLds<Fq>::WAVE_BYTES / 4);
  u32* lds_wave1 = lds_wave0 + GatherLds<Fq>::WAVE_BYTES / 4;
  const u32 gw = blockIdx.x * 4u + wave;
  if (gw >= n_groups) return;
  const u32 base = __builtin_amdgcn_readfirstlane(grp[gw].base), m = __builtin_amdgcn_readfirstlane(grp[gw].m);
  const u32* row = ents_t + base + lane;
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  if (m) {
    u32 e0 = row[0];
    u32 e1 = m > 1u ? row[64] : BPL_PAD;
    u32 e2 = m > 2u ? row[128] : BPL_PAD;
    gather_issue<Fq>(table, e0 & ENTRY_IDX, lds_wave0, lane);
    gather_issue<Fq>(table, e1 & ENTRY_IDX, lds_wave1, lane);
#pragma unroll 2
    for (u32 k = 0; k < m; k++) {
      const u32 e3_ld = row[(size_t)min(k + 3u, m - 1u) * 64u];  // always issued: the wait below counts it
      const u32 e3 = k + 3u < m ? e3_ld : BPL_PAD;
      u32* region = (k & 1u) ? lds_wave1 : lds_wave0;
      if (GatherLds<Fq>::N_INSTR == 4) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
      Affine<Fq> pt = gather_read<Fq>(region, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      gather_issue<Fq>(table, e2 & ENTRY_IDX, region, lane);
      if (!(e0 & BPL_PAD)) xyzz_madd<Fq>(acc, affine_neg_if<Fq>(pt, (e0 & ENTRY_NEG) != 0));
      e0 = e1;
      e1 = e2;
      e2 = e3;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  // the L parts of a bucket: butterfly over aligned groups of L lanes (both partners add in the same order)
#pragma unroll 1
  for (u32 w = (1u << log2_l) >> 1; w >= 1u; w >>= 1) {
    XYZZ<Fq> o;
#pragma unroll
    for (int i = 0; i < Fq::L; i++) {
      o.x.v[i] = __shfl_xor(acc.x.v[i], (int)w, 64);
      o.y.v[i] = __shfl_xor(acc.y.v[i], (int)w, 64);
      o.zz.v[i] = __shfl_xor(acc.zz.v[i], (int)w, 64);
      o.zzz.v[i] = __shfl_xor(acc.zzz.v[i], (int)w, 64);
    }
    const bool lowl = (lane & w) == 0;
    XYZZ<Fq> a = lowl ? acc : o;
    XYZZ<Fq> b2 = lowl ? o : acc;
    xyzz_add<Fq>(a, b2);
    acc = a;
  }
  // (position in the partition's size order -> bucket: k_prep_local_s)
  if ((lane & ((1u << log2_l) - 1u)) == 0u) xyzz_store<Fq>(buckets, (size_t)order[(gw * 64u + lane) >> log2_l], acc);
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
// lds must hold 4 XYZZ records (4 * 4*W u32).  Result valid in lane 0 of the workgroup.
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

// accumulate L2: heavy buckets (more than K1 partials: skewed digit distributions) in two steps so that
// the sequential chain per lane stays short whatever the skew: (a) L2_SLICES workgroups per heavy bucket each sum a
// slice of its partials (lane-strided sums, wave shuffles, LDS) into a scratch record, (b) one wave per heavy bucket
// folds the L2_SLICES records.  A single workgroup per bucket cost 3.8 ms for the 16 buckets of an all-equal 2^20
// vector (172 dependent additions per lane).
constexpr u32 L2_SLICES = 32;
template <class Fq>
__global__ void __launch_bounds__(256)
    k_accum_l2a(const u32* __restrict__ partials, const u32* __restrict__ items, const u32* __restrict__ item_off,
                const u32* __restrict__ heavy_count, const u32* __restrict__ heavy_list, u32* __restrict__ scratch) {
  __shared__ __attribute__((aligned(16))) u32 lds[4 * 4 * Fq::W];
  u32 nh = *heavy_count;
  const u32 sidx = blockIdx.x;  // slice
  for (u32 h = blockIdx.y; h < nh; h += gridDim.y) {
    u32 b = heavy_list[h];
    u32 n = items[b], off = item_off[b];
    u32 per = (n + L2_SLICES - 1) / L2_SLICES;
    u32 lo = min(sidx * per, n), hi = min(lo + per, n);
    XYZZ<Fq> acc = xyzz_inf<Fq>();
    for (u32 k = lo + threadIdx.x; k < hi; k += blockDim.x) {
      XYZZ<Fq> p = xyzz_load<Fq>(partials, off + k);
      xyzz_add<Fq>(acc, p);
    }
    block_reduce_xyzz<Fq>(acc, lds);
    if (threadIdx.x == 0) xyzz_store<Fq>(scratch, (size_t)h * L2_SLICES + sidx, acc);
    __syncthreads();
  }
}
template <class Fq>
__global__ void __launch_bounds__(64)
    k_accum_l2b(const u32* __restrict__ scratch, const u32* __restrict__ heavy_count, const u32* __restrict__ heavy_list,
                u32* __restrict__ buckets) {
  u32 nh = *heavy_count;
  for (u32 h = blockIdx.x; h < nh; h += gridDim.x) {
    XYZZ<Fq> acc = xyzz_inf<Fq>();
    if (threadIdx.x < L2_SLICES) acc = xyzz_load<Fq>(scratch, (size_t)h * L2_SLICES + threadIdx.x);
    wave_reduce_xyzz<Fq>(acc);
    if (threadIdx.x == 0) xyzz_store<Fq>(buckets, heavy_list[h], acc);
  }
}

// ---- quad-cooperative tail (ec.h: xyzz_add_quad / xyzz_dbl_quad): logical lane = an aligned quad of lanes ----
// butterfly over the 16 quads of a wave; every quad ends with the wave's sum
template <class Fq>
AMSM_DEV void wave_reduce_xyzz_quad(XYZZ<Fq>& acc) {
#pragma unroll 1
  for (int m = 32; m >= 4; m >>= 1) {
    XYZZ<Fq> o;
#pragma unroll
    for (int i = 0; i < Fq::L; i++) {
      o.x.v[i] = __shfl_xor(acc.x.v[i], m, 64);
      o.y.v[i] = __shfl_xor(acc.y.v[i], m, 64);
      o.zz.v[i] = __shfl_xor(acc.zz.v[i], m, 64);
      o.zzz.v[i] = __shfl_xor(acc.zzz.v[i], m, 64);
    }
    bool low = (threadIdx.x & m) == 0;  // both quads add in the same order
    XYZZ<Fq> a = low ? acc : o;
    XYZZ<Fq> b2 = low ? o : acc;
    xyzz_add_quad<Fq>(a, b2);
    acc = a;
  }
}
// lds: one XYZZ record per wave of the workgroup.  Result valid in the first quad of the workgroup.
template <class Fq>
AMSM_DEV void block_reduce_xyzz_quad(XYZZ<Fq>& acc, u32* lds) {
  wave_reduce_xyzz_quad<Fq>(acc);
  u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) xyzz_store<Fq>(lds, wave, acc);
  __syncthreads();
  if (threadIdx.x < 4) {
    for (u32 w = 1; w < (blockDim.x >> 6); w++) {
      XYZZ<Fq> o = xyzz_load<Fq>(lds, w);
      xyzz_add_quad<Fq>(acc, o);
    }
  }
}
template <class Fq>
AMSM_DEV XYZZ<Fq> xyzz_mul_small_quad(const XYZZ<Fq>& p, u32 k) {
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  if (k == 0) return acc;
  int top = 31 - __clz(k);
  for (int i = top; i >= 0; i--) {
    acc = xyzz_dbl_quad<Fq>(acc);
    if ((k >> i) & 1) xyzz_add_quad<Fq>(acc, p);
  }
  return acc;
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
  __shared__ __attribute__((aligned(16))) u32 lds[4 * 4 * Fq::W];
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

// ---------------------------------------------------------------------------------------------
// Round 4: the bucket reduction as ROW and COLUMN sums.  With j = 1024 a + b (A = nb / 1024 rows of 1024 columns),
//   sum_j (j + 1) B_j = sum_a (1024 a) R_a + sum_b (b + 1) C_b,   R_a = sum_b B_(a,b),  C_b = sum_a B_(a,b):
// every bucket enters two PLAIN sums -- the same two additions per bucket as the running-sum form above, but plain sums split
// into strips of any length and finish with a butterfly, so the dependent chain is a strip plus a few exchange levels instead of
// 2 x 32 additions and a 19-bit multiple per lane (the one-lane k_bucket_reduce + k_fold: 0.8 ms whatever the bucket count, which
// bounded batches of 2^18 / 2^19-pair MSMs; the quad pair: 0.33 ms exposed at 2^19 buckets).
//   k_red2_sums      wave = gw-lane groups: a group sums one row (its lanes take strips of `lrow` = 1024 / gw consecutive
//                    columns) or one column (strips of `srow` = A / gc consecutive rows, gc lanes), strip sequentially, then a
//                    butterfly over the group -> rc[set][0 .. A) = R_a, rc[set][A .. A + 1024) = C_b
//   k_red2_weighted  logical lane (a lane, or a quad) = one of the A + 1024 sums: its small multiple (1024 a: a 19-bit
//                    double-and-add; b + 1), a workgroup reduction -> one partial record per workgroup, folded by k_fold[_quad]
// ---------------------------------------------------------------------------------------------
struct Red2Geom {
  u32 A;      // rows = nb / 1024
  u32 gw;     // lanes per row group (power of two <= 64): row strips of 1024 / gw columns
  u32 gc;     // lanes per column group (power of two <= 64, <= A): column strips of A / gc rows
  u32 row_waves, col_waves;  // waves per set in each mode
};
// butterfly over aligned groups of `width` lanes (a power of two, runtime); every lane of a group ends with the group's sum
template <class Fq>
AMSM_DEV void group_reduce_xyzz_rt(XYZZ<Fq>& acc, u32 width) {
#pragma unroll 1
  for (u32 m = width >> 1; m >= 1u; m >>= 1) {
    XYZZ<Fq> o;
#pragma unroll
    for (int i = 0; i < Fq::L; i++) {
      o.x.v[i] = __shfl_xor(acc.x.v[i], (int)m, 64);
      o.y.v[i] = __shfl_xor(acc.y.v[i], (int)m, 64);
      o.zz.v[i] = __shfl_xor(acc.zz.v[i], (int)m, 64);
      o.zzz.v[i] = __shfl_xor(acc.zzz.v[i], (int)m, 64);
    }
    const bool low = (threadIdx.x & m) == 0;  // both partners add in the same order
    XYZZ<Fq> a = low ? acc : o;
    XYZZ<Fq> b2 = low ? o : acc;
    xyzz_add<Fq>(a, b2);
    acc = a;
  }
}
template <class Fq>
__global__ void __launch_bounds__(256) k_red2_sums(const u32* __restrict__ buckets, u32 nb, Red2Geom r, u32* __restrict__ rc) {
  const u32 set = blockIdx.y;
  const u32 wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
  const size_t base = (size_t)set * nb;
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  if (wave < r.row_waves) {  // rows: 64 / gw of them per wave
    const u32 per = 64u / r.gw, a = wave * per + lane / r.gw, strip = lane % r.gw, lrow = 1024u / r.gw;
    if (a < r.A) {
      const size_t j0 = base + (size_t)a * 1024u + (size_t)strip * lrow;
      acc = xyzz_load<Fq>(buckets, j0);
      for (u32 k = 1; k < lrow; k++) {
        XYZZ<Fq> b = xyzz_load<Fq>(buckets, j0 + k);
        xyzz_add<Fq>(acc, b);
      }
    }
    group_reduce_xyzz_rt<Fq>(acc, r.gw);
    if (a < r.A && strip == 0u) xyzz_store<Fq>(rc, (size_t)set * (r.A + 1024u) + a, acc);
  } else if (wave < r.row_waves + r.col_waves) {  // columns: 64 / gc of them per wave
    const u32 w = wave - r.row_waves, per = 64u / r.gc, b = w * per + lane / r.gc, strip = lane % r.gc, srow = r.A / r.gc;
    if (b < 1024u) {
      const size_t j0 = base + (size_t)strip * srow * 1024u + b;
      acc = xyzz_load<Fq>(buckets, j0);
      for (u32 k = 1; k < srow; k++) {
        XYZZ<Fq> x = xyzz_load<Fq>(buckets, j0 + (size_t)k * 1024u);
        xyzz_add<Fq>(acc, x);
      }
    }
    group_reduce_xyzz_rt<Fq>(acc, r.gc);
    if (b < 1024u && strip == 0u) xyzz_store<Fq>(rc, (size_t)set * (r.A + 1024u) + r.A + b, acc);
  }
}
template <class Fq, bool QUAD>
__global__ void __launch_bounds__(256) k_red2_weighted(const u32* __restrict__ rc, u32 A, u32* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) u32 lds[4 * 4 * Fq::W];
  const u32 set = blockIdx.y, items = A + 1024u;
  const u32 t = (blockIdx.x * blockDim.x + threadIdx.x) >> (QUAD ? 2 : 0);  // logical lane = item
  XYZZ<Fq> total = xyzz_inf<Fq>();
  if (t < items) {
    const XYZZ<Fq> v = xyzz_load<Fq>(rc, (size_t)set * items + t);
    const u32 w = t < A ? t * 1024u : (t - A) + 1u;
    total = QUAD ? xyzz_mul_small_quad<Fq>(v, w) : xyzz_mul_small<Fq>(v, w);
  }
  if (QUAD) block_reduce_xyzz_quad<Fq>(total, lds);
  else block_reduce_xyzz<Fq>(total, lds);
  if (threadIdx.x == 0) xyzz_store<Fq>(out, (size_t)set * gridDim.x + blockIdx.x, total);
}

// Sum of the generators whose scalar equals v: the MSM of a two-valued vector up to one scalar multiplication and its (at most
// eight) exceptions (vec_kernels.h: k_tv_probe found v and listed them).  Lane-strided mixed additions, one partial record per
// workgroup (k_fold sums them).  blockIdx.y = which of up to TV_BATCH vectors of a call (one launch for all: each is a
// latency-bound chain of a few additions and a workgroup reduction).  Workgroup 0 of a vector also copies out its exceptions:
// record t = the scalar as stored (8 words) | the generator (affine, C-ABI radix).
constexpr int TV_BATCH = 8;
struct TvBatch {
  const u32* scalars[TV_BATCH];
  const u32* probe[TV_BATCH];  // the vector's k_tv_probe words
  u32 n[TV_BATCH];
  u32 base_off[TV_BATCH];
};
template <class Fq>
__global__ void __launch_bounds__(256) k_tv_sum(const u32* __restrict__ table, TvBatch b, u32* __restrict__ out, u32* __restrict__ exc_out) {
  __shared__ __attribute__((aligned(16))) u32 lds[4 * 4 * Fq::W];
  const u32 v = blockIdx.y, n = b.n[v], base_off = b.base_off[v];
  const uint4* s4 = (const uint4*)b.scalars[v];
  const u32* pr = b.probe[v];
  const uint4 va = make_uint4(pr[8], pr[9], pr[10], pr[11]), vb = make_uint4(pr[12], pr[13], pr[14], pr[15]);
  const u32 stride = gridDim.x * 256u;
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  for (u32 i = blockIdx.x * 256u + threadIdx.x; i < n; i += stride) {
    const uint4 lo = s4[2 * (size_t)i], hi = s4[2 * (size_t)i + 1];
    if (lo.x == va.x && lo.y == va.y && lo.z == va.z && lo.w == va.w && hi.x == vb.x && hi.y == vb.y && hi.z == vb.z && hi.w == vb.w)
      xyzz_madd<Fq>(acc, affine_load<Fq>(table, (size_t)base_off + i));
  }
  if (blockIdx.x == 0 && threadIdx.x < pr[3] && threadIdx.x < 8u) {
    constexpr u32 REC = 8u + 2u * Fq::W;
    const u32 j = pr[16u + threadIdx.x];
    u32* o = exc_out + ((size_t)v * 8u + threadIdx.x) * REC;
    for (int k = 0; k < 8; k++) o[k] = b.scalars[v][(size_t)j * 8 + k];
    const Affine<Fq> g = affine_export<Fq>(affine_load<Fq>(table, (size_t)base_off + j));
    affine_store<Fq>(o + 8, 0, g);
  }
  block_reduce_xyzz<Fq>(acc, lds);
  if (threadIdx.x == 0) xyzz_store<Fq>(out, (size_t)v * gridDim.x + blockIdx.x, acc);
}

// ---------------------------------------------------------------------------------------------
// Direct sum (round 4): the latency regime.  An MSM of up to 2^14 pairs is a chain of ~14 launches on the sorted pipelines
// (0.28 ms at 2^12 pairs, 0.34 at 2^14) for arithmetic the chip does in 20 us; what it needs is the SHORTEST dependent chain, not
// the fewest additions.  Keys of up to 2^14 generators therefore also carry every multiple a 4-bit signed digit can ask for --
// table[(j - 1) * 64 + w][i] = j 2^(4 w) G_i, j = 1 .. 8, w = 0 .. 63 (512 affine points per generator) -- so that an MSM is a
// plain SUM of n * 64 * 15/16 table points: no buckets, hence no sort, no bucket reduction with its 2^(c-1)-fold weights, and no
// dependence on the digit distribution (a constant vector costs what a uniform one does).  One launch sums them -- lane (g, i)
// adds the points of scalar i's windows [g m, (g + 1) m) by mixed additions, two butterfly levels leave every quad of lanes
// with its sum, the quad-cooperative tree (ec.h) takes the workgroup's 64 quads to one record -- and k_fold_quad adds the
// workgroups' records.  Digits without a carry chain: s' = s + 0x0888...8 (an 8 under each of the 63 low windows), then digit
// w = nibble w of s' - 8 in [-8, 7], the top window's nibble taken as it is (0 .. 8 for every s < 8.47 * 2^252, which covers
// the canonical scalars of both fields; anything above is reported like a scalar that does not fit the other pipelines' windows).
// level w = 2^(4 w) G from the key's WINDOW table (levels 2^(c l) G, plain c-bit windows): the level at or below bit 4 w, then
// at most c - 1 doublings -- every (w, i) on its own lane (a chain of 252 doublings per generator from the generators alone)
// (windows [w0, w0 + wn) of the 64: the table is built in slabs so that the XYZZ scratch stays bounded, round 5)
template <class Fq>
__global__ void __launch_bounds__(256)
    k_ds_levels(const u32* __restrict__ win_table, u32 n, u32 c, u32 W, u32 w0, u32 wn, u32* __restrict__ xyzz_out) {
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * wn) return;
  const u32 w = w0 + t / n, i = t % n;
  u32 l = (4u * w) / c;
  if (l >= W) l = W - 1u;
  const u32 dbl = 4u * w - c * l;
  XYZZ<Fq> a = xyzz_from_affine<Fq>(affine_load<Fq>(win_table, (size_t)l * n + i));
  for (u32 k = 0; k < dbl; k++) a = xyzz_dbl<Fq>(a);
  xyzz_store<Fq>(xyzz_out, t, a);
}
// plane j - 1 (count records each) = j * plane 0, j = 2 .. 8
template <class Fq>
__global__ void __launch_bounds__(256) k_ds_multiples(u32* __restrict__ xyzz, u32 count) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const XYZZ<Fq> p1 = xyzz_load<Fq>(xyzz, i);
  XYZZ<Fq> p2 = xyzz_dbl<Fq>(p1), p3 = p2, p4 = xyzz_dbl<Fq>(p2);
  xyzz_add<Fq>(p3, p1);
  XYZZ<Fq> p5 = p4, p6 = xyzz_dbl<Fq>(p3), p8 = xyzz_dbl<Fq>(p4);
  xyzz_add<Fq>(p5, p1);
  XYZZ<Fq> p7 = p6;
  xyzz_add<Fq>(p7, p1);
  xyzz_store<Fq>(xyzz, (size_t)1 * count + i, p2);
  xyzz_store<Fq>(xyzz, (size_t)2 * count + i, p3);
  xyzz_store<Fq>(xyzz, (size_t)3 * count + i, p4);
  xyzz_store<Fq>(xyzz, (size_t)4 * count + i, p5);
  xyzz_store<Fq>(xyzz, (size_t)5 * count + i, p6);
  xyzz_store<Fq>(xyzz, (size_t)6 * count + i, p7);
  xyzz_store<Fq>(xyzz, (size_t)7 * count + i, p8);
}
template <class Fq, class Fr>
__global__ void __launch_bounds__(256)
    k_direct_sum(const u32* __restrict__ table, u32 key_n, DsBatch b, int mont, u32 m, int group_shift, u32* __restrict__ flags,
                 u32* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) u32 lds[4 * 4 * Fq::W];
  // blockIdx.y: which MSM of the batch -- or, grouped (two sums by bit group_shift of the scalar's index: the IPA rounds; one
  // vector), which index CLASS: this row of workgroups walks the n / 2 indices of its class (n a multiple of 2 << group_shift, the
  // launcher's condition).  Either way its records are partials[blockIdx.y * gridDim.x ...]
  const bool grouped = group_shift >= 0;
  const u32 v = grouped ? 0u : blockIdx.y, cls = grouped ? blockIdx.y : 0u;
  const u32 n = grouped ? b.n[0] >> 1 : b.n[v], base_off = b.base_off[v];
  const u32* __restrict__ scalars = b.scalars[v];
  const u32 L = blockIdx.x * 256u + threadIdx.x;
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  if (L < n * (DS_W / m)) {
    const u32 g = L / n;
    u32 i = L - g * n;
    if (grouped) {
      const u32 sh = (u32)group_shift;
      i = ((i >> sh) << (sh + 1u)) | (cls << sh) | (i & ((1u << sh) - 1u));
    }
    Fe<Fr> s = fe_load<Fr>(scalars + (size_t)i * 8);
    if (mont) s = fe_from_mont<Fr>(s);
    u32 cy = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) s.v[k] = __builtin_addc(s.v[k], k < 7 ? 0x88888888u : 0x08888888u, cy, &cy);
    if ((cy | ((s.v[7] >> 28) > 8u ? 1u : 0u)) && g == 0u) atomicOr(flags, 1u);
    const size_t col = (size_t)base_off + i;
    const u32 w0 = g * m;
    auto digit = [&](u32 w) -> int {
      const int nib = (int)((s.v[w >> 3] >> ((w & 7u) * 4u)) & 15u);
      return w == DS_W - 1u ? (nib > 8 ? 0 : nib) : nib - 8;  // (a top nibble above 8: reported above, never looked up)
    };
    auto fetch = [&](u32 w, int d) -> Affine<Fq> {
      const u32 mag = (u32)(d < 0 ? -d : d);
      return affine_load<Fq>(table, ((size_t)(mag ? mag - 1u : 0u) * DS_W + w) * key_n + col);  // (d = 0: loaded, not added)
    };
    int d = digit(w0);
    Affine<Fq> pt = fetch(w0, d);
    for (u32 k = 0; k < m; k++) {
      const u32 wn = w0 + (k + 1u < m ? k + 1u : k);
      const int dn = digit(wn);
      const Affine<Fq> nx = fetch(wn, dn);  // the next point is requested before this one's addition
      if (d != 0) xyzz_madd<Fq>(acc, affine_neg_if<Fq>(pt, d < 0));
      d = dn;
      pt = nx;
    }
  }
  group_reduce_xyzz<Fq, 4>(acc);  // one-lane additions: afterwards the four lanes of a quad hold the same sum
  block_reduce_xyzz_quad<Fq>(acc, lds);
  if (threadIdx.x == 0) xyzz_store<Fq>(partials, (size_t)blockIdx.y * gridDim.x + blockIdx.x, acc);
}

// fold: wave b sums records [b*n, (b+1)*n) (lane-strided + shuffle butterfly) and writes out[b] (C-ABI radix).
// mirror (may be null): page-locked HOST memory that receives the same record and flag words (round 4: the result used to leave
// by a 128-byte copy command, which queues behind whatever the copy engines are doing -- a 0.6 ms scalar upload of a host-slice
// batch -- and held the slot's `done` event back)
template <class Fq>
__global__ void __launch_bounds__(64) k_fold(const u32* __restrict__ in, u32 n, u32* __restrict__ out,
                                             const u32* __restrict__ flags, u32* __restrict__ mirror) {
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  for (u32 k = threadIdx.x; k < n; k += 64) {
    XYZZ<Fq> p = xyzz_load<Fq>(in, (size_t)blockIdx.x * n + k);
    xyzz_add<Fq>(acc, p);
  }
  wave_reduce_xyzz<Fq>(acc);
  if (threadIdx.x == 0) {
    // the folded record leaves the device: C-ABI Montgomery radix from here on (identity on saturated fields)
    XYZZ<Fq> e;
    e.x = fe_export<Fq>(acc.x);
    e.y = fe_export<Fq>(acc.y);
    e.zz = fe_export<Fq>(acc.zz);
    e.zzz = fe_export<Fq>(acc.zzz);
    xyzz_store<Fq>(out, blockIdx.x, e);
    if (mirror) xyzz_store<Fq>(mirror, blockIdx.x, e);
    // the MSM's flag words ride behind the records: one copy takes both to the host
    if (flags && blockIdx.x == 0) {
      out[(size_t)gridDim.x * (4 * Fq::W)] = flags[0];
      out[(size_t)gridDim.x * (4 * Fq::W) + 1] = flags[1];
      if (mirror) {
        mirror[(size_t)gridDim.x * (4 * Fq::W)] = flags[0];
        mirror[(size_t)gridDim.x * (4 * Fq::W) + 1] = flags[1];
      }
    }
    if (mirror) __threadfence_system();
  }
}

// The same two kernels with a quad of lanes per logical lane (the tail of a blocking MSM call is a chain of ~45 dependent
// point operations on a few waves: 0.24 ms; the quad schedule cuts the depth of each).  grid.x = 4x k_bucket_reduce's.
template <class Fq>
__global__ void __launch_bounds__(256)
    k_bucket_reduce_quad(const u32* __restrict__ buckets, MsmGeom g, u32* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) u32 lds[4 * 4 * Fq::W];
  u32 set = blockIdx.y;
  u32 t = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;  // logical lane: owns buckets [t*s, (t+1)*s)
  XYZZ<Fq> total = xyzz_inf<Fq>();
  if (t < g.red_threads) {
    u32 lo = t * g.red_s;
    XYZZ<Fq> run = xyzz_inf<Fq>(), sum = xyzz_inf<Fq>();
    XYZZ<Fq> bk = xyzz_load<Fq>(buckets, (size_t)set * g.nb + lo + g.red_s - 1);
    for (int k = (int)g.red_s - 1; k >= 0; k--) {
      XYZZ<Fq> nx = bk;  // the next bucket is requested before this one's two additions
      if (k > 0) nx = xyzz_load<Fq>(buckets, (size_t)set * g.nb + lo + k - 1);
      xyzz_add_quad<Fq>(run, bk);
      xyzz_add_quad<Fq>(sum, run);
      bk = nx;
    }
    total = xyzz_mul_small_quad<Fq>(run, lo);
    xyzz_add_quad<Fq>(total, sum);
  }
  block_reduce_xyzz_quad<Fq>(total, lds);
  if (threadIdx.x == 0) xyzz_store<Fq>(out, (size_t)set * gridDim.x + blockIdx.x, total);
}
// Round 6: k_bucket_reduce_quad and k_fold_quad as ONE launch.  The blocks of a set leave their partial records as before and
// take a ticket; the LAST block of the set to arrive folds the set's gridDim.x records (the quad fold's schedule), exports the
// sum to the C-ABI radix, mirrors it (and, for set 0, the MSM's flag words) into page-locked memory and clears the ticket for
// the slot's next MSM.  What it saves is the dependent launch between two latency chains -- the second kernel's dispatch, its
// drain and the ~5 us hand-over -- in every blocking MSM of up to 2^17 buckets and in every round of an IPA opening.
// partial: n_sets * gridDim.x records of scratch; ticket: n_sets words, zero on entry.
template <class Fq>
__global__ void __launch_bounds__(256)
    k_bucket_reduce_fold_quad(const u32* __restrict__ buckets, MsmGeom g, u32* partial, u32* __restrict__ ticket, u32* __restrict__ out,
                              const u32* __restrict__ flags, u32* __restrict__ mirror) {
  __shared__ __attribute__((aligned(16))) u32 lds[4 * 4 * Fq::W];
  __shared__ u32 s_last;
  const u32 set = blockIdx.y;
  {
    const u32 t = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;  // logical lane: owns buckets [t*s, (t+1)*s)
    XYZZ<Fq> total = xyzz_inf<Fq>();
    if (t < g.red_threads) {
      const u32 lo = t * g.red_s;
      XYZZ<Fq> run = xyzz_inf<Fq>(), sum = xyzz_inf<Fq>();
      XYZZ<Fq> bk = xyzz_load<Fq>(buckets, (size_t)set * g.nb + lo + g.red_s - 1);
      for (int k = (int)g.red_s - 1; k >= 0; k--) {
        XYZZ<Fq> nx = bk;
        if (k > 0) nx = xyzz_load<Fq>(buckets, (size_t)set * g.nb + lo + k - 1);
        xyzz_add_quad<Fq>(run, bk);
        xyzz_add_quad<Fq>(sum, run);
        bk = nx;
      }
      total = xyzz_mul_small_quad<Fq>(run, lo);
      xyzz_add_quad<Fq>(total, sum);
    }
    block_reduce_xyzz_quad<Fq>(total, lds);
    if (threadIdx.x == 0) {
      xyzz_store<Fq>(partial, (size_t)set * gridDim.x + blockIdx.x, total);
      __threadfence();  // the record is visible device-wide before the ticket is
      s_last = atomicAdd(ticket + set, 1u) == gridDim.x - 1u ? 1u : 0u;
    }
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();  // acquire: the other blocks' records (other XCDs' L2s) are read from memory
  const u32 n = gridDim.x;
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  const u32* in = partial + (size_t)set * n * (4 * Fq::W);
  if (n <= 4u) {
    // a handful of records (the small sets of 8-bit windows: ONE): the first quad adds them one after the other -- the workgroup
    // tree would be 7 more dependent additions on identities
    if (threadIdx.x < 4u)
      for (u32 k = 0; k < n; k++) {
        const XYZZ<Fq> p = xyzz_load<Fq>(in, k);  // (written by other workgroups of THIS launch: read behind the fence above)
        xyzz_add_quad<Fq>(acc, p);
      }
  } else {
    const u32 k0 = threadIdx.x >> 2, nq = blockDim.x >> 2;
    for (u32 k = k0; k < n; k += nq) {
      const XYZZ<Fq> p = xyzz_load<Fq>(in, k);
      xyzz_add_quad<Fq>(acc, p);
    }
    __syncthreads();  // (lds is reused)
    block_reduce_xyzz_quad<Fq>(acc, lds);
  }
  if (threadIdx.x == 0) {
    XYZZ<Fq> e;
    e.x = fe_export<Fq>(acc.x);
    e.y = fe_export<Fq>(acc.y);
    e.zz = fe_export<Fq>(acc.zz);
    e.zzz = fe_export<Fq>(acc.zzz);
    xyzz_store<Fq>(out, set, e);
    if (mirror) xyzz_store<Fq>(mirror, set, e);
    if (flags && set == 0) {
      const u32 f0 = flags[0], f1 = flags[1];
      out[(size_t)gridDim.y * (4 * Fq::W)] = f0;
      out[(size_t)gridDim.y * (4 * Fq::W) + 1] = f1;
      if (mirror) {
        mirror[(size_t)gridDim.y * (4 * Fq::W)] = f0;
        mirror[(size_t)gridDim.y * (4 * Fq::W) + 1] = f1;
      }
    }
    ticket[set] = 0;
    if (mirror) __threadfence_system();
  }
}
template <class Fq>
__global__ void __launch_bounds__(256) k_fold_quad(const u32* __restrict__ in, u32 n, u32* __restrict__ out,
                                                   const u32* __restrict__ flags, u32* __restrict__ mirror, u32 clear_flags) {
  // round 3: four waves (64 quads) instead of one -- the serial part of the fold drops from n / 16 to n / 64 additions per
  // quad (n = 256 partial records after the reduction of a 2^19-bucket set), then the quad butterfly and one LDS step
  __shared__ __attribute__((aligned(16))) u32 lds[4 * 4 * Fq::W];
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  {
    const u32 k0 = threadIdx.x >> 2, nq = blockDim.x >> 2;
    XYZZ<Fq> p = k0 < n ? xyzz_load<Fq>(in, (size_t)blockIdx.x * n + k0) : acc;
    for (u32 k = k0; k < n; k += nq) {
      XYZZ<Fq> nx = p;
      if (k + nq < n) nx = xyzz_load<Fq>(in, (size_t)blockIdx.x * n + k + nq);
      xyzz_add_quad<Fq>(acc, p);
      p = nx;
    }
  }
  block_reduce_xyzz_quad<Fq>(acc, lds);
  if (threadIdx.x == 0) {
    XYZZ<Fq> e;
    e.x = fe_export<Fq>(acc.x);
    e.y = fe_export<Fq>(acc.y);
    e.zz = fe_export<Fq>(acc.zz);
    e.zzz = fe_export<Fq>(acc.zzz);
    xyzz_store<Fq>(out, blockIdx.x, e);
    if (mirror) xyzz_store<Fq>(mirror, blockIdx.x, e);
    // the MSM's flag words ride behind the records: one copy takes both to the host
    if (flags && blockIdx.x == 0) {
      const u32 f0 = flags[0], f1 = flags[1];
      out[(size_t)gridDim.x * (4 * Fq::W)] = f0;
      out[(size_t)gridDim.x * (4 * Fq::W) + 1] = f1;
      if (mirror) {
        mirror[(size_t)gridDim.x * (4 * Fq::W)] = f0;
        mirror[(size_t)gridDim.x * (4 * Fq::W) + 1] = f1;
      }
      if (clear_flags) {  // the direct sum's flag words: left zeroed for the slot's next MSM (no fill command per MSM)
        const_cast<u32*>(flags)[0] = 0;
        const_cast<u32*>(flags)[1] = 0;
      }
    }
    if (mirror) __threadfence_system();
  }
}

// ---------------------------------------------------------------------------------------------
// Round 6: the JUMP FOLD of an IPA opening (ark_poly_commit::ipa_pc::open ext, under src/ipa_pc_as/mod.rs:454).  After j rounds the
// reference holds the folded key B_k = sum_t S_t G_(t m0 + k), k < m0 = n / 2^j, S_t = the product of the round challenges picked
// by the bits of t -- here the opening never folded the key (every round was a grouped MSM over the ORIGINAL generators), and its
// late rounds are latency chains of ~0.36 ms whatever their logical size.  This kernel computes all m0 folded generators at once so
// that the last log2(m0) rounds can run on the host: m0 MSMs of 2^j pairs that SHARE one scalar vector S and read their bases at
// stride m0.  Shared scalars => shared digits: the host recodes S once (signed c-bit digits over the key's window table
// T_w[i] = 2^(c w) G_i), splits every |digit| into two bytes (lo + 256 hi) and sorts the (t, w) pairs by byte value -- one list per
// (byte half, value), cut into PIECES of a few dozen entries so that the launch fills the chip.  A workgroup of JUMP_WAVES waves takes
// one piece: wave q adds entries q, q + JUMP_WAVES, .. for 64 outputs k at once (lane = k: the 64 points T_w[t m0 + k ..] of an entry are
// ONE contiguous row of the table: coalesced), an LDS tree adds the waves' sums, and the piece's sum lands in bucket (value - 1) of set
// ((piece number within its list, half), k) -- pieces of one list are buckets of the SAME weight in different sets.  The sets then go
// through the ordinary weighted bucket reduction (k_bucket_reduce_fold_quad: sum_v v bucket_v) and the host adds the sets of an
// output: lo + 2^8 hi.
// entries[e] = (w * stride + t * m0) | neg << 31;  list_off[l] .. list_off[l + 1]: piece l's entries;  list_slot[l] = (2 piece + half) nb +
// value - 1.  buckets: 2 n_pieces_max m0 sets of nb records, ZEROED by the launcher (a value without entries stays the identity).
// grid = (pieces, m0 / 64).
// ---------------------------------------------------------------------------------------------
constexpr u32 JUMP_WAVES = 4;  // (8 waves of 159 VGPRs: ONE workgroup per CU -- 250 us for 2^16 generators; 4: three per CU)
template <class Fq>
__global__ void __launch_bounds__(64 * JUMP_WAVES)
    k_ipa_jump_accum(const u32* __restrict__ table, const u32* __restrict__ entries, const u32* __restrict__ list_off,
                     const u32* __restrict__ list_slot, u32 m0, u32 nb, u32* __restrict__ buckets) {
  __shared__ __attribute__((aligned(16))) u32 lds[(JUMP_WAVES / 2) * 64 * 4 * Fq::W];
  const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const u32 k = blockIdx.y * 64u + lane;
  const u32 lo = list_off[blockIdx.x], hi = list_off[blockIdx.x + 1];
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  {
    // entry words run TWO iterations ahead of their row, rows one ahead of their addition: an iteration used to be the entry's
    // latency + max(the row's, the addition's) -- ~7 us, 100 us per workgroup (rocprofv3, round 6) -- with the address of the next row
    // already in a register it is the longer of the two
    u32 e = lo + wave;
    u32 w0 = e < hi ? entries[e] : 0u;
    u32 w1 = e + JUMP_WAVES < hi ? entries[e + JUMP_WAVES] : w0;
    Affine<Fq> pt = affine_load<Fq>(table, (size_t)(w0 & 0x7fffffffu) + k);
    for (; e < hi; e += JUMP_WAVES) {
      const u32 w2 = e + 2u * JUMP_WAVES < hi ? entries[e + 2u * JUMP_WAVES] : w1;
      const Affine<Fq> nx = affine_load<Fq>(table, (size_t)(w1 & 0x7fffffffu) + k);  // the next row is requested before this addition
      xyzz_madd<Fq>(acc, affine_neg_if<Fq>(pt, (w0 >> 31) != 0u));
      w0 = w1;
      w1 = w2;
      pt = nx;
    }
  }
  // 8 -> 4 -> 2 -> 1 waves through LDS (slot = wave within the receiving half, one record per lane)
#pragma unroll 1
  for (u32 h = JUMP_WAVES / 2; h >= 1u; h >>= 1) {
    if (wave >= h && wave < 2u * h) xyzz_store<Fq>(lds, (size_t)(wave - h) * 64u + lane, acc);
    __syncthreads();
    if (wave < h) {
      const XYZZ<Fq> o = xyzz_load<Fq>(lds, (size_t)wave * 64u + lane);
      xyzz_add<Fq>(acc, o);
    }
    __syncthreads();
  }
  if (wave == 0u) {
    const u32 slot = list_slot[blockIdx.x], grp = slot / nb, v = slot - grp * nb;  // grp = 2 * piece + half
    xyzz_store<Fq>(buckets, ((size_t)grp * m0 + k) * nb + v, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// Key preparation
// ---------------------------------------------------------------------------------------------
// Batched XYZZ -> affine (Montgomery's trick): lane g converts points g, g + T, g + 2T, ... (T = lanes of the grid, so a
// wave's loads stay coalesced), K per lane, with ONE field inversion: 9 multiplications per point + 1/K of the ~260-unit
// (Pallas) / ~480-unit (BLS12-381) Fermat inversion, instead of 5 + a whole one.  Identity (ZZ = 0) -> (0, 0).
// ABI: write the C-ABI radix.
template <class Fq, int K, bool ABI>
__global__ void __launch_bounds__(256) k_batch_to_affine(const u32* __restrict__ xyzz, u32 n, u32* __restrict__ out) {
  const u32 T = gridDim.x * blockDim.x, g = blockIdx.x * blockDim.x + threadIdx.x;
  const u32* zz0 = xyzz + 2 * Fq::W;  // ZZ | ZZZ of point 0
  Fe<Fq> prefix[K];
  Fe<Fq> run = fe_one<Fq>();
#pragma unroll
  for (int k = 0; k < K; k++) {
    u32 i = g + (u32)k * T;
    prefix[k] = run;
    if (i < n) {
      const u32* q = zz0 + (size_t)i * (4 * Fq::W);
      Fe<Fq> w = fe_mul<Fq>(fe_load<Fq>(q), fe_load<Fq>(q + Fq::W));
      if (!fe_is_zero<Fq>(w)) run = fe_mul<Fq>(run, w);
    }
  }
  Fe<Fq> inv = fe_inv<Fq>(run);
#pragma unroll
  for (int k = K - 1; k >= 0; k--) {
    u32 i = g + (u32)k * T;
    if (i < n) {
      XYZZ<Fq> p = xyzz_load<Fq>(xyzz, i);
      Fe<Fq> w = fe_mul<Fq>(p.zz, p.zzz);
      Affine<Fq> r;
      if (fe_is_zero<Fq>(w)) {
        r.x = fe_zero<Fq>();
        r.y = fe_zero<Fq>();
      } else {
        Fe<Fq> iw = fe_mul<Fq>(inv, prefix[k]);  // 1 / (ZZ * ZZZ) of this point
        inv = fe_mul<Fq>(inv, w);
        r.x = fe_mul<Fq>(p.x, fe_mul<Fq>(iw, p.zzz));
        r.y = fe_mul<Fq>(p.y, fe_mul<Fq>(iw, p.zz));
        if (ABI) r = affine_export<Fq>(r);
      }
      affine_store<Fq>(out, i, r);
    }
  }
}

// table[(level)*stride + i] = 2^c * table[(level-1)*stride + i]  (affine in, affine out).  XYZZ_OUT: leave the result
// unconverted in xyzz_out[i] for k_batch_to_affine (large keys: the per-point inversion is 2/3 of this kernel).
template <class Fq, bool XYZZ_OUT>
__global__ void __launch_bounds__(256)
    k_precompute_level(u32* __restrict__ table, u32 stride, u32 level, u32 c, u32* __restrict__ xyzz_out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= stride) return;
  Affine<Fq> p = affine_load<Fq>(table, (size_t)(level - 1) * stride + i);
  XYZZ<Fq> a = xyzz_inf<Fq>();
  if (!affine_is_inf<Fq>(p)) {
    a = xyzz_dbl_affine<Fq>(p);
    for (u32 k = 1; k < c; k++) a = xyzz_dbl<Fq>(a);
  }
  if (XYZZ_OUT) {
    xyzz_store<Fq>(xyzz_out, i, a);
  } else {
    affine_store<Fq>(table, (size_t)level * stride + i, affine_is_inf<Fq>(p) ? p : xyzz_to_affine<Fq>(a));
  }
}

// All levels of a SMALL key in one pass (round 4): lane i walks its generator through the W - 1 levels (c doublings each) and
// leaves them unconverted in xyzz_out[(w - 1) * n + i]; ONE k_batch_to_affine over (W - 1) n points follows.  The level-by-level
// form costs a kernel with a Fermat inversion per point and level: 31 launches of 150 us for a 2^12-generator key.
template <class Fq>
__global__ void __launch_bounds__(64) k_precompute_all_levels(const u32* __restrict__ table, u32 n, u32 c, u32 W, u32* __restrict__ xyzz_out) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Affine<Fq> p = affine_load<Fq>(table, i);
  XYZZ<Fq> a = xyzz_from_affine<Fq>(p);
  for (u32 w = 1; w < W; w++) {
    for (u32 k = 0; k < c; k++) a = xyzz_dbl<Fq>(a);
    xyzz_store<Fq>(xyzz_out, (size_t)(w - 1u) * n + i, a);
  }
}

// out[i] = l[i] + x * r[i] for affine point vectors (the commitment-key fold `key_l += key_r * xi` of the IPA
// opening, ark_poly_commit::ipa_pc ext, under src/ipa_pc_as/mod.rs:454): one lane per point, left-to-right
// double-and-add over the `nbits` low bits of the canonical scalar x, then one inversion back to affine.
struct FoldDigits {  // the scalar as kernel-argument digit masks (uniform over the grid), recoded once on the host
  // plain form: x in non-adjacent form (digits +1 / -1 / 0, a third non-zero) in pos1 / neg1, up to 256 digits.
  // GLV form (host_glv.h): x = k1 + k2 lambda, |k_i| ~ 2^128: NAF(k1) in pos1 / neg1, NAF(k2) in pos2 / neg2 (signs folded
  // in), and beta (C-ABI Montgomery words) with phi(x, y) = (beta x, y) = [lambda](x, y): half the doublings.
  u32 pos1[8], neg1[8], pos2[5], neg2[5];
  u32 nd;         // digit positions to run
  u32 beta[12];   // W words used
};
// ABI = true: l, r, out are caller-visible device buffers (C-ABI Montgomery radix in and out); false: key tables
// (device radix, amsm_bases_fold)
// XYZZ_OUT: leave the sums unconverted in out (XYZZ records, internal radix) for k_batch_to_affine.
// JAC: the ladder runs in Jacobian coordinates (ec.h: a doubling of 3M + 4S instead of 6M + 3S; the ladder is 3/4 doublings)
template <class Fq, bool ABI, bool XYZZ_OUT, bool GLV, bool JAC = false>
__global__ void __launch_bounds__(256)
    k_points_fold(const u32* __restrict__ l, const u32* __restrict__ r, u32 n, FoldDigits d, u32* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<Fq> pr = affine_load<Fq>(r, i);
  if (ABI) pr = affine_import<Fq>(pr);
  Affine<Fq> pr2 = pr;
  if (GLV) pr2.x = fe_mul<Fq>(fe_import<Fq>(fe_from_words<Fq>(d.beta)), pr.x);  // phi(r_i); (0, 0) stays (0, 0)   [< 1.1p]
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  Affine<Fq> pl = affine_load<Fq>(l, i);
  if (ABI) pl = affine_import<Fq>(pl);
  if constexpr (JAC) {
    Jac<Fq> ja = jac_inf<Fq>();
    for (int bit = (int)d.nd - 1; bit >= 0; bit--) {
      ja = jac_dbl<Fq>(ja);
      const u32 w = (u32)bit >> 5, sh = (u32)bit & 31u;
      const bool p1 = (d.pos1[w] >> sh) & 1u, m1 = (d.neg1[w] >> sh) & 1u;
      if (p1 | m1) jac_madd<Fq>(ja, affine_neg_if<Fq>(pr, m1));
      if (GLV) {
        const bool p2 = (d.pos2[w] >> sh) & 1u, m2 = (d.neg2[w] >> sh) & 1u;
        if (p2 | m2) jac_madd<Fq>(ja, affine_neg_if<Fq>(pr2, m2));
      }
    }
    jac_madd<Fq>(ja, pl);
    acc = xyzz_from_jac<Fq>(ja);
  } else {
    for (int bit = (int)d.nd - 1; bit >= 0; bit--) {
      acc = xyzz_dbl<Fq>(acc);
      const u32 w = (u32)bit >> 5, sh = (u32)bit & 31u;
      const bool p1 = (d.pos1[w] >> sh) & 1u, m1 = (d.neg1[w] >> sh) & 1u;
      if (p1 | m1) xyzz_madd<Fq>(acc, affine_neg_if<Fq>(pr, m1));
      if (GLV) {
        const bool p2 = (d.pos2[w] >> sh) & 1u, m2 = (d.neg2[w] >> sh) & 1u;
        if (p2 | m2) xyzz_madd<Fq>(acc, affine_neg_if<Fq>(pr2, m2));
      }
    }
    xyzz_madd<Fq>(acc, pl);
  }
  if (XYZZ_OUT) {
    xyzz_store<Fq>(out, i, acc);
    return;
  }
  Affine<Fq> res = xyzz_to_affine<Fq>(acc);
  if (ABI) res = affine_export<Fq>(res);
  affine_store<Fq>(out, i, res);
}

// The same fold over a key that carries its window multiples (level w = 2^(c w) G, the precomputed table of the MSM): with
// x = sum_w x_w 2^(c w), x r_i = sum_w x_w T_w[r_i] is a joint ladder of c + 1 doublings instead of `nbits` -- the additions
// (one per non-zero NAF digit of every x_w) stay what they were.  A 128-bit challenge over 17-bit windows: 18 doublings + ~45
// mixed additions instead of 128 + ~43 (the first fold of an IPA opening, which is half of all its fold work, always runs over
// the original key).  The host flattens the digits into one list of additions, most significant position first; they are
// uniform over the grid, so the next table row is requested while the current addition runs.
struct FoldTabOps {
  uint16_t op[160];  // bits 0..4: digit position, 5..9: level, 15: negate
  u32 n_ops, nd;
};
template <class Fq, bool XYZZ_OUT>
__global__ void __launch_bounds__(256)
    k_points_fold_tab(const u32* __restrict__ table, u32 stride, u32 n, FoldTabOps d, u32* __restrict__ out) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 r_i = n + i;  // r half of the key: generators [n, 2n)
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  u32 k = 0;
  Affine<Fq> nx = affine_load<Fq>(table, i);  // (the l point when there is nothing to add)
  if (d.n_ops) nx = affine_load<Fq>(table, (size_t)((d.op[0] >> 5) & 31u) * stride + r_i);
  for (int bit = (int)d.nd - 1; bit >= 0; bit--) {
    acc = xyzz_dbl<Fq>(acc);
    while (k < d.n_ops && (int)(d.op[k] & 31u) == bit) {
      const Affine<Fq> cur = nx;
      const bool neg = (d.op[k] >> 15) & 1u;
      k++;
      nx = k < d.n_ops ? affine_load<Fq>(table, (size_t)((d.op[k] >> 5) & 31u) * stride + r_i) : affine_load<Fq>(table, i);
      xyzz_madd<Fq>(acc, affine_neg_if<Fq>(cur, neg));
    }
  }
  xyzz_madd<Fq>(acc, nx);  // + l_i
  if (XYZZ_OUT) {
    xyzz_store<Fq>(out, i, acc);
    return;
  }
  affine_store<Fq>(out, i, xyzz_to_affine<Fq>(acc));
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

// C-ABI radix <-> internal radix of a point array (key load / key read; not launched for saturated fields)
template <class Fq>
__global__ void __launch_bounds__(256) k_points_import(const u32* __restrict__ src, u32* __restrict__ dst, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine_store<Fq>(dst, i, affine_import<Fq>(affine_load<Fq>(src, i)));
}
template <class Fq>
__global__ void __launch_bounds__(256) k_points_export(const u32* __restrict__ src, u32* __restrict__ dst, u32 n) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  affine_store<Fq>(dst, i, affine_export<Fq>(affine_load<Fq>(src, i)));
}

template <class Fq>
struct AffineWords {  // an affine point as kernel-argument words (C-ABI Montgomery form)
  u32 w[2 * Fq::W];
};

// table[i] = G_{first + i} = k * G, k = rng_scalar(seed, first + i); generator (gx, gy) passed in (C-ABI) Montgomery form.
// `first`: a shard of a multi-device key generates its own range of the stream.
template <class Fq>
__global__ void __launch_bounds__(256)
    k_generate_bases(u32* __restrict__ table, u64 seed, u32 first, u32 n, AffineWords<Fq> gw) {
  u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Affine<Fq> gen;
  gen.x = fe_import<Fq>(fe_from_words<Fq>(gw.w));
  gen.y = fe_import<Fq>(fe_from_words<Fq>(gw.w + Fq::W));
  u32 k[8];
  rng_scalar(seed, first + i, k);
  XYZZ<Fq> acc = xyzz_inf<Fq>();
  for (int bit = 253; bit >= 0; bit--) {
    acc = xyzz_dbl<Fq>(acc);
    if ((k[bit >> 5] >> (bit & 31)) & 1) xyzz_madd<Fq>(acc, gen);
  }
  affine_store<Fq>(table, i, xyzz_to_affine<Fq>(acc));
}

}  // namespace amsm
