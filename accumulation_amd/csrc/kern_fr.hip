// Scalar-field kernels (digit extraction, synthetic vectors, Hadamard / linear-combination / t-vector
// loops) instantiated for the scalar fields of both curves, plus the curve-independent bounds kernel.
#include "launch.h"
#include "vec_kernels.h"
#include "prep_kernels.h"

#include <algorithm>
#include <cmath>
#include <atomic>
#include <cstdlib>
#include <cstring>

namespace amsm {

static inline u32 cdiv_(u32 a, u32 b) { return (a + b - 1) / b; }
// grid of a streaming (grid-stride) kernel: enough 256-lane workgroups to fill the wave slots a few times over, not one
// per 256 elements beyond that (64 per CU: one element per lane up to 2^22 elements; round 2, 2^22 elements: 64 -> 0.57-0.79 of
// 8 TB/s, 8 -> 0.52-0.75)
static u32 stream_grid(u32 n) {
  constexpr u32 per_cu = 64;
  static const u32 cus = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256u;
    return (u32)std::max(1, prop.multiProcessorCount);
  }();
  return std::max(1u, std::min(cdiv_(n, 256), cus * per_cu));
}

void launch_bounds(hipStream_t st, const void* keys_sorted, bool keys16, u32* vals_sorted, MsmGeom g, u32* start,
                   u32* items) {
  if (keys16)
    hipLaunchKernelGGL((k_bounds<uint16_t>), dim3(cdiv_(g.B, 256)), dim3(256), 0, st, (const uint16_t*)keys_sorted,
                       vals_sorted, g, start, items);
  else
    hipLaunchKernelGGL((k_bounds<u32>), dim3(cdiv_(g.B, 256)), dim3(256), 0, st, (const u32*)keys_sorted, vals_sorted,
                       g, start, items);
}

// partitions of 2^SH consecutive buckets, at most PREP_MAX_P of them (about 512 when the bucket count allows)
constexpr u32 PREP_MAX_P = 4096;
static PrepGeom prep_geom(const MsmGeom& g) {
  PrepGeom pg;
  pg.SH = 4;
  while (((g.B + (1u << pg.SH) - 1u) >> pg.SH) > 512u && pg.SH < 10u) pg.SH++;
  while (((g.B + (1u << pg.SH) - 1u) >> pg.SH) > PREP_MAX_P) pg.SH++;
  pg.P = (g.B + (1u << pg.SH) - 1u) >> pg.SH;
  pg.SPB = g.S <= 16u ? 512u : 256u;  // SPB * S <= 8192 staged entries
  unsigned long long max_idx = (unsigned long long)g.base_off + g.n - 1ull +
                               (g.precomp ? (unsigned long long)(g.W - 1u) * g.table_stride : 0ull);
  if (g.idx_rel_bits) max_idx = ((unsigned long long)(g.W - 1u) << g.idx_rel_bits) | ((1ull << g.idx_rel_bits) - 1ull);
  pg.IB = 1;
  while ((max_idx >> pg.IB) != 0ull) pg.IB++;
  // the entry word has 31 bits for index + bucket-id low bits: trade partition size for partitions when it is tight
  while (pg.IB + pg.SH > 31u && pg.SH > 0u && ((g.B + (1u << (pg.SH - 1)) - 1u) >> (pg.SH - 1)) <= PREP_MAX_P) pg.SH--;
  pg.P = (g.B + (1u << pg.SH) - 1u) >> pg.SH;
  // k_prep_local stages a partition in LDS when it fits: a 148 KiB budget (counters + scan words + entries; one
  // 1024-lane workgroup per CU) holds the ~32 k entries of one of 512 partitions at 2^20 pairs with 4 k to spare
  const u32 budget_words = 37888u, fixed_words = 2u * (1u << pg.SH) + 1024u;
  pg.CAP = budget_words > fixed_words + 4096u ? budget_words - fixed_words : 0u;
  // heavy: several times the expected size AND big enough for 64 workgroups to beat one (below, one workgroup is fine)
  pg.HEAVY = std::max<u32>(4u * (g.E / pg.P + 1u), 1u << 17);
  pg.FIX = 0;
  return pg;
}
// words per per-bucket array of the heavy-partition path (every partition's 2^SH buckets, padded)
static size_t prep_heavy_words(const MsmGeom& g) { return (size_t)g.B + 4096 + 64; }
size_t prep_small_words(const MsmGeom& g) { return 4 * (PREP_MAX_P + 1) + 1 + 3 * prep_heavy_words(g) + PREP_MAX_HEAVY; }
static size_t prep_local_lds(const PrepGeom& pg) { return (2 * (size_t)(1u << pg.SH) + 1024 + pg.CAP) * sizeof(u32); }
// more than 64 KiB of dynamic LDS needs the attribute, once per DEVICE (the attribute is per device: a second context on
// another GPU of the same process needs its own opt-in)
constexpr size_t PREP_LDS_LIMIT = 160 * 1024;  // gfx950: 160 KiB per workgroup
static std::atomic<unsigned long long> prep_local_attr_devices{0};
static void prep_local_attr() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
  const unsigned long long bit = 1ull << dev;
  if (dev != 63 && (prep_local_attr_devices.load(std::memory_order_acquire) & bit)) return;
  (void)hipFuncSetAttribute((const void*)k_prep_local, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  prep_local_attr_devices.fetch_or(bit, std::memory_order_release);
}
static size_t prep_scatter_lds(const MsmGeom& g, const PrepGeom& pg) {
  const size_t cap = (size_t)pg.SPB * g.S;
  return (3 * (size_t)pg.P + cap) * sizeof(u32) + cap * sizeof(uint16_t);
}
bool prep_supported(const MsmGeom& g) {
  if (g.n == 0 || g.S > 32u) return false;  // S > 32 <=> c < 8: tiny problems, the rocPRIM chain is fine there
  PrepGeom pg = prep_geom(g);
  // entry word = negate | bucket-id low bits << IB | index; k_prep_local keeps 2 * 2^SH + 256 words in LDS; the dynamic
  // LDS of every kernel of the chain must fit a workgroup (wide windows / many bucket sets: fall back to rocPRIM)
  return pg.SH <= 12u && pg.IB + pg.SH <= 31u && prep_scatter_lds(g, pg) <= PREP_LDS_LIMIT &&
         prep_local_lds(pg) <= PREP_LDS_LIMIT && (pg.P + 256) * sizeof(u32) <= 64 * 1024;
}

// ---- bucket-per-lane prep (k_prep_local_t): partitions of exactly 1024 buckets, one 1024-lane workgroup each ----
// words of LDS k_prep_local_t keeps beside its CAP staged entries (the kernel carves the same layout)
constexpr u32 BPL_LOCAL_FIXED_WORDS = 3u * BPL_NB + 2u * BPL_BINS + 2u * BPL_GROUPS + 2u;
// round 4: the whole 160 KiB a workgroup may declare (round 3: 148 KiB, and 1792 more fixed words) -- a plain-key MSM of 2^20
// pairs over 16 bucket sets of 2^15 puts 32 768 entries into a partition, 36 200 into those of the top window on BLS12-381
constexpr u32 BPL_LOCAL_BUDGET_WORDS = 40960u;
static PrepGeom prep_bpl_geom(const MsmGeom& g) {
  PrepGeom pg;
  pg.SH = 10;
  pg.P = g.B >> pg.SH;
  pg.SPB = g.S <= 16u ? 512u : 256u;  // SPB * S <= 8192 staged entries
  unsigned long long max_idx = (unsigned long long)g.base_off + g.n - 1ull +
                               (g.precomp ? (unsigned long long)(g.W - 1u) * g.table_stride : 0ull);
  if (g.idx_rel_bits) max_idx = ((unsigned long long)(g.W - 1u) << g.idx_rel_bits) | ((1ull << g.idx_rel_bits) - 1ull);
  pg.IB = 1;
  while ((max_idx >> pg.IB) != 0ull) pg.IB++;
  pg.CAP = BPL_LOCAL_BUDGET_WORDS - BPL_LOCAL_FIXED_WORDS;  // ~35.3 k entries: a partition of a 2^20-pair MSM holds ~26.6 k
  pg.HEAVY = 0xffffffffu;
  // fixed partitions: what uniform digits bring plus a quarter and a constant, at most what the LDS stage takes (a partition
  // beyond it takes the skew fallback either way)
  const unsigned long long per = std::max<unsigned long long>(((unsigned long long)g.E + pg.P - 1) / std::max(1u, pg.P), g.part_max);
  pg.FIX = (u32)std::min<unsigned long long>(pg.CAP, (per + per / 4ull + 2048ull + 15ull) & ~15ull);
  return pg;
}
// 8-byte interchange entries the partition pass may write (fixed partitions reserve FIX each)
size_t prep_bpl_part_entries(const MsmGeom& g) {
  PrepGeom pg = prep_bpl_geom(g);
  return std::max<size_t>(g.E, pg.FIX ? (size_t)pg.P * pg.FIX : 0);
}
static size_t prep_bpl_local_lds(const PrepGeom& pg) { return (size_t)(BPL_LOCAL_FIXED_WORDS + pg.CAP) * sizeof(u32); }
static size_t prep_bpl_scatter_lds(const MsmGeom& g, const PrepGeom& pg) {
  const size_t cap = (size_t)pg.SPB * g.S;
  return (3 * (size_t)pg.P + 2 * cap) * sizeof(u32) + cap * sizeof(uint16_t);
}
u32 prep_bpl_partitions(const MsmGeom& g) { return g.B >> 10; }
u32 prep_bpl_groups_per_partition() { return BPL_GROUPS; }  // 17 groups of 64 lane slots (prep_kernels.h: BPL_SPLIT)
// transposed entries per partition: the expected partition size + 25 % (the padding to each group's largest bucket is ~6 %
// on uniform digits) + a constant; anything that does not fit is a skewed input and takes the fallback
u32 prep_bpl_stride(const MsmGeom& g) {
  const u32 P = std::max(1u, prep_bpl_partitions(g));
  const unsigned long long per = std::max<unsigned long long>(((unsigned long long)g.E + P - 1) / P, g.part_max);
  return (u32)((per + per / 4ull + 4096ull + 1023ull) & ~1023ull);
}
bool prep_bpl_supported(const MsmGeom& g) {
  // one bucket set per group over a precomputed key, W per group over a plain one (round 4); <= 16 entry slots per scalar
  if (g.S > 32u || g.n == 0 || g.n_sets != g.groups * sets_per_group(g)) return false;
  if (g.B < (1u << 16) || (g.B & 1023u) || (g.B >> 10) > PREP_MAX_P || ((g.B >> 10) & 3u)) return false;
  PrepGeom pg = prep_bpl_geom(g);
  if (pg.IB > 30u) return false;
  // the expected partition -- the fullest one -- must fit the LDS stage with room for the digits' spread (sigma ~ 180 entries at
  // 30 k: 1024 on the average partition, 5 sigma on the fullest)
  if ((unsigned long long)g.E / pg.P + 1024ull > pg.CAP) return false;
  if (g.part_max && (unsigned long long)g.part_max + 5ull * (unsigned long long)std::sqrt((double)g.part_max) > pg.CAP) return false;
  return prep_bpl_scatter_lds(g, pg) <= PREP_LDS_LIMIT && prep_bpl_local_lds(pg) <= PREP_LDS_LIMIT &&
         (unsigned long long)pg.P * prep_bpl_stride(g) < (1ull << 31);
}
static std::atomic<unsigned long long> prep_bpl_attr_devices{0};
template <class KS>
static void prep_bpl_attr(KS scatter_kernel) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
  const unsigned long long bit = 1ull << dev;
  (void)hipFuncSetAttribute((const void*)scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PREP_LDS_LIMIT);
  if (dev != 63 && (prep_bpl_attr_devices.load(std::memory_order_acquire) & bit)) return;
  static_assert((BPL_LOCAL_BUDGET_WORDS) * sizeof(u32) <= PREP_LDS_LIMIT, "k_prep_local_t's LDS must fit a workgroup");
  (void)hipFuncSetAttribute((const void*)k_prep_local_t, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PREP_LDS_LIMIT);
  prep_bpl_attr_devices.fetch_or(bit, std::memory_order_release);
}

// ---- bucket-split prep (k_prep_local_s): partitions of 1024 / L buckets = 1024 lanes, 32-bit interchange entries ----
// words of LDS k_prep_local_s keeps beside its CAP staged entries: cnt / cur / beg / ord (4 NBP), 1024 scan words, the size
// classes, rows and bases of the 16 groups -- ONE expression for the geometry, the launch size and the kernel's carving
// (round 3 budgeted 3 NBP + 1024 + 34 after the size ordering had added NBP + BPL_BINS: the top of stage[] lay past the
// allocation, reachable once a partition held more than CAP - NBP - 256 entries)
constexpr u32 bps_local_fixed_words(u32 nbp) { return 4u * nbp + 1024u + BPL_BINS + 2u * BPS_GROUPS + 2u; }
constexpr u32 BPS_LOCAL_BUDGET_WORDS = 37888u;  // 148 KiB
static PrepGeom prep_bps_geom(const MsmGeom& g, u32 log2_l) {
  PrepGeom pg;
  pg.SH = 10u - log2_l;
  pg.P = g.B >> pg.SH;
  pg.SPB = g.S <= 16u ? 512u : 256u;
  unsigned long long max_idx = (unsigned long long)g.base_off + g.n - 1ull +
                               (g.precomp ? (unsigned long long)(g.W - 1u) * g.table_stride : 0ull);
  if (g.idx_rel_bits) max_idx = ((unsigned long long)(g.W - 1u) << g.idx_rel_bits) | ((1ull << g.idx_rel_bits) - 1ull);
  pg.IB = 1;
  while ((max_idx >> pg.IB) != 0ull) pg.IB++;
  pg.CAP = BPS_LOCAL_BUDGET_WORDS - bps_local_fixed_words(1u << pg.SH);
  pg.HEAVY = 0xffffffffu;
  // fixed partitions (PrepGeom::FIX): half as much again as a uniform partition holds, at most what the LDS stage takes
  const unsigned long long per = ((unsigned long long)g.E + pg.P - 1) / std::max(1u, pg.P);
  pg.FIX = (u32)std::min<unsigned long long>(pg.CAP, (per + per / 2ull + 1024ull + 15ull) & ~15ull);
  return pg;
}
// 4-byte interchange entries the partition pass may write
size_t prep_bps_part_entries(const MsmGeom& g, u32 log2_l) {
  PrepGeom pg = prep_bps_geom(g, log2_l);
  return std::max<size_t>(g.E, pg.FIX ? (size_t)pg.P * pg.FIX : 0);
}
static size_t prep_bps_local_lds(const PrepGeom& pg) {
  return (size_t)(bps_local_fixed_words(1u << pg.SH) + pg.CAP) * sizeof(u32);
}
u32 prep_bps_partitions(const MsmGeom& g, u32 log2_l) { return g.B >> (10u - log2_l); }
u32 prep_bps_stride(const MsmGeom& g, u32 log2_l) {
  const u32 P = std::max(1u, prep_bps_partitions(g, log2_l));
  const unsigned long long per = ((unsigned long long)g.E + P - 1) / P;
  return (u32)((per + per / 2ull + 4096ull + 1023ull) & ~1023ull);
}
// lanes per bucket (log2): enough partitions that one fits the LDS stage and that the chip is filled (>= 128 partitions =
// 131 072 lanes), at most a wave per bucket.  -1: this geometry does not take the bucket-split pipeline.
int prep_bps_choose(const MsmGeom& g) {
  if (!g.precomp || g.n == 0 || g.S > 32u || g.B < 64u || (g.B & (g.B - 1u))) return -1;
  unsigned long long want = 128;
  while (want * 24576ull < g.E) want <<= 1;
  for (int l = 0; l <= 6; l++) {
    if ((10 - l) < 0 || (g.B >> (10 - l)) == 0) continue;
    const unsigned long long P = g.B >> (10 - l);
    if (P < want && l < 6) continue;
    if (P > PREP_MAX_P || (P & 1ull)) return -1;
    MsmGeom g2 = g;
    PrepGeom pg = prep_bps_geom(g2, (u32)l);
    if (pg.IB + pg.SH > 31u) return -1;
    if ((unsigned long long)g.E / P + 1024ull > pg.CAP) return -1;
    if (prep_scatter_lds(g, pg) > PREP_LDS_LIMIT || prep_bps_local_lds(pg) > PREP_LDS_LIMIT) return -1;
    if (P * (unsigned long long)prep_bps_stride(g, (u32)l) >= (1ull << 31)) return -1;
    return l;
  }
  return -1;
}
static std::atomic<unsigned long long> prep_bps_attr_devices{0};
static void prep_bps_attr() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 63;
  const unsigned long long bit = 1ull << dev;
  if (dev != 63 && (prep_bps_attr_devices.load(std::memory_order_acquire) & bit)) return;
  static_assert(BPS_LOCAL_BUDGET_WORDS * sizeof(u32) <= 152 * 1024, "k_prep_local_s's LDS must fit its attribute");
  (void)hipFuncSetAttribute((const void*)k_prep_local_s, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  prep_bps_attr_devices.fetch_or(bit, std::memory_order_release);
}

__global__ void k_mirror_flags(const u32* __restrict__ flags, u32* __restrict__ host_mirror) {
  if (threadIdx.x < 2u) host_mirror[threadIdx.x] = flags[threadIdx.x];
}
void launch_mirror_flags(hipStream_t st, const u32* flags, u32* host_mirror) {
  hipLaunchKernelGGL(k_mirror_flags, dim3(1), dim3(64), 0, st, flags, host_mirror);
}
void launch_vec_fill(hipStream_t st, u32* out, const u32 v[8], u32 n) {
  hipLaunchKernelGGL(k_vec_fill, dim3(cdiv_(n, 256)), dim3(256), 0, st, out, make_uint4(v[0], v[1], v[2], v[3]),
                     make_uint4(v[4], v[5], v[6], v[7]), n);
}

void launch_tv_probe(hipStream_t st, const u32* scalars, u32 n, u32* out16, const u32 one[8]) {
  static_assert(TV_PROBE_WORDS == TV_WORDS && TV_EXC_SLOTS == TV_EXC_MAX && TV_ONES_SAMPLE_COUNT == TV_ONES_SAMPLES,
                "launch.h mirrors vec_kernels.h");
  const u32 blocks = std::max(1u, std::min(512u, (n + 1023u) / 1024u));
  TvOne o;
  memcpy(o.w, one, 32);
  hipLaunchKernelGGL(k_tv_probe, dim3(blocks), dim3(256), 0, st, scalars, n, out16, o);
}

#define AMSM_FR_LAUNCHERS(FR)                                                                                        \
  template <>                                                                                                        \
  int launch_prep<FR>(hipStream_t st, const u32* scalars, int mont, MsmGeom g, const PrepBuffers& b) {               \
    PrepGeom pg = prep_geom(g);                                                                                      \
    prep_local_attr();                                                                                               \
    u32* part_total = b.d_small;                                                                                     \
    u32* part_start = b.d_small + (PREP_MAX_P + 1);                                                                  \
    u32* part_cursor = b.d_small + 2 * (PREP_MAX_P + 1);                                                             \
    u32* part_items = b.d_small + 3 * (PREP_MAX_P + 1);                                                              \
    PrepHeavy hv;                                                                                                    \
    hv.n = b.d_small + 4 * (PREP_MAX_P + 1);                                                                         \
    hv.cnt = hv.n + 1;                      /* zeroed with the words before it */                                    \
    hv.ids = hv.cnt + prep_heavy_words(g);                                                                           \
    hv.cur = hv.ids + PREP_MAX_HEAVY;                                                                                \
    hv.end = hv.cur + prep_heavy_words(g);                                                                           \
    /* one fill: the 16 flag words in front of d_small (b.err) and the arrays that must start at zero; a whole number */ \
    /* of 256-byte lines (an odd size made the runtime split the fill into two dispatches); the words it spills into */ \
    /* (hv.ids) are written before they are read */                                                                  \
    {                                                                                                                \
      size_t zero_bytes = (16 + 4 * (PREP_MAX_P + 1) + 1 + prep_heavy_words(g)) * sizeof(u32);                       \
      zero_bytes = (zero_bytes + 255) & ~(size_t)255;                                                                \
      if (b.err != b.d_small - 16) return -1;                                                                        \
      if (hipMemsetAsync(b.err, 0, zero_bytes, st) != hipSuccess) return -1;                                         \
    }                                                                                                                \
    u32 blocks = cdiv_(g.n, pg.SPB);                                                                                 \
    size_t lds_scatter = prep_scatter_lds(g, pg);                                                                    \
    if (lds_scatter > 64 * 1024) { /* rare (P above ~1300): opt in for this launch's instantiation */                \
      if (g.S <= 16u)                                                                                                \
        (void)hipFuncSetAttribute((const void*)k_prep_scatter<FR, 16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)PREP_LDS_LIMIT);                                                              \
      else                                                                                                           \
        (void)hipFuncSetAttribute((const void*)k_prep_scatter<FR, 32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)PREP_LDS_LIMIT);                                                              \
    }                                                                                                                \
    {  /* the histogram's blocking is independent of the scatter's: 1024 scalars, one per lane */                    \
      PrepGeom ph = pg;                                                                                              \
      ph.SPB = 1024;                                                                                                 \
      hipLaunchKernelGGL((k_prep_hist<FR>), dim3(cdiv_(g.n, 1024)), dim3(1024), pg.P * sizeof(u32), st, scalars,      \
                         mont, g, ph, part_total, b.err);                                                            \
    }                                                                                                                \
    hipLaunchKernelGGL(k_prep_scan, dim3(1), dim3(1024), 0, st, part_total, part_start, pg.P, pg.HEAVY, hv);         \
    if (g.S <= 16u)                                                                                                  \
      hipLaunchKernelGGL((k_prep_scatter<FR, 16, 1>), dim3(blocks), dim3(512), lds_scatter, st, scalars, mont, g, pg, \
                         part_start, part_cursor, b.part);                                                           \
    else                                                                                                             \
      hipLaunchKernelGGL((k_prep_scatter<FR, 32, 1>), dim3(blocks), dim3(256), lds_scatter, st, scalars, mont, g, pg, \
                         part_start, part_cursor, b.part);                                                           \
    const size_t lds_heavy = 2 * (size_t)(1u << pg.SH) * sizeof(u32);                                                \
    if (pg.HEAVY != 0xffffffffu)                                                                                     \
      hipLaunchKernelGGL(k_prep_heavy_count, dim3(PREP_HEAVY_GRID), dim3(256), lds_heavy, st, part_start, b.part, pg, hv); \
    hipLaunchKernelGGL(k_prep_local, dim3(pg.P), dim3(1024), prep_local_lds(pg), st,                                 \
                       part_start, b.part, g, pg, b.vals_sorted, b.start, b.items, b.item_off, part_items, hv);      \
    if (pg.HEAVY != 0xffffffffu)                                                                                     \
      hipLaunchKernelGGL(k_prep_heavy_place, dim3(PREP_HEAVY_GRID), dim3(256), lds_heavy, st, part_start, b.part, g, pg, hv, \
                         b.vals_sorted);                                                                             \
    hipLaunchKernelGGL(k_prep_offsets, dim3(cdiv_(g.B, 256)), dim3(256), (pg.P + 256) * sizeof(u32), st, part_start, \
                       part_items, pg, g, b.start, b.items, b.item_off, b.vals_sorted);                              \
    return 0;                                                                                                        \
  }                                                                                                                  \
  template <>                                                                                                        \
  int launch_prep_bps<FR>(hipStream_t st, const u32* scalars, int mont, MsmGeom g, u32 log2_l, const PrepBplBuffers& b) { \
    PrepGeom pg = prep_bps_geom(g, log2_l);                                                                          \
    u32* part_total = b.d_small;                                                                                     \
    u32* part_start = b.d_small + (PREP_MAX_P + 1);                                                                  \
    u32* part_cursor = b.d_small + 2 * (PREP_MAX_P + 1);                                                             \
    PrepHeavy hv;                                                                                                    \
    hv.n = b.d_small + 4 * (PREP_MAX_P + 1);                                                                         \
    hv.cnt = hv.n + 1;                                                                                               \
    hv.ids = hv.cnt + 64;                                                                                            \
    hv.cur = hv.end = hv.ids;                                                                                        \
    {                                                                                                                \
      size_t zero_bytes = (16 + 4 * (PREP_MAX_P + 1) + 1) * sizeof(u32);                                             \
      zero_bytes = (zero_bytes + 255) & ~(size_t)255;                                                                \
      if (b.err != b.d_small - 16) return -1;                                                                        \
      if (hipMemsetAsync(b.err, 0, zero_bytes, st) != hipSuccess) return -1;                                         \
    }                                                                                                                \
    prep_bps_attr();                                                                                                 \
    if (!pg.FIX) {                                                                                                   \
      PrepGeom ph = pg;                                                                                              \
      ph.SPB = 1024;                                                                                                 \
      hipLaunchKernelGGL((k_prep_hist<FR>), dim3(cdiv_(g.n, 1024)), dim3(1024), pg.P * sizeof(u32), st, scalars,      \
                         mont, g, ph, part_total, b.err);                                                            \
      hipLaunchKernelGGL(k_prep_scan, dim3(1), dim3(1024), 0, st, part_total, part_start, pg.P, pg.HEAVY, hv);       \
    }                                                                                                                \
    const size_t lds_scatter = prep_scatter_lds(g, pg);                                                              \
    if (g.S <= 16u) {                                                                                                \
      if (lds_scatter > 64 * 1024)                                                                                   \
        (void)hipFuncSetAttribute((const void*)k_prep_scatter<FR, 16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)PREP_LDS_LIMIT);                                                              \
      hipLaunchKernelGGL((k_prep_scatter<FR, 16, 1>), dim3(cdiv_(g.n, pg.SPB)), dim3(512), lds_scatter, st, scalars, \
                         mont, g, pg, part_start, part_cursor, b.part, b.err);                                       \
    } else {                                                                                                         \
      if (lds_scatter > 64 * 1024)                                                                                   \
        (void)hipFuncSetAttribute((const void*)k_prep_scatter<FR, 32, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  (int)PREP_LDS_LIMIT);                                                              \
      hipLaunchKernelGGL((k_prep_scatter<FR, 32, 1>), dim3(cdiv_(g.n, pg.SPB)), dim3(256), lds_scatter, st, scalars, \
                         mont, g, pg, part_start, part_cursor, b.part, b.err);                                       \
    }                                                                                                                \
    hipLaunchKernelGGL(k_prep_local_s, dim3(pg.P), dim3(1024), prep_bps_local_lds(pg), st,                           \
                       pg.FIX ? part_cursor : part_start, b.part, g, pg,                                             \
                       prep_bps_stride(g, log2_l), log2_l, b.ents_t, (BplGroup*)b.grp, b.order, b.err);              \
    return 0;                                                                                                        \
  }                                                                                                                  \
  template <>                                                                                                        \
  int launch_prep_bpl<FR>(hipStream_t st, const u32* scalars, int mont, MsmGeom g, const PrepBplBuffers& b) {        \
    PrepGeom pg = prep_bpl_geom(g);                                                                                  \
    u32* part_total = b.d_small;                                                                                     \
    u32* part_start = b.d_small + (PREP_MAX_P + 1);                                                                  \
    u32* part_cursor = b.d_small + 2 * (PREP_MAX_P + 1);                                                             \
    PrepHeavy hv;                                                                                                    \
    hv.n = b.d_small + 4 * (PREP_MAX_P + 1);                                                                         \
    hv.cnt = hv.n + 1;                                                                                               \
    hv.ids = hv.cnt + 64;                   /* never written: no partition exceeds HEAVY = 2^32 - 1 */               \
    hv.cur = hv.end = hv.ids;                                                                                        \
    {  /* one fill: the 16 flag words, the partition totals / starts / cursors and the heavy count */                \
      size_t zero_bytes = (16 + 4 * (PREP_MAX_P + 1) + 1) * sizeof(u32);                                             \
      zero_bytes = (zero_bytes + 255) & ~(size_t)255;                                                                \
      if (b.err != b.d_small - 16) return -1;                                                                        \
      if (hipMemsetAsync(b.err, 0, zero_bytes, st) != hipSuccess) return -1;                                         \
    }                                                                                                                \
    if (g.S <= 16u) prep_bpl_attr(k_prep_scatter<FR, 16, 1, true>);                                                  \
    else prep_bpl_attr(k_prep_scatter<FR, 32, 1, true>);                                                             \
    if (!pg.FIX) {                                                                                                   \
      PrepGeom ph = pg;                                                                                              \
      ph.SPB = 1024;                                                                                                 \
      hipLaunchKernelGGL((k_prep_hist<FR>), dim3(cdiv_(g.n, 1024)), dim3(1024), pg.P * sizeof(u32), st, scalars,      \
                         mont, g, ph, part_total, b.err);                                                            \
      hipLaunchKernelGGL(k_prep_scan, dim3(1), dim3(1024), 0, st, part_total, part_start, pg.P, pg.HEAVY, hv);       \
    }                                                                                                                \
    if (g.S <= 16u)                                                                                                  \
      hipLaunchKernelGGL((k_prep_scatter<FR, 16, 1, true>), dim3(cdiv_(g.n, pg.SPB)), dim3(512),                     \
                         prep_bpl_scatter_lds(g, pg), st, scalars, mont, g, pg, part_start, part_cursor, b.part, b.err); \
    else /* plain keys with 18 / 19 windows: 256 scalars per workgroup */                                            \
      hipLaunchKernelGGL((k_prep_scatter<FR, 32, 1, true>), dim3(cdiv_(g.n, pg.SPB)), dim3(256),                     \
                         prep_bpl_scatter_lds(g, pg), st, scalars, mont, g, pg, part_start, part_cursor, b.part, b.err); \
    hipLaunchKernelGGL(k_prep_local_t, dim3(pg.P), dim3(1024), prep_bpl_local_lds(pg), st,                           \
                       pg.FIX ? part_cursor : part_start, (const u64*)b.part, g, pg, prep_bpl_stride(g), b.ents_t,   \
                       (BplGroup*)b.grp, b.order, b.err);                                                            \
    return 0;                                                                                                        \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_digits<FR>(hipStream_t st, const u32* scalars, int mont, MsmGeom g, void* keys, bool keys16, u32* vals, \
                         u32* err) {                                                                                 \
    if (keys16)                                                                                                      \
      hipLaunchKernelGGL((k_digits<FR, uint16_t>), dim3(cdiv_(g.n, 256)), dim3(256), 0, st, scalars, mont, g,         \
                         (uint16_t*)keys, vals, err);                                                                \
    else                                                                                                             \
      hipLaunchKernelGGL((k_digits<FR, u32>), dim3(cdiv_(g.n, 256)), dim3(256), 0, st, scalars, mont, g, (u32*)keys,  \
                         vals, err);                                                                                 \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_skew_probe<FR>(hipStream_t st, const u32* scalars, int mont, u32 n, DigitWalk walk, u32* d_flag,       \
                             const u32* d_tv_words) {                                                               \
    hipLaunchKernelGGL((k_skew_probe<FR>), dim3(1), dim3(1024), 0, st, scalars, mont, n, walk, d_flag, d_tv_words);  \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_vec_random<FR>(hipStream_t st, u32* out, u64 seed, u32 n, int mont) {                                  \
    hipLaunchKernelGGL((k_vec_random<FR>), dim3(cdiv_(n, 256)), dim3(256), 0, st, out, seed, n, mont);               \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_vec_hadamard<FR>(hipStream_t st, const u32* a, const u32* b, u32* out, u32 n) {                        \
    hipLaunchKernelGGL((k_vec_hadamard<FR>), dim3(stream_grid(n)), dim3(256), 0, st, a, b, out, n);                  \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_vec_combine<FR>(hipStream_t st, const CombineArgs& a_in, u32* out) {                                   \
    dim3 grid(stream_grid(a_in.n)), block(256);                                                                      \
    const CombineArgs& a = a_in;                                                                                     \
    switch (a.n_vecs) {                                                                                              \
      case 0: /* only the hiding addend */                                                                          \
      case 1: hipLaunchKernelGGL((k_vec_combine<FR, 1>), grid, block, 0, st, a, out); break;                         \
      case 2: hipLaunchKernelGGL((k_vec_combine<FR, 2>), grid, block, 0, st, a, out); break;                         \
      case 3: hipLaunchKernelGGL((k_vec_combine<FR, 3>), grid, block, 0, st, a, out); break;                         \
      case 4: hipLaunchKernelGGL((k_vec_combine<FR, 4>), grid, block, 0, st, a, out); break;                         \
      case 5: hipLaunchKernelGGL((k_vec_combine<FR, 5>), grid, block, 0, st, a, out); break;                         \
      case 6: hipLaunchKernelGGL((k_vec_combine<FR, 6>), grid, block, 0, st, a, out); break;                         \
      case 7: hipLaunchKernelGGL((k_vec_combine<FR, 7>), grid, block, 0, st, a, out); break;                         \
      default: hipLaunchKernelGGL((k_vec_combine<FR, 8>), grid, block, 0, st, a, out); break;                        \
    }                                                                                                                \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_vec_powers<FR>(hipStream_t st, const u32 point[8], u32 n, u32* out) {                                  \
    CombineArgs a;                                                                                                   \
    memset(&a, 0, sizeof(a));                                                                                        \
    memcpy(a.coeff[0], point, 32);                                                                                   \
    a.n = n;                                                                                                         \
    hipLaunchKernelGGL((k_vec_powers<FR>), dim3(cdiv_(n, 256)), dim3(256), 0, st, a, out);                           \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_vec_inner_product<FR>(hipStream_t st, const u32* a, const u32* b, u32 n, u32 blocks, u32* out) {       \
    hipLaunchKernelGGL((k_vec_inner_product<FR>), dim3(blocks), dim3(256), 0, st, a, b, n, out);                     \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_vec_inner_product_pair<FR>(hipStream_t st, const u32* a0, const u32* b0, const u32* a1, const u32* b1, \
                                         u32 n, u32 blocks, u32* out) {                                              \
    hipLaunchKernelGGL((k_vec_inner_product_pair<FR>), dim3(blocks, 2), dim3(256), 0, st, a0, b0, a1, b1, n, out);   \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_ipa_fold_ip<FR>(hipStream_t st, u32* c, u32* z, u32 half, const u32 x[8], const u32 xinv[8], u32 blocks, \
                              u32* out) {                                                                            \
    IpaFoldArgs a;                                                                                                   \
    memcpy(a.x, x, 32);                                                                                              \
    memcpy(a.xinv, xinv, 32);                                                                                        \
    hipLaunchKernelGGL((k_ipa_fold_ip<FR>), dim3(blocks), dim3(256), 0, st, c, z, half, a, out);                     \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_check_poly_coeffs<FR>(hipStream_t st, const u32* xi, u32 k, u32* out) {                                \
    CheckPolyArgs a;                                                                                                 \
    memset(&a, 0, sizeof(a));                                                                                        \
    memcpy(a.xi, xi, (size_t)k * 32);                                                                                \
    a.k = k;                                                                                                         \
    hipLaunchKernelGGL((k_check_poly_coeffs<FR>), dim3(cdiv_(1u << k, 256)), dim3(256), 0, st, a, out);              \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_ipa_round_scalars<FR>(hipStream_t st, const u32* xi, u32 j, u32 log_n, const u32* c, u32* out_l,       \
                                    u32* out_r) {                                                                    \
    CheckPolyArgs a;                                                                                                 \
    memset(&a, 0, sizeof(a));                                                                                        \
    memcpy(a.xi, xi, (size_t)j * 32);                                                                                \
    a.k = j;                                                                                                         \
    hipLaunchKernelGGL((k_ipa_round_scalars<FR>), dim3(cdiv_(1u << log_n, 256)), dim3(256), 0, st, a, j, log_n, c,    \
                       out_l, out_r);                                                                                \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_spmv<FR>(hipStream_t st, const u32* row_ptr, const u32* col, const u32* val, const u32* input,         \
                       u32 n_input, const u32* witness, u32 n_witness, u32* out, u32 n_rows) {                       \
    hipLaunchKernelGGL((k_spmv<FR>), dim3(cdiv_(n_rows, 256)), dim3(256), 0, st, row_ptr, col, val, input, n_input,   \
                       witness, n_witness, out, n_rows);                                                             \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_hp_t_vecs<FR>(hipStream_t st, const TVecArgs& a_in, int n_inputs) {                                    \
    dim3 grid(stream_grid(a_in.len)), block(256);                                                                    \
    const TVecArgs& a = a_in;                                                                                        \
    switch (n_inputs) {                                                                                              \
      case 1: hipLaunchKernelGGL((k_hp_t_vecs<FR, 1>), grid, block, 0, st, a); break;                                \
      case 2: hipLaunchKernelGGL((k_hp_t_vecs<FR, 2>), grid, block, 0, st, a); break;                                \
      case 3: hipLaunchKernelGGL((k_hp_t_vecs<FR, 3>), grid, block, 0, st, a); break;                                \
      case 4: hipLaunchKernelGGL((k_hp_t_vecs<FR, 4>), grid, block, 0, st, a); break;                                \
      case 5: hipLaunchKernelGGL((k_hp_t_vecs<FR, 5>), grid, block, 0, st, a); break;                                \
      case 6: hipLaunchKernelGGL((k_hp_t_vecs<FR, 6>), grid, block, 0, st, a); break;                                \
      case 7: hipLaunchKernelGGL((k_hp_t_vecs<FR, 7>), grid, block, 0, st, a); break;                                \
      default: hipLaunchKernelGGL((k_hp_t_vecs<FR, 8>), grid, block, 0, st, a); break;                               \
    }                                                                                                                \
  }

AMSM_FR_LAUNCHERS(PallasFr)
AMSM_FR_LAUNCHERS(Bls12381Fr)

}  // namespace amsm
