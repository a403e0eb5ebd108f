// Which pipeline an MSM takes, and which table a key is built with: ONE place, two tables, no HIP -- plain C++ that the host
// launcher (api_pipeline.inc, api_keys.inc) includes and that tests/cpp_host/select_check.cpp compiles on its own to check every
// threshold edge without a GPU (tests/test_pipeline_select_cpu.py).  Every size threshold of the library lives here; the numbers
// are measured on MI355X (same-process A/B, DESIGN.md section 4.2 and profiles/), not derived.
//
// The four accumulation forms:
//   DIRECT_SUM       keys of up to 2^15 generators that carry every multiple a 4-bit signed digit can ask for: one launch that sums
//                    table points, no buckets (msm_kernels.h k_direct_sum)
//   BUCKET_SPLIT     2^16 .. 2^17 pairs over a precomputed key: every bucket on 2 .. 64 adjacent lanes (k_prep_local_s + k_accum_bps)
//   BUCKET_PER_LANE  (2^18, 2^20] pairs: a lane sums one whole bucket; 20-bit windows over a precomputed key of >= 2^20 generators,
//                    one bucket set per 16- / 15-bit window over a plain key (k_prep_local_t + k_accum_bpl)
//   CHUNKED          everything else, and every vector whose digits are skewed: equal-sized work items whatever the digit
//                    distribution (k_accum_l0 + k_accum_l1)
// Longer MSMs are cut into ranges first (`range`), each range is chosen again; the ranges of one vector over a bucket-per-lane key
// share one bucket set (api_types.h: struct Share).
#pragma once
#include <stddef.h>

namespace amsm {
namespace msel {

enum Pipeline { DIRECT_SUM = 0, BUCKET_SPLIT = 1, BUCKET_PER_LANE = 2, CHUNKED = 3 };
inline const char* pipeline_name(int p) {
  return p == DIRECT_SUM ? "direct_sum" : (p == BUCKET_SPLIT ? "bucket_split" : (p == BUCKET_PER_LANE ? "bucket_per_lane" : "chunked"));
}

inline int ilog2_ceil(size_t n) {
  int l = 0;
  while (((size_t)1 << l) < n) l++;
  return l;
}
constexpr size_t P2(int k) { return (size_t)1 << k; }

// ---- table 1: the window width a key's table is built for --------------------------------------------------------------------------
// Measured, not derived (tools/sweep_window.py): a width whose TOP window holds only 2-3 scalar bits (255 mod c small: 9, 11, 12, 14,
// 18, 19) puts 1/8 of that window's entries into a handful of buckets.
constexpr int BPL_WINDOW = 20;      // keys of >= 2^20 generators: 13 windows whose widths add up to 256 bits (9 x 20 + 4 x 19)
constexpr int DIRECT_MAX_LOG2 = 15; // largest key (log2 generators) that also carries the direct-sum table (AMSM_DIRECT_SUM_MAX_LOG2)
struct KeyWindowRow {
  int lg_lo, lg_hi;  // ceil(log2 generators) in [lg_lo, lg_hi]
  int c;
};
// precomputed keys (all windows share one bucket set)
constexpr KeyWindowRow kPrecomputedWindow[] = {
    {0, 12, 8}, {13, 14, 10}, {15, 15, 13}, {16, 16, 16}, {17, 17, 17}, {18, 19, 16}, {20, 64, 17 /* AMSM_BPL=0; else BPL_WINDOW */}};
// plain keys on the chunked pipeline (one bucket set per window; longer ones take the bucket-per-lane rows of table 2)
constexpr KeyWindowRow kPlainChunkedWindow[] = {{0, 8, 4}, {9, 19, 8}, {20, 20, 13}, {21, 21, 15}, {22, 64, 16}};
inline int key_window(size_t n_generators, bool precomputed, bool bpl_enabled) {
  const int lg = ilog2_ceil(n_generators < 2 ? 2 : n_generators);
  if (precomputed && bpl_enabled && lg >= 20) return BPL_WINDOW;
  if (precomputed) {
    for (const KeyWindowRow& r : kPrecomputedWindow)
      if (lg >= r.lg_lo && lg <= r.lg_hi) return r.c;
    return 17;
  }
  for (const KeyWindowRow& r : kPlainChunkedWindow)
    if (lg >= r.lg_lo && lg <= r.lg_hi) return r.c;
  return 16;
}

// ---- table 2: the pipeline of an MSM -----------------------------------------------------------------------------------------------
enum KeyKind {
  KEY_DIRECT = 0,  // precomputed, <= 2^15 generators, carries the direct-sum table
  KEY_BPL = 1,     // precomputed for 20-bit windows (>= 2^20 generators): amsm_bases::bpl
  KEY_TABLE = 2,   // any other precomputed key
  KEY_PLAIN = 3    // one copy of the generators
};
struct KeyDesc {
  size_t n;           // generators
  bool precomputed;
  bool bpl;           // 20-bit bucket-per-lane table
  bool direct_table;  // 512 points per generator present
  int c;              // the table's window width (0: plain)
};
inline KeyKind key_kind(const KeyDesc& k) {
  if (!k.precomputed) return KEY_PLAIN;
  if (k.direct_table) return KEY_DIRECT;
  return k.bpl ? KEY_BPL : KEY_TABLE;
}
struct Switches {              // the context's documented switches (include/amsm.h lists the environment variables)
  bool bpl = true;             // AMSM_BPL=0: no MSM takes the bucket-per-lane pipeline (and keys are built without 20-bit tables)
  bool bpl_plain = true;       // AMSM_BPL_PLAIN=0: plain keys stay on the chunked pipeline
  int bps = 2;                 // AMSM_BPS: 0 never, 1 grouped MSMs only, 2 every candidate
  int split_log2 = 21;         // AMSM_SPLIT_LOG2: range of an MSM over a key WITHOUT a 20-bit table (0: never cut)
  bool direct = true;          // AMSM_DIRECT_SUM_MAX_LOG2=0: no direct sums
  bool window_override = false;  // amsm_ctx_set_window: everything chunked with that width
};
struct Row {
  KeyKind kind;
  size_t lo, hi;       // pairs in (lo, hi]
  Pipeline pipeline;
  bool over_twin;      // KEY_BPL only: the MSM runs over the key's 17-bit twin (built on first use)
  int plain_window;    // KEY_PLAIN on the bucket-per-lane pipeline: its window width
  const char* why;
};
constexpr size_t BPS_MIN_PAIRS = P2(16), BPS_MAX_PAIRS = P2(17), RANGE_PAIRS = P2(20), SPLIT_MIN_LOG2 = 22;
// First matching row wins.  (2^20 pairs = one window of a 20-bit key; a range below a QUARTER of it fills the 2^19 buckets too thinly.)
constexpr Row kPipeline[] = {
    {KEY_DIRECT, 0, P2(DIRECT_MAX_LOG2), DIRECT_SUM, false, 0, "latency regime: one launch + the fold, 0.10-0.28 ms against 0.24-0.36"},
    {KEY_BPL, P2(18), RANGE_PAIRS, BUCKET_PER_LANE, false, 0, "13 gathered additions per pair, no partial records"},
    {KEY_BPL, BPS_MIN_PAIRS - 1, BPS_MAX_PAIRS, BUCKET_SPLIT, true, 0, "a short range of a long key: the twin's 17-bit windows"},
    {KEY_BPL, 0, P2(18), CHUNKED, true, 0, "a short range of a long key: the twin's 17-bit windows"},
    {KEY_TABLE, BPS_MIN_PAIRS - 1, BPS_MAX_PAIRS, BUCKET_SPLIT, false, 0, "three dispatches fewer in a latency-bound chain"},
    {KEY_TABLE, 0, ~(size_t)0, CHUNKED, false, 0, "round 2's path: 2^18 / 2^19 generators, BLS12-381 below 2^20, tiny keys"},
    {KEY_DIRECT, 0, ~(size_t)0, CHUNKED, false, 0, "(a direct-sum key asked for more pairs than it has: never reached, n <= key)"},
    {KEY_PLAIN, P2(18), RANGE_PAIRS, BUCKET_PER_LANE, false, 16, "16 x 16-bit windows, one bucket set each: the VariableBaseMSM shape"},
    {KEY_PLAIN, P2(17), P2(18), BUCKET_PER_LANE, false, 15, "18 windows whose widths add up to 256 bits (14 of 14 bits)"},
    {KEY_PLAIN, 0, P2(17), CHUNKED, false, 0, "8-bit windows (4 up to 2^8 pairs) win below 2^17 pairs"},
};
struct Choice {
  Pipeline pipeline;
  bool over_twin;
  int plain_window;
  size_t range;      // > 0: cut the MSM into ranges of this many pairs first (each range is chosen again)
  const Row* row;    // the table row (null: decided by a switch or by skew)
};
// How an MSM longer than the pipelines take is cut: windows of 2^20 pairs over a 20-bit key or a plain key, of 2^split_log2
// (from 2^22 pairs up) over any other precomputed key.
inline size_t range_of(const KeyDesc& k, size_t n, const Switches& sw) {
  if (!k.precomputed) return (sw.bpl && sw.bpl_plain && !sw.window_override && n > RANGE_PAIRS) ? RANGE_PAIRS : 0;
  if (k.bpl && sw.bpl) return n > RANGE_PAIRS ? RANGE_PAIRS : 0;
  return (sw.split_log2 > 0 && (n >> SPLIT_MIN_LOG2) != 0) ? P2(sw.split_log2) : 0;
}
// grouped: two sums by one bit of the index (the IPA rounds); grouped_regular: both classes hold n / 2 indices in the regular pattern
// (n a multiple of 2 << group_shift) -- what the direct sum needs; skewed: the digit probe's verdict (or a prep's overflow flag).
inline Choice choose(const KeyDesc& k, size_t n, bool grouped, bool grouped_regular, bool skewed, const Switches& sw) {
  Choice ch{CHUNKED, false, 0, 0, nullptr};
  const KeyKind kind = key_kind(k);
  ch.over_twin = kind == KEY_BPL;  // whatever does not take the 20-bit table runs over the twin
  if (n == 0) return ch;
  // the direct sum has no buckets: nothing to skew, nothing to cut
  if (kind == KEY_DIRECT && sw.direct && !sw.window_override && n <= k.n && (!grouped || grouped_regular)) {
    ch.pipeline = DIRECT_SUM;
    ch.over_twin = false;
    ch.row = &kPipeline[0];
    return ch;
  }
  ch.range = range_of(k, n, sw);
  if (ch.range) {
    ch.over_twin = false;
    return ch;  // (the pipeline is chosen per range)
  }
  if (skewed || sw.window_override) return ch;
  for (const Row& r : kPipeline) {
    if (r.kind != kind || n <= r.lo || n > r.hi || r.pipeline == DIRECT_SUM) continue;
    if (r.pipeline == BUCKET_PER_LANE && (!sw.bpl || (kind == KEY_PLAIN && !sw.bpl_plain))) continue;
    if (r.pipeline == BUCKET_SPLIT && (sw.bps == 0 || (sw.bps == 1 && !grouped))) continue;
    ch.pipeline = r.pipeline;
    ch.over_twin = r.over_twin;
    ch.plain_window = r.plain_window;
    ch.row = &r;
    return ch;
  }
  return ch;
}
// Is a digit probe worth its launch for this vector?  Only where a skewed vector would otherwise pay an aborted prep and a re-run:
// the sorted pipelines over precomputed keys (plain keys find out from the prep's overflow flag).
inline bool wants_skew_probe(const KeyDesc& k, size_t n, const Switches& sw) {
  if (!k.precomputed) return false;
  const size_t r = range_of(k, n, sw);
  const Choice ch = choose(k, r ? r : n, false, false, false, sw);
  return ch.pipeline == BUCKET_PER_LANE || ch.pipeline == BUCKET_SPLIT;
}

}  // namespace msel
}  // namespace amsm
