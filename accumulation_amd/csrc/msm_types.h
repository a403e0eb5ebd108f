// Plain-data types shared by the host launcher (api.hip) and the kernel translation units.
#pragma once
#include <stdint.h>

namespace amsm {

typedef uint32_t u32;
typedef uint64_t u64;

constexpr int VEC_MAX = 8;        // max vectors combined in one launch (reference uses 2..4)
constexpr int HP_MAX_INPUTS = 8;  // max inputs+accumulators of one hp_as t-vector launch (reference tests reach 6)

struct MsmGeom {
  u32 n;             // pairs in this call
  u32 c;             // window bits (2..24)
  u32 W;             // windows = floor(255 / c) + 1
  u32 S;             // entry slots per scalar (= W)
  u32 nb;            // buckets per set = 2^(c-1)
  u32 n_sets;        // bucket sets = groups * (1 when precomputed, else W)
  u32 groups;        // 1, or 2: scalar i adds into the sum of group (i >> group_shift) & 1 (two MSMs in one pass)
  u32 group_shift;
  u32 B;             // n_sets * nb  (key B = "digit 0", dropped)
  u32 E;             // n * S entries (upper bound; zero digits emit none)
  u32 base_off;      // first generator used
  u32 table_stride;  // generators per table level (key length)
  u32 precomp;       // table has W levels
  u32 idx_rel_bits;  // 0: the prep's interchange word carries the absolute table index base_off + i + w * table_stride.
                     // b > 0 (an MSM over a b-bit window of a longer precomputed key): it carries (w << b) | i -- the bits a
                     // partition pass has for bucket ids do not shrink with the KEY's length -- and k_prep_local /
                     // k_prep_heavy_place expand it to the absolute index when they write the list accumulate L0 reads
  u32 K0;            // entries per accumulate-L0 work item (chunk) of phase A: chunks [0, nA)
  u32 K0b;           // ... of phase B: chunks [nA, n_chunks).  Two sizes (K0 > K0b, both multiples of 4) let the grid be a
                     // WHOLE number of rounds of the resident wave slots: with one size the last round of a 2^20-pair
                     // launch was 56 % full (3.56 rounds of 24 entries: 11 % of the kernel's time on half-empty SIMDs)
  u32 nA;            // chunks of phase A (a multiple of 256: a workgroup never straddles the phases)
  u32 T0;            // entries covered by phase A = nA * K0
  u32 n_chunks;      // work items of accumulate L0 (upper bound: sized for E entries)
  u32 K1;            // max partials folded by one L1 lane
  u32 top_split;     // plain keys on the bucket-per-lane pipeline (round 4): the TOP window owns two bucket sets, the scalars with an
                     // even index feed the first, the odd ones the second (the host adds the two sums) -- its digits reach only
                     // r / 2^256 of their range and carry no sign to fold, so its partitions run 10 % fuller than the others' on
                     // uniform BLS12-381 scalars and twice as full on 254-bit ones (the synthetic stream of rng.h); sets per group
                     // = W + top_split
  u32 part_max;      // host only (bucket-per-lane prep): entries the fullest partition is expected to hold on uniform scalars -- the
                     // top window's digits cover only r / 2^256 of its range (no sign to fold), 10 % denser than the others'
                     // on BLS12-381; a plain key gives that window a bucket set of its own
  u32 top_shift;     // s > 0 (bucket-per-lane keys): the TOP window holds only 255 - c (W - 1) scalar bits, so its digits would all
                     // fall into the lowest 2^(that - 1) buckets -- a sixteenth of the partitions would receive a whole window.
                     // Its table level is 2^(c (W - 1) - s) G instead and its digit is shifted left by s: the same product, spread
                     // over every 2^s-th bucket of the whole range (raw digit <= 2^(c - 1 - s): never negative, no carry out)
  u32 n_narrow;      // round 4: the TOP n_narrow windows are c - 1 bits wide, so that the W widths add up to EXACTLY 256 bits (W c - 256
                     // narrow ones): every window is full -- no short top window -- and the recoding never carries out of the top.
                     // A narrow window's signed digit is DOUBLED (it lands on the even buckets of the full range) and what it
                     // multiplies stands one doubling short: table level w = 2^(e_w) G (precomputed key) / the set's sum counts
                     // 2^(e_w) (plain key), e_w = window_exponent().  0: the legacy walk (equal widths, top_shift)
  u32 bpl;           // 1: bucket-per-lane pipeline (k_prep_local_t + k_accum_bpl; 20-bit windows), 2: bucket-split pipeline
                     // (k_prep_local_s + k_accum_bps; small and medium MSMs), 0: chunked pipeline
  u32 bps_log2_l;    // bucket-split: log2 of the lanes per bucket
  u32 skip_ones;     // round 5: scalars equal to ONE emit no entries (their generators are summed apart, k_tv_sum: ark-ec's
                     // multi_scalar_mul adds the bases of unit scalars directly too) -- a witness with a share of boolean wires
                     // would otherwise put all of them into bucket 1 of the lowest window
  u32 red_s;         // buckets per reduce lane
  u32 red_threads;   // reduce lanes per set
};

// chunk <-> entry position of the two-phase chunking (host and device)
#if defined(__HIPCC__)
#define AMSM_GEOM_FN __host__ __device__ inline
#else
#define AMSM_GEOM_FN inline
#endif
// width of window w, and the power of two its digit multiplies (MsmGeom::n_narrow, MsmGeom::top_shift)
AMSM_GEOM_FN u32 window_bits_of(u32 c, u32 W, u32 n_narrow, u32 w) { return c - ((w + n_narrow >= W) ? 1u : 0u); }
// first bit of window w
AMSM_GEOM_FN u32 window_position_of(u32 c, u32 W, u32 n_narrow, u32 w) {
  const u32 R = W - n_narrow;  // regular windows
  return w < R ? w * c : R * c + (w - R) * (c - 1u);
}
AMSM_GEOM_FN u32 window_exponent_of(u32 c, u32 W, u32 n_narrow, u32 top_shift, u32 w) {
  // the window's position, minus the doubling a narrow window's digit carries, minus the top window's spread
  return window_position_of(c, W, n_narrow, w) - ((w + n_narrow >= W) ? 1u : 0u) - ((w == W - 1u) ? top_shift : 0u);
}
AMSM_GEOM_FN u32 window_exponent(const MsmGeom& g, u32 w) { return window_exponent_of(g.c, g.W, g.n_narrow, g.top_shift, w); }
// the parameters of the signed-digit walk (vec_kernels.h: digit_step), as every kernel that walks scalars receives them
struct DigitWalk {
  u32 c, W, n_narrow, top_shift;
  u32 skip_ones;  // (k_skew_probe: unit scalars are not looked at -- numerous ones are summed apart, few ones skew nothing)
};
AMSM_GEOM_FN DigitWalk digit_walk_of(const MsmGeom& g) {
  DigitWalk d;
  d.c = g.c;
  d.W = g.W;
  d.n_narrow = g.n_narrow;
  d.top_shift = g.top_shift;
  d.skip_ones = g.skip_ones;
  return d;
}
// narrow windows a width-c walk over 256 bits needs (0 when the widths divide evenly; ~0u when c - 1 is not narrow enough)
AMSM_GEOM_FN u32 narrow_windows_for(u32 c) {
  const u32 W = 255u / c + 1u, nn = W * c - 256u;
  return nn < W ? nn : ~0u;
}
// bucket sets per group / the set window w of scalar i feeds, relative to its group's first
AMSM_GEOM_FN u32 sets_per_group(const MsmGeom& g) { return g.precomp ? 1u : g.W + g.top_split; }
AMSM_GEOM_FN u32 window_set(const MsmGeom& g, u32 w, u32 i) {
  // (the bit that splits the top window must not be the bit that picks the group: a group would feed ONE of its two top sets)
  const u32 sb = (g.groups > 1u && g.group_shift == 0u) ? 1u : 0u;
  return g.precomp ? 0u : w + ((g.top_split && w == g.W - 1u) ? ((i >> sb) & 1u) : 0u);
}
// interchange index -> table index (see idx_rel_bits)
AMSM_GEOM_FN u32 entry_abs_index(const MsmGeom& g, u32 v) {
  return g.idx_rel_bits ? (g.base_off + (v & ((1u << g.idx_rel_bits) - 1u)) + (v >> g.idx_rel_bits) * g.table_stride) : v;
}
AMSM_GEOM_FN u32 chunk_of(const MsmGeom& g, u32 pos) { return pos < g.T0 ? pos / g.K0 : g.nA + (pos - g.T0) / g.K0b; }
AMSM_GEOM_FN u32 chunk_start(const MsmGeom& g, u32 c) { return c < g.nA ? c * g.K0 : g.T0 + (c - g.nA) * g.K0b; }

struct CombineArgs {
  const u32* vec[VEC_MAX];
  u32 len[VEC_MAX];
  u32 coeff[VEC_MAX][8];
  const u32* hiding;
  u32 hiding_len;
  u32 n_vecs;
  u32 n;
};

struct TVecArgs {
  const u32* a[HP_MAX_INPUTS];
  const u32* b[HP_MAX_INPUTS];
  u32 a_len[HP_MAX_INPUTS];
  u32 b_len[HP_MAX_INPUTS];
  u32 mu[HP_MAX_INPUTS + 1][8];
  const u32* hiding_a;
  const u32* hiding_b;
  u32 hiding_a_len, hiding_b_len;
  u32* t[2 * HP_MAX_INPUTS - 1];
  u32 len;
};

// direct sum over a small key's table of digit multiples (msm_kernels.h k_direct_sum): 64 four-bit windows, multiples 1 .. 8
constexpr u32 DS_W = 64, DS_MULT = 8;
// up to DS_BATCH MSMs over the same key in one launch (blockIdx.y = which): the provers' batched commits
constexpr int DS_BATCH = 16;
struct DsBatch {
  const u32* scalars[DS_BATCH];
  u32 n[DS_BATCH];
  u32 base_off[DS_BATCH];
};

}  // namespace amsm
