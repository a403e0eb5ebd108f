// Host-callable launchers for the device kernels.  The heavy elliptic-curve kernels are compiled in one
// translation unit per curve (kern_pallas.hip / kern_bls12_381.hip), the scalar-field kernels in
// kern_fr.hip, so the library builds in parallel; api.hip only sees these declarations.
#pragma once
#include <hip/hip_runtime.h>

#include "msm_types.h"

namespace amsm {

// ---- per-curve (Fq) launchers ------------------------------------------------------------------
template <class Fq>
void launch_accum_l0(hipStream_t st, const u32* table, const u32* vals_sorted, const u32* start,
                     const u32* item_off, MsmGeom g, u32* partials);
// bucket-per-lane accumulation (msm_kernels.h: k_accum_bpl) over the transposed layout k_prep_local_t wrote
template <class Fq>
void launch_accum_bpl(hipStream_t st, const u32* table, const u32* ents_t, const void* grp, const u32* order, u32 n_groups,
                      u32 groups_per_part, const u32* flags, u32* buckets, u32 wg_per_cu = 0, bool accumulate = false);
// resident 256-lane workgroups of accumulate L0 per CU (occupancy query)
template <class Fq>
int accum_l0_blocks_per_cu();
template <class Fq>
void launch_accum_l1(hipStream_t st, u32 lanes_per_bucket, const u32* partials, const u32* items, const u32* item_off, MsmGeom g,
                     u32* buckets, u32* heavy_count, u32* heavy_list);
template <class Fq>
void launch_accum_l2(hipStream_t st, const u32* partials, const u32* items, const u32* item_off,
                     const u32* heavy_count, const u32* heavy_list, u32* scratch, u32* buckets);
template <class Fq>
u32 accum_l2_slices();  // scratch records per heavy bucket
template <class Fq>
void launch_bucket_reduce(hipStream_t st, u32 red_blocks, const u32* buckets, MsmGeom g, u32* out);
// round 4: rows / columns form (msm_kernels.h: k_red2_sums, k_red2_weighted): nb a multiple of 1024; rc = n_sets * (nb / 1024 +
// 1024) scratch records; returns the partial records per set left in `out` (at most 4 * (nb / 1024 + 1024) / 256 + 1)
template <class Fq>
u32 launch_bucket_reduce2(hipStream_t st, const u32* buckets, MsmGeom g, bool quad, bool latency, u32* rc, u32* out);
template <class Fq>
// flags (may be null): two words copied behind the n_sets records (out needs 8 bytes more)
void launch_fold(hipStream_t st, u32 n_sets, const u32* in, u32 n_per_set, u32* out, const u32* flags, u32* host_mirror = nullptr);
// the same two kernels with a quad of lanes per logical lane (ec.h: xyzz_add_quad): red_blocks = 4x
template <class Fq>
void launch_bucket_reduce_quad(hipStream_t st, u32 red_blocks, const u32* buckets, MsmGeom g, u32* out);
// round 6: both in one launch (msm_kernels.h: k_bucket_reduce_fold_quad); partial: n_sets * red_blocks records, ticket: n_sets zeroed words
template <class Fq>
void launch_bucket_reduce_fold_quad(hipStream_t st, u32 red_blocks, const u32* buckets, MsmGeom g, u32* partial, u32* ticket, u32* out,
                                    const u32* flags, u32* host_mirror);
// round 6: the jump fold of an IPA opening (msm_kernels.h: k_ipa_jump_accum): n_lists lists, m0 (a multiple of 64) outputs
template <class Fq>
void launch_ipa_jump_accum(hipStream_t st, const u32* table, const u32* entries, const u32* list_off, const u32* list_slot, u32 n_lists,
                           u32 m0, u32 nb, u32* buckets);
template <class Fq>
void launch_fold_quad(hipStream_t st, u32 n_sets, const u32* in, u32 n_per_set, u32* out, const u32* flags, u32* host_mirror = nullptr,
                      bool clear_flags = false);  // clear_flags: the two words are zeroed once they have been copied out
template <class Fq>
// level = 2^c * mul_m * (level - 1); mul_m = 0 / 1: no small multiple (power-of-two windows)
void launch_precompute_level(hipStream_t st, u32* table, u32 stride, u32 level, u32 c, u32* xyzz_scratch);
// levels 1 .. W - 1 of a small key (plain c-bit windows) from level 0 in two launches; xyzz_scratch: (W - 1) n records
template <class Fq>
void launch_precompute_all_levels(hipStream_t st, u32* table, u32 n, u32 c, u32 W, u32* xyzz_scratch);
// Direct sum (msm_kernels.h k_direct_sum): the 512-points-per-generator table of a small key from its window table (W levels of
// plain c-bit windows, device radix), built in slabs of `slab_windows` of its 64 four-bit windows (xyzz_scratch: 8 * slab_windows * n
// records), and one MSM as cdiv(n * 64 / m, 256) partial records (returned) for launch_fold_quad
template <class Fq>
void launch_ds_table(hipStream_t st, const u32* win_table, u32 n, u32 c, u32 W, u32* xyzz_scratch, u32 slab_windows, u32* table);
// nv <= DS_BATCH MSMs over the key in one launch: MSM v's `blocks` records (the return value, sized by the longest vector) at
// partials[v * blocks ...].  group_shift >= 0 (nv = 1): two sums by that bit of the scalar's index (n a multiple of
// 2 << group_shift): class c's `blocks` records are partials[c * blocks ...]
template <class Fq>
u32 launch_direct_sum(hipStream_t st, const u32* table, u32 key_n, u32 nv, const u32* const* scalars, const u32* ns, const u32* base_offs,
                      int mont, u32 m, int group_shift, u32* flags, u32* partials);
template <class Fq>
void launch_apply_inf(hipStream_t st, u32* table, const uint8_t* is_inf, u32 n);
template <class Fq>
void launch_generate_bases(hipStream_t st, u32* table, u64 seed, u32 first, u32 n, const u32* gen_xy_mont);

// The device may keep points in an internal Montgomery radix (fpu.h): key tables, partials and buckets are in it,
// everything the C ABI exposes is not.  import/export convert a point array (src may equal dst); they do
// nothing when device_internal_radix<Fq>() is false.
template <class Fq>
bool device_internal_radix();
template <class Fq>
void launch_points_import(hipStream_t st, const u32* src, u32* dst, u32 n);
template <class Fq>
void launch_points_export(hipStream_t st, const u32* src, u32* dst, u32 n);

// xyzz_scratch (n XYZZ records, may be null): large folds / precompute levels leave their sums there and convert them to
// affine in a second kernel with one inversion per few points (batch_affine_pays(n) says when the scratch is used)
template <class Fq>
bool batch_affine_pays(u32 n);
template <class Fq>
void launch_points_fold(hipStream_t st, const u32* l, const u32* r, u32 n, const u32 x_canon[8], u32 nbits, u32* out,
                        bool abi_radix, u32* xyzz_scratch);

// Fold of a key that carries window multiples (table level w at w * stride points, = 2^(e_w) G with e_w =
// window_exponent_of(c, W, n_narrow, 0, w); levels 0 .. levels-1 usable): out[i] = table[i] + x * table[n + i], i < n.  False
// (nothing launched) when x does not fit the usable levels.
template <class Fq>
bool launch_points_fold_tab(hipStream_t st, const u32* table, u32 stride, u32 c, u32 W, u32 n_narrow, u32 levels, u32 n,
                            const u32 x_canon[8], u32 nbits, u32* out, u32* xyzz_scratch);

// two-valued vectors (vec_kernels.h k_tv_probe, msm_kernels.h k_tv_sum): exact probe into TV_PROBE_WORDS zeroed words, and the
// sum of the generators with non-zero scalars as `blocks` partial records
// out_words: TV_PROBE_WORDS words, zeroed; one: the unit scalar as the vector stores it (word 4 of the output counts how many of
// TV_ONES_SAMPLE_COUNT evenly spaced scalars equal it)
constexpr u32 TV_ONES_SAMPLE_COUNT = 1024;
void launch_tv_probe(hipStream_t st, const u32* scalars, u32 n, u32* out_words, const u32 one[8]);
// nv <= 8 vectors in one launch: vector v's `blocks` partial records at parts + v * blocks, its exception records (8 slots of
// 8 + 2 W words: scalar | generator in the C-ABI radix) at exc + v * 8 slots; probes[v] = the vector's k_tv_probe words
template <class Fq>
void launch_tv_sum(hipStream_t st, const u32* table, u32 nv, const u32* const* scalars, const u32* const* probes, const u32* ns,
                   const u32* base_offs, u32 blocks, u32* parts, u32* exc);
constexpr u32 TV_PROBE_WORDS = 32, TV_EXC_SLOTS = 8;  // == TV_WORDS / TV_EXC_MAX of vec_kernels.h

// ---- scalar-field (Fr) launchers ----------------------------------------------------------------
template <class Fr>
void launch_digits(hipStream_t st, const u32* scalars, int mont, MsmGeom g, void* keys, bool keys16, u32* vals, u32* err);
// Custom prep chain (prep_kernels.h): scalars -> sorted entry list + bucket table in 7 dispatches.  d_small holds
// prep_small_words(g) words (partition totals, starts, cursors, partial counts, the heavy-partition tables); the launcher
// zeroes what needs it.
size_t prep_small_words(const MsmGeom& g);
struct PrepBuffers {
  u32* d_small;       // >= prep_small_words(g) words (+ 256 bytes of slack), directly behind the 16 words at `err`
  u32* part;          // E words: entries grouped by partition
  u32* vals_sorted;   // E + 16 words
  u32* start;         // B + 2 words
  u32* items;         // B + 2 words
  u32* item_off;      // B + 2 words
  u32* err;           // == d_small - 16: the MSM's flag words, cleared by the same fill
};
bool prep_supported(const MsmGeom& g);
// Bucket-per-lane prep (prep_kernels.h: hist, scan, wide scatter, k_prep_local_t): 4 dispatches + the fill
struct PrepBplBuffers {
  u32* d_small;   // as PrepBuffers
  u32* part;      // 2 * E words: 64-bit interchange entries
  u32* ents_t;    // P * stride words: the transposed entry blocks
  void* grp;      // partitions x prep_bpl_groups_per_partition() group headers (8 bytes each)
  u32* order;     // 64 words per group: the bucket every lane slot sums
  u32* err;       // == d_small - 16; err[1] = overflow (the host falls back to the chunked pipeline)
};
bool prep_bpl_supported(const MsmGeom& g);
u32 prep_bpl_stride(const MsmGeom& g);  // entries per partition block of ents_t
u32 prep_bpl_partitions(const MsmGeom& g);
size_t prep_bpl_part_entries(const MsmGeom& g);
size_t prep_bps_part_entries(const MsmGeom& g, u32 log2_l);  // 4-byte interchange entries of the bucket-split prep  // 8-byte interchange entries the partition pass may write
u32 prep_bpl_groups_per_partition();  // group headers (and 64-lane blocks of `order`) per partition
template <class Fr>
int launch_prep_bpl(hipStream_t st, const u32* scalars, int mont, MsmGeom g, const PrepBplBuffers& b);
// Bucket-split prep for small / medium MSMs (prep_kernels.h: k_prep_local_s): every bucket on 2^log2_l adjacent lanes.
// prep_bps_choose: log2 of the lanes per bucket for this geometry, or -1 when it does not take this pipeline.
int prep_bps_choose(const MsmGeom& g);
u32 prep_bps_partitions(const MsmGeom& g, u32 log2_l);
u32 prep_bps_stride(const MsmGeom& g, u32 log2_l);
template <class Fr>
int launch_prep_bps(hipStream_t st, const u32* scalars, int mont, MsmGeom g, u32 log2_l, const PrepBplBuffers& b);
template <class Fq>
void launch_accum_bps(hipStream_t st, const u32* table, const u32* ents_t, const void* grp, const u32* order, u32 n_groups,
                      u32 log2_l, const u32* flags, u32* buckets);
template <class Fr>
int launch_prep(hipStream_t st, const u32* scalars, int mont, MsmGeom g, const PrepBuffers& b);

// *d_flag (zeroed by the caller) = 1 when the vector's c-bit digits look skewed (vec_kernels.h: k_skew_probe)
template <class Fr>
void launch_skew_probe(hipStream_t st, const u32* scalars, int mont, u32 n, DigitWalk walk, u32* d_flag,
                       const u32* d_tv_words = nullptr);
template <class Fr>
void launch_vec_random(hipStream_t st, u32* out, u64 seed, u32 n, int mont);
template <class Fr>
void launch_vec_hadamard(hipStream_t st, const u32* a, const u32* b, u32* out, u32 n);
template <class Fr>
void launch_vec_combine(hipStream_t st, const CombineArgs& a, u32* out);
template <class Fr>
void launch_hp_t_vecs(hipStream_t st, const TVecArgs& a, int n_inputs);

template <class Fr>
void launch_spmv(hipStream_t st, const u32* row_ptr, const u32* col, const u32* val, const u32* input, u32 n_input,
                 const u32* witness, u32 n_witness, u32* out, u32 n_rows);

template <class Fr>
void launch_vec_powers(hipStream_t st, const u32 point_mont[8], u32 n, u32* out);
template <class Fr>
void launch_vec_inner_product(hipStream_t st, const u32* a, const u32* b, u32 n, u32 blocks, u32* out);
template <class Fr>
void launch_vec_inner_product_pair(hipStream_t st, const u32* a0, const u32* b0, const u32* a1, const u32* b1, u32 n,
                                   u32 blocks, u32* out);
// in-place fold of c and z (2 * 2 * half elements -> 2 * half) + the two inner products of the folded halves
template <class Fr>
void launch_ipa_fold_ip(hipStream_t st, u32* c, u32* z, u32 half, const u32 x_mont[8], const u32 xinv_mont[8], u32 blocks,
                        u32* out);
template <class Fr>
void launch_ipa_round_scalars(hipStream_t st, const u32* xi_mont, u32 j, u32 log_n, const u32* c, u32* out_l, u32* out_r);
template <class Fr>
void launch_check_poly_coeffs(hipStream_t st, const u32* xi_mont, u32 k, u32* out);
void launch_vec_fill(hipStream_t st, u32* out, const u32 value[8], u32 n);
// the two flag words of an MSM slot (scalar-range error, prep overflow) into page-locked host memory: what k_fold's mirror does
// for an MSM with a tail, for a range that leaves its sums in a shared bucket set and has none (api_pipeline.inc: Share)
void launch_mirror_flags(hipStream_t st, const u32* flags, u32* host_mirror);
void launch_bounds(hipStream_t st, const void* keys_sorted, bool keys16, u32* vals_sorted, MsmGeom g, u32* start, u32* items);

}  // namespace amsm
