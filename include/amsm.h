/* amsm.h -- C ABI of the MI355X-native MSM engine for accumulation-scheme provers.
 *
 * This is the drop-in boundary for the ONE hot path of arkworks-rs/accumulation: the multi-scalar
 * multiplications / Pedersen commitments inside the hp_as, r1cs_nark_as, ipa_pc_as (and trivial_pc_as)
 * provers and deciders, plus the scalar-field vector loops that produce the MSM scalars.
 *
 * The reference has no FFI seam (src/lib.rs:24 `#![forbid(unsafe_code)]`); the calls replaced are
 * monomorphised Rust generics.  Each entry point below names the reference interface it stands in
 * for (paths relative to /root/reference; "ext" = a dependency whose source is not in that tree).
 * INTEGRATION.md shows the Rust `extern "C"` adapter a maintainer would add.
 *
 * Data formats (identical to ark-ff 0.2 memory, so Rust slices can be passed zero-copy):
 *   - field element  : little-endian u64 limbs of x*R mod m (Montgomery form), R = 2^(64*limbs);
 *                      limbs = 4 for Pallas Fq/Fr and BLS12-381 Fr, 6 for BLS12-381 Fq.
 *   - scalar (BigInt): 4 little-endian u64 limbs of the canonical integer in [0, r)  (`into_repr()`),
 *                      or Montgomery form when the `scalars_mont` argument is non-zero (raw `Vec<Fr>`).
 *   - affine point   : 2*limbs u64 = x_mont | y_mont, plus a separate is_inf byte (ark-ec
 *                      `GroupAffine{x,y,infinity}` is not repr(C), so the adapter marshals it).
 *
 * Ownership: the caller owns every host buffer; calls are synchronous (results are on the host when
 * the call returns) unless the name ends in `_device`/`_async`.  Device memory lives behind the
 * opaque handles.  A ctx is NOT thread-safe (one thread at a time per context; different contexts may run on different
 * threads); an amsm_bases is immutable after creation.  The context-free host helpers (amsm_host_lincomb[_batch], amsm_fr_*,
 * amsm_*_serialize / _deserialize, amsm_poseidon_* on distinct sponges) may be called from any thread concurrently, and from a
 * fork()ed child of a process that used them.
 * Errors: 0 = OK, negative = AMSM_E_*; nothing throws across the boundary.
 * Nothing falls back to the CPU implicitly: with no usable gfx950 GPU amsm_ctx_create(.., device_id >= 0, ..) fails with
 * AMSM_E_NO_DEVICE.  A caller that WANTS the host backend asks for it (device_id = AMSM_DEVICE_HOST, or amsm_ctx_create_multi with
 * n_dev = 0: SURVEY.md section 8(b), BASELINE.json config 1 "plumbing, no GPU"): every entry point below then computes on the host
 * cores -- same arguments, same results; "device" pointers are host memory from amsm_dev_alloc; keys are plain (flags are hints).
 *
 * Multi-GPU: ONE process drives the GPUs of a node through a multi-device context (amsm_ctx_create_multi, below): the
 * committer key is sharded over the devices by contiguous index ranges, every MSM entry point that takes a key accepts a
 * sharded one, and the devices' partial sums are exchanged inside the library (RCCL all-gather over xGMI, peer copies as
 * the fallback).  The one-process-per-GPU form (amsm_msm_partial*_device + the caller's own collective) remains.
 */
#ifndef AMSM_H
#define AMSM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct amsm_ctx amsm_ctx;
typedef struct amsm_bases amsm_bases;

enum amsm_curve {
  AMSM_PALLAS = 0,       /* ark_pallas::Affine -- the only curve the reference exercises (Cargo.toml:39) */
  AMSM_BLS12_381_G1 = 1, /* BASELINE.json config 3 (384-bit base field); extension, no reference harness */
};

enum amsm_status {
  AMSM_OK = 0,
  AMSM_E_INVALID_ARG = -1,
  AMSM_E_OOM = -2,
  AMSM_E_HIP = -3,
  AMSM_E_UNSUPPORTED = -4,
  AMSM_E_NO_DEVICE = -5,
  AMSM_E_SCALAR_RANGE = -6, /* a scalar was >= 2^255 (not a canonical `into_repr()` value) */
  AMSM_E_RCCL = -7,         /* the RCCL collective of a multi-device context failed */
};

/* flags for amsm_bases_load / amsm_bases_generate */
enum amsm_bases_flags {
  AMSM_BASES_DEFAULT = 0,      /* library picks: precomputed whenever the table fits (round 4: at every size -- keys of up to
                                  2^15 generators also get the direct-sum table, see amsm_ctx_direct_sum_msms) */
  AMSM_BASES_PRECOMPUTE = 1,   /* keep 2^(c*w)*G_i for every window w resident in HBM (W x memory); creation fails with
                                  AMSM_E_OOM / AMSM_E_UNSUPPORTED (n * W >= 2^30) when the table cannot be built --
                                  only AMSM_BASES_DEFAULT falls back to a plain key (amsm_bases_precomputed() tells) */
  AMSM_BASES_NO_PRECOMPUTE = 2, /* one copy of the key; windows are combined on the host */
  /* Round 5 -- the OPTIONAL tables are the caller's decision (OR them into any of the above):
   *   a precomputed key of up to 2^15 generators also holds 512 affine points per generator (32 KiB each: 1.07 GB for 2^15 Pallas
   *   generators, 1.6 GB BLS12-381) from which its MSMs are summed without buckets (2-4x faster there);
   *   a key with the 20-bit table (>= 2^20 generators) builds a 17-bit TWIN of about the same size (1 GiB at 2^20) on the first
   *   call that needs it -- a range below a quarter of 2^20 pairs, a vector with skewed digits. */
  AMSM_BASES_NO_DIRECT_TABLE = 4, /* never build the direct-sum table: such a key's MSMs take the windowed pipelines */
  AMSM_BASES_NO_TWIN = 8,         /* never build the twin: the calls that would need it return AMSM_E_UNSUPPORTED */
  /* Round 6 -- multi-device contexts only (ignored elsewhere): every device holds the WHOLE key instead of a shard of it, and the
   * entry points that carry several independent MSMs over one key (amsm_msm_batch_device, amsm_msm_batch,
   * amsm_pedersen_commit_batch) deal them to the devices round-robin -- MSM v runs whole on device v mod n_dev, no partial sums,
   * no exchange, results in call order.  What a SMALL key wants: point-sharding 2^18 generators 8 ways leaves 2^15 pairs per GPU
   * per MSM (the latency regime), while `r1cs_nark_as::prove` issues 2-8 independent commitments per round
   * (src/r1cs_nark_as/r1cs_nark/mod.rs:216-218,234-236,251,261; src/r1cs_nark_as/mod.rs:394-410).  Every other entry point uses
   * the primary device's copy, so such a key is accepted wherever a single-device key is (grouped MSMs, the IPA round, key folds). */
  AMSM_BASES_REPLICATE = 16
};

const char* amsm_strerror(int status);
/* Number of usable gfx950 devices (0 when there is none); never initialises a context. */
int amsm_device_count(void);
/* device_id of the host backend (accumulation_amd/csrc/api_cpu.inc: window-parallel Pippenger + plain loops on the library's own
 * host field arithmetic; what ark-ec's CPU MSM is to the reference).  Explicit only. */
#define AMSM_DEVICE_HOST (-1)

/* ---- context -------------------------------------------------------------------------------- */
/* One context = one GPU + one HIP stream + a grow-only workspace.
 * `stream` is a hipStream_t owned by the caller (e.g. torch.cuda.current_stream().cuda_stream) or
 * NULL to let the context create its own non-blocking stream. */
int amsm_ctx_create(amsm_ctx** out, int curve, int device_id, void* stream);
/* 1 for a context of the host backend (device_id == AMSM_DEVICE_HOST / n_dev == 0), else 0. */
int amsm_ctx_is_host(const amsm_ctx* ctx);
/* Multi-device context (reference: `prove` is ONE synchronous call in ONE process,
 * src/lib.rs:163-249): n_dev >= 1 devices driven by the calling process.  Device device_ids[0] is the PRIMARY: scalar
 * vectors handed to the `_device` entry points, the scalar-field vector kernels and every result live there.  Keys created
 * through this context are SHARDED: device g holds generators [lo_g, hi_g) = [g*n/n_dev, (g+1)*n/n_dev) (its own
 * precomputed table), and amsm_msm / amsm_msm_device / amsm_msm_batch_device / amsm_pedersen_commit[_device] split the
 * scalars the same way (host slices are uploaded straight to their device, primary-device vectors are peer-copied), run the
 * shards concurrently (one host worker thread per device), gather the per-device 128/192-byte partial sums on the primary
 * with ONE RCCL all-gather per call (communicators created from the device list; hipMemcpyPeerAsync when RCCL is
 * unavailable, AMSM_COLLECTIVE=peer, or a device is listed twice) and fold + normalise once.  Results are bit-identical to
 * the single-device ones.  Grouped MSMs and the IPA round shard too since round 5 (every shard sums its part of both index
 * classes -- its two sums come back to the host, where they are folded: no device exchange on that latency-bound path --, the round's scalars are peer-copied slice by slice: an `ipa_pc` opening over a sharded key runs every round as one
 * grouped MSM over the ORIGINAL key and never folds it).  Multi-offset MSMs go out as one sharded batch per run of jobs
 * over the same key range.  Entry points that do not shard (key folds, amsm_bases_device_ptr) return AMSM_E_UNSUPPORTED / NULL for a sharded key.
 * A device id may appear more than once (two shards on one GPU: how a 1-GPU box exercises this path).
 * n_dev == 0 (device_ids ignored): a context of the host backend, as amsm_ctx_create(.., AMSM_DEVICE_HOST, NULL). */
int amsm_ctx_create_multi(amsm_ctx** out, int curve, const int* device_ids, int n_dev);
/* 1 for amsm_ctx_create contexts. */
int amsm_ctx_num_devices(const amsm_ctx* ctx);
/* The single-device context of shard g (borrowed: destroyed with its parent; g = 0 is the primary, i.e. usable exactly
 * like `ctx` itself): lets the caller allocate / fill / transform scalar vectors ON device g with the ordinary entry
 * points, e.g. the slices amsm_msm_batch_sharded_device takes.  NULL when g is out of range. */
amsm_ctx* amsm_ctx_shard(amsm_ctx* ctx, int g);
/* Keys of up to `n_generators` generators created through this multi-device context from now on are REPLICATED
 * (AMSM_BASES_REPLICATE) instead of sharded, whatever their flags; 0 (the default): only keys that ask for it. */
int amsm_ctx_set_replicate_below(amsm_ctx* ctx, size_t n_generators);
/* Devices that hold a full copy of this key: 1 for single-device and sharded keys, n_dev for a replicated one. */
int amsm_bases_replicas(const amsm_bases* bases);
/* MSMs of batch calls over replicated keys that ran on a device other than the primary so far. */
unsigned long long amsm_ctx_replicated_msms(const amsm_ctx* ctx);
/* "rccl", "peer-copy" (multi-device contexts) or "none". */
const char* amsm_ctx_collective(const amsm_ctx* ctx);
/* Exchanges of partial records (one RCCL all-gather, or one round of peer copies) this multi-device context has run:
 * exactly ONE per sharded MSM / commit call however many vectors the call carries, so a prover's exchange count is its
 * number of DEPENDENT commit rounds (tests/test_cpp_multi_device.py counts them).  0 for single-device contexts. */
unsigned long long amsm_ctx_collectives(const amsm_ctx* ctx);
/* MSMs of device vectors that took the two-valued form so far: every scalar 0 or one value v -> v * (sum of the generators
 * with a non-zero scalar).  ark-ec's multi_scalar_mul special-cases the scalars 0 and 1 the same way (ark-ec ^0.2.0 msm, not
 * in /root/reference: Cargo.toml:15); the reference's hp_as inputs are `vec![rand; n]` (src/hp_as/mod.rs:189-190). */
unsigned long long amsm_ctx_two_valued_msms(const amsm_ctx* ctx);
/* MSMs of device vectors whose UNIT scalars were summed apart (round 5): a vector that is not two-valued but holds a share of
 * ones -- the boolean wires of an R1CS witness -- would put all of them into bucket 1 of the lowest window (2.2x slower than a
 * uniform vector at 2^18 pairs with 10 % booleans).  The same exact pass that tests for two-valued vectors counts the ones among
 * 1024 evenly spaced scalars; from 8 up the generators under unit scalars are summed by themselves (n_ones additions) and the
 * windowed pipelines skip those scalars -- what ark-ec's multi_scalar_mul does with them.  Results do not depend on it. */
unsigned long long amsm_ctx_unit_scalar_msms(const amsm_ctx* ctx);
/* MSMs that were summed straight from a small key's table of digit multiples (round 4: precomputed keys of up to 2^15 generators --
 * AMSM_DIRECT_SUM_MAX_LOG2, at most 16, 0 turns it off -- also hold j 2^(4w) G_i for j = 1 .. 8, w = 0 .. 63, 512 affine points per
 * generator, so that an MSM is one launch of mixed additions and a tree: no buckets, no sort, no dependence on the digit
 * distribution).  Grouped MSMs (amsm_msm_grouped_device, the IPA rounds) are two such sums in one launch -- one row of workgroups
 * per index class -- whenever n is a multiple of 2 << group_shift (both classes then hold n / 2 indices), else they take the
 * windowed pipelines.  Results do not depend on the path. */
unsigned long long amsm_ctx_direct_sum_msms(const amsm_ctx* ctx);
/* MSMs longer than the 2^c-pair window of their bucket-per-lane key (2^22 pairs over a 20-bit key: BASELINE.json config 5, the
 * commitments of src/hp_as/mod.rs:372-385,911-918) that ran over ONE bucket set: range 1 writes the set, ranges 2 .. k add to it,
 * one bucket reduction and one fold per MSM instead of one per range (round 5).  A range whose digits turn out skewed sends the
 * whole MSM back to independent ranges; results do not depend on the path. */
unsigned long long amsm_ctx_shared_bucket_msms(const amsm_ctx* ctx);
/* Which accumulation pipeline the context's MSMs took so far: *n_bucket_per_lane = MSMs enqueued on the bucket-per-lane
 * pipeline (keys of >= 2^20 generators, MSMs of (2^19, 2^20] pairs -- longer ones as windows of 2^20; 20-bit windows, 13
 * gathered additions per pair), *n_fallbacks = those whose scalars turned out skewed (a digit value shared by a large part of
 * a window: constant vectors) and were re-run through the chunked pipeline over the key's 17-bit-window twin,
 * which is built on the first such call.  Results do not depend on the path.  Either pointer may be NULL. */
int amsm_ctx_pipeline_stats(const amsm_ctx* ctx, unsigned long long* n_bucket_per_lane, unsigned long long* n_fallbacks);
/* The same counters for the bucket-split pipeline that MSMs of 2^13 .. 2^19 pairs over a precomputed key take (every bucket on
 * 2 .. 64 adjacent lanes, no partial records): enqueued / re-run through the chunked pipeline over the same key because the
 * scalars were skewed. */
int amsm_ctx_pipeline_stats_small(const amsm_ctx* ctx, unsigned long long* n_bucket_split, unsigned long long* n_fallbacks);
void amsm_ctx_destroy(amsm_ctx* ctx);
int amsm_ctx_curve(const amsm_ctx* ctx);
/* limbs (u64) of a base-field element: 4 (Pallas) or 6 (BLS12-381). */
int amsm_ctx_fq_limbs(const amsm_ctx* ctx);
/* Override the Pippenger window width c (bits); 0 restores the automatic choice. */
int amsm_ctx_set_window(amsm_ctx* ctx, int c_bits);
int amsm_ctx_synchronize(amsm_ctx* ctx);

/* Device memory the context holds: the MSM workspace (grow-only: sized by the largest MSM so far, three pipeline slots), the
 * bytes of live amsm_dev_alloc buffers and the bytes parked on the allocator's free lists (amsm_dev_free keeps buffers for
 * reuse because hipMalloc / hipFree synchronise the device; AMSM_POOL_MAX_MB caps the free lists, default 16384).
 * amsm_ctx_trim synchronises and releases the workspace and the free lists (live buffers and keys stay). */
int amsm_ctx_memory(const amsm_ctx* ctx, size_t* workspace_bytes, size_t* vectors_live_bytes, size_t* vectors_pooled_bytes);
int amsm_ctx_trim(amsm_ctx* ctx);
/* Device memory a key holds: its table (W levels of affine points when precomputed -- plus 512 points per generator for a key of up
 * to 2^15 generators, the direct-sum table -- the generators otherwise), the C-ABI-radix
 * copy amsm_bases_device_ptr made (0 if never asked), and the 17-bit-window TWIN of a 20-bit key -- built lazily, inside the
 * first MSM that needs it (a range below a quarter of 2^20 pairs, a vector with skewed digits; round 3 also grouped MSMs),
 * about as large as the table itself: 0 until then.  amsm_bases_prebuild_twin builds it NOW (at key-creation time, on the
 * caller's schedule) so that no prove-time call allocates or stalls; AMSM_OK without effect for keys that have none.
 * Sharded keys report the sum over their shards. */
int amsm_bases_memory(const amsm_bases* bases, size_t* table_bytes, size_t* abi_copy_bytes, size_t* twin_bytes);
int amsm_bases_prebuild_twin(amsm_ctx* ctx, const amsm_bases* bases);
/* Which tables a key holds, one by one (round 5).  out[0] = the window table (W levels of affine points; the generators alone for a
 * plain key), out[1] = the direct-sum table (0: none), out[2] = the twin (0: not built), out[3] = the C-ABI-radix copy, out[4] = W,
 * out[5] = why there is no direct-sum table although the key qualified: 0 (there is one / it never qualified), 1
 * AMSM_BASES_NO_DIRECT_TABLE, 2 the context's table budget, 3 the allocation failed; out[6] = the same for the twin when a call was
 * refused one.  Sharded keys: sums over the shards, the largest reason code. */
int amsm_bases_tables(const amsm_bases* bases, size_t out[7]);
/* Bytes ONE key may spend on its optional tables (each table separately; default: no limit).  Keys created afterwards obey it; a
 * table that does not fit is not built (direct sum) or refused (twin: AMSM_E_UNSUPPORTED from the call that needs it), and
 * amsm_ctx_tables_denied counts it.  Eight ranks of a node each hold their own tables: budget accordingly. */
int amsm_ctx_set_table_budget(amsm_ctx* ctx, size_t bytes_per_table);
/* Tables denied so far by the budget or by a failed allocation (not by the caller's own flags): a key that silently stays on the
 * 2-4x slower path is visible here. */
unsigned long long amsm_ctx_tables_denied(const amsm_ctx* ctx);

/* Per-stage device timings of the LAST msm call (hipEvent pairs on the context's stream).
 * Enable with on != 0; stage names: amsm_stage_name(i), i in [0, amsm_stage_count()). */
int amsm_ctx_set_profiling(amsm_ctx* ctx, int on);
int amsm_stage_count(void);
const char* amsm_stage_name(int stage);
int amsm_ctx_stage_ms(amsm_ctx* ctx, int stage, float* ms);

/* ---- committer key (generators) ------------------------------------------------------------- */
/* Replaces the generator vector of `ark_poly_commit::trivial_pc::CommitterKey` (ext) built by
 * `PedersenCommitment::setup/trim` -- call sites src/hp_as/mod.rs:640-641,
 * src/r1cs_nark_as/r1cs_nark/mod.rs:107-108.  Generators are static per key, so they are copied to
 * HBM once and stay resident.  xy_mont: n * 2 * limbs u64; is_inf: n bytes or NULL. */
int amsm_bases_load(amsm_ctx* ctx, const uint64_t* xy_mont, const uint8_t* is_inf, size_t n, unsigned flags,
                    amsm_bases** out);
/* Synthetic key: G_i = k_i * G with k_i from the counter-based splitmix64 stream `seed`
 * (oracle/pyref.py:rng_points is the definition).  Stands in for `PedersenCommitment::setup(n)`
 * (ext; ark-poly-commit hashes to the curve -- any fixed set of distinct subgroup points is
 * equivalent for this path). */
int amsm_bases_generate(amsm_ctx* ctx, uint64_t seed, size_t n, unsigned flags, amsm_bases** out);
/* Copy generators [off, off+n) back to the host (affine, Montgomery). */
int amsm_bases_read(amsm_ctx* ctx, const amsm_bases* bases, size_t off, size_t n, uint64_t* xy_mont, uint8_t* is_inf);
size_t amsm_bases_len(const amsm_bases* bases);
/* Shards of a key (1 for a single-device key) and the generator range [*lo, *hi) shard g holds. */
int amsm_bases_num_shards(const amsm_bases* bases);
int amsm_bases_shard_range(const amsm_bases* bases, int g, size_t* lo, size_t* hi);
int amsm_bases_precomputed(const amsm_bases* bases);
/* Window width c of a precomputed key (its table holds ceil-ish(256 / c) window multiples of every generator; fixed at
 * creation from the key's size), 0 for a plain key.  For reporting: an MSM over it gathers one point per non-zero c-bit
 * digit of every scalar. */
int amsm_bases_window_bits(const amsm_bases* bases);
void amsm_bases_free(amsm_bases* bases);

/* ---- MSM ------------------------------------------------------------------------------------ */
/* Replaces `ark_ec::msm::VariableBaseMSM::multi_scalar_mul(&bases[off..], &scalars)` (ext, ark-ec
 * ^0.2.0, Cargo.toml:15) followed by `.into_affine()`, i.e. what every
 * `PedersenCommitment::commit(ck, v, None)` call site needs: src/hp_as/mod.rs:377,911-918;
 * src/r1cs_nark_as/mod.rs:394-410,1081-1093; src/r1cs_nark_as/r1cs_nark/mod.rs:216-218,234-236,251,261,375-403;
 * and the (d+1)-point MSM under src/ipa_pc_as/mod.rs:836.
 * Semantics kept: uses min(n, len - base_off) pairs; zero scalars and identity bases contribute
 * nothing; n = 0 returns the identity; the result does not depend on pair order.
 * scalars: n * 4 u64 (host).  scalars_mont != 0 => Montgomery form (device performs `into_repr`).
 * out_xy_mont: 2 * limbs u64 (zeroed when the result is the identity); out_is_inf: 1 byte. */
int amsm_msm(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const uint64_t* scalars, size_t n,
             int scalars_mont, uint64_t* out_xy_mont, uint8_t* out_is_inf);
/* The literal `VariableBaseMSM::multi_scalar_mul(bases: &[G], scalars: &[BigInt]) -> G::Projective` (ext, ark-ec ^0.2.0,
 * Cargo.toml:15; SURVEY.md section 8(b) "where a replacement can attach (1)": a `[patch]` of ark-ec's msm body): BOTH slices are
 * host memory and belong to this call only -- no amsm_bases handle, nothing left resident (the generators pass through a
 * grow-only buffer of the context that amsm_ctx_trim releases).  Uses min(n_bases, n_scalars) pairs like ark-ec; identity bases
 * (is_inf[i] != 0, or x = y = 0) and zero scalars contribute nothing; n = 0 returns the identity.  Ranges of 2^19 pairs: the
 * generators (64 / 96 B each) and scalars (32 B) of range j + 1 cross the link while range j's MSM runs over a plain key (no
 * precomputed multiples: they would cost more than the MSM), so a long call is bound by the link -- 96 B per Pallas pair -- not by
 * upload + MSM in series (bench.py: config.pairs_per_s_oneshot_host_bases and its fraction of the measured H2D rate).  A caller
 * whose generators are a static committer key should amsm_bases_load them ONCE instead.
 * Multi-device contexts split the pairs over the devices; the host backend computes it on the host cores. */
int amsm_msm_oneshot(amsm_ctx* ctx, const uint64_t* bases_xy_mont, const uint8_t* bases_is_inf, size_t n_bases,
                     const uint64_t* scalars, size_t n_scalars, int scalars_mont, uint64_t* out_xy_mont, uint8_t* out_is_inf);
/* Same with the scalar vector already resident in HBM (device pointer, n * 32 bytes, 16-byte
 * aligned) -- the form the field-vector kernels below feed, and the form bench.py times. */
int amsm_msm_device(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const void* d_scalars, size_t n,
                    int scalars_mont, uint64_t* out_xy_mont, uint8_t* out_is_inf);
/* n_vecs independent MSMs over the same generators (replaces the sequential loop of
 * `compute_product_poly_comm`, src/hp_as/mod.rs:354-388).  d_scalars[v] are device pointers;
 * outputs are n_vecs consecutive points. */
int amsm_msm_batch_device(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const void* const* d_scalars,
                          size_t n_vecs, size_t n, int scalars_mont, uint64_t* out_xy_mont, uint8_t* out_is_inf);

/* amsm_msm_batch_device over a WHOLE sharded key with the scalars already sharded: d_slices[v * n_dev + g] is the slice of
 * vector v for shard g -- (hi_g - lo_g) scalars resident on device g (allocated / produced through amsm_ctx_shard(ctx, g)).
 * No scalar crosses a device boundary: only the partial sums do (weak scaling: what bench.py --gpus N times in its
 * single-process mode).  A single-device context accepts n_dev = 1.  A key WITHOUT shards on a multi-device context (a
 * replicated or folded key: its vectors are whole) has no slices to take: AMSM_E_INVALID_ARG. */
int amsm_msm_batch_sharded_device(amsm_ctx* ctx, const amsm_bases* bases, const void* const* d_slices, size_t n_vecs,
                                  int scalars_mont, uint64_t* out_xy_mont, uint8_t* out_is_inf);

/* n_msms independent MSMs over windows of ONE key, pipelined like amsm_msm_batch_device: MSM v uses generators
 * [base_offs[v], base_offs[v] + ns[v]) and the device scalars d_scalars[v].  The two cross commitments of an IPA
 * round, <c_r, key_l> and <c_l, key_r> (ark_poly_commit::ipa_pc ext, under src/ipa_pc_as/mod.rs:454), are one call. */
int amsm_msm_multi_device(amsm_ctx* ctx, const amsm_bases* bases, size_t n_msms, const size_t* base_offs,
                          const void* const* d_scalars, const size_t* ns, int scalars_mont, uint64_t* out_xy_mont,
                          uint8_t* out_is_inf);

/* Two MSMs over disjoint index classes of ONE scalar vector in one pass: out[g] (g = 0, 1) = sum over the i with
 * ((i >> group_shift) & 1) == g of scalars[i] * generators[base_off + i].  Same kernels as amsm_msm_device with two
 * bucket sets; one prep, one accumulate, one tail instead of two (the cross commitments of an IPA round). */
int amsm_msm_grouped_device(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const void* d_scalars, size_t n,
                            int scalars_mont, unsigned group_shift, uint64_t* out_xy_mont, uint8_t* out_is_inf);

/* Multi-GPU (one process per GPU): each rank runs the MSM over its shard of the key and leaves a
 * fixed-size un-normalised partial in device memory; ranks all-gather the partials (RCCL, raw bytes)
 * and every rank folds them.  amsm_partial_bytes() is the per-rank record size. */
size_t amsm_partial_bytes(const amsm_ctx* ctx);
int amsm_msm_partial_device(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const void* d_scalars, size_t n,
                            int scalars_mont, void* d_partial_out);
int amsm_partials_combine(amsm_ctx* ctx, const void* d_partials, size_t n_partials, uint64_t* out_xy_mont,
                          uint8_t* out_is_inf);
/* Batched forms (the prover commits several vectors back to back, src/hp_as/mod.rs:354-388): n_vecs MSMs over the
 * rank's shard, pipelined like amsm_msm_batch_device, leave n_vecs consecutive records at d_partials_out; after ONE
 * all-gather of all records, amsm_partials_combine_batch folds n_groups groups of `count` consecutive records
 * (group g = the `count` ranks' records of MSM g) into n_groups consecutive affine points. */
int amsm_msm_partial_batch_device(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const void* const* d_scalars,
                                  size_t n_vecs, size_t n, int scalars_mont, void* d_partials_out);
int amsm_partials_combine_batch(amsm_ctx* ctx, const void* d_partials, size_t n_groups, size_t count,
                                uint64_t* out_xy_mont, uint8_t* out_is_inf);

/* n_vecs MSMs over the same generators whose scalar vectors are HOST slices (scalars[v]: n * 4 u64) -- the form the
 * reference's provers hand their commitments over in, back to back: src/hp_as/mod.rs:372-385 (the t-vectors), :196-214 (the
 * hiding commitments), src/r1cs_nark_as/r1cs_nark/mod.rs:216-218 (comm_a, comm_b, comm_c).  The upload of vector v + 1 runs on
 * a copy stream while MSM v computes, so a batch pays PCIe once per vector only where it is longer than the MSM (32 MiB at
 * 2^20: ~0.6 ms against ~1.1 ms); results as amsm_msm_batch_device.  The slices need not be pinned. */
int amsm_msm_batch(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const uint64_t* const* scalars, size_t n_vecs,
                   size_t n, int scalars_mont, uint64_t* out_xy_mont, uint8_t* out_is_inf);

/* Page-locking caller memory for the host-slice entry points (amsm_msm, amsm_msm_batch, amsm_pedersen_commit[_batch]).
 * Since round 5 amsm_host_register / amsm_host_unregister are NO-OPS kept for binary compatibility (AMSM_OK; AMSM_E_INVALID_ARG
 * for a null pointer / zero bytes): on MI355X / ROCm 7.2 a 32 MiB scalar vector crosses the link at 55-56 GB/s from pageable,
 * registered and hipHostMalloc'ed memory alike, so page-locking cannot gain throughput, and it lost some -- the first copy out
 * of a freshly registered region stalls ~2 ms (lazy pinning), on some boxes every copy does (host-slice batches -20 % .. -27 %).
 * Evidence: profiles/r05_host_slices.md.  Hand the slices over as they are (a Rust `&[Fr]`); memory the caller has page-locked
 * itself still works. */
int amsm_host_register(void* ptr, size_t bytes);
int amsm_host_unregister(void* ptr);
/* 1: `ptr` is page-locked (hipHostRegister'ed or hipHostMalloc'ed by the caller) */
int amsm_host_is_pinned(const void* ptr);

/* Replaces `PedersenCommitment::commit(ck, elems, Some(r))` (ext): MSM over ck.generators[..n] plus
 * r * hiding_generator (single scalar-mul, done on the host like SURVEY.md section 8(a) row a11).
 * elems_mont: raw `&[Fr]` memory (Montgomery).  randomizer_mont / hiding_xy_mont may be NULL (no
 * hiding term, `commit(.., None)`). */
int amsm_pedersen_commit(amsm_ctx* ctx, const amsm_bases* ck, const uint64_t* elems_mont, size_t n,
                         const uint64_t* randomizer_mont, const uint64_t* hiding_xy_mont, uint64_t* out_xy_mont,
                         uint8_t* out_is_inf);

/* n_vecs commitments over one key in one call: elems_mont[v] = ns[v] Montgomery elements on the HOST (raw `&[Fr]` memory,
 * lengths may differ), randomizers_mont = NULL or n_vecs pointers of which any may be NULL (`commit(.., None)`); uploads
 * overlapped with the previous commitment's MSM as in amsm_msm_batch, the r_v * hiding_generator terms on host threads, ONE
 * batched normalisation.  What a `[patch]`ed `PedersenCommitment::commit` loop calls. */
int amsm_pedersen_commit_batch(amsm_ctx* ctx, const amsm_bases* ck, const uint64_t* const* elems_mont, const size_t* ns,
                               size_t n_vecs, const uint64_t* const* randomizers_mont, const uint64_t* hiding_xy_mont,
                               uint64_t* out_xy_mont, uint8_t* out_is_inf);

/* Same with the committed vector already resident in HBM (Montgomery form). */
int amsm_pedersen_commit_device(amsm_ctx* ctx, const amsm_bases* ck, const void* d_elems_mont, size_t n,
                                const uint64_t* randomizer_mont, const uint64_t* hiding_xy_mont,
                                uint64_t* out_xy_mont, uint8_t* out_is_inf);

/* Host-side linear combination sum_i s_i * P_i of a HANDFUL of points (no GPU, no context): the O(#inputs)
 * scalar-muls + `batch_normalization_into_affine` of `combine_commitments` /
 * `compute_combined_hp_commitments` (src/hp_as/mod.rs:391-479) and their r1cs_nark_as / ipa_pc_as
 * counterparts (these stay on the host).  scalars_mont: n*4 u64 Montgomery. */
int amsm_host_lincomb(int curve, const uint64_t* xy_mont, const uint8_t* is_inf, const uint64_t* scalars_mont,
                      size_t n, uint64_t* out_xy_mont, uint8_t* out_is_inf);
/* n_jobs INDEPENDENT combinations in one call: job j is sum_i scalars_mont[j][i] * xy_mont[j][i] over n_terms[j] points
 * (is_inf may be NULL, or hold NULL for a job without points at infinity); out_xy_mont[j], out_is_inf[j] its result.
 * The groups the provers form -- the four blinded commitments of every input and the three beta-combinations of
 * src/r1cs_nark_as/mod.rs:220-286,452-542, the three combined commitments of src/hp_as/mod.rs:391-479 -- run on a small
 * pool of host threads (AMSM_HOST_THREADS, default up to 3 helpers beside the caller; the reference's `parallel` feature
 * does the same with rayon) and are normalised with one inversion.  A single amsm_host_lincomb of 8 points or more (the IPA
 * verifier's 2 log n + 2 point combination) is split over the same pool.  Results do not depend on the thread count. */
int amsm_host_lincomb_batch(int curve, size_t n_jobs, const size_t* n_terms, const uint64_t* const* xy_mont,
                            const uint8_t* const* is_inf, const uint64_t* const* scalars_mont, uint64_t* out_xy_mont,
                            uint8_t* out_is_inf);
/* Helper threads of the process-wide host pool behind the two calls above (and behind the host backend): AMSM_HOST_THREADS if set,
 * else min(7, cores / LOCAL_WORLD_SIZE - 1) -- a node's ranks (one process per GPU under torchrun) share its cores. */
int amsm_host_threads(void);

/* ---- device buffers for scalar-field vectors ------------------------------------------------- */
int amsm_dev_alloc(amsm_ctx* ctx, size_t bytes, void** d_ptr);
int amsm_dev_free(amsm_ctx* ctx, void* d_ptr);
int amsm_dev_upload(amsm_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int amsm_dev_download(amsm_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);
/* Fill d_out with n synthetic scalars of stream `seed` (oracle/pyref.py:rng_scalar), canonical
 * integers (mont == 0) or the Montgomery form of the same integers (mont != 0). */
int amsm_vec_random(amsm_ctx* ctx, uint64_t seed, size_t n, int mont, void* d_out);

/* out[i] = value for i < n  (`vec![x; len]`, src/hp_as/mod.rs:189-190).  value_mont: 4 u64. */
int amsm_vec_fill(amsm_ctx* ctx, const uint64_t* value_mont, size_t n, void* d_out);

/* ---- scalar-field (Fr) vector kernels; all operands Montgomery, n elements of 32 bytes -------- */
/* out[i] = a[i]*b[i]                      -- `compute_hp`, src/hp_as/mod.rs:278-285 */
int amsm_vec_hadamard(amsm_ctx* ctx, const void* d_a, const void* d_b, void* d_out, size_t n);
/* out[i] = sum_j coeff[j]*vecs[j][i] (+ hiding[i])  -- `combine_vectors` / `scale_vector`,
 * src/hp_as/mod.rs:482-512.  coeffs_mont: host, n_vecs*4 u64.  d_hiding may be NULL.
 * lens[j] (host, may be NULL = all n) gives ragged lengths; missing entries read as zero. */
int amsm_vec_combine(amsm_ctx* ctx, const void* const* d_vecs, const size_t* lens, size_t n_vecs,
                     const uint64_t* coeffs_mont, const void* d_hiding, size_t hiding_len, void* d_out, size_t n);
/* t-vectors of `compute_t_vecs`, src/hp_as/mod.rs:288-349: for every position li the 2n-1
 * coefficients of (sum_j mu_j a_j[li] X^j) * (sum_j b_{n-1-j}[li] X^j), hiding terms optional.
 * d_t[k] (k = 0..2n-2) receive the coefficient vectors; d_t[n-1] may be NULL (it is never
 * committed, src/hp_as/mod.rs:373-375).  a_lens/b_lens (host, NULL = all `len`) give ragged witness
 * lengths: missing entries read as zero like `.get(li)` at :306-318.  Returns AMSM_E_INVALID_ARG when
 * n_inputs + (hiding ? 1 : 0) > n_mu (the reference's assert at :295). */
int amsm_hp_t_vecs(amsm_ctx* ctx, const void* const* d_a, const size_t* a_lens, const void* const* d_b,
                   const size_t* b_lens, size_t n_inputs, const uint64_t* mu_mont, size_t n_mu, const void* d_hiding_a,
                   size_t hiding_a_len, const void* d_hiding_b, size_t hiding_b_len, void* const* d_t, size_t len);

/* ---- host scalar-field helpers (no device work) -------------------------------------------------------------- */
/* Elementwise over n scalars of 4 u64 each: what a scheme driver needs for its O(#inputs) challenge arithmetic
 * (powers and products of squeezed challenges, src/hp_as/mod.rs:233-275) without its own field code.  The
 * canonical value must be < r for amsm_fr_to_mont. */
int amsm_fr_mul(int curve, const uint64_t* a_mont, const uint64_t* b_mont, size_t n, uint64_t* out_mont);
int amsm_fr_add(int curve, const uint64_t* a_mont, const uint64_t* b_mont, size_t n, uint64_t* out_mont);
int amsm_fr_sub(int curve, const uint64_t* a_mont, const uint64_t* b_mont, size_t n, uint64_t* out_mont);
int amsm_fr_inv(int curve, const uint64_t* a_mont, size_t n, uint64_t* out_mont); /* 0 -> 0 */
int amsm_fr_to_mont(int curve, const uint64_t* canonical, size_t n, uint64_t* out_mont);
int amsm_fr_from_mont(int curve, const uint64_t* a_mont, size_t n, uint64_t* out_canonical);

/* ---- wire format: ark-serialize 0.2 `CanonicalSerialize` / `CanonicalDeserialize` (ext) ---
 * The reference derives it for every instance / witness / proof type (src/hp_as/data_structures.rs:13,53,76,94,
 * src/r1cs_nark_as/data_structures.rs:105,155,217,249, src/ipa_pc_as/data_structures.rs:55,76) and prints
 * `serialized_size()` at examples/scaling-as.rs:123-131.  Host only (no context, no device).  The composite types are
 * assembled from these in include/amsm_serialize.hpp.  PARITY UNPINNED (accumulation_amd/csrc/host_serialize.h).
 *   field element : canonical integer, little-endian, 32 bytes (Fr of both curves)
 *   point         : compressed = x with 2 flag bits in the top of the last byte (bit 7: y is the larger root, bit 6:
 *                   infinity) -- 33 bytes (Pallas) / 48 (BLS12-381 G1); uncompressed = x | y+flags -- 65 / 96 bytes.
 * Deserialisation returns AMSM_E_INVALID_ARG for a non-canonical integer, an x without a point, a point off the curve or
 * outside the prime-order subgroup, or both flag bits set. */
size_t amsm_fr_serialized_size(int curve);
size_t amsm_point_serialized_size(int curve, int compressed);
int amsm_fr_serialize(int curve, const uint64_t* a_mont, size_t n, uint8_t* out);
int amsm_fr_deserialize(int curve, const uint8_t* in, size_t n, uint64_t* out_mont);
int amsm_points_serialize(int curve, const uint64_t* xy_mont, const uint8_t* is_inf, size_t n, int compressed, uint8_t* out);
int amsm_points_deserialize(int curve, const uint8_t* in, size_t n, int compressed, uint64_t* xy_mont, uint8_t* is_inf);

/* ---- Poseidon sponge over the curve's base field (ark-sponge `PoseidonSponge<ConstraintF<G>>`, ext) ------------------------
 * The `S` every scheme test of the reference instantiates (src/hp_as/mod.rs:1047-1055, src/r1cs_nark_as/mod.rs:1279-1287,
 * examples/scaling-as.rs:27,36); call sites of absorb / squeeze: src/hp_as/mod.rs:233-275,753-780, src/r1cs_nark_as/mod.rs:
 * 423-448, src/r1cs_nark_as/r1cs_nark/mod.rs:49-72, src/ipa_pc_as/mod.rs:254-388.  Host only: O(#inputs) hashing, SURVEY.md
 * section 8(f) rank 2.  Parameters and encodings are stated in accumulation_amd/csrc/host_poseidon.h (PARITY UNPINNED: ark-sponge
 * @ branch `accumulation-experimental` is not in /root/reference).  include/amsm_poseidon.hpp wraps this as the `Sponge`
 * template argument of the scheme drivers. */
typedef struct amsm_sponge amsm_sponge;
int amsm_poseidon_new(int curve, amsm_sponge** out);                       /* PoseidonSponge::new() */
int amsm_poseidon_clone(const amsm_sponge* s, amsm_sponge** out);
void amsm_poseidon_free(amsm_sponge* s);
/* `fork(domain)`: a clone that absorbed (domain.len() as u64 LE || domain) as a byte string. */
int amsm_poseidon_fork(const amsm_sponge* s, const uint8_t* domain, size_t n, amsm_sponge** out);
/* absorb: native elements (base field, Montgomery); one usize / bool / Option tag; a byte string (31-byte LE chunks for
 * Pallas, 47 for BLS12-381, one element each); affine points (x, y, infinity each; a flagged identity absorbs as 0, 1, 1 whatever
 * xy_mont holds: ark-ec ^0.2.0's `GroupAffine::zero()`). */
int amsm_poseidon_absorb_native(amsm_sponge* s, const uint64_t* fq_mont, size_t n);
int amsm_poseidon_absorb_u64(amsm_sponge* s, uint64_t v);
int amsm_poseidon_absorb_bytes(amsm_sponge* s, const uint8_t* bytes, size_t n);
int amsm_poseidon_absorb_points(amsm_sponge* s, const uint64_t* xy_mont, const uint8_t* is_inf, size_t n);
int amsm_poseidon_squeeze_native(amsm_sponge* s, size_t n, uint64_t* out_fq_mont);
/* n_bits bits, little-endian, packed into ceil(n_bits / 8) bytes. */
int amsm_poseidon_squeeze_bits(amsm_sponge* s, size_t n_bits, uint8_t* out_bytes);
/* `squeeze_nonnative_field_elements_with_sizes(&[Truncated(n_bits); count])` -> count canonical integers < 2^n_bits
 * (4 u64 each); n_bits <= 254.  ONE squeeze for the whole batch (the windows are consecutive bits of it). */
int amsm_poseidon_squeeze_nonnative(amsm_sponge* s, unsigned n_bits, size_t count, uint64_t* out_canonical);
/* The bare permutation on 3 elements (Montgomery), and the 39 * 3 round constants, for known-answer tests. */
int amsm_poseidon_permute(int curve, uint64_t* state_mont);
int amsm_poseidon_round_constants(int curve, uint64_t* out_mont);

/* ---- inner-product-argument opening (ark_poly_commit::ipa_pc, ext) ------------ */
/* Committer key living in device memory (the folded keys of the IPA rounds change every round, so they are
 * never precomputed): wraps a COPY of n affine points (Montgomery x|y, (0,0) = identity) at d_xy. */
int amsm_bases_from_device(amsm_ctx* ctx, const void* d_xy_mont, size_t n, unsigned flags, amsm_bases** out);
/* Device pointer to generators [0, len) of a key (affine, Montgomery), valid while the key lives. */
const void* amsm_bases_device_ptr(const amsm_bases* bases);
/* d_out[i] = d_l[i] + x * d_r[i], i < n, affine in / affine out: the key fold `key_l += key_r * xi` of
 * `open_individual_opening_challenges` (ext; under src/ipa_pc_as/mod.rs:454).  x_mont: 4 u64; nbits = number of
 * significant bits of the canonical x (128 for the truncated round challenges, 255 in general). */
int amsm_points_fold(amsm_ctx* ctx, const void* d_l, const void* d_r, size_t n, const uint64_t* x_mont,
                     unsigned nbits, void* d_out);
/* The same fold from key to key, with no detour through caller-visible buffers: *out = a new (never precomputed) key of
 * n_half generators, out[i] = key[i] + x * key[n_half + i].  Stream-ordered on the context's stream. */
int amsm_bases_fold(amsm_ctx* ctx, const amsm_bases* key, size_t n_half, const uint64_t* x_mont, unsigned nbits,
                    amsm_bases** out);
/* out_mont = sum_i a[i]*b[i]  (`inner_product` of the IPA rounds; polynomial evaluation as <coeffs, powers>). */
int amsm_vec_inner_product(amsm_ctx* ctx, const void* d_a, const void* d_b, size_t n, uint64_t* out_mont);
/* d_out[i] = point^i, i < n  (the evaluation vector z of the opening). */
int amsm_vec_powers(amsm_ctx* ctx, const uint64_t* point_mont, size_t n, void* d_out);
/* Scalars of the two cross commitments L_j = <c_r, key_l>, R_j = <c_l, key_r> of opening round j (0-based) expressed
 * over the ORIGINAL key of n = 2^log_n generators, so that no key is folded between rounds: with xi_0..xi_{j-1} the
 * previous rounds' challenges (key_l += xi * key_r each round) and d_coeffs the current coefficient vector of
 * length n / 2^j, d_out_l / d_out_r (n elements each, half of them zero) satisfy
 * L_j = msm(key, d_out_l), R_j = msm(key, d_out_r).  The final folded key is msm(key, amsm_ipa_check_poly_coeffs(xi)).
 * Same points as the reference's round-by-round folding (ext, under src/ipa_pc_as/mod.rs:454), n MSM pairs per
 * round on the precomputed key instead of n / 2^j 128-bit scalar multiplications with an inversion each.
 * d_out_r may be NULL: d_out_l then receives the ONE vector u with L_j = sum over k with bit (log_n-1-j) clear of
 * u[k] key[k] and R_j = the sum over the others, which amsm_msm_grouped_device(group_shift = log_n-1-j) computes in
 * one pass. */
int amsm_ipa_round_scalars(amsm_ctx* ctx, const uint64_t* xi_mont, size_t j, size_t log_n, const void* d_coeffs,
                           void* d_out_l, void* d_out_r);
/* One whole opening round with one synchronisation (ext, under src/ipa_pc_as/mod.rs:454): the scalars of round j over
 * `key` (2^log_key generators; amsm_ipa_round_scalars with d_out_r = NULL into the scratch d_u, 2^log_key scalars), the
 * grouped MSM (L_j, R_j without their h' terms -> out_lr_xy[2][2L], out_lr_inf[2]) and the two inner products
 * out_ip_mont[0] = <c_r, z_l>, out_ip_mont[1] = <c_l, z_r> of the current coefficient / evaluation vectors (length
 * 2^(log_key - j) each), which run beside the MSM's tail.  xi_mont: the j challenges since `key` was formed. */
int amsm_ipa_round(amsm_ctx* ctx, const amsm_bases* key, const uint64_t* xi_mont, size_t j, size_t log_key,
                   const void* d_coeffs, const void* d_z, void* d_u, uint64_t* out_lr_xy, uint8_t* out_lr_inf,
                   uint64_t* out_ip_mont);
/* amsm_ipa_round with the round's host algebra behind the same call (the opening loop of ext, under
 * src/ipa_pc_as/mod.rs:454, per round: `l = cm_commit(key_l, c_r) + h' <c_r, z_l>`, `r = ...`, then
 * `c_l += x^-1 c_r; z_l += x z_r`):
 *  - fold_x_mont != NULL: the PREVIOUS round's challenge x; d_coeffs / d_z still hold that round's vectors
 *    (2^(log_key - j + 1) elements) and are folded in place first, c[i] += x^-1 c[cur + i], z[i] += x z[cur + i],
 *    cur = 2^(log_key - j); the round then runs on their first cur elements.  x = 0 is AMSM_E_INVALID_ARG.
 *  - h_prime_xy != NULL (affine, Montgomery, finite): out_lr = L_j + <c_r, z_l> h', R_j + <c_l, z_r> h' -- the two
 *    multiples are computed on the host while the MSM's tail runs, added in extended coordinates and L, R normalised
 *    with ONE inversion (amsm_ipa_round + two amsm_host_lincomb calls: three).
 * With both NULL it is amsm_ipa_round.  The driver's loop is one call per round plus the round's challenge.
 * Errors: argument errors and AMSM_E_OOM (every allocation of the round is made first) leave d_coeffs / d_z untouched, so the
 * call can be retried; any later error (AMSM_E_SCALAR_RANGE, AMSM_E_HIP) returns with the fold ALREADY applied -- the
 * vectors are consumed, do not retry with fold_x_mont set. */
int amsm_ipa_round_fused(amsm_ctx* ctx, const amsm_bases* key, const uint64_t* xi_mont, size_t j, size_t log_key,
                         void* d_coeffs, void* d_z, const uint64_t* fold_x_mont, const uint64_t* h_prime_xy, void* d_u,
                         uint64_t* out_lr_xy, uint8_t* out_lr_inf, uint64_t* out_ip_mont);
/* d_out[p] (p < 2^k) = coefficients of prod_{i=1..k} (1 + xi_i X^(2^(k-i))):
 * `SuccinctCheckPolynomial::compute_coeffs` (ext), call sites src/ipa_pc_as/mod.rs:400 and under :836. k <= 32. */
int amsm_ipa_check_poly_coeffs(amsm_ctx* ctx, const uint64_t* xi_mont, size_t k, void* d_out);
/* Round 6 -- the JUMP FOLD of an opening that never folded its key: after j rounds (challenges xi_mont[0 .. j), the first one pairs the
 * two halves of the key) the reference holds the folded key `key_l[i] += key_r[i] * xi` applied j times (ark_poly_commit::ipa_pc ext,
 * under src/ipa_pc_as/mod.rs:454): m0 = 2^(log_key - j) generators B_k = sum_t S_t G_(t m0 + k), S_t = the product of the challenges
 * the bits of t pick.  This call computes all m0 of them from the key's window table in one pass (m0 MSMs of 2^j pairs that share
 * one scalar vector: one digit sort on the host, one accumulation launch whose waves read 64 outputs' points as one contiguous
 * table row, the ordinary weighted bucket reduction) so that the driver can run the LAST log2(m0) rounds -- latency chains of
 * ~0.36 ms each on the device whatever their logical size -- on the host over those few generators, and take the final folded key
 * from there instead of from one more full-size MSM.  out_xy_mont: m0 * 2 * limbs u64.
 * AMSM_E_UNSUPPORTED when the key does not qualify (a single-device precomputed key with equal-width windows covering 256 bits --
 * the tables of keys of up to 2^16 and of 2^18 / 2^19 generators --, m0 a multiple of 64; the host backend takes any key): the
 * caller then keeps running its rounds through amsm_ipa_round_fused.  Proofs do not depend on the choice. */
int amsm_ipa_jump_fold(amsm_ctx* ctx, const amsm_bases* key, size_t log_key, const uint64_t* xi_mont, size_t j,
                       uint64_t* out_xy_mont, uint8_t* out_is_inf);

/* ---- R1CS matrices (row-sparse) for the NARK prover -------------------------------------------- */
typedef struct amsm_matrix amsm_matrix;
/* Replaces `Matrix<F> = Vec<Vec<(F, usize)>>` of `IndexProverKey{a,b,c}`
 * (src/r1cs_nark_as/r1cs_nark/data_structures.rs:33-48), flattened to CSR by the adapter:
 * row_ptr (n_rows+1), col_idx (nnz), vals_mont (nnz*4 u64, Montgomery).  Copied to HBM once per index. */
int amsm_matrix_load(amsm_ctx* ctx, const uint32_t* row_ptr, const uint32_t* col_idx, const uint64_t* vals_mont,
                     size_t n_rows, size_t nnz, amsm_matrix** out);
size_t amsm_matrix_rows(const amsm_matrix* m);
void amsm_matrix_free(amsm_matrix* m);
/* out[r] = sum_k val[k] * (input || witness)[col[k]]  -- `matrix_vec_mul(matrix, input, witness)`,
 * src/r1cs_nark_as/r1cs_nark/mod.rs:443-462; call sites :183-185,194-196, src/r1cs_nark_as/mod.rs:329-339.
 * d_input / d_witness / d_out are device vectors of Montgomery Fr elements; indices beyond
 * n_input + n_witness read as zero. */
int amsm_matrix_vec_mul(amsm_ctx* ctx, const amsm_matrix* m, const void* d_input, size_t n_input,
                        const void* d_witness, size_t n_witness, void* d_out);

#ifdef __cplusplus
}
#endif
#endif /* AMSM_H */
