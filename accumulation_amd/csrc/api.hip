// C ABI implementation (include/amsm.h): contexts, resident committer keys, the MSM pipeline launcher,
// host-side finalisation.  One context = one GPU + one HIP stream + a grow-only HBM workspace.
#include "../../include/amsm.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <dlfcn.h>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <pthread.h>
#include <thread>
#include <unordered_map>
#include <rocprim/rocprim.hpp>
#include <vector>

#include "fp.h"
#include "host_field.h"
#include "host_poseidon.h"
#include "host_serialize.h"
#include "launch.h"

using namespace amsm;

namespace {

enum Stage { ST_DIGITS = 0, ST_SORT, ST_BOUNDS, ST_ACCUM_L0, ST_ACCUM_L12, ST_REDUCE, ST_COUNT };
// Elapsed device time between the stage's first and last kernel on the stream the stage runs on (hipEvent pairs).  The first
// three are the prep stream: with the short prep chain everything is in "prep_chain" (the other two only have work in the
// rocPRIM fallback).  Inside a batch these are the times the kernels take WHILE SHARING the GPU with the other MSMs in flight
// (prep_chain stretches to about one accumulate-L0 duration); a blocking call gives the stand-alone times.
const char* kStageNames[ST_COUNT] = {"prep_chain", "prep_sort_rocprim", "prep_bounds_rocprim", "accum_l0", "accum_l1_l2",
                                     "bucket_reduce_fold"};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

// One pipeline slot = one stream + one private workspace, so two MSMs of a batch can be in flight:
// the latency-bound tail of MSM i (fold partials, bucket reduce) overlaps the throughput-bound head of
// MSM i+1 on the other slot's stream.
constexpr int N_SLOTS = 3;  // MSMs of one batch in flight

struct Slot {  // buffers and events of one MSM in flight (the streams belong to the context: one per pipeline stage)
  hipEvent_t l0_done = nullptr, prep_done = nullptr;
  hipEvent_t ev[ST_COUNT + 1] = {};  // stage begins (each on the stream its stage runs on)
  hipEvent_t ev_prep_end = nullptr, ev_l0_end = nullptr;  // ends of the stages whose successor starts on ANOTHER stream
  hipEvent_t done = nullptr;
  DevBuf keys_a, keys_b, vals_a, vals_b, start, items, item_off, partials, buckets, red_out, fold_out, heavy, misc,
      sort_tmp, scan_tmp, prep_small, heavy_scratch;
  void* h_pinned = nullptr;
  size_t h_pinned_bytes = 0;
  MsmGeom geom = {};
  bool busy = false;
  hipStream_t tail = nullptr;  // the stream this slot's tail (and its result copy) was queued on
};

struct ShardWorker {  // one persistent host thread per non-primary shard
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  bool has_job = false, stop = false, idle = true;
  ShardWorker() {
    th = std::thread([this] {
      for (;;) {
        std::function<void()> j;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return has_job || stop; });
          if (stop) return;
          j = std::move(job);
          has_job = false;
        }
        j();
        {
          std::lock_guard<std::mutex> lk(mu);
          idle = true;
        }
        cv.notify_all();
      }
    });
  }
  void submit(std::function<void()> j) {
    {
      std::lock_guard<std::mutex> lk(mu);
      job = std::move(j);
      has_job = true;
      idle = false;
    }
    cv.notify_all();
  }
  void wait() {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return idle; });
  }
  ~ShardWorker() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};

// Host threads for the schemes' host-side group algebra (amsm_host_lincomb[_batch]: rows a11 of the scope table -- the
// blinded commitments, the beta-combinations, the IPA verifier's 2 log n + 2 point combination are 50-500 us each and come
// in independent groups).  A small persistent pool; the caller works too.  AMSM_HOST_THREADS=0 disables it (default: up
// to 3 helpers).  One parallel region at a time: a second caller that finds the pool busy runs its tasks itself.
struct HostPool {
  std::vector<std::unique_ptr<ShardWorker>> workers;
  std::mutex busy;
  HostPool() {
    int want = 3;
    if (const char* e = getenv("AMSM_HOST_THREADS")) want = atoi(e);
    const int hw = (int)std::thread::hardware_concurrency();
    want = std::max(0, std::min(want, std::min(15, hw > 1 ? hw - 1 : 0)));
    for (int i = 0; i < want; i++) workers.emplace_back(new ShardWorker());
    // a fork()ed child (Python multiprocessing, a host that forks verifier workers) inherits this object but none of its
    // threads: the child drops the workers WITHOUT joining them and runs every region on the calling thread
    pthread_atfork(nullptr, nullptr, [] {
      HostPool& p = HostPool::get();
      for (auto& w : p.workers) (void)w.release();
      p.workers.clear();
      new (&p.busy) std::mutex();  // the parent may have held it at the fork
    });
  }
  static HostPool& get() {
    static HostPool pool;
    return pool;
  }
  // fn(i) for i in [0, n), each exactly once
  template <class F>
  void run(size_t n, F&& fn) {
    std::unique_lock<std::mutex> lk(busy, std::try_to_lock);
    if (!lk.owns_lock() || workers.empty() || n < 2) {
      for (size_t i = 0; i < n; i++) fn(i);
      return;
    }
    std::atomic<size_t> next{0};
    auto loop = [&] {
      for (size_t i; (i = next.fetch_add(1, std::memory_order_relaxed)) < n;) fn(i);
    };
    const size_t helpers = std::min(workers.size(), n - 1);
    for (size_t w = 0; w < helpers; w++) workers[w]->submit(loop);
    loop();
    for (size_t w = 0; w < helpers; w++) workers[w]->wait();
  }
};

struct amsm_ctx {
  int curve = 0;
  int device = 0;
  // One stream per pipeline STAGE, shared by all MSMs in flight (in-order per stage, so consecutive MSMs pipeline:
  // prep(k+1) and tail(k-1) run beside accumulate L0 of MSM k).  One stream per MSM instead made the overlap depend on
  // which hardware queues the runtime happened to map the streams to (measured 610-700 Mpairs/s for the same code).
  hipStream_t stream = nullptr;  // main: the caller's stream -- accumulate L0 and every non-MSM kernel
  hipStream_t s_prep = nullptr;  // digits, sort, bounds, scan (memory-bound)
  hipStream_t s_tail = nullptr;  // fold partials, bucket reduce, fold, D2H (latency-bound)
  bool own_stream = false;
  bool custom_prep = true;  // AMSM_PREP=rocprim: digits + rocPRIM radix sort + bounds + rocPRIM scan instead (A/B, fallback)
  int window_override = 0;
  int K0 = 0;          // 0 = automatic (see make_geom)
  int cu_count = 256;
  // AMSM_L0_SPREAD=1: small launches as ONE round at 1-2 workgroups per CU (residency capped with unused LDS).  Built on the
  // theory that the dispatcher packs a CU to the kernel's occupancy before moving on; measured in round 2 it changes nothing
  // (2^16: 0.460 vs 0.447 ms per blocking call) -- the small-launch time was the per-flush bucket search.  Off.
  bool small_spread = false;
  u32 l0_lds_pad = 0;  // AMSM_L0_LDS_PAD: dynamic LDS bytes per accumulate-L0 workgroup that only cap its residency
  int K0_max = 32;     // automatic choice: largest chunk (AMSM_K0_MAX); round 2, batches of 2^20-pair MSMs: 32 -> 811-816, 24 -> 805 Mpairs/s
  bool two_phase = true;  // automatic choice: two chunk sizes so that the grid is a whole number of rounds (AMSM_K0_2PHASE=0: A/B)
  int wave_slots = 4096;  // resident accumulate-L0 waves: CUs x resident 256-lane blocks per CU x 4
  int K1 = 1024;  // buckets with more partials than this go to the workgroup-per-bucket path (extreme skew only)
  int red_s = 4;
  int split_log2 = 21;      // MSMs of 2^split_min_log2 pairs and more over a precomputed key run as windows of 2^split_log2
  int split_min_log2 = 22;  // generators (msm_multi_split_xyzz; AMSM_SPLIT_LOG2=0 disables, AMSM_SPLIT_MIN_LOG2)
  bool one_stream = true;  // a lone blocking MSM runs its whole chain on the caller's stream (AMSM_ONE_STREAM=0: per-stage streams, A/B)
  bool tail_quad = true;  // bucket reduce / fold with a quad of lanes per logical lane (AMSM_TAIL_QUAD=0: one lane, A/B)
  bool profiling = false;
  float stage_ms[ST_COUNT] = {};  // mean over the MSMs of the last call
  float stage_acc[ST_COUNT] = {};
  int stage_n = 0;
  Slot slot[N_SLOTS];
  hipEvent_t fork = nullptr;
  hipEvent_t ip_ready = nullptr;  // amsm_ipa_round_fused: the inner products have reached the host
  DevBuf scalars;
  DevBuf xyzz_scratch;  // unconverted sums of large key folds / precompute levels (launch.h: batch_affine_pays)
  // ---- multi-device (amsm_ctx_create_multi) ----
  // shard_ctx[0] == this (the primary); shard_ctx[g >= 1] are owned single-device contexts, each served by one host
  // worker thread so that the blocking single-device pipeline runs on all devices at once.
  std::vector<amsm_ctx*> shard_ctx;
  std::vector<ShardWorker*> workers;  // workers[g - 1] drives shard_ctx[g]
  amsm_ctx* parent = nullptr;
  int collective = 0;            // 0 none, 1 RCCL all-gather, 2 peer copies
  void** rccl_comms = nullptr;   // ncclComm_t per shard
  DevBuf rec_send, rec_recv, stage;  // per device: this shard's partial records / the gathered ones / scalar slices
  hipEvent_t multi_fork = nullptr;
  // ---- caching allocator behind amsm_dev_alloc / amsm_dev_free ----
  // hipMalloc / hipFree synchronise the device: a scheme driver that allocates its vectors per call (every `Vec<F>` the
  // reference builds) would serialise the GPU on each one.  Freed buffers go to size-keyed free lists and are handed out
  // again; reuse is safe in stream order because every kernel that touches them runs on this context's streams and the MSM
  // calls that read them from the prep stream are blocking.  amsm_ctx_trim releases everything.
  std::unordered_map<size_t, std::vector<void*>> pool;    // rounded size -> free buffers
  std::unordered_map<void*, size_t> pool_size;            // every live or pooled buffer -> its rounded size
  size_t pool_free_bytes = 0, pool_live_bytes = 0;
  size_t pool_cap_bytes = (size_t)16 << 30;               // free-list budget (AMSM_POOL_MAX_MB); beyond it frees are real
};

struct amsm_bases {
  int curve = 0;
  int device = 0;
  size_t n = 0;
  int precomp = 0;
  int c = 0;  // window bits fixed at creation when precomputed
  int W = 0;
  u32* d_table = nullptr;          // device-internal Montgomery radix (launch.h: device_internal_radix)
  // C-ABI-radix copy of generators [0, n), made on the first amsm_bases_device_ptr (under abi_mu: the handle is
  // shareable between threads)
  mutable u32* d_abi = nullptr;
  mutable std::mutex abi_mu;
  // recorded by amsm_bases_fold behind the kernel that writes d_table (it returns without synchronising): consumers that
  // are not ordered behind the folding context's stream (amsm_bases_device_ptr) wait for it
  hipEvent_t ready = nullptr;
  // sharded key of a multi-device context: shard g (a single-device key on shard_ctx[g]'s device) holds generators
  // [bound[g], bound[g + 1]); n is the total, d_table stays null
  std::vector<amsm_bases*> shards;
  std::vector<size_t> bound;
  const amsm_ctx* owner = nullptr;
};

struct amsm_sponge {  // host-side Poseidon sponge over the curve's base field (host_poseidon.h)
  int curve = 0;
  host::PoseidonSponge<PallasFq> pallas;
  host::PoseidonSponge<Bls12381Fq> bls;
};

struct amsm_matrix {
  int curve = 0;
  int device = 0;
  size_t n_rows = 0, nnz = 0;
  u32* d_row_ptr = nullptr;
  u32* d_col = nullptr;
  u32* d_val = nullptr;
};

namespace {

#define HIP_TRY(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      fprintf(stderr, "[amsm] HIP error %s (%d) at %s:%d: %s\n", hipGetErrorName(_e), (int)_e, __FILE__, \
              __LINE__, #expr);                                                                        \
      return _e == hipErrorOutOfMemory ? AMSM_E_OOM : AMSM_E_HIP;                                      \
    }                                                                                                  \
  } while (0)

#define TRY(expr)            \
  do {                       \
    int _s = (expr);         \
    if (_s != AMSM_OK) return _s; \
  } while (0)

int ensure(DevBuf& b, size_t bytes) {
  if (b.bytes >= bytes && b.p) return AMSM_OK;
  if (b.p) HIP_TRY(hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  size_t want = bytes + bytes / 8 + 256;
  HIP_TRY(hipMalloc(&b.p, want));
  b.bytes = want;
  return AMSM_OK;
}

int ensure_pinned(Slot* sl, size_t bytes) {
  if (sl->h_pinned_bytes >= bytes) return AMSM_OK;
  if (sl->h_pinned) HIP_TRY(hipHostFree(sl->h_pinned));
  sl->h_pinned = nullptr;
  HIP_TRY(hipHostMalloc(&sl->h_pinned, bytes + 4096, hipHostMallocDefault));
  sl->h_pinned_bytes = bytes + 4096;
  return AMSM_OK;
}

inline u32 cdiv(u32 a, u32 b) { return (a + b - 1) / b; }
inline int ilog2_ceil(size_t n) {
  int l = 0;
  while (((size_t)1 << l) < n) l++;
  return l;
}

// Window width.  Measured on MI355X (tools/sweep_window.py, Pallas, batches of MSMs), not derived: a width whose TOP
// window holds only 2-3 scalar bits (255 mod c small: c = 9, 11, 12, 14, 18, 19) concentrates 2^-3 of all entries of
// that window in a handful of buckets, which then take the heavy-bucket path; c = 8, 10, 13, 15, 16 do not.
//   precomputed key (all windows share one bucket set): 2^10-2^12 -> 8, 2^13-2^14 -> 10, 2^15 -> 13, 2^16 -> 15 (round 2: 16),
//   >= 2^17 -> 16 (2^18: 518 vs 361 Mpairs/s at the old lg-4 rule; 2^22: 774 vs 440).
//   Round 2: c = 17 from 2^20 up (and at 2^17).  15 windows of 17 bits cover the 255-bit scalars, so the 16th holds only the
//   recoding's carry -- never set for Pallas (r < 2^254 + 2^126), set for the 45 % of BLS12-381 scalars above 2^254, whose
//   entries share ONE bucket (heavy-bucket path): 15 instead of 16 entries per scalar.  Same-process A/B against c = 16
//   (tools/ab_pipeline.py, batches): 2^17 380 vs 370 Mpairs/s, 2^18 562 vs 585, 2^19 757 vs 767, 2^20 862 vs 838, 2^21 870 vs 824,
//   2^22 854 vs 829; BLS12-381 2^20 379 vs 372.  (c = 18 / 19 need more than the 31 bits of the prep's entry word.)
//   plain key (one bucket set per window): lg - 6 moved to the nearest width of that list.
int choose_window(size_t n, bool precomp) {
  int lg = ilog2_ceil(n < 2 ? 2 : n);
  if (precomp) {
    if (lg >= 20 || lg == 17) return 17;
    if (lg >= 16) return 16;  // 2^16: 16 (207 Mpairs/s in batches; 13: 210, 15: 193; a blocking IPA round is within 1 % for all three)
    if (lg == 15) return 13;
    if (lg >= 13) return 10;
    return 8;
  }
  int c = lg - 6;
  if (c < 4) c = 4;
  if (c > 16) c = 16;
  static const int good[13] = {4, 5, 6, 7, 8, 8, 10, 10, 10, 13, 13, 15, 16};  // index c - 4
  return good[c - 4];
}
inline int windows_for(int c) { return 255 / c + 1; }  // W*c >= 256 (signed digits need one spare bit)
inline int slots_for(int c) { return windows_for(c); }   // entry slots per scalar

template <class Fq>
constexpr size_t affine_bytes() {
  return 2 * Fq::L * 4;
}
template <class Fq>
constexpr size_t xyzz_bytes() {
  return 4 * Fq::L * 4;
}

// generator coordinates in Montgomery form (computed from canonical constants at first use)
template <class Fq>
std::vector<u32> generator_mont(int curve) {
  using H = host::HFe<Fq>;
  H gx = host::h_zero<Fq>(), gy = host::h_zero<Fq>();
  if (curve == AMSM_PALLAS) {
    // (-1, 2)
    H one = host::h_zero<Fq>();
    one.v[0] = 1;
    H two = host::h_zero<Fq>();
    two.v[0] = 2;
    gx = host::h_neg<Fq>(host::h_to_mont<Fq>(one));
    gy = host::h_to_mont<Fq>(two);
  } else {
    static const u64 X[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull,
                             0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};
    static const u64 Y[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull,
                             0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};
    for (int i = 0; i < H::N && i < 6; i++) {
      gx.v[i] = X[i];
      gy.v[i] = Y[i];
    }
    gx = host::h_to_mont<Fq>(gx);
    gy = host::h_to_mont<Fq>(gy);
  }
  std::vector<u32> g(2 * Fq::L);
  memcpy(g.data(), gx.v, 4 * Fq::L);
  memcpy(g.data() + Fq::L, gy.v, 4 * Fq::L);
  return g;
}

// ---------------------------------------------------------------------------------------------
// The MSM pipeline.  Leaves n_sets folded XYZZ records in ctx->fold_out.
// ---------------------------------------------------------------------------------------------
struct RunInfo {
  MsmGeom g;
};

int make_geom(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, size_t n, MsmGeom* out, int group_shift = -1) {
  MsmGeom g;
  memset(&g, 0, sizeof(g));
  int c, W;
  if (bases->precomp) {
    c = bases->c;
    W = bases->W;
  } else {
    c = ctx->window_override ? ctx->window_override : choose_window(n, false);
    W = windows_for(c);
  }
  if (c < 2 || c > 24) return AMSM_E_INVALID_ARG;
  g.n = (u32)n;
  g.c = (u32)c;
  g.W = (u32)W;
  g.nb = 1u << (c - 1);
  g.groups = group_shift >= 0 ? 2u : 1u;
  g.group_shift = group_shift >= 0 ? (u32)group_shift : 0u;
  g.n_sets = g.groups * (bases->precomp ? 1u : (u32)W);
  g.B = g.n_sets * g.nb;
  g.S = (u32)slots_for(c);
  if ((unsigned long long)n * g.S >= (1ull << 30)) return AMSM_E_UNSUPPORTED;  // entry words carry a 30-bit index
  g.E = (u32)(n * g.S);
  g.base_off = (u32)base_off;
  g.table_stride = (u32)bases->n;
  g.precomp = (u32)bases->precomp;
  if ((unsigned long long)bases->n * (bases->precomp ? W : 1) >= (1ull << 30)) return AMSM_E_UNSUPPORTED;
  // chunk length of accumulate L0: the grid should be a whole number of rounds of the resident wave
  // slots (queried from the kernel's occupancy), so that no SIMD idles while a partial last round drains
  const unsigned long long lanes_per_cu_block = 256ull * (unsigned long long)std::max(1, ctx->cu_count);
  if (ctx->K0 > 0) {
    g.K0 = ((u32)ctx->K0 + 3u) & ~3u;  // accumulate L0 reads entries in groups of 4
    g.K0b = g.K0;
    g.nA = 0xffffff00u;
  } else if (ctx->small_spread && (unsigned long long)g.E <= lanes_per_cu_block * 2ull * 24ull) {
    // Small MSMs (an IPA round at d + 1 = 2^16: 1.1 M entries): fewer workgroups than the GPU holds.  The dispatcher fills a
    // CU to the kernel's occupancy (3 workgroups) before it moves on, so 384 workgroups ran on 128 of the 256 CUs, three
    // waves per SIMD sharing one multiplier: 144 us for 47 us of dependent work per lane (rocprofv3 timeline, round 2).
    // ONE round at 1 (or 2) workgroups per CU instead: the residency is capped with unused dynamic LDS, and the chunk is
    // the lane's whole share.
    const unsigned long long cap = (unsigned long long)g.E <= lanes_per_cu_block * 24ull ? 1ull : 2ull;
    const unsigned long long share = (g.E + lanes_per_cu_block * cap - 1) / (lanes_per_cu_block * cap);
    g.K0 = (u32)std::max<unsigned long long>(4ull, (share + 3ull) & ~3ull);
    g.K0b = g.K0;
    g.nA = 0xffffff00u;
    g.l0_per_cu = (u32)cap;
  } else {
    // <= K0_max entries per lane per round of resident waves (24 measured best on MI355X for batches of MSMs,
    // tools/ab_pipeline.py: shorter chunks let the other MSMs' prep / tail kernels in sooner, longer ones save partials),
    // and the grid a WHOLE number of rounds: every resident lane gets `rounds` chunks whose sizes (multiples of 4: the
    // entries are read 16 bytes at a time) add up to its share of the list -- ra rounds of kb + 4 entries, then
    // rounds - ra of kb.  One size for all rounds left the last round of a 2^20-pair launch 56 % full.
    const unsigned long long lanes = (unsigned long long)ctx->wave_slots * 64ull;
    const unsigned long long kmax = (unsigned long long)std::max(4, ctx->K0_max);
    const unsigned long long share = (g.E + lanes - 1) / lanes;  // entries per resident lane
    const unsigned long long rounds = std::max<unsigned long long>(1, (share + kmax - 1) / kmax);
    unsigned long long kb = (share / rounds) & ~3ull;
    if (!ctx->two_phase || kb < 12ull) {  // small problems: one size, at least 12 (2^16: 12 -> 0.57 ms per blocking call, 16 -> 0.59)
      unsigned long long k = (share + rounds - 1) / rounds;
      g.K0 = ((u32)std::min<unsigned long long>(std::max<unsigned long long>(k, 12ull), kmax) + 3u) & ~3u;
      g.K0b = g.K0;
      g.nA = 0xffffff00u;
    } else {
      const unsigned long long ra = (share - rounds * kb + 3ull) / 4ull;  // rounds that take 4 more entries (<= rounds)
      g.K0 = (u32)(kb + 4ull);
      g.K0b = (u32)kb;
      g.nA = (u32)(ra * lanes);  // lanes = 64 * wave_slots, wave_slots a multiple of 4: a multiple of 256
      if (ra >= rounds) {
        g.K0b = g.K0;
        g.nA = 0xffffff00u;
      }
    }
  }
  {
    unsigned long long t0 = (unsigned long long)g.nA * g.K0;
    if (t0 >= g.E) {  // phase A covers everything
      g.nA = (g.E + g.K0 - 1) / g.K0;
      g.nA = (g.nA + 255u) & ~255u;
      g.T0 = g.nA * g.K0;  // < E + 256 * K0: fits
      g.n_chunks = (g.E + g.K0 - 1) / g.K0;
    } else {
      g.T0 = (u32)t0;
      g.n_chunks = g.nA + (g.E - g.T0 + g.K0b - 1) / g.K0b;
    }
  }
  g.K1 = (u32)ctx->K1;
  g.red_s = std::min<u32>((u32)ctx->red_s, g.nb);
  g.red_threads = g.nb / g.red_s;
  *out = g;
  return AMSM_OK;
}

void stage_mark(amsm_ctx* ctx, Slot* sl, int idx, hipStream_t st) {
  if (ctx->profiling) (void)hipEventRecord(sl->ev[idx], st);
}

// lanes cooperating on one bucket in accumulate L1 (tree over partials): more lanes = lower latency,
// but only worth it when buckets have several partials each
// `hidden`: the tail runs behind the next MSM's accumulation (inside a batch): what counts is its ALU work, not its
// depth -- one lane per bucket does no butterfly additions at all when there are buckets enough to fill the lanes.
// Measured (round 2, 2^20 Pallas, stand-alone kernel): 16 lanes per bucket 0.23 ms, 4 lanes 0.107 ms for ~21-26 partials per
// bucket -- the wide tree only pays when there are too few buckets to occupy the chip with 4 lanes each.
u32 l1_lanes(const MsmGeom& g, bool hidden) {
  double avg = (double)g.n_chunks / g.B;
  if (hidden && g.B >= 16384u) return avg >= 48.0 ? 4u : 1u;
  if (avg >= 24.0 && g.B * 16ull <= 65536ull) return 16u;
  return avg >= 3.0 ? 4u : 1u;
}

// AMSM_DEBUG=1: synchronise after every stage of the pipeline and report it on stderr (finds the faulting kernel)
#define AMSM_DBG(name)                                    \
  do {                                                    \
    if (getenv("AMSM_DEBUG")) {                           \
      fprintf(stderr, "[amsm] %s ...", name);             \
      hipError_t e_ = hipDeviceSynchronize();             \
      fprintf(stderr, " %s\n", hipGetErrorString(e_));    \
    }                                                     \
  } while (0)

// Work already queued on the caller's stream (e.g. the kernels that produced the scalars) must be
// visible to the prep stream.  Called ONCE per API call, before its MSMs are enqueued -- never between the MSMs of
// a batch: an event recorded on the caller's stream there would sit behind the previous MSM's accumulate L0.
int prep_fork(amsm_ctx* ctx) {
  HIP_TRY(hipEventRecord(ctx->fork, ctx->stream));
  HIP_TRY(hipStreamWaitEvent(ctx->s_prep, ctx->fork, 0));
  return AMSM_OK;
}

// Enqueue the whole pipeline for one MSM on slot `sl` (asynchronous); leaves n_sets folded XYZZ records
// in sl->fold_out and queues their D2H into sl->h_pinned.
template <class Fq, class Fr>
int msm_enqueue(amsm_ctx* ctx, Slot* sl, const amsm_bases* bases, size_t base_off, const void* d_scalars, size_t n,
                int scalars_mont, int group_shift = -1, bool exposed_tail = true, bool alone = true) {
  // exposed_tail: nothing is queued behind this MSM, so the caller waits for its tail (bucket reduce + fold, a chain of
  // dependent point operations on a few waves): run it on the quad-cooperative kernels (-0.08 ms).  Inside a batch the
  // tail is hidden behind the next MSM's accumulation and the one-lane kernels cost less ALU time (measured: 1 % of the
  // batch throughput).
  const bool quad = ctx->tail_quad && exposed_tail;
  MsmGeom g;
  TRY(make_geom(ctx, bases, base_off, n, &g, group_shift));
  sl->geom = g;
  // alone: the only MSM of a blocking call.  Nothing can overlap, and every hand-over between streams (an event wait on
  // another hardware queue) costs ~10 us of idle GPU -- three of them in a chain of 0.4 ms at 2^16 (rocprofv3 timeline of an
  // IPA round, round 2): the whole chain goes on the caller's stream.
  const bool one = ctx->one_stream && alone && exposed_tail;
  hipStream_t st = one ? ctx->stream : ctx->s_prep;  // digits / sort / bounds
  hipStream_t sm = ctx->stream;                      // accumulate L0
  const u32 max_items = g.n_chunks + g.B + 1;
  const u32 red_blocks = cdiv(g.red_threads * (quad ? 4u : 1u), 256);
  TRY(ensure(sl->vals_a, (size_t)g.E * 4 + 64));
  TRY(ensure(sl->vals_b, (size_t)g.E * 4 + 64));  // read in groups of 4 entries
  TRY(ensure(sl->start, (size_t)(g.B + 2) * 4));
  TRY(ensure(sl->items, (size_t)(g.B + 2) * 4));
  TRY(ensure(sl->item_off, (size_t)(g.B + 2) * 4));
  TRY(ensure(sl->partials, (size_t)max_items * xyzz_bytes<Fq>()));
  TRY(ensure(sl->buckets, (size_t)g.B * xyzz_bytes<Fq>()));
  TRY(ensure(sl->red_out, (size_t)g.n_sets * red_blocks * xyzz_bytes<Fq>()));
  TRY(ensure(sl->fold_out, (size_t)g.n_sets * xyzz_bytes<Fq>() + 64));
  TRY(ensure(sl->heavy, (size_t)(g.B + 1) * 4));
  // at most max_items / K1 buckets can hold more than K1 partials each
  TRY(ensure(sl->heavy_scratch, ((size_t)max_items / std::max<u32>(g.K1, 1u) + 2) * accum_l2_slices<Fq>() * xyzz_bytes<Fq>()));
  size_t rec = xyzz_bytes<Fq>();
  TRY(ensure_pinned(sl, g.n_sets * rec + 64));
  // 16 flag words (scalar-range error, heavy-bucket count).  With the short prep chain they lead its block of small
  // arrays, so that ONE fill clears both (a fill is a 5 us dispatch at the head of every MSM's chain).
  // An MSM over a window of a longer precomputed key: the interchange word of the partition pass carries the index relative
  // to the window (MsmGeom::idx_rel_bits), so that its bucket-id bits -- the number of partitions -- follow the MSM's size,
  // not the key's
  if (ctx->custom_prep && g.precomp && g.groups == 1u) {
    u32 b = 1;
    while ((1ull << b) < n) b++;
    unsigned long long abs_max = (unsigned long long)g.base_off + n - 1ull + (unsigned long long)(g.W - 1u) * g.table_stride;
    unsigned long long rel_max = ((unsigned long long)(g.W - 1u) << b) | ((1ull << b) - 1ull);
    if (rel_max < abs_max) {
      MsmGeom g2 = g;
      g2.idx_rel_bits = b;
      if (prep_supported(g2)) sl->geom = g = g2;
    }
  }
  const bool short_prep = ctx->custom_prep && prep_supported(g);
  if (short_prep) TRY(ensure(sl->prep_small, 64 + prep_small_words(g) * sizeof(u32) + 256));
  else TRY(ensure(sl->misc, 64));
  u32* d_err = short_prep ? (u32*)sl->prep_small.p : (u32*)sl->misc.p;
  u32* d_heavy_count = d_err + 1;
  u32* keys_a = (u32*)sl->keys_a.p;
  u32* keys_b = (u32*)sl->keys_b.p;
  u32* vals_a = (u32*)sl->vals_a.p;
  u32* vals_b = (u32*)sl->vals_b.p;

  if (!short_prep) HIP_TRY(hipMemsetAsync(sl->misc.p, 0, 64, st));
  stage_mark(ctx, sl, ST_DIGITS, st);
  if (short_prep) {
    // short prep chain (prep_kernels.h: 5 dispatches + 2 for skewed inputs); the stage marks keep their names: "sort" = scatter + local sort
    PrepBuffers pb;
    pb.d_small = d_err + 16;
    pb.part = vals_a;
    pb.vals_sorted = vals_b;
    pb.start = (u32*)sl->start.p;
    pb.items = (u32*)sl->items.p;
    pb.item_off = (u32*)sl->item_off.p;
    pb.err = d_err;
    if (launch_prep<Fr>(st, (const u32*)d_scalars, scalars_mont, g, pb) != 0) return AMSM_E_HIP;
    stage_mark(ctx, sl, ST_SORT, st);
    stage_mark(ctx, sl, ST_BOUNDS, st);
  } else {
  TRY(ensure(sl->keys_a, (size_t)g.E * 4));  // only the rocPRIM chain sorts (key, value) pairs
  TRY(ensure(sl->keys_b, (size_t)g.E * 4));
  keys_a = (u32*)sl->keys_a.p;
  keys_b = (u32*)sl->keys_b.p;
  const bool keys16 = g.B < 65536u;  // every key (incl. the "digit 0" key B) fits 16 bits
  launch_digits<Fr>(st, (const u32*)d_scalars, scalars_mont, g, keys_a, keys16, vals_a, d_err);
  stage_mark(ctx, sl, ST_SORT, st);
  {
    int bits = 1;
    while ((1u << bits) <= g.B) bits++;
    size_t tmp = 0;
    if (keys16) {
      uint16_t* ka = (uint16_t*)keys_a;
      uint16_t* kb = (uint16_t*)keys_b;
      HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, ka, kb, vals_a, vals_b, (size_t)g.E, 0u, (unsigned)bits, st));
      TRY(ensure(sl->sort_tmp, tmp));
      tmp = sl->sort_tmp.bytes;
      HIP_TRY(rocprim::radix_sort_pairs(sl->sort_tmp.p, tmp, ka, kb, vals_a, vals_b, (size_t)g.E, 0u, (unsigned)bits, st));
    } else {
      HIP_TRY(rocprim::radix_sort_pairs(nullptr, tmp, keys_a, keys_b, vals_a, vals_b, (size_t)g.E, 0u, (unsigned)bits, st));
      TRY(ensure(sl->sort_tmp, tmp));
      tmp = sl->sort_tmp.bytes;
      HIP_TRY(rocprim::radix_sort_pairs(sl->sort_tmp.p, tmp, keys_a, keys_b, vals_a, vals_b, (size_t)g.E, 0u,
                                        (unsigned)bits, st));
    }
  }
  stage_mark(ctx, sl, ST_BOUNDS, st);
  launch_bounds(st, (const void*)keys_b, keys16, vals_b, g, (u32*)sl->start.p, (u32*)sl->items.p);
  {
    size_t tmp = 0;
    HIP_TRY(rocprim::exclusive_scan(nullptr, tmp, (u32*)sl->items.p, (u32*)sl->item_off.p, 0u, (size_t)(g.B + 1),
                                    rocprim::plus<u32>(), st));
    TRY(ensure(sl->scan_tmp, tmp));
    tmp = sl->scan_tmp.bytes;
    HIP_TRY(rocprim::exclusive_scan(sl->scan_tmp.p, tmp, (u32*)sl->items.p, (u32*)sl->item_off.p, 0u,
                                    (size_t)(g.B + 1), rocprim::plus<u32>(), st));
  }
  AMSM_DBG("pre-l0");
  }
  if (ctx->profiling) (void)hipEventRecord(sl->ev_prep_end, st);
  if (!one) {
    HIP_TRY(hipEventRecord(sl->prep_done, st));
    HIP_TRY(hipStreamWaitEvent(sm, sl->prep_done, 0));
  }
  stage_mark(ctx, sl, ST_ACCUM_L0, sm);
  launch_accum_l0<Fq>(sm, (const u32*)bases->d_table, (const u32*)vals_b, (const u32*)sl->start.p,
                      (const u32*)sl->item_off.p, g, (u32*)sl->partials.p,
                      // 160 KiB of LDS per CU, 32 KiB static per workgroup: 64 KiB of padding lets one workgroup in, 24 KiB two
                      g.l0_per_cu == 1 ? 65536u : (g.l0_per_cu == 2 ? 24576u : ctx->l0_lds_pad));
  AMSM_DBG("l0");
  if (ctx->profiling) (void)hipEventRecord(sl->ev_l0_end, sm);
  hipStream_t tl = one ? ctx->stream : ctx->s_tail;
  sl->tail = tl;
  if (!one) {
    HIP_TRY(hipEventRecord(sl->l0_done, sm));
    HIP_TRY(hipStreamWaitEvent(tl, sl->l0_done, 0));
  }
  if (ctx->profiling) (void)hipEventRecord(sl->ev[ST_ACCUM_L12], tl);
  launch_accum_l1<Fq>(tl, l1_lanes(g, !exposed_tail), (const u32*)sl->partials.p, (const u32*)sl->items.p, (const u32*)sl->item_off.p,
                      g, (u32*)sl->buckets.p, d_heavy_count, (u32*)sl->heavy.p);
  AMSM_DBG("l1");
  launch_accum_l2<Fq>(tl, (const u32*)sl->partials.p, (const u32*)sl->items.p, (const u32*)sl->item_off.p,
                      (const u32*)d_heavy_count, (const u32*)sl->heavy.p, (u32*)sl->heavy_scratch.p,
                      (u32*)sl->buckets.p);
  AMSM_DBG("l2");
  if (ctx->profiling) (void)hipEventRecord(sl->ev[ST_REDUCE], tl);
  if (quad)
    launch_bucket_reduce_quad<Fq>(tl, red_blocks, (const u32*)sl->buckets.p, g, (u32*)sl->red_out.p);
  else
    launch_bucket_reduce<Fq>(tl, red_blocks, (const u32*)sl->buckets.p, g, (u32*)sl->red_out.p);
  AMSM_DBG("reduce");
  if (quad)
    launch_fold_quad<Fq>(tl, g.n_sets, (const u32*)sl->red_out.p, red_blocks, (u32*)sl->fold_out.p, d_err);
  else
    launch_fold<Fq>(tl, g.n_sets, (const u32*)sl->red_out.p, red_blocks, (u32*)sl->fold_out.p, d_err);
  AMSM_DBG("fold");
  if (ctx->profiling) (void)hipEventRecord(sl->ev[ST_COUNT], tl);
  HIP_TRY(hipGetLastError());
  u32* h = (u32*)sl->h_pinned;
  HIP_TRY(hipMemcpyAsync(h, sl->fold_out.p, g.n_sets * rec + 8, hipMemcpyDeviceToHost, tl));  // records + flag words
  HIP_TRY(hipEventRecord(sl->done, tl));
  sl->busy = true;
  return AMSM_OK;
}

// Wait for slot `sl`, then host Horner over the window sums (plain key) -> one XYZZ per group on the host
// (`out` has room for sl->geom.groups results).
template <class Fq>
int msm_collect(amsm_ctx* ctx, Slot* sl, host::HXYZZ<Fq>* out) {
  const MsmGeom& g = sl->geom;
  size_t rec = xyzz_bytes<Fq>();
  HIP_TRY(hipEventSynchronize(sl->done));
  sl->busy = false;
  if (ctx->profiling) {
    for (int s = 0; s < ST_COUNT; s++) {
      float ms = 0;
      // a stage ends on its own stream: the next stage's begin mark sits on another stream for prep -> L0 -> tail, and
      // would add that stream's queueing (the tail stream is still busy with the previous MSM inside a batch)
      hipEvent_t end = s == ST_BOUNDS ? sl->ev_prep_end : (s == ST_ACCUM_L0 ? sl->ev_l0_end : sl->ev[s + 1]);
      (void)hipEventElapsedTime(&ms, sl->ev[s], end);
      ctx->stage_acc[s] += ms;
    }
    ctx->stage_n++;
  }
  u32* h = (u32*)sl->h_pinned;
  u32 err = *(u32*)((char*)h + g.n_sets * rec);
  if (err) return AMSM_E_SCALAR_RANGE;
  // one result per group: Horner over the group's window sums (plain key) or its single record (precomputed key)
  const u32 per = g.n_sets / g.groups;
  for (u32 grp = 0; grp < g.groups; grp++) {
    const u32* base = h + (size_t)grp * per * (rec / 4);
    host::HXYZZ<Fq> acc = host::hx_from_device<Fq>(base + (size_t)(per - 1) * (rec / 4));
    for (int w = (int)per - 2; w >= 0; w--) {
      for (u32 k = 0; k < g.c; k++) acc = host::hx_dbl<Fq>(acc);
      acc = host::hx_add<Fq>(acc, host::hx_from_device<Fq>(base + (size_t)w * (rec / 4)));
    }
    out[grp] = acc;
  }
  return AMSM_OK;
}

void stage_begin(amsm_ctx* ctx) {
  for (int s = 0; s < ST_COUNT; s++) ctx->stage_acc[s] = 0;
  ctx->stage_n = 0;
}
void stage_end(amsm_ctx* ctx) {
  if (!ctx->stage_n) return;
  for (int s = 0; s < ST_COUNT; s++) ctx->stage_ms[s] = ctx->stage_acc[s] / ctx->stage_n;
}

template <class Fq, class Fr>
int msm_multi_split_xyzz(amsm_ctx* ctx, const amsm_bases* bases, size_t k, const size_t* offs, const void* const* d_scalars,
                         const size_t* ns, int scalars_mont, std::vector<host::HXYZZ<Fq>>* out);

template <class Fq, class Fr>
int msm_device_xyzz(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const void* d_scalars, size_t n,
                    int scalars_mont, host::HXYZZ<Fq>* out) {
  if (base_off > bases->n) return AMSM_E_INVALID_ARG;
  n = std::min(n, bases->n - base_off);
  if (n == 0) {
    *out = host::hx_inf<Fq>();
    return AMSM_OK;
  }
  if (ctx->split_log2 > 0 && bases->precomp && (n >> ctx->split_min_log2) != 0) {  // large: pipelined windows of the key
    std::vector<host::HXYZZ<Fq>> r;
    int rc = msm_multi_split_xyzz<Fq, Fr>(ctx, bases, 1, &base_off, &d_scalars, &n, scalars_mont, &r);
    if (rc == AMSM_OK) *out = r[0];
    return rc;
  }
  stage_begin(ctx);
  TRY(prep_fork(ctx));
  TRY((msm_enqueue<Fq, Fr>(ctx, &ctx->slot[0], bases, base_off, d_scalars, n, scalars_mont)));
  int rc = msm_collect<Fq>(ctx, &ctx->slot[0], out);
  stage_end(ctx);
  return rc;
}

// k MSMs over (windows of) the same key, N_SLOTS in flight on the per-stage streams.  MSM v uses generators
// [offs[v], offs[v] + ns[v]) and the scalars at d_scalars[v].
template <class Fq, class Fr>
int msm_multi_xyzz(amsm_ctx* ctx, const amsm_bases* bases, size_t k, const size_t* offs, const void* const* d_scalars,
                   const size_t* ns, int scalars_mont, std::vector<host::HXYZZ<Fq>>* out) {
  out->assign(k, host::hx_inf<Fq>());
  std::vector<size_t> len(k);
  for (size_t v = 0; v < k; v++) {
    if (offs[v] > bases->n) return AMSM_E_INVALID_ARG;
    len[v] = std::min(ns[v], bases->n - offs[v]);
  }
  if (k == 0) return AMSM_OK;
  stage_begin(ctx);
  TRY(prep_fork(ctx));
  int rc = AMSM_OK;
  std::vector<long> owner(N_SLOTS, -1);  // which MSM a busy slot carries
  size_t slot_rr = 0;
  size_t last = 0;  // the last non-empty MSM: the only one whose tail the caller waits for
  size_t n_live = 0;
  for (size_t v = 0; v < k; v++)
    if (len[v]) last = v, n_live++;
  for (size_t v = 0; v < k && rc == AMSM_OK; v++) {
    if (len[v] == 0) continue;  // identity
    Slot* sl = &ctx->slot[slot_rr % N_SLOTS];
    if (sl->busy) rc = msm_collect<Fq>(ctx, sl, &(*out)[owner[slot_rr % N_SLOTS]]);
    if (rc == AMSM_OK) {
      rc = msm_enqueue<Fq, Fr>(ctx, sl, bases, offs[v], d_scalars[v], len[v], scalars_mont, -1, v == last, n_live == 1);
      owner[slot_rr % N_SLOTS] = (long)v;
      slot_rr++;
    }
  }
  for (int j = 0; j < N_SLOTS; j++) {  // drain in enqueue order
    size_t s = (slot_rr + j) % N_SLOTS;
    Slot* sl = &ctx->slot[s];
    if (sl->busy) {
      int r2 = msm_collect<Fq>(ctx, sl, &(*out)[owner[s]]);
      if (rc == AMSM_OK) rc = r2;
    }
  }
  stage_end(ctx);
  return rc;
}

// Large MSMs over a precomputed key run as pipelined sub-MSMs over windows of 2^split_log2 generators, summed on the host.
// The partition pass of the prep sorts entries by (bucket, index) in one 32-bit word: above 2^21 pairs the index bits leave
// too few for bucket ids and the pass needs 2048-4096 partitions (8- to 4-byte runs: prep 7.2 ms at 2^23; 2^24 fell back to
// the rocPRIM sort).  A window carries window-relative indices (MsmGeom::idx_rel_bits) and sorts like a 2^21-pair MSM
// whatever the key's length; the extra bucket reductions hide behind the next window's accumulation, and a blocking call of
// that size becomes a pipeline.  Measured (batches, M pairs/s, whole / 2^20 windows / 2^21 windows): 2^22 815 / 822 / 845,
// 2^23 507 / 841 / 868, 2^24 725 / 848 / 874; the workspace of a 2^24-pair MSM drops from 25.6 GB to 1.5 GB.
template <class Fq, class Fr>
int msm_multi_split_xyzz(amsm_ctx* ctx, const amsm_bases* bases, size_t k, const size_t* offs, const void* const* d_scalars,
                         const size_t* ns, int scalars_mont, std::vector<host::HXYZZ<Fq>>* out) {
  const size_t chunk = ctx->split_log2 > 0 ? (size_t)1 << ctx->split_log2 : 0;
  bool any = false;
  if (chunk && bases->precomp)
    for (size_t v = 0; v < k; v++) any = any || (std::min(ns[v], offs[v] <= bases->n ? bases->n - offs[v] : 0) >> ctx->split_min_log2) != 0;
  if (!any) return msm_multi_xyzz<Fq, Fr>(ctx, bases, k, offs, d_scalars, ns, scalars_mont, out);
  std::vector<size_t> s_offs, s_ns, owner;
  std::vector<const void*> s_ptrs;
  for (size_t v = 0; v < k; v++) {
    if (offs[v] > bases->n) return AMSM_E_INVALID_ARG;
    const size_t len = std::min(ns[v], bases->n - offs[v]);
    const bool split = (len >> ctx->split_min_log2) != 0;
    const size_t step = split ? chunk : std::max<size_t>(len, 1);
    for (size_t lo = 0; lo < len; lo += step) {
      s_offs.push_back(offs[v] + lo);
      s_ns.push_back(std::min(step, len - lo));
      s_ptrs.push_back((const char*)d_scalars[v] + lo * 32);
      owner.push_back(v);
    }
  }
  std::vector<host::HXYZZ<Fq>> part;
  int rc = msm_multi_xyzz<Fq, Fr>(ctx, bases, s_ns.size(), s_offs.data(), s_ptrs.data(), s_ns.data(), scalars_mont, &part);
  out->assign(k, host::hx_inf<Fq>());
  if (rc != AMSM_OK) return rc;
  for (size_t j = 0; j < part.size(); j++) (*out)[owner[j]] = host::hx_add<Fq>((*out)[owner[j]], part[j]);
  return AMSM_OK;
}

// n_vecs MSMs over the same generators (the prover's back-to-back commits)
template <class Fq, class Fr>
int msm_batch_xyzz(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const void* const* d_scalars,
                   size_t n_vecs, size_t n, int scalars_mont, std::vector<host::HXYZZ<Fq>>* out) {
  std::vector<size_t> offs(n_vecs, base_off), ns(n_vecs, n);
  return msm_multi_split_xyzz<Fq, Fr>(ctx, bases, n_vecs, offs.data(), d_scalars, ns.data(), scalars_mont, out);
}

// batch_normalization_into_affine (src/hp_as/mod.rs:468): one inversion for the whole batch
template <class Fq>
void write_affine_batch(const std::vector<host::HXYZZ<Fq>>& pts, uint64_t* out_xy, uint8_t* out_is_inf) {
  using H = host::HFe<Fq>;
  constexpr int N = H::N;
  size_t k = pts.size();
  std::vector<H> z(k), pre(k);
  H run = host::h_one<Fq>();
  for (size_t i = 0; i < k; i++) {
    bool inf = host::hx_is_inf<Fq>(pts[i]);
    z[i] = inf ? host::h_one<Fq>() : host::h_mul<Fq>(pts[i].zz, pts[i].zzz);
    pre[i] = run;
    run = host::h_mul<Fq>(run, z[i]);
  }
  H inv = host::h_inv<Fq>(run);
  for (size_t i = k; i-- > 0;) {
    H zi = host::h_mul<Fq>(inv, pre[i]);  // 1 / (zz*zzz)
    inv = host::h_mul<Fq>(inv, z[i]);
    uint64_t* o = out_xy + i * 2 * N;
    if (host::hx_is_inf<Fq>(pts[i])) {
      memset(o, 0, 16 * N);
      if (out_is_inf) out_is_inf[i] = 1;
      continue;
    }
    H x = host::h_mul<Fq>(pts[i].x, host::h_mul<Fq>(zi, pts[i].zzz));
    H y = host::h_mul<Fq>(pts[i].y, host::h_mul<Fq>(zi, pts[i].zz));
    memcpy(o, x.v, 8 * N);
    memcpy(o + N, y.v, 8 * N);
    if (out_is_inf) out_is_inf[i] = 0;
  }
}

template <class Fq, class Fr>
int bases_finish(amsm_ctx* ctx, amsm_bases* b, unsigned flags) {
  // decide precomputation, then build the table levels on device
  bool pre;
  if (flags & AMSM_BASES_PRECOMPUTE) pre = true;
  else if (flags & AMSM_BASES_NO_PRECOMPUTE) pre = false;
  else pre = b->n >= 256;
  if (!pre) return AMSM_OK;
  // An EXPLICIT request (AMSM_BASES_PRECOMPUTE) that cannot be honoured is an error, not a silent 10x slower key; the
  // library's own choice (AMSM_BASES_DEFAULT) degrades to the plain key.
  const bool requested = (flags & AMSM_BASES_PRECOMPUTE) != 0;
  int c = ctx->window_override ? ctx->window_override : choose_window(b->n, true);
  int W = windows_for(c);
  if ((unsigned long long)b->n * W >= (1ull << 30))  // entry words carry a 30-bit table index
    return requested ? AMSM_E_UNSUPPORTED : AMSM_OK;
  u32* table = nullptr;
  hipError_t e = hipMalloc((void**)&table, (size_t)b->n * W * affine_bytes<Fq>());
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return requested ? AMSM_E_OOM : AMSM_OK;  // not enough HBM for W copies
  }
  HIP_TRY(hipMemcpyAsync(table, b->d_table, b->n * affine_bytes<Fq>(), hipMemcpyDeviceToDevice, ctx->stream));
  u32* scratch = nullptr;
  if (batch_affine_pays<Fq>((u32)b->n) && ensure(ctx->xyzz_scratch, b->n * xyzz_bytes<Fq>()) == AMSM_OK)
    scratch = (u32*)ctx->xyzz_scratch.p;
  for (int w = 1; w < W; w++) {
    launch_precompute_level<Fq>(ctx->stream, table, (u32)b->n, (u32)w, (u32)c, scratch);
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(hipGetLastError());
  if (scratch) {  // a key-sized buffer: not worth keeping between the rare key loads
    (void)hipFree(ctx->xyzz_scratch.p);
    ctx->xyzz_scratch = DevBuf();
  }
  HIP_TRY(hipFree(b->d_table));
  b->d_table = table;
  b->precomp = 1;
  b->c = c;
  b->W = W;
  return AMSM_OK;
}

template <class Fq, class Fr>
int bases_load_impl(amsm_ctx* ctx, const uint64_t* xy, const uint8_t* is_inf, size_t n, unsigned flags,
                    amsm_bases** out) {
  amsm_bases* b = new (std::nothrow) amsm_bases();
  if (!b) return AMSM_E_OOM;
  b->curve = ctx->curve;
  b->device = ctx->device;
  b->n = n;
  size_t bytes = std::max<size_t>(n, 1) * affine_bytes<Fq>();
  hipError_t e = hipMalloc((void**)&b->d_table, bytes);
  if (e != hipSuccess) {
    delete b;
    return AMSM_E_OOM;
  }
  int s = AMSM_OK;
  do {
    if (n) {
      if (hipMemcpyAsync(b->d_table, xy, n * affine_bytes<Fq>(), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        s = AMSM_E_HIP;
        break;
      }
      if (is_inf) {
        s = ensure(ctx->scalars, n);
        if (s) break;
        if (hipMemcpyAsync(ctx->scalars.p, is_inf, n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
          s = AMSM_E_HIP;
          break;
        }
        launch_apply_inf<Fq>(ctx->stream, b->d_table, (const uint8_t*)ctx->scalars.p, (u32)n);
      }
      launch_points_import<Fq>(ctx->stream, b->d_table, b->d_table, (u32)n);  // C-ABI radix -> device radix
      if (hipStreamSynchronize(ctx->stream) != hipSuccess) {
        s = AMSM_E_HIP;
        break;
      }
      s = bases_finish<Fq, Fr>(ctx, b, flags);
    }
  } while (0);
  if (s != AMSM_OK) {
    (void)hipFree(b->d_table);
    delete b;
    return s;
  }
  *out = b;
  return AMSM_OK;
}

template <class Fq, class Fr>
int bases_generate_impl(amsm_ctx* ctx, uint64_t seed, size_t n, unsigned flags, amsm_bases** out, size_t first = 0) {
  amsm_bases* b = new (std::nothrow) amsm_bases();
  if (!b) return AMSM_E_OOM;
  b->curve = ctx->curve;
  b->device = ctx->device;
  b->n = n;
  hipError_t e = hipMalloc((void**)&b->d_table, std::max<size_t>(n, 1) * affine_bytes<Fq>());
  if (e != hipSuccess) {
    delete b;
    return AMSM_E_OOM;
  }
  int s = AMSM_OK;
  if (n) {
    std::vector<u32> gen = generator_mont<Fq>(ctx->curve);
    launch_generate_bases<Fq>(ctx->stream, b->d_table, seed, (u32)first, (u32)n, gen.data());
    if (hipStreamSynchronize(ctx->stream) != hipSuccess || hipGetLastError() != hipSuccess) s = AMSM_E_HIP;
    if (s == AMSM_OK) s = bases_finish<Fq, Fr>(ctx, b, flags);
  }
  if (s != AMSM_OK) {
    (void)hipFree(b->d_table);
    delete b;
    return s;
  }
  *out = b;
  return AMSM_OK;
}

template <class Fq>
int bases_read_impl(amsm_ctx* ctx, const amsm_bases* b, size_t off, size_t n, uint64_t* xy, uint8_t* is_inf) {
  if (off > b->n || n > b->n - off) return AMSM_E_INVALID_ARG;
  if (!n) return AMSM_OK;
  const char* src = (const char*)b->d_table + off * affine_bytes<Fq>();
  if (device_internal_radix<Fq>()) {  // back to the C-ABI radix through a scratch buffer
    TRY(ensure(ctx->scalars, n * affine_bytes<Fq>()));
    launch_points_export<Fq>(ctx->stream, (const u32*)src, (u32*)ctx->scalars.p, (u32)n);
    src = (const char*)ctx->scalars.p;
  }
  HIP_TRY(hipMemcpyAsync(xy, src, n * affine_bytes<Fq>(), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (is_inf) {
    constexpr int N = 2 * Fq::L / 2;
    for (size_t i = 0; i < n; i++) {
      u64 o = 0;
      for (int k = 0; k < N; k++) o |= xy[i * N + k];
      is_inf[i] = o == 0;
    }
  }
  return AMSM_OK;
}

template <class Fq>
void write_affine(const host::HXYZZ<Fq>& p, uint64_t* out_xy, uint8_t* out_is_inf) {
  uint8_t inf = 0;
  host::hx_to_affine<Fq>(p, out_xy, &inf);
  if (out_is_inf) *out_is_inf = inf;
}

template <class Fq, class Fr>
int msm_host_scalars(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const uint64_t* scalars, size_t n,
                     int mont, host::HXYZZ<Fq>* out) {
  if (base_off > bases->n) return AMSM_E_INVALID_ARG;
  n = std::min(n, bases->n - base_off);
  if (n == 0) {
    *out = host::hx_inf<Fq>();
    return AMSM_OK;
  }
  TRY(ensure(ctx->scalars, n * 32));
  HIP_TRY(hipMemcpyAsync(ctx->scalars.p, scalars, n * 32, hipMemcpyHostToDevice, ctx->stream));
  return msm_device_xyzz<Fq, Fr>(ctx, bases, base_off, ctx->scalars.p, n, mont, out);
}

template <class Fq, class Fr>
int pedersen_impl(amsm_ctx* ctx, const amsm_bases* ck, const uint64_t* elems, size_t n, const uint64_t* rand_mont,
                  const uint64_t* hiding_xy, uint64_t* out_xy, uint8_t* out_inf) {
  host::HXYZZ<Fq> acc;
  TRY((msm_host_scalars<Fq, Fr>(ctx, ck, 0, elems, n, 1, &acc)));
  if (rand_mont && hiding_xy) {
    host::HFe<Fr> r;
    memcpy(r.v, rand_mont, 32);
    r = host::h_from_mont<Fr>(r);
    host::HXYZZ<Fq> h = host::hx_from_affine<Fq>(hiding_xy, false);
    acc = host::hx_add<Fq>(acc, host::hx_mul<Fq>(h, r.v));
  }
  write_affine<Fq>(acc, out_xy, out_inf);
  return AMSM_OK;
}

template <class Fq, class Fr>
int pedersen_device_impl(amsm_ctx* ctx, const amsm_bases* ck, const void* d_elems, size_t n, const uint64_t* rand_mont,
                         const uint64_t* hiding_xy, uint64_t* out_xy, uint8_t* out_inf) {
  host::HXYZZ<Fq> acc;
  TRY((msm_device_xyzz<Fq, Fr>(ctx, ck, 0, d_elems, n, 1, &acc)));
  if (rand_mont && hiding_xy) {
    host::HFe<Fr> r;
    memcpy(r.v, rand_mont, 32);
    r = host::h_from_mont<Fr>(r);
    host::HXYZZ<Fq> h = host::hx_from_affine<Fq>(hiding_xy, false);
    acc = host::hx_add<Fq>(acc, host::hx_mul<Fq>(h, r.v));
  }
  write_affine<Fq>(acc, out_xy, out_inf);
  return AMSM_OK;
}

// Bases that recur with full-size scalars get a fixed-base table (HFixedBase) on their third use; a handful per thread
// and field, least recently used evicted.  AMSM_HOST_FIXED_BASE=0 turns the cache off (A/B, tests).
template <class Fq>
struct FixedBaseCache {
  static constexpr size_t SLOTS = 4;
  static constexpr unsigned BUILD_AT = 3;
  std::vector<host::HFixedBase<Fq>> entries;
  uint64_t call = 0;  // entries touched by the current host_lincomb call (stamp == call) are never evicted
  FixedBaseCache() { entries.reserve(SLOTS); }  // no reallocation: pointers handed out stay valid
  const host::HFixedBase<Fq>* lookup(const uint64_t* xy, const host::HXYZZ<Fq>& p) {
    constexpr int N = host::HFe<Fq>::N;
    for (auto& c : entries)
      if (memcmp(c.key, xy, 16 * N) == 0) {
        c.stamp = call;
        if (c.tab.empty() && ++c.uses >= BUILD_AT) c.build(p);
        return c.tab.empty() ? nullptr : &c;
      }
    host::HFixedBase<Fq>* victim = nullptr;
    if (entries.size() < SLOTS) {
      entries.emplace_back();
      victim = &entries.back();
    } else {
      for (auto& c : entries)
        if (c.stamp != call && (!victim || c.stamp < victim->stamp)) victim = &c;
      if (!victim) return nullptr;  // every slot is in use by this very call
    }
    memcpy(victim->key, xy, 16 * N);
    victim->tab.clear();
    victim->uses = 1;
    victim->stamp = call;
    return nullptr;
  }
};

inline bool fixed_base_enabled() {
  static const bool on = [] {
    const char* e = getenv("AMSM_HOST_FIXED_BASE");
    return !(e && atoi(e) == 0);
  }();
  return on;
}
template <class Fq>
FixedBaseCache<Fq>& fixed_base_cache() {
  thread_local FixedBaseCache<Fq> cache;
  return cache;
}
// k * P for a point that recurs over calls (h' of an IPA opening): the thread's fixed-base table once it exists
template <class Fq, class Fr>
host::HXYZZ<Fq> host_mul_cached(const uint64_t* xy, const host::HFe<Fr>& k_mont) {
  host::HFe<Fr> k = host::h_from_mont<Fr>(k_mont);
  host::HXYZZ<Fq> p = host::hx_from_affine<Fq>(xy, false);
  const host::HFixedBase<Fq>* fb = nullptr;
  if (fixed_base_enabled() && host::hx_scalar_bits(k.v) > 128) {
    FixedBaseCache<Fq>& cache = fixed_base_cache<Fq>();
    cache.call++;
    fb = cache.lookup(xy, p);
  }
  return fb ? fb->mul(k.v) : host::hx_mul<Fq>(p, k.v);
}

// sum_i k_i P_i as an XYZZ point (k Montgomery).  Jobs of LINCOMB_SPLIT_MIN points and more are split over the host pool:
// every part pays the shared doublings again (128 for challenge-sized scalars) but the additions -- 32 per point -- divide.
constexpr size_t LINCOMB_SPLIT_MIN = 8;
template <class Fq, class Fr>
host::HXYZZ<Fq> host_lincomb_xyzz(const uint64_t* xy, const uint8_t* is_inf, const uint64_t* scalars_mont, size_t n,
                                  bool may_split) {
  constexpr int N = host::HFe<Fq>::N;
  if (may_split && n >= LINCOMB_SPLIT_MIN && !HostPool::get().workers.empty()) {
    const size_t parts = std::min(HostPool::get().workers.size() + 1, n / (LINCOMB_SPLIT_MIN / 2));
    std::vector<host::HXYZZ<Fq>> part(parts, host::hx_inf<Fq>());
    HostPool::get().run(parts, [&](size_t t) {
      const size_t lo = n * t / parts, hi = n * (t + 1) / parts;
      part[t] = host_lincomb_xyzz<Fq, Fr>(xy + lo * 2 * N, is_inf ? is_inf + lo : nullptr, scalars_mont + 4 * lo, hi - lo, false);
    });
    host::HXYZZ<Fq> acc = part[0];
    for (size_t t = 1; t < parts; t++) acc = host::hx_add<Fq>(acc, part[t]);
    return acc;
  }
  std::vector<host::HXYZZ<Fq>> pts(n);
  std::vector<std::array<uint64_t, 4>> ks(n);
  std::vector<const host::HFixedBase<Fq>*> fixed(n, nullptr);
  const bool use_cache = fixed_base_enabled();
  FixedBaseCache<Fq>& cache = fixed_base_cache<Fq>();  // per thread
  cache.call++;
  for (size_t i = 0; i < n; i++) {
    host::HFe<Fr> s;
    memcpy(s.v, scalars_mont + 4 * i, 32);
    s = host::h_from_mont<Fr>(s);
    memcpy(ks[i].data(), s.v, 32);
    pts[i] = host::hx_from_affine<Fq>(xy + i * 2 * N, is_inf && is_inf[i]);
    if (use_cache && !host::hx_is_inf<Fq>(pts[i]) && host::hx_scalar_bits(s.v) > 128)
      fixed[i] = cache.lookup(xy + i * 2 * N, pts[i]);
  }
  return host::hx_lincomb<Fq>(pts.data(), reinterpret_cast<const uint64_t(*)[4]>(ks.data()), n, fixed.data());
}

template <class Fq, class Fr>
int host_lincomb_impl(const uint64_t* xy, const uint8_t* is_inf, const uint64_t* scalars_mont, size_t n,
                      uint64_t* out_xy, uint8_t* out_inf) {
  write_affine<Fq>(host_lincomb_xyzz<Fq, Fr>(xy, is_inf, scalars_mont, n, true), out_xy, out_inf);
  return AMSM_OK;
}

// independent combinations, one per pool task (large ones are not split again: the pool is taken), ONE batched normalisation
template <class Fq, class Fr>
int host_lincomb_batch_impl(size_t n_jobs, const size_t* n_terms, const uint64_t* const* xy, const uint8_t* const* is_inf,
                            const uint64_t* const* scalars_mont, uint64_t* out_xy, uint8_t* out_inf) {
  std::vector<host::HXYZZ<Fq>> r(n_jobs, host::hx_inf<Fq>());
  HostPool::get().run(n_jobs, [&](size_t j) {
    r[j] = host_lincomb_xyzz<Fq, Fr>(xy[j], is_inf ? is_inf[j] : nullptr, scalars_mont[j], n_terms[j], false);
  });
  if (out_inf) memset(out_inf, 0, n_jobs);
  write_affine_batch<Fq>(r, out_xy, out_inf);
  return AMSM_OK;
}

template <class Fq>
int partials_combine_impl(amsm_ctx* ctx, const void* d_partials, size_t count, uint64_t* out_xy, uint8_t* out_inf) {
  size_t rec = xyzz_bytes<Fq>();
  Slot* sl = &ctx->slot[0];
  TRY(ensure_pinned(sl, count * rec));
  if (count) {
    HIP_TRY(hipMemcpyAsync(sl->h_pinned, d_partials, count * rec, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  host::HXYZZ<Fq> acc = host::hx_inf<Fq>();
  for (size_t i = 0; i < count; i++)
    acc = host::hx_add<Fq>(acc, host::hx_from_device<Fq>((const u32*)sl->h_pinned + i * (rec / 4)));
  write_affine<Fq>(acc, out_xy, out_inf);
  return AMSM_OK;
}

template <class Fq, class Fr>
int msm_grouped_impl(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const void* d_scalars, size_t n, int mont,
                     unsigned group_shift, uint64_t* out_xy, uint8_t* out_inf) {
  if (base_off > bases->n) return AMSM_E_INVALID_ARG;
  n = std::min(n, bases->n - base_off);
  std::vector<host::HXYZZ<Fq>> r(2, host::hx_inf<Fq>());
  if (n) {
    stage_begin(ctx);
    TRY(prep_fork(ctx));
    TRY((msm_enqueue<Fq, Fr>(ctx, &ctx->slot[0], bases, base_off, d_scalars, n, mont, (int)group_shift)));
    int rc = msm_collect<Fq>(ctx, &ctx->slot[0], r.data());
    stage_end(ctx);
    if (rc != AMSM_OK) return rc;
  }
  write_affine_batch<Fq>(r, out_xy, out_inf);
  return AMSM_OK;
}

// One IPA opening round with ONE synchronisation: the round's scalar expansion, the grouped MSM over the key and the two
// inner products <c_r, z_l>, <c_l, z_r>; the inner-product kernels are queued behind accumulate L0 on the context's
// stream and run beside the MSM's tail.
template <class Fq, class Fr>
int ipa_round_impl(amsm_ctx* ctx, const amsm_bases* key, const uint64_t* xi_mont, size_t j, size_t log_key, const void* d_coeffs,
                   const void* d_z, size_t half, void* d_u, uint64_t* out_lr_xy, uint8_t* out_lr_inf, uint64_t* out_ip_mont,
                   const uint64_t* fold_x_mont = nullptr, const uint64_t* h_prime_xy = nullptr) {
  const size_t n = (size_t)1 << log_key;
  if (n > key->n) return AMSM_E_INVALID_ARG;
  // The two inner products first: their per-workgroup partial sums go straight to pinned host memory (no copy in the
  // chain), and the host turns them into the h' multiples while the MSM runs.  With a fold pending (the previous round's
  // c <- c_l + x^-1 c_r, z <- z_l + x z_r over the 4 * half elements the buffers still hold) the same launch folds in place.
  const u32 blocks = std::min<u32>(1024u, cdiv((u32)half, 256));
  Slot* aux = &ctx->slot[1];  // only its pinned buffer: slot 1 carries no MSM during a single-MSM call
  TRY(ensure_pinned(aux, (size_t)2 * blocks * 32));
  u32* co = (u32*)d_coeffs;
  u32* z = (u32*)d_z;
  u32* part = (u32*)aux->h_pinned;
  if (fold_x_mont) {
    host::HFe<Fr> x, xinv;
    memcpy(x.v, fold_x_mont, 32);
    xinv = host::h_inv<Fr>(x);
    launch_ipa_fold_ip<Fr>(ctx->stream, co, z, (u32)half, (const u32*)x.v, (const u32*)xinv.v, blocks, part);
  } else {
    launch_vec_inner_product_pair<Fr>(ctx->stream, co + half * 8, z, co, z + half * 8, (u32)half, blocks, part);  // <c_r, z_l>, <c_l, z_r>
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ctx->ip_ready, ctx->stream));
  launch_ipa_round_scalars<Fr>(ctx->stream, (const u32*)xi_mont, (u32)j, (u32)log_key, (const u32*)d_coeffs, (u32*)d_u, nullptr);
  HIP_TRY(hipGetLastError());
  std::vector<host::HXYZZ<Fq>> r(2, host::hx_inf<Fq>());
  stage_begin(ctx);
  TRY(prep_fork(ctx));
  TRY((msm_enqueue<Fq, Fr>(ctx, &ctx->slot[0], key, 0, d_u, n, 1, (int)(log_key - 1 - j))));
  HIP_TRY(hipEventSynchronize(ctx->ip_ready));
  const u64* h = (const u64*)aux->h_pinned;
  host::HXYZZ<Fq> hterm[2] = {host::hx_inf<Fq>(), host::hx_inf<Fq>()};
  for (int k = 0; k < 2; k++) {
    host::HFe<Fr> acc = host::h_zero<Fr>(), t;
    for (u32 i = 0; i < blocks; i++) {
      memcpy(t.v, h + 4 * ((size_t)k * blocks + i), 32);
      acc = host::h_add<Fr>(acc, t);
    }
    memcpy(out_ip_mont + 4 * k, acc.v, 32);
    if (h_prime_xy) hterm[k] = host_mul_cached<Fq, Fr>(h_prime_xy, acc);
  }
  int rc = msm_collect<Fq>(ctx, &ctx->slot[0], r.data());
  stage_end(ctx);
  if (rc != AMSM_OK) return rc;
  if (h_prime_xy)
    for (int k = 0; k < 2; k++) r[k] = host::hx_add<Fq>(r[k], hterm[k]);
  write_affine_batch<Fq>(r, out_lr_xy, out_lr_inf);
  return AMSM_OK;
}

// groups of `count` consecutive records -> one affine point per group (one D2H copy, one batched normalisation)
template <class Fq>
int partials_combine_batch_impl(amsm_ctx* ctx, const void* d_partials, size_t n_groups, size_t count, uint64_t* out_xy,
                                uint8_t* out_inf) {
  size_t rec = xyzz_bytes<Fq>();
  Slot* sl = &ctx->slot[0];
  size_t total = n_groups * count;
  TRY(ensure_pinned(sl, total * rec + 64));
  if (total) {
    HIP_TRY(hipMemcpyAsync(sl->h_pinned, d_partials, total * rec, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  std::vector<host::HXYZZ<Fq>> pts(n_groups, host::hx_inf<Fq>());
  for (size_t g = 0; g < n_groups; g++)
    for (size_t i = 0; i < count; i++)
      pts[g] = host::hx_add<Fq>(pts[g], host::hx_from_device<Fq>((const u32*)sl->h_pinned + (g * count + i) * (rec / 4)));
  write_affine_batch<Fq>(pts, out_xy, out_inf);
  return AMSM_OK;
}

// n_vecs MSMs (pipelined like msm_batch_xyzz) -> n_vecs consecutive device records at d_out
template <class Fq, class Fr>
int msm_partial_batch_impl(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const void* const* d_scalars,
                           size_t n_vecs, size_t n, int mont, void* d_out) {
  std::vector<host::HXYZZ<Fq>> r;
  TRY((msm_batch_xyzz<Fq, Fr>(ctx, bases, base_off, d_scalars, n_vecs, n, mont, &r)));
  if (!n_vecs) return AMSM_OK;
  size_t rec = xyzz_bytes<Fq>();
  Slot* sl = &ctx->slot[0];
  TRY(ensure_pinned(sl, n_vecs * rec + 64));
  u32* h = (u32*)sl->h_pinned;
  for (size_t v = 0; v < n_vecs; v++) {
    u32* o = h + v * (rec / 4);
    memcpy(o, r[v].x.v, rec / 4);
    memcpy(o + Fq::L, r[v].y.v, rec / 4);
    memcpy(o + 2 * Fq::L, r[v].zz.v, rec / 4);
    memcpy(o + 3 * Fq::L, r[v].zzz.v, rec / 4);
  }
  HIP_TRY(hipMemcpyAsync(d_out, h, n_vecs * rec, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return AMSM_OK;
}

// store a host XYZZ as a device record
template <class Fq>
int upload_xyzz(amsm_ctx* ctx, const host::HXYZZ<Fq>& p, void* d_out) {
  size_t rec = xyzz_bytes<Fq>();
  Slot* sl = &ctx->slot[0];
  TRY(ensure_pinned(sl, rec + 64));
  u32* h = (u32*)sl->h_pinned;
  memcpy(h, p.x.v, rec / 4);
  memcpy(h + Fq::L, p.y.v, rec / 4);
  memcpy(h + 2 * Fq::L, p.zz.v, rec / 4);
  memcpy(h + 3 * Fq::L, p.zzz.v, rec / 4);
  HIP_TRY(hipMemcpyAsync(d_out, h, rec, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return AMSM_OK;
}

template <class Fq, class Fr>
int msm_partial_impl(amsm_ctx* ctx, const amsm_bases* bases, size_t base_off, const void* d_scalars, size_t n, int mont,
                     void* d_out) {
  if (base_off > bases->n) return AMSM_E_INVALID_ARG;
  n = std::min(n, bases->n - base_off);
  if (n == 0) return upload_xyzz<Fq>(ctx, host::hx_inf<Fq>(), d_out);
  Slot* sl = &ctx->slot[0];
  stage_begin(ctx);
  TRY(prep_fork(ctx));
  TRY((msm_enqueue<Fq, Fr>(ctx, sl, bases, base_off, d_scalars, n, mont)));
  if (sl->geom.n_sets == 1) {
    // single folded record: it stays on the device (copied record-to-record); the host only waits for
    // the range flag
    HIP_TRY(hipMemcpyAsync(d_out, sl->fold_out.p, xyzz_bytes<Fq>(), hipMemcpyDeviceToDevice, sl->tail));
    HIP_TRY(hipEventRecord(sl->done, sl->tail));
    host::HXYZZ<Fq> unused;
    int rc = msm_collect<Fq>(ctx, sl, &unused);
    stage_end(ctx);
    return rc;
  }
  host::HXYZZ<Fq> acc;
  int rc = msm_collect<Fq>(ctx, sl, &acc);
  stage_end(ctx);
  if (rc != AMSM_OK) return rc;
  return upload_xyzz<Fq>(ctx, acc, d_out);
}

template <class Fr>
int vec_combine_impl(amsm_ctx* ctx, const void* const* d_vecs, const size_t* lens, size_t n_vecs,
                     const uint64_t* coeffs, const void* d_hiding, size_t hiding_len, void* d_out, size_t n) {
  if (!n) return AMSM_OK;
  // more than VEC_MAX vectors (the reference has no limit): groups of VEC_MAX, the running sum carried as the "hiding"
  // addend of the next launch (in place: a lane reads its own element of d_out before it writes it)
  size_t done = 0;
  do {
    const size_t k = std::min(n_vecs - done, (size_t)VEC_MAX);
    CombineArgs a;
    memset(&a, 0, sizeof(a));
    for (size_t j = 0; j < k; j++) {
      a.vec[j] = (const u32*)d_vecs[done + j];
      a.len[j] = (u32)(lens ? std::min(lens[done + j], n) : n);
      memcpy(a.coeff[j], coeffs + 4 * (done + j), 32);
    }
    a.hiding = done ? (const u32*)d_out : (const u32*)d_hiding;
    a.hiding_len = done ? (u32)n : (u32)hiding_len;
    a.n_vecs = (u32)k;
    a.n = (u32)n;
    launch_vec_combine<Fr>(ctx->stream, a, (u32*)d_out);
    HIP_TRY(hipGetLastError());
    done += k;
  } while (done < n_vecs);
  return AMSM_OK;
}

template <class Fr>
int t_vecs_impl(amsm_ctx* ctx, const void* const* d_a, const size_t* a_lens, const void* const* d_b,
                const size_t* b_lens, size_t n_in, const uint64_t* mu, size_t n_mu, const void* d_ha, size_t ha_len,
                const void* d_hb, size_t hb_len, void* const* d_t, size_t len) {
  if (n_in < 1 || n_in > (size_t)HP_MAX_INPUTS) return AMSM_E_UNSUPPORTED;
  bool hiding = d_ha != nullptr || d_hb != nullptr;
  // assert!(num_inputs + hiding <= mu_challenges.len())  (src/hp_as/mod.rs:295)
  if (n_in + (hiding ? 1 : 0) > n_mu) return AMSM_E_INVALID_ARG;
  if (!len) return AMSM_OK;
  TVecArgs a;
  memset(&a, 0, sizeof(a));
  for (size_t j = 0; j < n_in; j++) {
    a.a[j] = (const u32*)d_a[j];
    a.b[j] = (const u32*)d_b[j];
    a.a_len[j] = (u32)(a_lens ? std::min(a_lens[j], len) : len);
    a.b_len[j] = (u32)(b_lens ? std::min(b_lens[j], len) : len);
  }
  for (size_t j = 0; j < std::min<size_t>(n_mu, HP_MAX_INPUTS + 1); j++) memcpy(a.mu[j], mu + 4 * j, 32);
  a.hiding_a = (const u32*)d_ha;
  a.hiding_b = (const u32*)d_hb;
  a.hiding_a_len = (u32)ha_len;
  a.hiding_b_len = (u32)hb_len;
  for (size_t k = 0; k < 2 * n_in - 1; k++) a.t[k] = (u32*)d_t[k];
  a.len = (u32)len;
  launch_hp_t_vecs<Fr>(ctx->stream, a, (int)n_in);
  HIP_TRY(hipGetLastError());
  return AMSM_OK;
}

// =============================================================================================
// Multi-device contexts: ONE process drives the GPUs of a node.  The key is sharded over the devices by contiguous ranges
// (the split of accumulation_amd/dist.py: shard_bounds); every device runs the ordinary single-device pipeline on its
// shard, driven by its own host thread; the per-device partial sums (one folded XYZZ record per MSM) are gathered on the
// primary device with one RCCL all-gather (raw bytes: EC addition is not an RCCL reduce op) or peer copies, folded and
// normalised once.
// =============================================================================================
// RCCL is bound at run time (dlopen): libamsm.so carries no link-time dependency on it, a process that already loaded a
// copy (PyTorch ships its own) keeps using that one, and a box without RCCL still gets the peer-copy exchange.
struct RcclApi {
  typedef int (*comm_init_all_t)(void** comms, int ndev, const int* devlist);
  typedef int (*comm_destroy_t)(void* comm);
  typedef int (*all_gather_t)(const void* send, void* recv, size_t count, int datatype, void* comm, hipStream_t st);
  typedef int (*group_t)();
  typedef const char* (*err_t)(int);
  comm_init_all_t comm_init_all = nullptr;
  comm_destroy_t comm_destroy = nullptr;
  all_gather_t all_gather = nullptr;
  group_t group_start = nullptr, group_end = nullptr;
  err_t error_string = nullptr;
  bool ok = false;
  static const RcclApi& get() {
    static RcclApi api = [] {
      RcclApi a;
      void* h = nullptr;
      const char* names[] = {"librccl.so", "librccl.so.1"};
      for (const char* n : names)  // a copy that is already in the process first
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
      for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (!h) return a;
      a.comm_init_all = (comm_init_all_t)dlsym(h, "ncclCommInitAll");
      a.comm_destroy = (comm_destroy_t)dlsym(h, "ncclCommDestroy");
      a.all_gather = (all_gather_t)dlsym(h, "ncclAllGather");
      a.group_start = (group_t)dlsym(h, "ncclGroupStart");
      a.group_end = (group_t)dlsym(h, "ncclGroupEnd");
      a.error_string = (err_t)dlsym(h, "ncclGetErrorString");
      a.ok = a.comm_init_all && a.comm_destroy && a.all_gather && a.group_start && a.group_end;
      return a;
    }();
    return api;
  }
};
constexpr int kNcclUint8 = 1;  // rccl.h: ncclUint8

inline size_t n_shards(const amsm_ctx* c) { return c->shard_ctx.empty() ? 1 : c->shard_ctx.size(); }
inline bool key_sharded(const amsm_bases* b) { return !b->shards.empty(); }

// dist.shard_bounds: the first n % world shards get one extra element
std::vector<size_t> shard_bounds(size_t n, size_t world) {
  std::vector<size_t> b(world + 1);
  size_t q = n / world, r = n % world;
  for (size_t g = 0; g <= world; g++) b[g] = g * q + std::min(g, r);
  return b;
}

// Run job(g) for every shard: shard 0 on the calling thread, the others on their workers.  Returns the first error.
template <class F>
int for_each_shard(amsm_ctx* c, F&& job) {
  const size_t N = n_shards(c);
  std::vector<int> rc(N, AMSM_OK);
  for (size_t g = 1; g < N; g++) c->workers[g - 1]->submit([&, g] { rc[g] = job(g); });
  rc[0] = job(0);
  for (size_t g = 1; g < N; g++) c->workers[g - 1]->wait();
  (void)hipSetDevice(c->device);
  for (size_t g = 0; g < N; g++)
    if (rc[g] != AMSM_OK) return rc[g];
  return AMSM_OK;
}

enum SliceKind {
  SLICE_HOST,     // srcs[v]: host scalars of vector v (whole vector)
  SLICE_PRIMARY,  // srcs[v]: device pointer on the primary device (whole vector)
  SLICE_SHARDED   // srcs[v * N + g]: shard g's slice of vector v, resident on device g
};

// n_vecs MSMs over generators [base_off, base_off + n) of a sharded key -> one XYZZ per vector on the host.
template <class Fq, class Fr>
int msm_sharded(amsm_ctx* c, const amsm_bases* key, size_t base_off, size_t n, size_t n_vecs, int mont, SliceKind kind,
                const void* const* srcs, std::vector<host::HXYZZ<Fq>>* out) {
  const size_t N = n_shards(c);
  out->assign(n_vecs, host::hx_inf<Fq>());
  if (base_off > key->n) return AMSM_E_INVALID_ARG;
  const size_t n_eff = std::min(n, key->n - base_off);
  if (n_vecs == 0 || n_eff == 0) return AMSM_OK;
  const size_t rec = xyzz_bytes<Fq>(), per = n_vecs * rec;
  if (kind == SLICE_PRIMARY) HIP_TRY(hipEventRecord(c->multi_fork, c->stream));  // the producers of the vectors
  const bool rccl = c->collective == 1;
  TRY(for_each_shard(c, [&](size_t g) -> int {
    amsm_ctx* cg = c->shard_ctx[g];
    HIP_TRY(hipSetDevice(cg->device));
    const size_t lo = std::max(key->bound[g], base_off), hi = std::min(key->bound[g + 1], base_off + n_eff);
    const size_t cnt = hi > lo ? hi - lo : 0;
    TRY(ensure(cg->rec_send, per));
    if (rccl) TRY(ensure(cg->rec_recv, N * per));
    if (cnt == 0) {  // this shard holds none of the range: identity records (ZZ = 0)
      HIP_TRY(hipMemsetAsync(cg->rec_send.p, 0, per, cg->stream));
      HIP_TRY(hipStreamSynchronize(cg->stream));
      return AMSM_OK;
    }
    std::vector<const void*> ptrs(n_vecs);
    const size_t skip = (lo - base_off) * 32, bytes = cnt * 32;
    if (kind == SLICE_SHARDED) {
      for (size_t v = 0; v < n_vecs; v++) ptrs[v] = srcs[v * N + g];
    } else if (kind == SLICE_PRIMARY && g == 0) {
      for (size_t v = 0; v < n_vecs; v++) ptrs[v] = (const char*)srcs[v] + skip;
    } else {
      TRY(ensure(cg->stage, n_vecs * bytes));
      if (kind == SLICE_PRIMARY) HIP_TRY(hipStreamWaitEvent(cg->stream, c->multi_fork, 0));
      for (size_t v = 0; v < n_vecs; v++) {
        char* dst = (char*)cg->stage.p + v * bytes;
        const char* src = (const char*)srcs[v] + skip;
        if (kind == SLICE_HOST)
          HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, cg->stream));
        else
          HIP_TRY(hipMemcpyPeerAsync(dst, cg->device, src, c->device, bytes, cg->stream));
        ptrs[v] = dst;
      }
    }
    return msm_partial_batch_impl<Fq, Fr>(cg, key->shards[g], lo - key->bound[g], ptrs.data(), n_vecs, cnt, mont,
                                          cg->rec_send.p);
  }));
  // exchange: every shard's records -> the primary (rank-major: record (g, v) at (g * n_vecs + v) * rec)
  TRY(ensure(c->rec_recv, N * per));
  if (rccl) {
    const RcclApi& api = RcclApi::get();
    int e = api.group_start();
    for (size_t g = 0; g < N && e == 0; g++) {
      amsm_ctx* cg = c->shard_ctx[g];
      HIP_TRY(hipSetDevice(cg->device));
      e = api.all_gather(cg->rec_send.p, cg->rec_recv.p, per, kNcclUint8, c->rccl_comms[g], cg->stream);
    }
    int e2 = api.group_end();
    (void)hipSetDevice(c->device);
    if (e != 0 || e2 != 0) {
      fprintf(stderr, "[amsm] RCCL all-gather failed: %s\n", api.error_string ? api.error_string(e ? e : e2) : "?");
      return AMSM_E_RCCL;
    }
  } else {
    for (size_t g = 0; g < N; g++) {
      amsm_ctx* cg = c->shard_ctx[g];  // its records are complete: msm_partial_batch_impl synchronised cg's stream
      HIP_TRY(hipMemcpyPeerAsync((char*)c->rec_recv.p + g * per, c->device, cg->rec_send.p, cg->device, per, c->stream));
    }
  }
  Slot* sl = &c->slot[0];
  TRY(ensure_pinned(sl, N * per + 64));
  HIP_TRY(hipMemcpyAsync(sl->h_pinned, c->rec_recv.p, N * per, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const u32* h = (const u32*)sl->h_pinned;
  for (size_t v = 0; v < n_vecs; v++)
    for (size_t g = 0; g < N; g++)
      (*out)[v] = host::hx_add<Fq>((*out)[v], host::hx_from_device<Fq>(h + (g * n_vecs + v) * (rec / 4)));
  return AMSM_OK;
}

// A sharded key: make_shard(g, ctx_g, lo, cnt, &shard) builds shard g on its device.
template <class F>
int bases_create_sharded(amsm_ctx* c, size_t n, amsm_bases** out, F&& make_shard) {
  const size_t N = n_shards(c);
  amsm_bases* b = new (std::nothrow) amsm_bases();
  if (!b) return AMSM_E_OOM;
  b->curve = c->curve;
  b->device = c->device;
  b->n = n;
  b->owner = c;
  b->bound = shard_bounds(n, N);
  b->shards.assign(N, nullptr);
  int rc = for_each_shard(c, [&](size_t g) -> int {
    amsm_ctx* cg = c->shard_ctx[g];
    HIP_TRY(hipSetDevice(cg->device));
    return make_shard(g, cg, b->bound[g], b->bound[g + 1] - b->bound[g], &b->shards[g]);
  });
  if (rc != AMSM_OK) {
    for (amsm_bases* s : b->shards) amsm_bases_free(s);
    delete b;
    return rc;
  }
  b->precomp = 1;
  for (amsm_bases* s : b->shards) b->precomp &= s->precomp;
  *out = b;
  return AMSM_OK;
}

// Does (ctx, key) form a valid pair?  A sharded key needs the multi-device context that made it.
inline bool key_matches(const amsm_ctx* c, const amsm_bases* b) {
  if (b->curve != c->curve) return false;
  if (key_sharded(b)) return b->owner == c && b->shards.size() == n_shards(c);
  return b->device == c->device;
}

// commit(ck, v, r) over a sharded key: the sharded MSM, then + r * hiding_generator on the host (a11)
template <class Fq, class Fr>
int pedersen_sharded_impl(amsm_ctx* c, const amsm_bases* ck, size_t n, SliceKind kind, const void* const* src,
                                 const uint64_t* rand_mont, const uint64_t* hiding_xy, uint64_t* out_xy, uint8_t* out_inf) {
  std::vector<host::HXYZZ<Fq>> r;
  TRY((msm_sharded<Fq, Fr>(c, ck, 0, n, 1, 1, kind, src, &r)));
  host::HXYZZ<Fq> acc = r[0];
  if (rand_mont && hiding_xy) {
    host::HFe<Fr> k;
    memcpy(k.v, rand_mont, 32);
    k = host::h_from_mont<Fr>(k);
    acc = host::hx_add<Fq>(acc, host::hx_mul<Fq>(host::hx_from_affine<Fq>(hiding_xy, false), k.v));
  }
  write_affine<Fq>(acc, out_xy, out_inf);
  return AMSM_OK;
}
#define DISPATCH(ctx, CALL_P, CALL_B)                   \
  ((ctx)->curve == AMSM_PALLAS ? (CALL_P) : (CALL_B))

int bind_device(const amsm_ctx* ctx) {
  HIP_TRY(hipSetDevice(ctx->device));
  return AMSM_OK;
}

}  // namespace

// =============================================================================================
// extern "C"
// =============================================================================================
// Host scalar-field helpers (no device work): what a scheme driver needs for its O(#inputs) challenge arithmetic.
template <class Fr, class F>
static void fr_map2(const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out, F&& f) {
  for (size_t i = 0; i < n; i++) {
    host::HFe<Fr> x, y;
    memcpy(x.v, a + 4 * i, 32);
    if (b) memcpy(y.v, b + 4 * i, 32);
    host::HFe<Fr> r = f(x, y);
    memcpy(out + 4 * i, r.v, 32);
  }
}
#define AMSM_FR_OP(NAME, EXPR)                                                                             \
  static int NAME(int curve, const uint64_t* a, const uint64_t* b, size_t n, uint64_t* out) {                      \
    if ((curve != AMSM_PALLAS && curve != AMSM_BLS12_381_G1) || (n && (!a || !out))) return AMSM_E_INVALID_ARG; \
    if (curve == AMSM_PALLAS) {                                                                            \
      using F = PallasFr;                                                                                  \
      fr_map2<F>(a, b, n, out, [](const host::HFe<F>& x, const host::HFe<F>& y) { (void)y; return EXPR; });  \
    } else {                                                                                               \
      using F = Bls12381Fr;                                                                                \
      fr_map2<F>(a, b, n, out, [](const host::HFe<F>& x, const host::HFe<F>& y) { (void)y; return EXPR; });  \
    }                                                                                                      \
    return AMSM_OK;                                                                                        \
  }
AMSM_FR_OP(amsm_fr_mul_impl, host::h_mul<F>(x, y))
AMSM_FR_OP(amsm_fr_add_impl, host::h_add<F>(x, y))
AMSM_FR_OP(amsm_fr_sub_impl, host::h_sub<F>(x, y))
AMSM_FR_OP(amsm_fr_inv_impl, host::h_inv<F>(x))
AMSM_FR_OP(amsm_fr_to_mont_impl, host::h_to_mont<F>(x))
AMSM_FR_OP(amsm_fr_from_mont_impl, host::h_from_mont<F>(x))
extern "C" {

const char* amsm_strerror(int s) {
  switch (s) {
    case AMSM_OK: return "ok";
    case AMSM_E_INVALID_ARG: return "invalid argument";
    case AMSM_E_OOM: return "out of (device) memory";
    case AMSM_E_HIP: return "HIP runtime error";
    case AMSM_E_UNSUPPORTED: return "unsupported size or configuration";
    case AMSM_E_NO_DEVICE: return "no usable gfx950 device (there is no CPU fallback)";
    case AMSM_E_SCALAR_RANGE: return "scalar out of range (not a canonical into_repr value)";
    case AMSM_E_RCCL: return "RCCL collective failed";
    default: return "unknown error";
  }
}

int amsm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int amsm_ctx_create(amsm_ctx** out, int curve, int device_id, void* stream) {
  if (!out) return AMSM_E_INVALID_ARG;
  *out = nullptr;
  if (curve != AMSM_PALLAS && curve != AMSM_BLS12_381_G1) return AMSM_E_INVALID_ARG;
  int ndev = amsm_device_count();
  if (ndev <= 0) return AMSM_E_NO_DEVICE;
  if (device_id < 0 || device_id >= ndev) return AMSM_E_INVALID_ARG;
  HIP_TRY(hipSetDevice(device_id));
  amsm_ctx* c = new (std::nothrow) amsm_ctx();
  if (!c) return AMSM_E_OOM;
  c->curve = curve;
  c->device = device_id;
  bool ok = true;
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    ok = ok && hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    c->own_stream = ok;
  }
  {
    // prep and tail run beside another MSM's accumulate L0, which fills every wave slot: high priority lets their
    // kernels take the slots L0 frees first
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    ok = ok && hipStreamCreateWithPriority(&c->s_prep, hipStreamNonBlocking, prio_hi) == hipSuccess;
    ok = ok && hipStreamCreateWithPriority(&c->s_tail, hipStreamNonBlocking, prio_hi) == hipSuccess;
  }
  for (int k = 0; k < N_SLOTS && ok; k++) {
    for (int i = 0; i <= ST_COUNT && ok; i++) ok = hipEventCreate(&c->slot[k].ev[i]) == hipSuccess;
    ok = ok && hipEventCreate(&c->slot[k].ev_prep_end) == hipSuccess && hipEventCreate(&c->slot[k].ev_l0_end) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->slot[k].done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->slot[k].l0_done, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->slot[k].prep_done, hipEventDisableTiming) == hipSuccess;
  }
  ok = ok && hipEventCreateWithFlags(&c->fork, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ip_ready, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    amsm_ctx_destroy(c);
    return AMSM_E_HIP;
  }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0) {
      c->cu_count = prop.multiProcessorCount;
      if (const char* e = getenv("AMSM_L0_LDS_PAD")) c->l0_lds_pad = (u32)std::max(0, atoi(e));
      int per_cu = curve == AMSM_PALLAS ? accum_l0_blocks_per_cu<PallasFq>(c->l0_lds_pad)
                                        : accum_l0_blocks_per_cu<Bls12381Fq>(c->l0_lds_pad);
      if (getenv("AMSM_DEBUG")) fprintf(stderr, "[amsm] accumulate L0: %d workgroups per CU\n", per_cu);
      c->wave_slots = prop.multiProcessorCount * std::max(1, per_cu) * 4;
    }
  }
  if (const char* e = getenv("AMSM_PREP")) c->custom_prep = strcmp(e, "rocprim") != 0;
  if (const char* e = getenv("AMSM_K0")) c->K0 = std::max(0, atoi(e));
  if (const char* e = getenv("AMSM_K0_MAX")) c->K0_max = std::max(4, atoi(e));
  if (const char* e = getenv("AMSM_POOL_MAX_MB")) c->pool_cap_bytes = (size_t)std::max(0, atoi(e)) << 20;
  if (const char* e = getenv("AMSM_L0_SPREAD")) c->small_spread = atoi(e) != 0;
  if (const char* e = getenv("AMSM_K0_2PHASE")) c->two_phase = atoi(e) != 0;
  if (const char* e = getenv("AMSM_TAIL_QUAD")) c->tail_quad = atoi(e) != 0;
  if (const char* e = getenv("AMSM_ONE_STREAM")) c->one_stream = atoi(e) != 0;
  if (const char* e = getenv("AMSM_SPLIT_LOG2")) c->split_log2 = std::max(0, std::min(28, atoi(e)));
  if (const char* e = getenv("AMSM_SPLIT_MIN_LOG2")) c->split_min_log2 = std::max(10, std::min(40, atoi(e)));
  if (const char* e = getenv("AMSM_K1")) c->K1 = std::max(1, atoi(e));
  if (const char* e = getenv("AMSM_RED_S")) c->red_s = std::max(1, atoi(e));
  if (const char* e = getenv("AMSM_WINDOW")) c->window_override = atoi(e);
  *out = c;
  return AMSM_OK;
}

static void pool_release_all(amsm_ctx* c);
void amsm_ctx_destroy(amsm_ctx* c) {
  if (!c) return;
  if (c->parent) return;  // a shard context is borrowed: it goes with its parent
  for (ShardWorker* w : c->workers) delete w;  // joins the threads
  c->workers.clear();
  if (c->rccl_comms) {
    const RcclApi& api = RcclApi::get();
    for (size_t g = 0; g < c->shard_ctx.size(); g++)
      if (c->rccl_comms[g] && api.comm_destroy) (void)api.comm_destroy(c->rccl_comms[g]);
    delete[] c->rccl_comms;
    c->rccl_comms = nullptr;
  }
  for (size_t g = 1; g < c->shard_ctx.size(); g++) {
    c->shard_ctx[g]->parent = nullptr;
    amsm_ctx_destroy(c->shard_ctx[g]);
  }
  c->shard_ctx.clear();
  (void)hipSetDevice(c->device);
  if (c->s_prep) (void)hipStreamSynchronize(c->s_prep);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->s_tail) (void)hipStreamSynchronize(c->s_tail);
  for (int k = 0; k < N_SLOTS; k++) {
    Slot* sl = &c->slot[k];
    DevBuf* bufs[] = {&sl->keys_a, &sl->keys_b, &sl->vals_a, &sl->vals_b, &sl->start, &sl->items, &sl->item_off,
                      &sl->partials, &sl->buckets, &sl->red_out, &sl->fold_out, &sl->heavy, &sl->misc, &sl->sort_tmp,
                      &sl->scan_tmp, &sl->prep_small, &sl->heavy_scratch};
    for (DevBuf* b : bufs)
      if (b->p) (void)hipFree(b->p);
    if (sl->h_pinned) (void)hipHostFree(sl->h_pinned);
    for (int i = 0; i <= ST_COUNT; i++)
      if (sl->ev[i]) (void)hipEventDestroy(sl->ev[i]);
    if (sl->ev_prep_end) (void)hipEventDestroy(sl->ev_prep_end);
    if (sl->ev_l0_end) (void)hipEventDestroy(sl->ev_l0_end);
    if (sl->done) (void)hipEventDestroy(sl->done);
    if (sl->l0_done) (void)hipEventDestroy(sl->l0_done);
    if (sl->prep_done) (void)hipEventDestroy(sl->prep_done);
  }
  if (c->fork) (void)hipEventDestroy(c->fork);
  if (c->ip_ready) (void)hipEventDestroy(c->ip_ready);
  if (c->s_prep) (void)hipStreamDestroy(c->s_prep);
  if (c->s_tail) (void)hipStreamDestroy(c->s_tail);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  if (c->scalars.p) (void)hipFree(c->scalars.p);
  if (c->xyzz_scratch.p) (void)hipFree(c->xyzz_scratch.p);
  pool_release_all(c);
  for (DevBuf* b : {&c->rec_send, &c->rec_recv, &c->stage})
    if (b->p) (void)hipFree(b->p);
  if (c->multi_fork) (void)hipEventDestroy(c->multi_fork);
  delete c;
}

int amsm_ctx_create_multi(amsm_ctx** out, int curve, const int* device_ids, int n_dev) {
  if (!out) return AMSM_E_INVALID_ARG;
  *out = nullptr;
  if (!device_ids || n_dev < 1 || n_dev > 64) return AMSM_E_INVALID_ARG;
  amsm_ctx* c = nullptr;
  TRY(amsm_ctx_create(&c, curve, device_ids[0], nullptr));
  c->shard_ctx.push_back(c);
  bool ok = hipEventCreateWithFlags(&c->multi_fork, hipEventDisableTiming) == hipSuccess;
  for (int g = 1; g < n_dev && ok; g++) {
    amsm_ctx* cg = nullptr;
    int rc = amsm_ctx_create(&cg, curve, device_ids[g], nullptr);
    if (rc != AMSM_OK) {
      amsm_ctx_destroy(c);
      return rc;
    }
    cg->parent = c;
    c->shard_ctx.push_back(cg);
    c->workers.push_back(new ShardWorker());
  }
  if (!ok) {
    amsm_ctx_destroy(c);
    return AMSM_E_HIP;
  }
  (void)hipSetDevice(c->device);
  if (n_dev > 1) {
    bool distinct = true;
    for (int a = 0; a < n_dev; a++)
      for (int b = a + 1; b < n_dev; b++) distinct = distinct && device_ids[a] != device_ids[b];
    const char* want = getenv("AMSM_COLLECTIVE");  // "rccl" (fail if unavailable) | "peer" | unset (RCCL when possible)
    const bool force_rccl = want && strcmp(want, "rccl") == 0, force_peer = want && strcmp(want, "peer") == 0;
    c->collective = 2;
    if (!force_peer && distinct) {
      const RcclApi& api = RcclApi::get();
      int e = -1;
      if (api.ok) {
        c->rccl_comms = new void*[n_dev]();
        e = api.comm_init_all(c->rccl_comms, n_dev, device_ids);
        (void)hipSetDevice(c->device);
        if (e != 0) {
          fprintf(stderr, "[amsm] ncclCommInitAll failed (%s)%s\n", api.error_string ? api.error_string(e) : "?",
                  force_rccl ? "" : ": falling back to peer copies");
          delete[] c->rccl_comms;
          c->rccl_comms = nullptr;
        }
      }
      if (e == 0) c->collective = 1;
    }
    if (c->collective != 1 && force_rccl) {
      amsm_ctx_destroy(c);
      return AMSM_E_RCCL;
    }
    if (c->collective == 2 && distinct) {  // direct peer copies where the fabric allows (xGMI); else HIP stages them
      for (int a = 0; a < n_dev; a++)
        for (int b = 0; b < n_dev; b++) {
          int can = 0;
          if (a != b && hipDeviceCanAccessPeer(&can, device_ids[a], device_ids[b]) == hipSuccess && can) {
            (void)hipSetDevice(device_ids[a]);
            (void)hipDeviceEnablePeerAccess(device_ids[b], 0);
            (void)hipGetLastError();
          }
        }
      (void)hipSetDevice(c->device);
    }
  }
  *out = c;
  return AMSM_OK;
}

int amsm_ctx_num_devices(const amsm_ctx* c) { return c ? (int)n_shards(c) : AMSM_E_INVALID_ARG; }
amsm_ctx* amsm_ctx_shard(amsm_ctx* c, int g) {
  if (!c || g < 0 || (size_t)g >= n_shards(c)) return nullptr;
  return c->shard_ctx.empty() ? c : c->shard_ctx[g];
}
const char* amsm_ctx_collective(const amsm_ctx* c) {
  if (!c) return "none";
  return c->collective == 1 ? "rccl" : (c->collective == 2 ? "peer-copy" : "none");
}

int amsm_ctx_curve(const amsm_ctx* c) { return c ? c->curve : AMSM_E_INVALID_ARG; }
int amsm_ctx_fq_limbs(const amsm_ctx* c) { return !c ? AMSM_E_INVALID_ARG : (c->curve == AMSM_PALLAS ? 4 : 6); }
int amsm_ctx_set_window(amsm_ctx* c, int bits) {
  if (!c || (bits != 0 && (bits < 2 || bits > 24))) return AMSM_E_INVALID_ARG;
  c->window_override = bits;
  return AMSM_OK;
}
int amsm_ctx_synchronize(amsm_ctx* c) {
  if (!c) return AMSM_E_INVALID_ARG;
  for (size_t g = 1; g < c->shard_ctx.size(); g++) TRY(amsm_ctx_synchronize(c->shard_ctx[g]));
  TRY(bind_device(c));
  HIP_TRY(hipStreamSynchronize(c->s_prep));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipStreamSynchronize(c->s_tail));
  return AMSM_OK;
}
int amsm_ctx_set_profiling(amsm_ctx* c, int on) {
  if (!c) return AMSM_E_INVALID_ARG;
  c->profiling = on != 0;
  return AMSM_OK;
}
int amsm_stage_count(void) { return ST_COUNT; }
const char* amsm_stage_name(int s) { return (s >= 0 && s < ST_COUNT) ? kStageNames[s] : ""; }
int amsm_ctx_stage_ms(amsm_ctx* c, int s, float* ms) {
  if (!c || !ms || s < 0 || s >= ST_COUNT) return AMSM_E_INVALID_ARG;
  *ms = c->stage_ms[s];
  return AMSM_OK;
}

int amsm_bases_load(amsm_ctx* c, const uint64_t* xy, const uint8_t* is_inf, size_t n, unsigned flags, amsm_bases** out) {
  if (!c || !out || (n && !xy)) return AMSM_E_INVALID_ARG;
  if (n >= (1ull << 31)) return AMSM_E_UNSUPPORTED;
  TRY(bind_device(c));
  if (n_shards(c) > 1) {  // shard g copies its own range of the caller's arrays
    const size_t L2 = 2 * (size_t)amsm_ctx_fq_limbs(c);
    return bases_create_sharded(c, n, out, [&](size_t, amsm_ctx* cg, size_t lo, size_t cnt, amsm_bases** o) {
      // the impl, not the entry point: shard 0's context IS this (multi-device) context
      return DISPATCH(cg, (bases_load_impl<PallasFq, PallasFr>(cg, xy + lo * L2, is_inf ? is_inf + lo : nullptr, cnt, flags, o)),
                      (bases_load_impl<Bls12381Fq, Bls12381Fr>(cg, xy + lo * L2, is_inf ? is_inf + lo : nullptr, cnt, flags, o)));
    });
  }
  return DISPATCH(c, (bases_load_impl<PallasFq, PallasFr>(c, xy, is_inf, n, flags, out)),
                  (bases_load_impl<Bls12381Fq, Bls12381Fr>(c, xy, is_inf, n, flags, out)));
}
int amsm_bases_generate(amsm_ctx* c, uint64_t seed, size_t n, unsigned flags, amsm_bases** out) {
  if (!c || !out) return AMSM_E_INVALID_ARG;
  if (n >= (1ull << 31)) return AMSM_E_UNSUPPORTED;
  TRY(bind_device(c));
  if (n_shards(c) > 1)  // shard g generates its own range of the synthetic stream
    return bases_create_sharded(c, n, out, [&](size_t, amsm_ctx* cg, size_t lo, size_t cnt, amsm_bases** o) {
      return DISPATCH(cg, (bases_generate_impl<PallasFq, PallasFr>(cg, seed, cnt, flags, o, lo)),
                      (bases_generate_impl<Bls12381Fq, Bls12381Fr>(cg, seed, cnt, flags, o, lo)));
    });
  return DISPATCH(c, (bases_generate_impl<PallasFq, PallasFr>(c, seed, n, flags, out)),
                  (bases_generate_impl<Bls12381Fq, Bls12381Fr>(c, seed, n, flags, out)));
}
int amsm_bases_read(amsm_ctx* c, const amsm_bases* b, size_t off, size_t n, uint64_t* xy, uint8_t* is_inf) {
  if (!c || !b || (n && !xy) || b->curve != c->curve) return AMSM_E_INVALID_ARG;
  if (key_sharded(b)) {
    if (!key_matches(c, b) || off > b->n || n > b->n - off) return AMSM_E_INVALID_ARG;
    const size_t L2 = 2 * (size_t)amsm_ctx_fq_limbs(c);
    for (size_t g = 0; g < b->shards.size(); g++) {
      const size_t lo = std::max(b->bound[g], off), hi = std::min(b->bound[g + 1], off + n);
      if (hi > lo)
        TRY(amsm_bases_read(c->shard_ctx[g], b->shards[g], lo - b->bound[g], hi - lo, xy + (lo - off) * L2,
                            is_inf ? is_inf + (lo - off) : nullptr));
    }
    return bind_device(c);
  }
  TRY(bind_device(c));
  return DISPATCH(c, (bases_read_impl<PallasFq>(c, b, off, n, xy, is_inf)),
                  (bases_read_impl<Bls12381Fq>(c, b, off, n, xy, is_inf)));
}
size_t amsm_bases_len(const amsm_bases* b) { return b ? b->n : 0; }
int amsm_bases_precomputed(const amsm_bases* b) { return b ? b->precomp : 0; }
int amsm_bases_window_bits(const amsm_bases* b) {
  if (!b) return 0;
  if (!b->shards.empty()) return b->shards[0] ? b->shards[0]->c : 0;
  return b->precomp ? b->c : 0;
}
int amsm_bases_num_shards(const amsm_bases* b) { return !b ? 0 : (key_sharded(b) ? (int)b->shards.size() : 1); }
int amsm_bases_shard_range(const amsm_bases* b, int g, size_t* lo, size_t* hi) {
  if (!b || !lo || !hi || g < 0 || g >= amsm_bases_num_shards(b)) return AMSM_E_INVALID_ARG;
  *lo = key_sharded(b) ? b->bound[g] : 0;
  *hi = key_sharded(b) ? b->bound[g + 1] : b->n;
  return AMSM_OK;
}
void amsm_bases_free(amsm_bases* b) {
  if (!b) return;
  for (amsm_bases* s : b->shards) amsm_bases_free(s);
  b->shards.clear();
  (void)hipSetDevice(b->device);
  if (b->ready) {
    (void)hipEventSynchronize(b->ready);
    (void)hipEventDestroy(b->ready);
  }
  if (b->d_table) (void)hipFree(b->d_table);
  if (b->d_abi) (void)hipFree(b->d_abi);
  delete b;
}

// n_vecs MSMs over a sharded key -> n_vecs affine points (shared by every entry point that accepts one)
static int msm_sharded_affine(amsm_ctx* c, const amsm_bases* b, size_t off, size_t n, size_t n_vecs, int mont, SliceKind kind,
                              const void* const* srcs, uint64_t* out_xy, uint8_t* out_inf) {
  TRY(bind_device(c));
  if (c->curve == AMSM_PALLAS) {
    std::vector<host::HXYZZ<PallasFq>> r;
    TRY((msm_sharded<PallasFq, PallasFr>(c, b, off, n, n_vecs, mont, kind, srcs, &r)));
    write_affine_batch<PallasFq>(r, out_xy, out_inf);
  } else {
    std::vector<host::HXYZZ<Bls12381Fq>> r;
    TRY((msm_sharded<Bls12381Fq, Bls12381Fr>(c, b, off, n, n_vecs, mont, kind, srcs, &r)));
    write_affine_batch<Bls12381Fq>(r, out_xy, out_inf);
  }
  return AMSM_OK;
}

int amsm_msm(amsm_ctx* c, const amsm_bases* b, size_t off, const uint64_t* scalars, size_t n, int mont, uint64_t* out_xy,
             uint8_t* out_inf) {
  if (!c || !b || !out_xy || (n && !scalars) || !key_matches(c, b)) return AMSM_E_INVALID_ARG;
  if (key_sharded(b)) {
    const void* src = scalars;
    return msm_sharded_affine(c, b, off, n, 1, mont, SLICE_HOST, &src, out_xy, out_inf);
  }
  TRY(bind_device(c));
  if (c->curve == AMSM_PALLAS) {
    host::HXYZZ<PallasFq> r;
    TRY((msm_host_scalars<PallasFq, PallasFr>(c, b, off, scalars, n, mont, &r)));
    write_affine<PallasFq>(r, out_xy, out_inf);
  } else {
    host::HXYZZ<Bls12381Fq> r;
    TRY((msm_host_scalars<Bls12381Fq, Bls12381Fr>(c, b, off, scalars, n, mont, &r)));
    write_affine<Bls12381Fq>(r, out_xy, out_inf);
  }
  return AMSM_OK;
}

int amsm_msm_device(amsm_ctx* c, const amsm_bases* b, size_t off, const void* d_scalars, size_t n, int mont,
                    uint64_t* out_xy, uint8_t* out_inf) {
  if (!c || !b || !out_xy || (n && !d_scalars) || !key_matches(c, b)) return AMSM_E_INVALID_ARG;
  if (key_sharded(b)) return msm_sharded_affine(c, b, off, n, 1, mont, SLICE_PRIMARY, &d_scalars, out_xy, out_inf);
  TRY(bind_device(c));
  if (c->curve == AMSM_PALLAS) {
    host::HXYZZ<PallasFq> r;
    TRY((msm_device_xyzz<PallasFq, PallasFr>(c, b, off, d_scalars, n, mont, &r)));
    write_affine<PallasFq>(r, out_xy, out_inf);
  } else {
    host::HXYZZ<Bls12381Fq> r;
    TRY((msm_device_xyzz<Bls12381Fq, Bls12381Fr>(c, b, off, d_scalars, n, mont, &r)));
    write_affine<Bls12381Fq>(r, out_xy, out_inf);
  }
  return AMSM_OK;
}

int amsm_msm_batch_device(amsm_ctx* c, const amsm_bases* b, size_t off, const void* const* d_scalars, size_t n_vecs,
                          size_t n, int mont, uint64_t* out_xy, uint8_t* out_inf) {
  if (!c || !b || (n_vecs && (!d_scalars || !out_xy)) || !key_matches(c, b)) return AMSM_E_INVALID_ARG;
  for (size_t v = 0; v < n_vecs; v++)
    if (n && !d_scalars[v]) return AMSM_E_INVALID_ARG;
  if (key_sharded(b)) return msm_sharded_affine(c, b, off, n, n_vecs, mont, SLICE_PRIMARY, d_scalars, out_xy, out_inf);
  TRY(bind_device(c));
  if (c->curve == AMSM_PALLAS) {
    std::vector<host::HXYZZ<PallasFq>> r;
    TRY((msm_batch_xyzz<PallasFq, PallasFr>(c, b, off, d_scalars, n_vecs, n, mont, &r)));
    write_affine_batch<PallasFq>(r, out_xy, out_inf);
  } else {
    std::vector<host::HXYZZ<Bls12381Fq>> r;
    TRY((msm_batch_xyzz<Bls12381Fq, Bls12381Fr>(c, b, off, d_scalars, n_vecs, n, mont, &r)));
    write_affine_batch<Bls12381Fq>(r, out_xy, out_inf);
  }
  return AMSM_OK;
}

int amsm_msm_multi_device(amsm_ctx* c, const amsm_bases* b, size_t n_msms, const size_t* base_offs,
                          const void* const* d_scalars, const size_t* ns, int mont, uint64_t* out_xy, uint8_t* out_inf) {
  if (!c || !b || (n_msms && (!base_offs || !d_scalars || !ns || !out_xy)) || b->curve != c->curve ||
      (!key_sharded(b) && b->device != c->device))
    return AMSM_E_INVALID_ARG;
  if (key_sharded(b)) return AMSM_E_UNSUPPORTED;
  for (size_t v = 0; v < n_msms; v++)
    if (ns[v] && !d_scalars[v]) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  if (c->curve == AMSM_PALLAS) {
    std::vector<host::HXYZZ<PallasFq>> r;
    TRY((msm_multi_xyzz<PallasFq, PallasFr>(c, b, n_msms, base_offs, d_scalars, ns, mont, &r)));
    write_affine_batch<PallasFq>(r, out_xy, out_inf);
  } else {
    std::vector<host::HXYZZ<Bls12381Fq>> r;
    TRY((msm_multi_xyzz<Bls12381Fq, Bls12381Fr>(c, b, n_msms, base_offs, d_scalars, ns, mont, &r)));
    write_affine_batch<Bls12381Fq>(r, out_xy, out_inf);
  }
  return AMSM_OK;
}

int amsm_msm_grouped_device(amsm_ctx* c, const amsm_bases* b, size_t off, const void* d_scalars, size_t n, int mont,
                            unsigned group_shift, uint64_t* out_xy, uint8_t* out_inf) {
  if (!c || !b || !out_xy || (n && !d_scalars) || group_shift > 31 || b->curve != c->curve || b->device != c->device)
    return AMSM_E_INVALID_ARG;
  if (key_sharded(b)) return AMSM_E_UNSUPPORTED;  // does not shard (include/amsm.h: amsm_ctx_create_multi)
  TRY(bind_device(c));
  return DISPATCH(c, (msm_grouped_impl<PallasFq, PallasFr>(c, b, off, d_scalars, n, mont, group_shift, out_xy, out_inf)),
                  (msm_grouped_impl<Bls12381Fq, Bls12381Fr>(c, b, off, d_scalars, n, mont, group_shift, out_xy, out_inf)));
}

size_t amsm_partial_bytes(const amsm_ctx* c) {
  if (!c) return 0;
  return c->curve == AMSM_PALLAS ? xyzz_bytes<PallasFq>() : xyzz_bytes<Bls12381Fq>();
}

int amsm_msm_partial_device(amsm_ctx* c, const amsm_bases* b, size_t off, const void* d_scalars, size_t n, int mont,
                            void* d_out) {
  if (!c || !b || !d_out || (n && !d_scalars) || b->curve != c->curve || b->device != c->device)
    return AMSM_E_INVALID_ARG;
  if (key_sharded(b)) return AMSM_E_UNSUPPORTED;  // does not shard (include/amsm.h: amsm_ctx_create_multi)
  TRY(bind_device(c));
  return DISPATCH(c, (msm_partial_impl<PallasFq, PallasFr>(c, b, off, d_scalars, n, mont, d_out)),
                  (msm_partial_impl<Bls12381Fq, Bls12381Fr>(c, b, off, d_scalars, n, mont, d_out)));
}

int amsm_msm_partial_batch_device(amsm_ctx* c, const amsm_bases* b, size_t off, const void* const* d_scalars, size_t n_vecs,
                                  size_t n, int mont, void* d_out) {
  if (!c || !b || (n_vecs && (!d_scalars || !d_out)) || b->curve != c->curve || b->device != c->device)
    return AMSM_E_INVALID_ARG;
  for (size_t v = 0; v < n_vecs; v++)
    if (n && !d_scalars[v]) return AMSM_E_INVALID_ARG;
  if (key_sharded(b)) return AMSM_E_UNSUPPORTED;  // does not shard (include/amsm.h: amsm_ctx_create_multi)
  TRY(bind_device(c));
  return DISPATCH(c, (msm_partial_batch_impl<PallasFq, PallasFr>(c, b, off, d_scalars, n_vecs, n, mont, d_out)),
                  (msm_partial_batch_impl<Bls12381Fq, Bls12381Fr>(c, b, off, d_scalars, n_vecs, n, mont, d_out)));
}

int amsm_partials_combine_batch(amsm_ctx* c, const void* d_partials, size_t n_groups, size_t count, uint64_t* out_xy,
                                uint8_t* out_inf) {
  if (!c || (n_groups && !out_xy) || (n_groups && count && !d_partials)) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  return DISPATCH(c, (partials_combine_batch_impl<PallasFq>(c, d_partials, n_groups, count, out_xy, out_inf)),
                  (partials_combine_batch_impl<Bls12381Fq>(c, d_partials, n_groups, count, out_xy, out_inf)));
}

int amsm_partials_combine(amsm_ctx* c, const void* d_partials, size_t count, uint64_t* out_xy, uint8_t* out_inf) {
  if (!c || !out_xy || (count && !d_partials)) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  return DISPATCH(c, (partials_combine_impl<PallasFq>(c, d_partials, count, out_xy, out_inf)),
                  (partials_combine_impl<Bls12381Fq>(c, d_partials, count, out_xy, out_inf)));
}

static int pedersen_sharded(amsm_ctx* c, const amsm_bases* ck, size_t n, SliceKind kind, const void* const* src,
                            const uint64_t* rand_mont, const uint64_t* hiding_xy, uint64_t* out_xy, uint8_t* out_inf) {
  TRY(bind_device(c));
  return DISPATCH(c, (pedersen_sharded_impl<PallasFq, PallasFr>(c, ck, n, kind, src, rand_mont, hiding_xy, out_xy, out_inf)),
                  (pedersen_sharded_impl<Bls12381Fq, Bls12381Fr>(c, ck, n, kind, src, rand_mont, hiding_xy, out_xy, out_inf)));
}

int amsm_msm_batch_sharded_device(amsm_ctx* c, const amsm_bases* b, const void* const* d_slices, size_t n_vecs, int mont,
                                  uint64_t* out_xy, uint8_t* out_inf) {
  if (!c || !b || (n_vecs && (!d_slices || !out_xy)) || !key_matches(c, b)) return AMSM_E_INVALID_ARG;
  if (!key_sharded(b)) return amsm_msm_batch_device(c, b, 0, d_slices, n_vecs, b->n, mont, out_xy, out_inf);
  const size_t N = b->shards.size();
  for (size_t v = 0; v < n_vecs; v++)
    for (size_t g = 0; g < N; g++)
      if (b->bound[g + 1] > b->bound[g] && !d_slices[v * N + g]) return AMSM_E_INVALID_ARG;
  return msm_sharded_affine(c, b, 0, b->n, n_vecs, mont, SLICE_SHARDED, d_slices, out_xy, out_inf);
}

int amsm_pedersen_commit(amsm_ctx* c, const amsm_bases* ck, const uint64_t* elems, size_t n, const uint64_t* rand_mont,
                         const uint64_t* hiding_xy, uint64_t* out_xy, uint8_t* out_inf) {
  if (!c || !ck || !out_xy || (n && !elems) || !key_matches(c, ck)) return AMSM_E_INVALID_ARG;
  if ((rand_mont == nullptr) != (hiding_xy == nullptr)) return AMSM_E_INVALID_ARG;
  if (key_sharded(ck)) {
    const void* src = elems;
    return pedersen_sharded(c, ck, n, SLICE_HOST, &src, rand_mont, hiding_xy, out_xy, out_inf);
  }
  TRY(bind_device(c));
  return DISPATCH(c, (pedersen_impl<PallasFq, PallasFr>(c, ck, elems, n, rand_mont, hiding_xy, out_xy, out_inf)),
                  (pedersen_impl<Bls12381Fq, Bls12381Fr>(c, ck, elems, n, rand_mont, hiding_xy, out_xy, out_inf)));
}

int amsm_pedersen_commit_device(amsm_ctx* c, const amsm_bases* ck, const void* d_elems, size_t n,
                                const uint64_t* rand_mont, const uint64_t* hiding_xy, uint64_t* out_xy, uint8_t* out_inf) {
  if (!c || !ck || !out_xy || (n && !d_elems) || !key_matches(c, ck)) return AMSM_E_INVALID_ARG;
  if ((rand_mont == nullptr) != (hiding_xy == nullptr)) return AMSM_E_INVALID_ARG;
  if (key_sharded(ck)) return pedersen_sharded(c, ck, n, SLICE_PRIMARY, &d_elems, rand_mont, hiding_xy, out_xy, out_inf);
  TRY(bind_device(c));
  return DISPATCH(c, (pedersen_device_impl<PallasFq, PallasFr>(c, ck, d_elems, n, rand_mont, hiding_xy, out_xy, out_inf)),
                  (pedersen_device_impl<Bls12381Fq, Bls12381Fr>(c, ck, d_elems, n, rand_mont, hiding_xy, out_xy,
                                                                out_inf)));
}

int amsm_host_lincomb(int curve, const uint64_t* xy, const uint8_t* is_inf, const uint64_t* scalars_mont, size_t n,
                      uint64_t* out_xy, uint8_t* out_inf) {
  if (!out_xy || (n && (!xy || !scalars_mont))) return AMSM_E_INVALID_ARG;
  if (curve == AMSM_PALLAS) return host_lincomb_impl<PallasFq, PallasFr>(xy, is_inf, scalars_mont, n, out_xy, out_inf);
  if (curve == AMSM_BLS12_381_G1)
    return host_lincomb_impl<Bls12381Fq, Bls12381Fr>(xy, is_inf, scalars_mont, n, out_xy, out_inf);
  return AMSM_E_INVALID_ARG;
}

int amsm_host_lincomb_batch(int curve, size_t n_jobs, const size_t* n_terms, const uint64_t* const* xy, const uint8_t* const* is_inf,
                            const uint64_t* const* scalars_mont, uint64_t* out_xy, uint8_t* out_inf) {
  if (n_jobs && (!n_terms || !xy || !scalars_mont || !out_xy)) return AMSM_E_INVALID_ARG;
  for (size_t j = 0; j < n_jobs; j++)
    if (n_terms[j] && (!xy[j] || !scalars_mont[j])) return AMSM_E_INVALID_ARG;
  if (curve == AMSM_PALLAS)
    return host_lincomb_batch_impl<PallasFq, PallasFr>(n_jobs, n_terms, xy, is_inf, scalars_mont, out_xy, out_inf);
  if (curve == AMSM_BLS12_381_G1)
    return host_lincomb_batch_impl<Bls12381Fq, Bls12381Fr>(n_jobs, n_terms, xy, is_inf, scalars_mont, out_xy, out_inf);
  return AMSM_E_INVALID_ARG;
}

int amsm_fr_mul(int curve, const uint64_t* a_mont, const uint64_t* b_mont, size_t n, uint64_t* out_mont) {
  if (n && !b_mont) return AMSM_E_INVALID_ARG;
  return amsm_fr_mul_impl(curve, a_mont, b_mont, n, out_mont);
}
int amsm_fr_add(int curve, const uint64_t* a_mont, const uint64_t* b_mont, size_t n, uint64_t* out_mont) {
  if (n && !b_mont) return AMSM_E_INVALID_ARG;
  return amsm_fr_add_impl(curve, a_mont, b_mont, n, out_mont);
}
int amsm_fr_sub(int curve, const uint64_t* a_mont, const uint64_t* b_mont, size_t n, uint64_t* out_mont) {
  if (n && !b_mont) return AMSM_E_INVALID_ARG;
  return amsm_fr_sub_impl(curve, a_mont, b_mont, n, out_mont);
}
int amsm_fr_inv(int curve, const uint64_t* a_mont, size_t n, uint64_t* out_mont) {
  return amsm_fr_inv_impl(curve, a_mont, nullptr, n, out_mont);
}
int amsm_fr_to_mont(int curve, const uint64_t* canonical, size_t n, uint64_t* out_mont) {
  return amsm_fr_to_mont_impl(curve, canonical, nullptr, n, out_mont);
}
int amsm_fr_from_mont(int curve, const uint64_t* a_mont, size_t n, uint64_t* out_canonical) {
  return amsm_fr_from_mont_impl(curve, a_mont, nullptr, n, out_canonical);
}

// ---- wire format (host_serialize.h) ------------------------------------------------------------------------------
size_t amsm_fr_serialized_size(int curve) {
  if (curve == AMSM_PALLAS) return host::h_serialized_size<PallasFr>(0);
  if (curve == AMSM_BLS12_381_G1) return host::h_serialized_size<Bls12381Fr>(0);
  return 0;
}
size_t amsm_point_serialized_size(int curve, int compressed) {
  if (curve == AMSM_PALLAS) return host::point_serialized_size<PallasFq>(compressed != 0);
  if (curve == AMSM_BLS12_381_G1) return host::point_serialized_size<Bls12381Fq>(compressed != 0);
  return 0;
}
int amsm_fr_serialize(int curve, const uint64_t* a_mont, size_t n, uint8_t* out) {
  if ((curve != AMSM_PALLAS && curve != AMSM_BLS12_381_G1) || (n && (!a_mont || !out))) return AMSM_E_INVALID_ARG;
  for (size_t i = 0; i < n; i++) {
    if (curve == AMSM_PALLAS) {
      host::HFe<PallasFr> x;
      memcpy(x.v, a_mont + 4 * i, 32);
      host::h_write_le<PallasFr>(x, out + 32 * i, 32);
    } else {
      host::HFe<Bls12381Fr> x;
      memcpy(x.v, a_mont + 4 * i, 32);
      host::h_write_le<Bls12381Fr>(x, out + 32 * i, 32);
    }
  }
  return AMSM_OK;
}
int amsm_fr_deserialize(int curve, const uint8_t* in, size_t n, uint64_t* out_mont) {
  if ((curve != AMSM_PALLAS && curve != AMSM_BLS12_381_G1) || (n && (!in || !out_mont))) return AMSM_E_INVALID_ARG;
  for (size_t i = 0; i < n; i++) {
    bool ok;
    if (curve == AMSM_PALLAS) {
      host::HFe<PallasFr> x;
      ok = host::h_read_le<PallasFr>(in + 32 * i, 32, &x);
      memcpy(out_mont + 4 * i, x.v, 32);
    } else {
      host::HFe<Bls12381Fr> x;
      ok = host::h_read_le<Bls12381Fr>(in + 32 * i, 32, &x);
      memcpy(out_mont + 4 * i, x.v, 32);
    }
    if (!ok) return AMSM_E_INVALID_ARG;
  }
  return AMSM_OK;
}
int amsm_points_serialize(int curve, const uint64_t* xy_mont, const uint8_t* is_inf, size_t n, int compressed, uint8_t* out) {
  if ((curve != AMSM_PALLAS && curve != AMSM_BLS12_381_G1) || (n && (!xy_mont || !out))) return AMSM_E_INVALID_ARG;
  const size_t sz = amsm_point_serialized_size(curve, compressed);
  for (size_t i = 0; i < n; i++) {
    const bool inf = is_inf && is_inf[i];
    if (curve == AMSM_PALLAS) host::point_serialize<PallasFq>(xy_mont + i * 8, inf, compressed != 0, out + i * sz);
    else host::point_serialize<Bls12381Fq>(xy_mont + i * 12, inf, compressed != 0, out + i * sz);
  }
  return AMSM_OK;
}
int amsm_points_deserialize(int curve, const uint8_t* in, size_t n, int compressed, uint64_t* xy_mont, uint8_t* is_inf) {
  if ((curve != AMSM_PALLAS && curve != AMSM_BLS12_381_G1) || (n && (!in || !xy_mont || !is_inf))) return AMSM_E_INVALID_ARG;
  const size_t sz = amsm_point_serialized_size(curve, compressed);
  for (size_t i = 0; i < n; i++) {
    bool ok;
    if (curve == AMSM_PALLAS) {  // y^2 = x^3 + 5, cofactor 1
      ok = host::point_deserialize<PallasFq>(in + i * sz, compressed != 0, 5, false, nullptr, xy_mont + i * 8, is_inf + i);
    } else {  // y^2 = x^3 + 4; G1 is the order-r subgroup (cofactor != 1)
      u64 r[4];
      for (int k = 0; k < 4; k++) r[k] = host::hmod<Bls12381Fr>(k);
      ok = host::point_deserialize<Bls12381Fq>(in + i * sz, compressed != 0, 4, true, r, xy_mont + i * 12, is_inf + i);
    }
    if (!ok) return AMSM_E_INVALID_ARG;
  }
  return AMSM_OK;
}

// ---- Poseidon sponge (host_poseidon.h) ---------------------------------------------------------------------------------
#define SPONGE_DO(s, EXPR_P, EXPR_B) \
  do {                               \
    if ((s)->curve == AMSM_PALLAS) { \
      auto& sp = (s)->pallas;        \
      using FQ = PallasFq;           \
      (void)sizeof(FQ);              \
      EXPR_P;                        \
    } else {                         \
      auto& sp = (s)->bls;           \
      using FQ = Bls12381Fq;         \
      (void)sizeof(FQ);              \
      EXPR_B;                        \
    }                                \
  } while (0)
int amsm_poseidon_new(int curve, amsm_sponge** out) {
  if (!out || (curve != AMSM_PALLAS && curve != AMSM_BLS12_381_G1)) return AMSM_E_INVALID_ARG;
  amsm_sponge* s = new (std::nothrow) amsm_sponge();
  if (!s) return AMSM_E_OOM;
  s->curve = curve;
  *out = s;
  return AMSM_OK;
}
int amsm_poseidon_clone(const amsm_sponge* s, amsm_sponge** out) {
  if (!s || !out) return AMSM_E_INVALID_ARG;
  amsm_sponge* c = new (std::nothrow) amsm_sponge(*s);
  if (!c) return AMSM_E_OOM;
  *out = c;
  return AMSM_OK;
}
void amsm_poseidon_free(amsm_sponge* s) { delete s; }
int amsm_poseidon_absorb_native(amsm_sponge* s, const uint64_t* fq_mont, size_t n) {
  if (!s || (n && !fq_mont)) return AMSM_E_INVALID_ARG;
  SPONGE_DO(s, sp.absorb_words(fq_mont, n), sp.absorb_words(fq_mont, n));
  return AMSM_OK;
}
int amsm_poseidon_absorb_u64(amsm_sponge* s, uint64_t v) {
  if (!s) return AMSM_E_INVALID_ARG;
  SPONGE_DO(s, sp.absorb_u64(v), sp.absorb_u64(v));
  return AMSM_OK;
}
int amsm_poseidon_absorb_bytes(amsm_sponge* s, const uint8_t* b, size_t n) {
  if (!s || (n && !b)) return AMSM_E_INVALID_ARG;
  SPONGE_DO(s, sp.absorb_bytes(b, n), sp.absorb_bytes(b, n));
  return AMSM_OK;
}
int amsm_poseidon_absorb_points(amsm_sponge* s, const uint64_t* xy_mont, const uint8_t* is_inf, size_t n) {
  if (!s || (n && !xy_mont)) return AMSM_E_INVALID_ARG;
  SPONGE_DO(s, sp.absorb_points(xy_mont, is_inf, n), sp.absorb_points(xy_mont, is_inf, n));
  return AMSM_OK;
}
int amsm_poseidon_fork(const amsm_sponge* s, const uint8_t* domain, size_t n, amsm_sponge** out) {
  if (!s || !out || (n && !domain)) return AMSM_E_INVALID_ARG;
  TRY(amsm_poseidon_clone(s, out));
  std::vector<uint8_t> input(8 + n);  // `domain.len()` as u64 little-endian, then the domain
  for (int i = 0; i < 8; i++) input[i] = (uint8_t)((uint64_t)n >> (8 * i));
  if (n) memcpy(input.data() + 8, domain, n);
  return amsm_poseidon_absorb_bytes(*out, input.data(), input.size());
}
int amsm_poseidon_squeeze_native(amsm_sponge* s, size_t n, uint64_t* out_fq_mont) {
  if (!s || (n && !out_fq_mont)) return AMSM_E_INVALID_ARG;
  SPONGE_DO(s, sp.squeeze_words(n, out_fq_mont), sp.squeeze_words(n, out_fq_mont));
  return AMSM_OK;
}
int amsm_poseidon_squeeze_bits(amsm_sponge* s, size_t n_bits, uint8_t* out_bytes) {
  if (!s || (n_bits && !out_bytes)) return AMSM_E_INVALID_ARG;
  std::vector<uint8_t> b;
  SPONGE_DO(s, b = sp.squeeze_bits(n_bits), b = sp.squeeze_bits(n_bits));
  if (!b.empty()) memcpy(out_bytes, b.data(), b.size());
  return AMSM_OK;
}
int amsm_poseidon_squeeze_nonnative(amsm_sponge* s, unsigned n_bits, size_t count, uint64_t* out_canonical) {
  if (!s || (count && !out_canonical) || n_bits == 0 || n_bits > 254) return AMSM_E_INVALID_ARG;
  std::vector<uint8_t> b;
  const size_t total = (size_t)n_bits * count;
  SPONGE_DO(s, b = sp.squeeze_bits(total), b = sp.squeeze_bits(total));
  memset(out_canonical, 0, count * 32);
  for (size_t k = 0; k < count; k++)
    for (unsigned i = 0; i < n_bits; i++) {
      const size_t bit = k * n_bits + i;
      if ((b[bit >> 3] >> (bit & 7)) & 1) out_canonical[4 * k + (i >> 6)] |= 1ull << (i & 63);
    }
  return AMSM_OK;
}
int amsm_poseidon_permute(int curve, uint64_t* state_mont) {
  if (!state_mont || (curve != AMSM_PALLAS && curve != AMSM_BLS12_381_G1)) return AMSM_E_INVALID_ARG;
  amsm_sponge s;
  s.curve = curve;
  SPONGE_DO(&s, {
    constexpr int N = host::HFe<FQ>::N;
    for (int i = 0; i < 3; i++) memcpy(sp.state[i].v, state_mont + i * N, 8 * N);
    sp.permute();
    for (int i = 0; i < 3; i++) memcpy(state_mont + i * N, sp.state[i].v, 8 * N);
  }, {
    constexpr int N = host::HFe<FQ>::N;
    for (int i = 0; i < 3; i++) memcpy(sp.state[i].v, state_mont + i * N, 8 * N);
    sp.permute();
    for (int i = 0; i < 3; i++) memcpy(state_mont + i * N, sp.state[i].v, 8 * N);
  });
  return AMSM_OK;
}
int amsm_poseidon_round_constants(int curve, uint64_t* out_mont) {  // (8 + 31) * 3 elements, round-major
  if (!out_mont || (curve != AMSM_PALLAS && curve != AMSM_BLS12_381_G1)) return AMSM_E_INVALID_ARG;
  if (curve == AMSM_PALLAS) {
    const auto& p = host::PoseidonParams<PallasFq>::get();
    for (int r = 0; r < 39; r++)
      for (int i = 0; i < 3; i++) memcpy(out_mont + (r * 3 + i) * 4, p.ark[r][i].v, 32);
  } else {
    const auto& p = host::PoseidonParams<Bls12381Fq>::get();
    for (int r = 0; r < 39; r++)
      for (int i = 0; i < 3; i++) memcpy(out_mont + (r * 3 + i) * 6, p.ark[r][i].v, 48);
  }
  return AMSM_OK;
}

int amsm_vec_fill(amsm_ctx* c, const uint64_t* value_mont, size_t n, void* d_out) {
  if (!c || !value_mont || (n && !d_out) || n >= (1ull << 32)) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  if (!n) return AMSM_OK;
  u32 v[8];
  memcpy(v, value_mont, 32);
  launch_vec_fill(c->stream, (u32*)d_out, v, (u32)n);
  HIP_TRY(hipGetLastError());
  return AMSM_OK;
}

static size_t pool_round(size_t bytes) {  // 256-byte granules up to 1 MiB, 64-KiB granules above
  bytes = std::max<size_t>(bytes, 16);
  const size_t g = bytes <= ((size_t)1 << 20) ? 256 : 65536;
  return (bytes + g - 1) / g * g;
}
static void pool_release_all(amsm_ctx* c) {
  for (auto& kv : c->pool)
    for (void* p : kv.second) {
      (void)hipFree(p);
      c->pool_size.erase(p);
    }
  c->pool.clear();
  c->pool_free_bytes = 0;
}
int amsm_dev_alloc(amsm_ctx* c, size_t bytes, void** d_ptr) {
  if (!c || !d_ptr) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  const size_t sz = pool_round(bytes);
  auto it = c->pool.find(sz);
  if (it != c->pool.end() && !it->second.empty()) {
    *d_ptr = it->second.back();
    it->second.pop_back();
    c->pool_free_bytes -= sz;
    c->pool_live_bytes += sz;
    return AMSM_OK;
  }
  hipError_t e = hipMalloc(d_ptr, sz);
  if (e != hipSuccess) {  // give the free lists back to the driver and try once more
    (void)hipGetLastError();
    (void)hipStreamSynchronize(c->stream);
    pool_release_all(c);
    e = hipMalloc(d_ptr, sz);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return AMSM_E_OOM;
  }
  c->pool_size[*d_ptr] = sz;
  c->pool_live_bytes += sz;
  return AMSM_OK;
}
int amsm_dev_free(amsm_ctx* c, void* d_ptr) {
  if (!c) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  if (!d_ptr) return AMSM_OK;
  auto it = c->pool_size.find(d_ptr);
  if (it == c->pool_size.end()) {  // not one of ours (allocated before a trim raced, or foreign): plain free
    HIP_TRY(hipFree(d_ptr));
    return AMSM_OK;
  }
  const size_t sz = it->second;
  c->pool_live_bytes -= sz;
  if (c->pool_free_bytes + sz > c->pool_cap_bytes) {
    c->pool_size.erase(it);
    HIP_TRY(hipFree(d_ptr));
    return AMSM_OK;
  }
  c->pool[sz].push_back(d_ptr);
  c->pool_free_bytes += sz;
  return AMSM_OK;
}
int amsm_ctx_memory(const amsm_ctx* c, size_t* workspace_bytes, size_t* vectors_live_bytes, size_t* vectors_pooled_bytes) {
  if (!c) return AMSM_E_INVALID_ARG;
  size_t ws = c->scalars.bytes + c->xyzz_scratch.bytes + c->rec_send.bytes + c->rec_recv.bytes + c->stage.bytes;
  for (int k = 0; k < N_SLOTS; k++) {
    const Slot* sl = &c->slot[k];
    const DevBuf* bufs[] = {&sl->keys_a, &sl->keys_b, &sl->vals_a, &sl->vals_b, &sl->start, &sl->items, &sl->item_off,
                            &sl->partials, &sl->buckets, &sl->red_out, &sl->fold_out, &sl->heavy, &sl->misc, &sl->sort_tmp,
                            &sl->scan_tmp, &sl->prep_small, &sl->heavy_scratch};
    for (const DevBuf* b : bufs) ws += b->bytes;
  }
  if (workspace_bytes) *workspace_bytes = ws;
  if (vectors_live_bytes) *vectors_live_bytes = c->pool_live_bytes;
  if (vectors_pooled_bytes) *vectors_pooled_bytes = c->pool_free_bytes;
  return AMSM_OK;
}
int amsm_ctx_trim(amsm_ctx* c) {
  if (!c) return AMSM_E_INVALID_ARG;
  for (size_t g = 1; g < c->shard_ctx.size(); g++) TRY(amsm_ctx_trim(c->shard_ctx[g]));
  TRY(amsm_ctx_synchronize(c));
  pool_release_all(c);
  DevBuf* own[] = {&c->scalars, &c->xyzz_scratch, &c->rec_send, &c->rec_recv, &c->stage};
  for (DevBuf* b : own)
    if (b->p) {
      (void)hipFree(b->p);
      *b = DevBuf();
    }
  for (int k = 0; k < N_SLOTS; k++) {
    Slot* sl = &c->slot[k];
    DevBuf* bufs[] = {&sl->keys_a, &sl->keys_b, &sl->vals_a, &sl->vals_b, &sl->start, &sl->items, &sl->item_off,
                      &sl->partials, &sl->buckets, &sl->red_out, &sl->fold_out, &sl->heavy, &sl->misc, &sl->sort_tmp,
                      &sl->scan_tmp, &sl->prep_small, &sl->heavy_scratch};
    for (DevBuf* b : bufs)
      if (b->p) {
        (void)hipFree(b->p);
        *b = DevBuf();
      }
  }
  return AMSM_OK;
}
int amsm_dev_upload(amsm_ctx* c, void* d_dst, const void* h_src, size_t bytes) {
  if (!c || (bytes && (!d_dst || !h_src))) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  if (bytes) {
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return AMSM_OK;
}
int amsm_dev_download(amsm_ctx* c, void* h_dst, const void* d_src, size_t bytes) {
  if (!c || (bytes && (!h_dst || !d_src))) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  if (bytes) {
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return AMSM_OK;
}

int amsm_vec_random(amsm_ctx* c, uint64_t seed, size_t n, int mont, void* d_out) {
  if (!c || (n && !d_out) || n >= (1ull << 32)) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  if (!n) return AMSM_OK;
  if (c->curve == AMSM_PALLAS)
    launch_vec_random<PallasFr>(c->stream, (u32*)d_out, seed, (u32)n, mont);
  else
    launch_vec_random<Bls12381Fr>(c->stream, (u32*)d_out, seed, (u32)n, mont);
  HIP_TRY(hipGetLastError());
  return AMSM_OK;
}

int amsm_vec_hadamard(amsm_ctx* c, const void* d_a, const void* d_b, void* d_out, size_t n) {
  if (!c || (n && (!d_a || !d_b || !d_out)) || n >= (1ull << 32)) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  if (!n) return AMSM_OK;
  if (c->curve == AMSM_PALLAS)
    launch_vec_hadamard<PallasFr>(c->stream, (const u32*)d_a, (const u32*)d_b, (u32*)d_out, (u32)n);
  else
    launch_vec_hadamard<Bls12381Fr>(c->stream, (const u32*)d_a, (const u32*)d_b, (u32*)d_out, (u32)n);
  HIP_TRY(hipGetLastError());
  return AMSM_OK;
}

int amsm_vec_combine(amsm_ctx* c, const void* const* d_vecs, const size_t* lens, size_t n_vecs, const uint64_t* coeffs,
                     const void* d_hiding, size_t hiding_len, void* d_out, size_t n) {
  if (!c || (n && !d_out) || (n_vecs && (!d_vecs || !coeffs)) || n >= (1ull << 32)) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  return DISPATCH(c, (vec_combine_impl<PallasFr>(c, d_vecs, lens, n_vecs, coeffs, d_hiding, hiding_len, d_out, n)),
                  (vec_combine_impl<Bls12381Fr>(c, d_vecs, lens, n_vecs, coeffs, d_hiding, hiding_len, d_out, n)));
}

int amsm_bases_from_device(amsm_ctx* c, const void* d_xy, size_t n, unsigned flags, amsm_bases** out) {
  if (!c || !out || (n && !d_xy) || n >= (1ull << 30)) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  amsm_bases* b = new (std::nothrow) amsm_bases();
  if (!b) return AMSM_E_OOM;
  b->curve = c->curve;
  b->device = c->device;
  b->n = n;
  size_t pb = (c->curve == AMSM_PALLAS) ? affine_bytes<PallasFq>() : affine_bytes<Bls12381Fq>();
  if (hipMalloc((void**)&b->d_table, std::max<size_t>(n, 1) * pb) != hipSuccess) {
    (void)hipGetLastError();
    delete b;
    return AMSM_E_OOM;
  }
  int s = AMSM_OK;
  if (n) {
    if (hipMemcpyAsync(b->d_table, d_xy, n * pb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) s = AMSM_E_HIP;
    if (s == AMSM_OK) {
      if (c->curve == AMSM_PALLAS) launch_points_import<PallasFq>(c->stream, b->d_table, b->d_table, (u32)n);
      else launch_points_import<Bls12381Fq>(c->stream, b->d_table, b->d_table, (u32)n);
      if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) s = AMSM_E_HIP;
    }
    if (s == AMSM_OK)
      s = DISPATCH(c, (bases_finish<PallasFq, PallasFr>(c, b, flags ? flags : AMSM_BASES_NO_PRECOMPUTE)),
                   (bases_finish<Bls12381Fq, Bls12381Fr>(c, b, flags ? flags : AMSM_BASES_NO_PRECOMPUTE)));
  }
  if (s != AMSM_OK) {
    (void)hipFree(b->d_table);
    delete b;
    return s;
  }
  *out = b;
  return AMSM_OK;
}
const void* amsm_bases_device_ptr(const amsm_bases* b) {
  if (!b || key_sharded(b)) return nullptr;
  bool internal = b->curve == AMSM_PALLAS ? device_internal_radix<PallasFq>() : device_internal_radix<Bls12381Fq>();
  // the table may still be written by the fold that created the key (queued on that context's non-blocking stream,
  // which the NULL stream below does not order behind)
  if (b->ready && hipEventSynchronize(b->ready) != hipSuccess) return nullptr;
  if (!internal) return b->d_table;
  std::lock_guard<std::mutex> lock(b->abi_mu);
  if (!b->d_abi && b->n) {  // the table is in the device radix: hand out a C-ABI-radix copy of level 0
    size_t pb = (b->curve == AMSM_PALLAS) ? affine_bytes<PallasFq>() : affine_bytes<Bls12381Fq>();
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(b->device) != hipSuccess) return nullptr;
    u32* p = nullptr;
    bool ok = hipMalloc((void**)&p, b->n * pb) == hipSuccess;
    if (ok) {
      if (b->curve == AMSM_PALLAS) launch_points_export<PallasFq>(nullptr, b->d_table, p, (u32)b->n);
      else launch_points_export<Bls12381Fq>(nullptr, b->d_table, p, (u32)b->n);
      ok = hipStreamSynchronize(nullptr) == hipSuccess && hipGetLastError() == hipSuccess;
      if (!ok) (void)hipFree(p);
    } else {
      (void)hipGetLastError();
    }
    (void)hipSetDevice(prev);
    if (ok) b->d_abi = p;
  }
  return b->n ? b->d_abi : b->d_table;
}

static void canonical_scalar(int curve, const uint64_t* x_mont, u32 out[8]) {
  if (curve == AMSM_PALLAS) {
    host::HFe<PallasFr> x;
    memcpy(x.v, x_mont, 32);
    x = host::h_from_mont<PallasFr>(x);
    memcpy(out, x.v, 32);
  } else {
    host::HFe<Bls12381Fr> x;
    memcpy(x.v, x_mont, 32);
    x = host::h_from_mont<Bls12381Fr>(x);
    memcpy(out, x.v, 32);
  }
}

// scratch for the unconverted sums of a large fold (null: the kernel converts in place); stream-ordered reuse is safe
// because every user runs on the context's stream
static u32* fold_scratch(amsm_ctx* c, size_t n) {
  bool pays = c->curve == AMSM_PALLAS ? batch_affine_pays<PallasFq>((u32)n) : batch_affine_pays<Bls12381Fq>((u32)n);
  size_t rec = c->curve == AMSM_PALLAS ? xyzz_bytes<PallasFq>() : xyzz_bytes<Bls12381Fq>();
  if (!pays || ensure(c->xyzz_scratch, n * rec) != AMSM_OK) return nullptr;
  return (u32*)c->xyzz_scratch.p;
}

int amsm_points_fold(amsm_ctx* c, const void* d_l, const void* d_r, size_t n, const uint64_t* x_mont, unsigned nbits,
                     void* d_out) {
  if (!c || !x_mont || (n && (!d_l || !d_r || !d_out)) || n >= (1ull << 32) || nbits > 256) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  if (!n) return AMSM_OK;
  u32 canon[8];
  canonical_scalar(c->curve, x_mont, canon);
  u32* scratch = fold_scratch(c, n);
  if (c->curve == AMSM_PALLAS)
    launch_points_fold<PallasFq>(c->stream, (const u32*)d_l, (const u32*)d_r, (u32)n, canon, nbits, (u32*)d_out, true, scratch);
  else
    launch_points_fold<Bls12381Fq>(c->stream, (const u32*)d_l, (const u32*)d_r, (u32)n, canon, nbits, (u32*)d_out, true,
                                   scratch);
  HIP_TRY(hipGetLastError());
  return AMSM_OK;
}

int amsm_bases_fold(amsm_ctx* c, const amsm_bases* key, size_t n_half, const uint64_t* x_mont, unsigned nbits,
                    amsm_bases** out) {
  if (!c || !key || !out || !x_mont || key->curve != c->curve || key->device != c->device || nbits > 256 ||
      n_half == 0 || 2 * n_half > key->n)
    return AMSM_E_INVALID_ARG;
  if (key_sharded(key)) return AMSM_E_UNSUPPORTED;
  TRY(bind_device(c));
  size_t pb = (c->curve == AMSM_PALLAS) ? affine_bytes<PallasFq>() : affine_bytes<Bls12381Fq>();
  amsm_bases* b = new (std::nothrow) amsm_bases();
  if (!b) return AMSM_E_OOM;
  b->curve = c->curve;
  b->device = c->device;
  b->n = n_half;
  if (hipMalloc((void**)&b->d_table, n_half * pb) != hipSuccess) {
    (void)hipGetLastError();
    delete b;
    return AMSM_E_OOM;
  }
  u32 canon[8];
  canonical_scalar(c->curve, x_mont, canon);
  const u32* l = key->d_table;  // level 0 of a precomputed table is the key itself
  const u32* r = (const u32*)((const char*)key->d_table + n_half * pb);
  u32* scratch = fold_scratch(c, n_half);
  if (c->curve == AMSM_PALLAS)
    launch_points_fold<PallasFq>(c->stream, l, r, (u32)n_half, canon, nbits, b->d_table, false, scratch);
  else
    launch_points_fold<Bls12381Fq>(c->stream, l, r, (u32)n_half, canon, nbits, b->d_table, false, scratch);
  if (hipGetLastError() != hipSuccess || hipEventCreateWithFlags(&b->ready, hipEventDisableTiming) != hipSuccess ||
      hipEventRecord(b->ready, c->stream) != hipSuccess) {
    (void)hipGetLastError();
    if (b->ready) (void)hipEventDestroy(b->ready);
    (void)hipFree(b->d_table);
    delete b;
    return AMSM_E_HIP;
  }
  *out = b;  // stream-ordered: MSM prep and further folds are ordered behind the context's stream; others wait for `ready`
  return AMSM_OK;
}

int amsm_vec_inner_product(amsm_ctx* c, const void* d_a, const void* d_b, size_t n, uint64_t* out_mont) {
  if (!c || !out_mont || (n && (!d_a || !d_b)) || n >= (1ull << 32)) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  memset(out_mont, 0, 32);
  if (!n) return AMSM_OK;
  u32 blocks = std::min<u32>(1024u, cdiv((u32)n, 256));
  Slot* sl = &c->slot[0];
  TRY(ensure(sl->red_out, (size_t)blocks * 32 + 4096));
  TRY(ensure_pinned(sl, (size_t)blocks * 32));
  if (c->curve == AMSM_PALLAS)
    launch_vec_inner_product<PallasFr>(c->stream, (const u32*)d_a, (const u32*)d_b, (u32)n, blocks, (u32*)sl->red_out.p);
  else
    launch_vec_inner_product<Bls12381Fr>(c->stream, (const u32*)d_a, (const u32*)d_b, (u32)n, blocks,
                                         (u32*)sl->red_out.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(sl->h_pinned, sl->red_out.p, (size_t)blocks * 32, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const u64* h = (const u64*)sl->h_pinned;
  if (c->curve == AMSM_PALLAS) {
    host::HFe<PallasFr> acc = host::h_zero<PallasFr>(), t;
    for (u32 i = 0; i < blocks; i++) {
      memcpy(t.v, h + 4 * i, 32);
      acc = host::h_add<PallasFr>(acc, t);
    }
    memcpy(out_mont, acc.v, 32);
  } else {
    host::HFe<Bls12381Fr> acc = host::h_zero<Bls12381Fr>(), t;
    for (u32 i = 0; i < blocks; i++) {
      memcpy(t.v, h + 4 * i, 32);
      acc = host::h_add<Bls12381Fr>(acc, t);
    }
    memcpy(out_mont, acc.v, 32);
  }
  return AMSM_OK;
}

int amsm_vec_powers(amsm_ctx* c, const uint64_t* point_mont, size_t n, void* d_out) {
  if (!c || !point_mont || (n && !d_out) || n >= (1ull << 32)) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  if (!n) return AMSM_OK;
  u32 pt[8];
  memcpy(pt, point_mont, 32);
  if (c->curve == AMSM_PALLAS) launch_vec_powers<PallasFr>(c->stream, pt, (u32)n, (u32*)d_out);
  else launch_vec_powers<Bls12381Fr>(c->stream, pt, (u32)n, (u32*)d_out);
  HIP_TRY(hipGetLastError());
  return AMSM_OK;
}

int amsm_ipa_check_poly_coeffs(amsm_ctx* c, const uint64_t* xi_mont, size_t k, void* d_out) {
  if (!c || !d_out || (k && !xi_mont) || k > 30) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  if (c->curve == AMSM_PALLAS) launch_check_poly_coeffs<PallasFr>(c->stream, (const u32*)xi_mont, (u32)k, (u32*)d_out);
  else launch_check_poly_coeffs<Bls12381Fr>(c->stream, (const u32*)xi_mont, (u32)k, (u32*)d_out);
  HIP_TRY(hipGetLastError());
  return AMSM_OK;
}

int amsm_ipa_round_scalars(amsm_ctx* c, const uint64_t* xi_mont, size_t j, size_t log_n, const void* d_coeffs,
                           void* d_out_l, void* d_out_r) {
  if (!c || !d_coeffs || !d_out_l || (j && !xi_mont) || log_n == 0 || log_n > 30 || j >= log_n)
    return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  if (c->curve == AMSM_PALLAS)
    launch_ipa_round_scalars<PallasFr>(c->stream, (const u32*)xi_mont, (u32)j, (u32)log_n, (const u32*)d_coeffs,
                                       (u32*)d_out_l, (u32*)d_out_r);
  else
    launch_ipa_round_scalars<Bls12381Fr>(c->stream, (const u32*)xi_mont, (u32)j, (u32)log_n, (const u32*)d_coeffs,
                                         (u32*)d_out_l, (u32*)d_out_r);
  HIP_TRY(hipGetLastError());
  return AMSM_OK;
}

int amsm_ipa_round(amsm_ctx* c, const amsm_bases* key, const uint64_t* xi_mont, size_t j, size_t log_key, const void* d_coeffs,
                   const void* d_z, void* d_u, uint64_t* out_lr_xy, uint8_t* out_lr_inf, uint64_t* out_ip_mont) {
  if (!c || !key || !d_coeffs || !d_z || !d_u || !out_lr_xy || !out_ip_mont || (j && !xi_mont) || log_key == 0 || log_key > 30 ||
      j >= log_key || key->curve != c->curve || key->device != c->device)
    return AMSM_E_INVALID_ARG;
  if (key_sharded(key)) return AMSM_E_UNSUPPORTED;
  TRY(bind_device(c));
  const size_t half = (size_t)1 << (log_key - j - 1);
  return DISPATCH(c, (ipa_round_impl<PallasFq, PallasFr>(c, key, xi_mont, j, log_key, d_coeffs, d_z, half, d_u, out_lr_xy, out_lr_inf,
                                                          out_ip_mont)),
                  (ipa_round_impl<Bls12381Fq, Bls12381Fr>(c, key, xi_mont, j, log_key, d_coeffs, d_z, half, d_u, out_lr_xy,
                                                          out_lr_inf, out_ip_mont)));
}

int amsm_ipa_round_fused(amsm_ctx* c, const amsm_bases* key, const uint64_t* xi_mont, size_t j, size_t log_key, void* d_coeffs,
                         void* d_z, const uint64_t* fold_x_mont, const uint64_t* h_prime_xy, void* d_u, uint64_t* out_lr_xy,
                         uint8_t* out_lr_inf, uint64_t* out_ip_mont) {
  if (!c || !key || !d_coeffs || !d_z || !d_u || !out_lr_xy || !out_ip_mont || (j && !xi_mont) || log_key == 0 || log_key > 30 ||
      j >= log_key || key->curve != c->curve || key->device != c->device)
    return AMSM_E_INVALID_ARG;
  if (key_sharded(key)) return AMSM_E_UNSUPPORTED;
  if (fold_x_mont) {
    bool zero = true;
    for (int k = 0; k < 4; k++) zero = zero && fold_x_mont[k] == 0;
    if (zero) return AMSM_E_INVALID_ARG;  // a zero challenge has no inverse (the reference's verifier rejects it too)
  }
  TRY(bind_device(c));
  const size_t half = (size_t)1 << (log_key - j - 1);
  return DISPATCH(c, (ipa_round_impl<PallasFq, PallasFr>(c, key, xi_mont, j, log_key, d_coeffs, d_z, half, d_u, out_lr_xy, out_lr_inf,
                                                          out_ip_mont, fold_x_mont, h_prime_xy)),
                  (ipa_round_impl<Bls12381Fq, Bls12381Fr>(c, key, xi_mont, j, log_key, d_coeffs, d_z, half, d_u, out_lr_xy,
                                                          out_lr_inf, out_ip_mont, fold_x_mont, h_prime_xy)));
}

int amsm_matrix_load(amsm_ctx* c, const uint32_t* row_ptr, const uint32_t* col_idx, const uint64_t* vals, size_t n_rows,
                     size_t nnz, amsm_matrix** out) {
  if (!c || !out || !row_ptr || (nnz && (!col_idx || !vals)) || n_rows >= (1ull << 32) || nnz >= (1ull << 32))
    return AMSM_E_INVALID_ARG;
  if (row_ptr[0] != 0 || row_ptr[n_rows] != nnz) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  amsm_matrix* m = new (std::nothrow) amsm_matrix();
  if (!m) return AMSM_E_OOM;
  m->curve = c->curve;
  m->device = c->device;
  m->n_rows = n_rows;
  m->nnz = nnz;
  bool ok = hipMalloc((void**)&m->d_row_ptr, (n_rows + 1) * 4) == hipSuccess &&
            hipMalloc((void**)&m->d_col, std::max<size_t>(nnz, 1) * 4) == hipSuccess &&
            hipMalloc((void**)&m->d_val, std::max<size_t>(nnz, 1) * 32) == hipSuccess;
  ok = ok && hipMemcpyAsync(m->d_row_ptr, row_ptr, (n_rows + 1) * 4, hipMemcpyHostToDevice, c->stream) == hipSuccess;
  if (ok && nnz) {
    ok = hipMemcpyAsync(m->d_col, col_idx, nnz * 4, hipMemcpyHostToDevice, c->stream) == hipSuccess &&
         hipMemcpyAsync(m->d_val, vals, nnz * 32, hipMemcpyHostToDevice, c->stream) == hipSuccess;
  }
  ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
  if (!ok) {
    (void)hipGetLastError();
    amsm_matrix_free(m);
    return AMSM_E_OOM;
  }
  *out = m;
  return AMSM_OK;
}
size_t amsm_matrix_rows(const amsm_matrix* m) { return m ? m->n_rows : 0; }
void amsm_matrix_free(amsm_matrix* m) {
  if (!m) return;
  (void)hipSetDevice(m->device);
  if (m->d_row_ptr) (void)hipFree(m->d_row_ptr);
  if (m->d_col) (void)hipFree(m->d_col);
  if (m->d_val) (void)hipFree(m->d_val);
  delete m;
}
int amsm_matrix_vec_mul(amsm_ctx* c, const amsm_matrix* m, const void* d_input, size_t n_input, const void* d_witness,
                        size_t n_witness, void* d_out) {
  if (!c || !m || m->curve != c->curve || m->device != c->device || (m->n_rows && !d_out) ||
      (n_input && !d_input) || (n_witness && !d_witness) || n_input >= (1ull << 32) || n_witness >= (1ull << 32))
    return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  if (!m->n_rows) return AMSM_OK;
  if (c->curve == AMSM_PALLAS)
    launch_spmv<PallasFr>(c->stream, m->d_row_ptr, m->d_col, m->d_val, (const u32*)d_input, (u32)n_input,
                          (const u32*)d_witness, (u32)n_witness, (u32*)d_out, (u32)m->n_rows);
  else
    launch_spmv<Bls12381Fr>(c->stream, m->d_row_ptr, m->d_col, m->d_val, (const u32*)d_input, (u32)n_input,
                            (const u32*)d_witness, (u32)n_witness, (u32*)d_out, (u32)m->n_rows);
  HIP_TRY(hipGetLastError());
  return AMSM_OK;
}

int amsm_hp_t_vecs(amsm_ctx* c, const void* const* d_a, const size_t* a_lens, const void* const* d_b,
                   const size_t* b_lens, size_t n_inputs, const uint64_t* mu_mont, size_t n_mu, const void* d_hiding_a,
                   size_t hiding_a_len, const void* d_hiding_b, size_t hiding_b_len, void* const* d_t, size_t len) {
  if (!c || !d_a || !d_b || !mu_mont || !d_t || len >= (1ull << 32)) return AMSM_E_INVALID_ARG;
  TRY(bind_device(c));
  return DISPATCH(c,
                  (t_vecs_impl<PallasFr>(c, d_a, a_lens, d_b, b_lens, n_inputs, mu_mont, n_mu, d_hiding_a, hiding_a_len,
                                         d_hiding_b, hiding_b_len, d_t, len)),
                  (t_vecs_impl<Bls12381Fr>(c, d_a, a_lens, d_b, b_lens, n_inputs, mu_mont, n_mu, d_hiding_a,
                                           hiding_a_len, d_hiding_b, hiding_b_len, d_t, len)));
}

}  // extern "C"
