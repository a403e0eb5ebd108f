// Private to api.hip (one translation unit): the context, key, slot and host-thread types of the C ABI implementation.
#pragma once
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

// Measured constants of the launcher (environment knobs until round 5: each was A/B'd, the numbers are in DESIGN.md / profiles/)
namespace tune {
constexpr int K0_MAX = 32;                 // largest chunk of accumulate L0 (24 -> 805, 32 -> 811-816 M pairs/s in 2^20 batches)
constexpr amsm::u32 K1 = 1024;             // buckets with more partials than this go to the workgroup-per-bucket path (extreme skew)
constexpr size_t ONESHOT_RANGE = (size_t)1 << 19;  // amsm_msm_oneshot: pairs per range (upload of range j + 1 beside the MSM of range j)
constexpr int TAIL_QUAD_HIDDEN_LOG2 = 17;  // bucket tables up to 2^this take the quad tail inside a batch too (2^16 234 -> 283 M pairs/s)
}  // namespace tune

// One range of an MSM that is longer than the 2^c-pair window of its bucket-per-lane key (round 5; 2^22 pairs over the 20-bit key:
// four ranges).  The ranges used to be independent MSMs -- four preps, four 2^19-bucket reductions, four folds, summed on the
// host -- although they share the key's bucket index space: now range 1 writes the MSM's bucket table, ranges 2 .. k add to it
// (k_accum_bpl<ACC>) and only the last one carries a tail.  The table is one of two the context owns (consecutive long MSMs
// alternate; the first range of the next user waits for the previous user's tail through `shared_free`).
struct Share {
  int buf = -1;  // which of amsm_ctx::shared_buckets; -1: an ordinary MSM
  bool first = false, last = false;
};

// One pipeline slot = one stream + one private workspace, so two MSMs of a batch can be in flight:
// the latency-bound tail of MSM i (fold partials, bucket reduce) overlaps the throughput-bound head of
// MSM i+1 on the other slot's stream.
constexpr int N_SLOTS = 3;  // MSMs of one batch in flight (4 measured no better)

struct Slot {  // buffers and events of one MSM in flight (the streams belong to the context: one per pipeline stage)
  hipEvent_t l0_done = nullptr, prep_done = nullptr;
  hipEvent_t ev[ST_COUNT + 1] = {};  // stage begins (each on the stream its stage runs on)
  hipEvent_t ev_prep_end = nullptr, ev_l0_end = nullptr;  // ends of the stages whose successor starts on ANOTHER stream
  hipEvent_t done = nullptr;
  DevBuf keys_a, keys_b, vals_a, vals_b, start, items, item_off, partials, buckets, red_out, fold_out, heavy, misc,
      sort_tmp, scan_tmp, prep_small, heavy_scratch;
  DevBuf red2_rc;  // row / column sums of the bucket reduction (k_red2_sums)
  DevBuf red_ticket;  // one arrival counter per bucket set (k_bucket_reduce_fold_quad; zeroed at allocation, left clear by the kernel)
  DevBuf ds_flags;  // the direct sum's two flag words (zeroed at allocation, left clear by k_fold_quad)
  DevBuf bpl_grp, bpl_order;  // bucket-per-lane pipeline: group headers, bucket order (entries live in vals_a / vals_b)
  // the MSM this slot carries, kept until it is collected: a bucket-per-lane MSM whose prep reports a skewed input is
  // re-run from here through the chunked pipeline (msm_collect)
  struct Job {
    Share share;  // this MSM is one range of a longer one whose ranges share a bucket set (buf >= 0)
    const amsm_bases* bases = nullptr;
    size_t base_off = 0, n = 0;
    const void* d_scalars = nullptr;
    int mont = 0;
    int group_shift = -1;
    bool force_chunked = false;  // a bucket-split MSM whose prep overflowed: the re-run must not choose that pipeline again
    int (*rerun)(amsm_ctx*, Slot*) = nullptr;
  } job;
  void* h_pinned = nullptr;
  size_t h_pinned_bytes = 0;
  MsmGeom geom = {};
  bool busy = false;
  bool share_overflow = false;  // set by msm_collect: a range of a shared MSM found skewed digits -- the whole MSM is re-run unshared
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
// to 7 helpers).  One parallel region at a time: a second caller that finds the pool busy runs its tasks itself.
struct HostPool {
  std::vector<std::unique_ptr<ShardWorker>> workers;
  std::mutex busy;
  HostPool() {
    int want = 7;  // (round 4: 3 -> 7 -- the harness shapes of hp_as / r1cs_nark_as prove 10-20 % faster, same box)
    int hw = (int)std::thread::hardware_concurrency();
    // one process per GPU (torchrun exports LOCAL_WORLD_SIZE): the node's cores are shared by that many pools -- eight ranks of
    // seven helpers each would be 64 threads on whatever the box has (round 5: the driver's first 8-rank run must not find out)
    if (const char* e = getenv("LOCAL_WORLD_SIZE")) {
      const int lw = atoi(e);
      if (lw > 1) hw = std::max(1, hw / lw);
    }
    if (const char* e = getenv("AMSM_HOST_THREADS")) want = atoi(e);  // explicit: taken as given (still below the cap of 15)
    else want = std::min(want, hw > 1 ? hw - 1 : 0);
    want = std::max(0, std::min(want, 15));
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
  bool host_only = false;  // the host backend (api_cpu.inc): device == AMSM_DEVICE_HOST, no stream, "device" pointers are host memory
  // One stream per pipeline STAGE, shared by all MSMs in flight (in-order per stage, so consecutive MSMs pipeline:
  // prep(k+1) and tail(k-1) run beside accumulate L0 of MSM k).  One stream per MSM instead made the overlap depend on
  // which hardware queues the runtime happened to map the streams to (measured 610-700 Mpairs/s for the same code).
  hipStream_t stream = nullptr;  // main: the caller's stream -- accumulate L0 and every non-MSM kernel
  hipStream_t s_prep = nullptr;  // digits, sort, bounds, scan (memory-bound)
  hipStream_t s_tail = nullptr;  // fold partials, bucket reduce, fold, D2H (latency-bound)
  bool own_stream = false;
  int window_override = 0;  // amsm_ctx_set_window: every MSM chunked with this width (tests, sweeps)
  int cu_count = 256;
  int wave_slots = 4096;  // resident accumulate-L0 waves: CUs x resident 256-lane blocks per CU x 4
  // ---- the documented switches (include/amsm.h "Environment"; msm_select.h: Switches) -- everything else that used to be an
  // environment knob is a measured constant now (namespace tune below; the A/Bs are in profiles/ and DESIGN.md) ----
  bool bpl = true;        // AMSM_BPL=0: no 20-bit tables, no bucket-per-lane pipeline (17-bit windows + the chunked pipeline)
  bool bpl_plain = true;  // AMSM_BPL_PLAIN=0: plain keys stay on the chunked pipeline
  int bps = 2;            // AMSM_BPS: bucket-split pipeline 0 never, 1 grouped MSMs only, 2 every candidate of 2^16 .. 2^17 pairs
  int split_log2 = 21;    // AMSM_SPLIT_LOG2: MSMs of 2^22 pairs and more over a key without a 20-bit table run as ranges of 2^this (0: whole)
  unsigned long long n_bps = 0, n_bps_fallbacks = 0;
  unsigned long long n_direct = 0;  // MSMs summed straight from a small key's 512-points-per-generator table (k_direct_sum)
  int direct_max_log2 = 15;         // keys of up to 2^this generators carry that table (AMSM_DIRECT_SUM_MAX_LOG2; 0: none)
  unsigned direct_rr = 0;           // which stream the next direct sum of a batch takes
  bool host_halves = true;  // AMSM_HOST_HALVES=0: a lone host slice over a 20-bit key as ONE range (upload, then the MSM: rounds 1-5; A/B)
  bool fused_fold = true;  // AMSM_FUSED_FOLD=0: the quad bucket reduction and its fold as two launches (round 5's tail; A/B)
  bool bpl_probe = true;  // sample every candidate vector's digits first and send skewed ones straight to the chunked pipeline
                          // (AMSM_BPL_PROBE=0: find out from the prep's overflow flag only -- the safety net either way)
  unsigned long long n_bpl = 0, n_bpl_fallbacks = 0;  // MSMs that took it / that were re-run chunked (skewed digits)
  // MSMs longer than the key's window: one bucket set for all their ranges (struct Share; AMSM_SHARE_BUCKETS=0: independent ranges, A/B)
  bool share_buckets = true;
  DevBuf shared_buckets[2];
  hipEvent_t shared_free[2] = {};  // recorded behind the tail that last read the table
  bool shared_used[2] = {false, false};
  unsigned shared_rr = 0;
  unsigned long long n_shared = 0;  // long MSMs that ran over one bucket set
  // the OPTIONAL tables of a key (the 512-points-per-generator direct-sum table, the 17-bit twin) are a decision, not a side effect
  // (round 5): bytes one key may spend on them (amsm_ctx_set_table_budget; default: no limit) and how often one was denied
  size_t table_budget = ~(size_t)0;
  unsigned long long n_tables_denied = 0;
  bool profiling = false;
  float stage_ms[ST_COUNT] = {};  // mean over the MSMs of the last call
  float stage_acc[ST_COUNT] = {};
  int stage_n = 0;
  Slot slot[N_SLOTS];
  hipEvent_t fork = nullptr;
  hipEvent_t ip_ready = nullptr;  // amsm_ipa_round_fused: the inner products have reached the host
  DevBuf scalars;
  DevBuf probe_flags;  // one word per vector probed for skew (api_pipeline.inc: bpl_probe_device)
  // host-slice batches (amsm_msm_batch, amsm_pedersen_commit_batch): a ring of device staging buffers, filled on a copy stream
  // while the previous MSMs compute (created on first use).  Uploads run ONE vector ahead of the MSM being enqueued (two ahead
  // was measured slower: 0.77 instead of 0.86-0.88 of the device-resident rate -- the second blocking copy delays the enqueue)
  static constexpr int STAGE_RING = N_SLOTS + 2;
  DevBuf stage_ring[STAGE_RING];
  hipEvent_t up_ev[STAGE_RING] = {};
  hipStream_t s_copy = nullptr;
  // amsm_msm_oneshot (round 6): the generators arrive WITH the call (`multi_scalar_mul(&[G], &[BigInt])`); they are imported range by
  // range into this grow-only buffer (the call's temporary plain key) behind the copy stream, and their infinity flags here
  DevBuf oneshot_table, oneshot_inf;
  DevBuf xyzz_scratch;  // unconverted sums of large key folds / precompute levels (launch.h: batch_affine_pays)
  // two-valued device vectors (api_pipeline.inc: msm_two_valued_pass): every scalar is 0 or one value v -> v * (sum of the
  // generators with a non-zero scalar), on its own stream beside the batch's other MSMs.  AMSM_TWO_VALUED=0 turns it off.
  int two_valued = 1;
  DevBuf tv_flags, tv_parts, tv_out;
  void* tv_pinned = nullptr;
  size_t tv_pinned_bytes = 0;
  hipStream_t s_tv = nullptr;
  hipEvent_t tv_done = nullptr;
  unsigned long long n_two_valued = 0;
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
  // vectors whose unit scalars are summed apart during the current msm_multi_split_xyzz call: [first byte, one past the last) of
  // their scalars -- msm_enqueue sets MsmGeom::skip_ones for every (range of a) vector that lies inside one
  std::vector<std::pair<const char*, const char*>> skip_ones_ranges;
  unsigned long long n_ones_split = 0;  // MSMs that took that form (amsm_ctx_unit_scalar_msms)
  size_t replicate_below = 0;             // amsm_ctx_set_replicate_below: keys up to this many generators are replicated, not sharded
  unsigned long long n_replicated = 0;    // MSMs of batch calls over replicated keys that ran off the primary device
  unsigned long long n_collectives = 0;  // exchanges of partial records so far (amsm_ctx_collectives: one per sharded call)
  uint64_t n_host_gathers = 0;  // grouped MSMs / IPA rounds over sharded keys: the shards' class sums folded on the host
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
  bool host = false;  // key of a host context: d_table is a host copy of the generators (C-ABI radix), nothing else is set
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
  // bpl: the table holds the multiples for 20-bit windows (13 levels) and MSMs of (2^19, 2^20] pairs over it run the
  // bucket-per-lane pipeline.  Everything that pipeline does not take (shorter ranges, grouped MSMs, skewed scalars) runs
  // over `alt`, the same generators precomputed for 17-bit windows, built on first use (under alt_mu: keys are shared)
  int bpl = 0;
  int top_shift = 0;  // MsmGeom::top_shift of the table (level W - 1 = 2^(c (W - 1) - top_shift) G)
  int n_narrow = 0;   // MsmGeom::n_narrow of the table (level w = 2^(window_exponent) G)
  // small keys (round 4): j 2^(4 w) G_i for j = 1 .. 8, w = 0 .. 63 at [((j - 1) 64 + w) n + i] -- every MSM over such a key that is
  // not grouped is a plain sum of table points (msm_kernels.h k_direct_sum); null: none
  u32* d_small = nullptr;
  int direct_denied = 0;          // why a small key has no such table (api_keys.inc: TableDenied); 0: it has one, or never qualified
  bool no_twin = false;           // AMSM_BASES_NO_TWIN
  mutable int twin_denied = 0;    // a call needed the twin and was refused (flag / budget)
  mutable amsm_bases* alt = nullptr;
  mutable std::mutex alt_mu;
  // sharded key of a multi-device context: shard g (a single-device key on shard_ctx[g]'s device) holds generators
  // [bound[g], bound[g + 1]); n is the total, d_table stays null
  std::vector<amsm_bases*> shards;
  std::vector<size_t> bound;
  // REPLICATED key of a multi-device context (AMSM_BASES_REPLICATE, round 6): this object is the PRIMARY device's ordinary key;
  // replicas[g] (g >= 1, owned) is the same key on shard_ctx[g]'s device, replicas[0] stays null.  Empty: not replicated.
  std::vector<amsm_bases*> replicas;
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
  bool host = false;  // matrix of a host context: the three arrays are host memory
  size_t n_rows = 0, nnz = 0;
  u32* d_row_ptr = nullptr;
  u32* d_col = nullptr;
  u32* d_val = nullptr;
};
