// Multi-device context through the C ABI ONLY (include/amsm.h): a key sharded over n "devices" must give results that are
// bit-identical to the single-device ones for every entry point that accepts a sharded key.  On a 1-GPU box every shard
// maps to device 0 (amsm_ctx_create_multi accepts a repeated device id), on a node the distinct devices are used and the
// partial sums travel over RCCL.
//
//   multi_device_check [n_shards (default 2)] [n (default 50000)] [curve (0 Pallas | 1 BLS12-381 G1)]
//
// Replaces: the reference's `prove` is one synchronous call in one process (src/lib.rs:163-249); its MSMs
// (ark_ec::msm::VariableBaseMSM::multi_scalar_mul, ext) run on the calling process's cores.  Here the calling process drives
// all the GPUs of the node.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "amsm.h"

#define CHECK(expr)                                                                         \
  do {                                                                                      \
    int s_ = (expr);                                                                        \
    if (s_ != AMSM_OK) {                                                                    \
      fprintf(stderr, "%s:%d: %s -> %s (%d)\n", __FILE__, __LINE__, #expr, amsm_strerror(s_), s_); \
      return 1;                                                                             \
    }                                                                                       \
  } while (0)
#define EXPECT(cond)                                               \
  do {                                                             \
    if (!(cond)) {                                                 \
      fprintf(stderr, "%s:%d: FAILED %s\n", __FILE__, __LINE__, #cond); \
      return 1;                                                    \
    }                                                              \
  } while (0)

struct Pt {
  std::vector<uint64_t> xy;
  uint8_t inf = 0;
  bool operator==(const Pt& o) const { return inf == o.inf && xy == o.xy; }
};

int main(int argc, char** argv) {
  const int n_shards = argc > 1 ? atoi(argv[1]) : 2;
  const size_t n = argc > 2 ? (size_t)atoll(argv[2]) : 50000;
  const int curve = argc > 3 ? atoi(argv[3]) : AMSM_PALLAS;
  const int n_gpu = amsm_device_count();
  if (n_gpu < 1) {
    fprintf(stderr, "no GPU: %s\n", amsm_strerror(AMSM_E_NO_DEVICE));
    return 2;
  }
  std::vector<int> devs(n_shards);
  for (int g = 0; g < n_shards; g++) devs[g] = g % n_gpu;  // distinct while they last, then repeated
  const uint64_t SEED_KEY = 0x5EED1001, SEED_V = 0x5EED0001;
  const size_t K = 3;

  // ---- single-device reference ------------------------------------------------------------------
  amsm_ctx* one = nullptr;
  CHECK(amsm_ctx_create(&one, curve, 0, nullptr));
  const size_t L2 = 2 * (size_t)amsm_ctx_fq_limbs(one);
  amsm_bases* key1 = nullptr;
  CHECK(amsm_bases_generate(one, SEED_KEY, n, AMSM_BASES_DEFAULT, &key1));
  std::vector<void*> d_v(K);
  std::vector<std::vector<uint64_t>> h_v(K, std::vector<uint64_t>(4 * n));
  for (size_t v = 0; v < K; v++) {
    CHECK(amsm_dev_alloc(one, n * 32, &d_v[v]));
    CHECK(amsm_vec_random(one, SEED_V + v, n, /*mont=*/1, d_v[v]));
    CHECK(amsm_dev_download(one, h_v[v].data(), d_v[v], n * 32));
  }
  auto msm1 = [&](size_t off, const void* d, size_t cnt, Pt* out) {
    out->xy.assign(L2, 0);
    return amsm_msm_device(one, key1, off, d, cnt, 1, out->xy.data(), &out->inf);
  };
  std::vector<Pt> ref(K);
  for (size_t v = 0; v < K; v++) CHECK(msm1(0, d_v[v], n, &ref[v]));
  // a window that starts inside shard 0 and ends inside the last shard
  const size_t w_off = n / 7, w_n = n - n / 7 - n / 11;
  Pt ref_win;
  CHECK(msm1(w_off, d_v[0], w_n, &ref_win));
  // a window inside one shard only
  const size_t s_off = 3, s_n = n / (2 * (size_t)n_shards) > 4 ? n / (2 * (size_t)n_shards) - 4 : 1;
  Pt ref_small;
  CHECK(msm1(s_off, d_v[1], s_n, &ref_small));
  // Pedersen commitment with a hiding term
  std::vector<uint64_t> hid(L2), rnd(4);
  {
    uint8_t inf = 0;
    CHECK(amsm_bases_read(one, key1, n - 1, 1, hid.data(), &inf));  // any point will do as the hiding generator
    memcpy(rnd.data(), h_v[2].data() + 4 * 5, 32);
  }
  Pt ref_ped;
  ref_ped.xy.assign(L2, 0);
  CHECK(amsm_pedersen_commit_device(one, key1, d_v[0], n, rnd.data(), hid.data(), ref_ped.xy.data(), &ref_ped.inf));
  std::vector<uint64_t> key_xy(n * L2);
  std::vector<uint8_t> key_inf(n);
  CHECK(amsm_bases_read(one, key1, 0, n, key_xy.data(), key_inf.data()));

  // ---- the same through a multi-device context ----------------------------------------------------
  amsm_ctx* multi = nullptr;
  CHECK(amsm_ctx_create_multi(&multi, curve, devs.data(), n_shards));
  EXPECT(amsm_ctx_num_devices(multi) == n_shards);
  printf("collective %s\n", amsm_ctx_collective(multi));
  amsm_bases* keyN = nullptr;
  CHECK(amsm_bases_generate(multi, SEED_KEY, n, AMSM_BASES_DEFAULT, &keyN));
  EXPECT(amsm_bases_len(keyN) == n);
  EXPECT(amsm_bases_num_shards(keyN) == n_shards);
  {  // the shards tile [0, n) like accumulation_amd/dist.py: shard_bounds
    size_t prev = 0;
    for (int g = 0; g < n_shards; g++) {
      size_t lo = 0, hi = 0;
      CHECK(amsm_bases_shard_range(keyN, g, &lo, &hi));
      EXPECT(lo == prev && hi >= lo && hi - lo <= n / n_shards + 1);
      prev = hi;
    }
    EXPECT(prev == n);
  }
  {  // the sharded key holds the same generators
    std::vector<uint64_t> xy(n * L2);
    std::vector<uint8_t> inf(n);
    CHECK(amsm_bases_read(multi, keyN, 0, n, xy.data(), inf.data()));
    EXPECT(xy == key_xy && inf == key_inf);
  }
  // vectors on the primary device of the multi context
  std::vector<void*> d_p(K);
  for (size_t v = 0; v < K; v++) {
    CHECK(amsm_dev_alloc(multi, n * 32, &d_p[v]));
    CHECK(amsm_dev_upload(multi, d_p[v], h_v[v].data(), n * 32));
  }
  Pt got;
  got.xy.assign(L2, 0);
  // (1) host scalars
  CHECK(amsm_msm(multi, keyN, 0, h_v[0].data(), n, 1, got.xy.data(), &got.inf));
  EXPECT(got == ref[0]);
  // (2) primary-device scalars
  CHECK(amsm_msm_device(multi, keyN, 0, d_p[1], n, 1, got.xy.data(), &got.inf));
  EXPECT(got == ref[1]);
  // (3) batch
  {
    std::vector<uint64_t> xy(K * L2);
    std::vector<uint8_t> inf(K);
    CHECK(amsm_msm_batch_device(multi, keyN, 0, (const void* const*)d_p.data(), K, n, 1, xy.data(), inf.data()));
    for (size_t v = 0; v < K; v++) {
      Pt p;
      p.xy.assign(xy.begin() + v * L2, xy.begin() + (v + 1) * L2);
      p.inf = inf[v];
      EXPECT(p == ref[v]);
    }
  }
  // (4) windows of the key
  CHECK(amsm_msm_device(multi, keyN, w_off, d_p[0], w_n, 1, got.xy.data(), &got.inf));
  EXPECT(got == ref_win);
  CHECK(amsm_msm_device(multi, keyN, s_off, d_p[1], s_n, 1, got.xy.data(), &got.inf));
  EXPECT(got == ref_small);
  // (5) Pedersen commitment (device and host elements)
  CHECK(amsm_pedersen_commit_device(multi, keyN, d_p[0], n, rnd.data(), hid.data(), got.xy.data(), &got.inf));
  EXPECT(got == ref_ped);
  CHECK(amsm_pedersen_commit(multi, keyN, h_v[0].data(), n, rnd.data(), hid.data(), got.xy.data(), &got.inf));
  EXPECT(got == ref_ped);
  // (6) scalars already sharded: slice (v, g) lives on device g
  {
    std::vector<const void*> slices(K * n_shards, nullptr);
    std::vector<void*> owned;
    for (int g = 0; g < n_shards; g++) {
      amsm_ctx* cg = amsm_ctx_shard(multi, g);
      EXPECT(cg != nullptr);
      size_t lo = 0, hi = 0;
      CHECK(amsm_bases_shard_range(keyN, g, &lo, &hi));
      for (size_t v = 0; v < K; v++) {
        void* d = nullptr;
        CHECK(amsm_dev_alloc(cg, (hi - lo) * 32, &d));
        CHECK(amsm_dev_upload(cg, d, h_v[v].data() + 4 * lo, (hi - lo) * 32));
        slices[v * n_shards + g] = d;
        owned.push_back(d);
      }
    }
    std::vector<uint64_t> xy(K * L2);
    std::vector<uint8_t> inf(K);
    CHECK(amsm_msm_batch_sharded_device(multi, keyN, slices.data(), K, 1, xy.data(), inf.data()));
    for (size_t v = 0; v < K; v++) {
      Pt p;
      p.xy.assign(xy.begin() + v * L2, xy.begin() + (v + 1) * L2);
      p.inf = inf[v];
      EXPECT(p == ref[v]);
    }
    size_t k = 0;
    for (int g = 0; g < n_shards; g++)
      for (size_t v = 0; v < K; v++) CHECK(amsm_dev_free(amsm_ctx_shard(multi, g), owned[k++]));
  }
  // (7) grouped MSMs over the sharded key (round 5: every shard sums its part of both index classes) against the single-device
  // grouped MSM: group periods below, at and above a shard's length, ragged lengths and offsets (the shard slices then start
  // off the group period: msm_grouped_partial's head / tail pieces)
  {
    unsigned top = 0;
    while (((size_t)2 << top) < n) top++;
    struct G {
      size_t off, cnt;
      unsigned shift;
    } cases[] = {{0, n, 0}, {0, n, 3}, {0, n, top}, {0, n, top > 1 ? top - 1 : 0}, {0, n > 64 ? n - 5 : n, 2}, {n > 64 ? (size_t)7 : 0, n > 64 ? n - 20 : n, 4},
                 {w_off, w_n, top > 2 ? top - 2 : 1}, {s_off, s_n, 1}};
    for (const G& gc : cases) {
      if (gc.off + gc.cnt > n) continue;
      uint8_t i1[2] = {0, 0}, iN[2] = {0, 0};
      std::vector<uint64_t> x1(2 * L2), xN(2 * L2);
      CHECK(amsm_msm_grouped_device(one, key1, gc.off, d_v[0], gc.cnt, 1, gc.shift, x1.data(), i1));
      CHECK(amsm_msm_grouped_device(multi, keyN, gc.off, d_p[0], gc.cnt, 1, gc.shift, xN.data(), iN));
      EXPECT(x1 == xN && i1[0] == iN[0] && i1[1] == iN[1]);
    }
  }
  // (7a) multi-offset MSMs (the IPA cross commitments, trivial_pc's witness commitments): runs of jobs over one key range share
  // an exchange; mixed offsets and lengths, an empty job in the middle
  {
    const size_t offs[5] = {0, 0, w_off, s_off, n > 64 ? (size_t)9 : 0};
    const size_t lens[5] = {n, n, w_n, 0, n > 64 ? n - 30 : n};
    const void* vecs[5] = {d_v[0], d_v[1], d_v[2], d_v[0], d_v[1]};
    const void* vecsN[5] = {d_p[0], d_p[1], d_p[2], d_p[0], d_p[1]};
    std::vector<uint64_t> x1(5 * L2), xN(5 * L2);
    uint8_t i1[5], iN[5];
    CHECK(amsm_msm_multi_device(one, key1, 5, offs, vecs, lens, 1, x1.data(), i1));
    const uint64_t before = amsm_ctx_collectives(multi);
    CHECK(amsm_msm_multi_device(multi, keyN, 5, offs, vecsN, lens, 1, xN.data(), iN));
    EXPECT(x1 == xN && memcmp(i1, iN, 5) == 0 && iN[3] == 1);
    EXPECT(n_shards == 1 || amsm_ctx_collectives(multi) - before <= 4);
  }
  // (7b) what does not shard says so instead of crashing (one shard = an ordinary key)
  if (n_shards > 1) {
    EXPECT(amsm_bases_device_ptr(keyN) == nullptr);
    amsm_bases* folded = nullptr;
    EXPECT(amsm_bases_fold(multi, keyN, n / 2, rnd.data(), 128, &folded) == AMSM_E_UNSUPPORTED);
    // a sharded key does not work with a foreign context
    EXPECT(amsm_msm_device(one, keyN, 0, d_v[0], n, 1, got.xy.data(), &got.inf) == AMSM_E_INVALID_ARG);
  }
  // (8) a sharded key loaded from the caller's arrays (amsm_bases_load), not generated
  {
    amsm_bases* keyL = nullptr;
    CHECK(amsm_bases_load(multi, key_xy.data(), key_inf.data(), n, AMSM_BASES_DEFAULT, &keyL));
    EXPECT(amsm_bases_num_shards(keyL) == n_shards && amsm_bases_len(keyL) == n);
    CHECK(amsm_msm_device(multi, keyL, 0, d_p[1], n, 1, got.xy.data(), &got.inf));
    EXPECT(got == ref[1]);
    CHECK(amsm_msm_device(multi, keyL, w_off, d_p[0], w_n, 1, got.xy.data(), &got.inf));
    EXPECT(got == ref_win);
    amsm_bases_free(keyL);
  }
  // (9) empty ranges
  CHECK(amsm_msm_device(multi, keyN, n, d_p[0], 10, 1, got.xy.data(), &got.inf));
  EXPECT(got.inf == 1);
  CHECK(amsm_ctx_synchronize(multi));
  for (size_t v = 0; v < K; v++) {
    CHECK(amsm_dev_free(multi, d_p[v]));
    CHECK(amsm_dev_free(one, d_v[v]));
  }
  amsm_bases_free(keyN);
  amsm_bases_free(key1);
  amsm_ctx_destroy(multi);
  amsm_ctx_destroy(one);
  printf("ok shards=%d n=%zu curve=%d\n", n_shards, n, curve);
  return 0;
}
