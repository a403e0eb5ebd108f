// amsm.hpp -- header-only C++17 host-side mirror of the reference interfaces on the MSM hot path, over the C
// ABI of amsm.h.  The reference is compiled code (Rust); this is the compiled-language twin of the Python mirror
// in accumulation_amd/engine.py + hp_as.py: same names, argument meaning and error behaviour as
//   ark_ec::msm::VariableBaseMSM::multi_scalar_mul                       (ext)
//   ark_poly_commit::trivial_pc::{PedersenCommitment, CommitterKey}      (ext, a2; src/hp_as/mod.rs:196,377,911)
//   ASForHadamardProducts::{compute_hp, combine_vectors, scale_vector, compute_t_vecs,
//                           compute_product_poly_comm} and the decider's three commitments
//                                                                        (src/hp_as/mod.rs:278-512, 354-388, 894-925)
// Errors: every non-zero ABI status is thrown as amsm::Error (the Rust adapter maps it to BoxedError).
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "amsm.h"

namespace amsm {

struct Error : std::runtime_error {
  int status;
  Error(int s, const char* where) : std::runtime_error(std::string(where) + ": " + amsm_strerror(s)), status(s) {}
};
inline void check(int s, const char* where) {
  if (s != AMSM_OK) throw Error(s, where);
}

// Affine point in the ABI's format: x_mont | y_mont (2*limbs u64) + infinity flag.
struct Affine {
  std::vector<uint64_t> xy;
  bool infinity = true;
  bool operator==(const Affine& o) const { return infinity == o.infinity && (infinity || xy == o.xy); }
};
using Fr = std::array<uint64_t, 4>;  // Montgomery limbs of one scalar-field element (raw ark-ff memory)

class Context {
 public:
  explicit Context(int curve = AMSM_PALLAS, int device = 0, void* stream = nullptr) {
    check(amsm_ctx_create(&h_, curve, device, stream), "amsm_ctx_create");
  }
  // One context over several devices (amsm_ctx_create_multi): keys loaded through it are sharded, every MSM / commit / grouped
  // MSM / IPA round runs on all of them with one exchange of partial sums.  An empty list = the host backend.
  // replicate_below > 0: keys of up to that many generators are REPLICATED on every device instead (amsm.h AMSM_BASES_REPLICATE:
  // the independent MSMs of a commit round are dealt to the devices whole, no exchange)
  Context(int curve, const std::vector<int>& devices, size_t replicate_below = 0) {
    check(amsm_ctx_create_multi(&h_, curve, devices.data(), (int)devices.size()), "amsm_ctx_create_multi");
    if (replicate_below) check(amsm_ctx_set_replicate_below(h_, replicate_below), "amsm_ctx_set_replicate_below");
  }
  ~Context() { amsm_ctx_destroy(h_); }
  int num_devices() const { return amsm_ctx_num_devices(h_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  amsm_ctx* get() const { return h_; }
  int fq_limbs() const { return amsm_ctx_fq_limbs(h_); }
  void synchronize() { check(amsm_ctx_synchronize(h_), "amsm_ctx_synchronize"); }

 private:
  amsm_ctx* h_ = nullptr;
};

// `Vec<G::ScalarField>` resident in HBM.
class FrVector {
 public:
  FrVector(Context& ctx, size_t n) : ctx_(&ctx), n_(n) {
    check(amsm_dev_alloc(ctx.get(), n * 32, &p_), "amsm_dev_alloc");
  }
  FrVector(Context& ctx, const std::vector<Fr>& host) : FrVector(ctx, host.size()) {
    if (n_) check(amsm_dev_upload(ctx.get(), p_, host.data(), n_ * 32), "amsm_dev_upload");
  }
  static FrVector random(Context& ctx, uint64_t seed, size_t n, bool mont) {
    FrVector v(ctx, n);
    check(amsm_vec_random(ctx.get(), seed, n, mont ? 1 : 0, v.p_), "amsm_vec_random");
    return v;
  }
  ~FrVector() {
    if (p_) amsm_dev_free(ctx_->get(), p_);
  }
  FrVector(FrVector&& o) noexcept : ctx_(o.ctx_), n_(o.n_), p_(o.p_) { o.p_ = nullptr; }
  FrVector& operator=(FrVector&& o) noexcept {
    if (this != &o) {
      if (p_) amsm_dev_free(ctx_->get(), p_);
      ctx_ = o.ctx_, n_ = o.n_, p_ = o.p_;
      o.p_ = nullptr;
    }
    return *this;
  }
  FrVector(const FrVector&) = delete;
  FrVector& operator=(const FrVector&) = delete;
  size_t len() const { return n_; }
  const void* ptr() const { return p_; }
  void* ptr() { return p_; }
  std::vector<Fr> to_host() const {
    std::vector<Fr> out(n_);
    if (n_) check(amsm_dev_download(ctx_->get(), out.data(), p_, n_ * 32), "amsm_dev_download");
    return out;
  }
  Context& ctx() const { return *ctx_; }

 private:
  Context* ctx_;
  size_t n_;
  void* p_ = nullptr;
};

// trivial_pc::CommitterKey{generators, hiding_generator}: generators live in HBM (precomputed window multiples).
class CommitterKey {
 public:
  static CommitterKey load(Context& ctx, const std::vector<uint64_t>& xy_mont, const std::vector<uint8_t>* is_inf,
                           unsigned flags = AMSM_BASES_DEFAULT) {
    CommitterKey k(ctx);
    size_t n = xy_mont.size() / (2 * (size_t)ctx.fq_limbs());
    check(amsm_bases_load(ctx.get(), xy_mont.data(), is_inf ? is_inf->data() : nullptr, n, flags, &k.h_),
          "amsm_bases_load");
    return k;
  }
  ~CommitterKey() { amsm_bases_free(h_); }
  CommitterKey(CommitterKey&& o) noexcept : hiding_generator(std::move(o.hiding_generator)), ctx_(o.ctx_), h_(o.h_) {
    o.h_ = nullptr;
  }
  CommitterKey(const CommitterKey&) = delete;
  size_t supported_num_elems() const { return amsm_bases_len(h_); }
  const amsm_bases* get() const { return h_; }
  Context& ctx() const { return *ctx_; }
  std::vector<uint64_t> read(size_t off, size_t n) const {
    std::vector<uint64_t> xy(n * 2 * (size_t)ctx_->fq_limbs());
    check(amsm_bases_read(ctx_->get(), h_, off, n, xy.data(), nullptr), "amsm_bases_read");
    return xy;
  }
  // key[i] + x * key[n_half + i], key to key on the device: the `key_l += key_r * xi` of the IPA opening
  // (ark_poly_commit::ipa_pc ext, under src/ipa_pc_as/mod.rs:454).  x in Montgomery form.
  CommitterKey fold(size_t n_half, const Fr& x_mont, unsigned nbits = 255) const {
    CommitterKey k(*ctx_);
    check(amsm_bases_fold(ctx_->get(), h_, n_half, x_mont.data(), nbits, &k.h_), "amsm_bases_fold");
    return k;
  }
  std::vector<uint64_t> hiding_generator;  // affine x|y (Montgomery), host side

 private:
  friend struct PedersenCommitment;
  explicit CommitterKey(Context& ctx) : ctx_(&ctx) {}
  Context* ctx_;
  amsm_bases* h_ = nullptr;
};

struct VariableBaseMSM {
  // multi_scalar_mul(&bases, &scalars): scalars = canonical BigInt limbs (`into_repr()`), n*4 u64 on the host.
  static Affine multi_scalar_mul(const CommitterKey& bases, const std::vector<Fr>& scalars) {
    Affine out;
    out.xy.assign(2 * (size_t)bases.ctx().fq_limbs(), 0);
    uint8_t inf = 0;
    check(amsm_msm(bases.ctx().get(), bases.get(), 0, reinterpret_cast<const uint64_t*>(scalars.data()), scalars.size(),
                   0, out.xy.data(), &inf),
          "amsm_msm");
    out.infinity = inf != 0;
    return out;
  }
  // the ark-ec call shape itself: `multi_scalar_mul(bases: &[G], scalars: &[BigInt])` -- both host slices, nothing kept
  // (amsm_msm_oneshot).  bases_xy: n * 2 * limbs u64 (x_mont | y_mont), is_inf: n flags or null; min(n_bases, scalars.size()) pairs.
  static Affine multi_scalar_mul(Context& ctx, const uint64_t* bases_xy, const uint8_t* is_inf, size_t n_bases,
                                 const std::vector<Fr>& scalars) {
    Affine out;
    out.xy.assign(2 * (size_t)ctx.fq_limbs(), 0);
    uint8_t inf = 0;
    check(amsm_msm_oneshot(ctx.get(), bases_xy, is_inf, n_bases, reinterpret_cast<const uint64_t*>(scalars.data()), scalars.size(), 0,
                           out.xy.data(), &inf),
          "amsm_msm_oneshot");
    out.infinity = inf != 0;
    return out;
  }
  // device-resident Montgomery scalars (the form the vector kernels produce)
  static Affine multi_scalar_mul(const CommitterKey& bases, const FrVector& scalars_mont) {
    Affine out;
    out.xy.assign(2 * (size_t)bases.ctx().fq_limbs(), 0);
    uint8_t inf = 0;
    check(amsm_msm_device(bases.ctx().get(), bases.get(), 0, scalars_mont.ptr(), scalars_mont.len(), 1, out.xy.data(),
                          &inf),
          "amsm_msm_device");
    out.infinity = inf != 0;
    return out;
  }
};

// Several MSMs in one call (pipelined on the device).  All take device-resident Montgomery scalars.
struct MsmBatch {
  // k MSMs over host slices of canonical scalars (`&[BigInt]`), uploads overlapped with the MSMs: amsm_msm_batch
  static std::vector<Affine> same_bases_host(const CommitterKey& bases, const std::vector<const std::vector<Fr>*>& vecs,
                                             bool mont = false) {
    const size_t k = vecs.size(), n = k ? vecs[0]->size() : 0;
    std::vector<const uint64_t*> ptrs(k);
    for (size_t v = 0; v < k; v++) {
      if (vecs[v]->size() != n) throw Error(AMSM_E_INVALID_ARG, "same_bases_host: vectors of one length");
      ptrs[v] = reinterpret_cast<const uint64_t*>(vecs[v]->data());
    }
    return run(bases, k, [&](uint64_t* xy, uint8_t* inf) {
      return amsm_msm_batch(bases.ctx().get(), bases.get(), 0, ptrs.data(), k, n, mont ? 1 : 0, xy, inf);
    }, "amsm_msm_batch");
  }
  // the same generators, several scalar vectors (the prover's back-to-back commits, src/hp_as/mod.rs:354-388)
  static std::vector<Affine> same_bases(const CommitterKey& bases, const std::vector<const FrVector*>& vecs) {
    std::vector<const void*> ptrs;
    size_t len = 0;
    for (const FrVector* v : vecs) {
      ptrs.push_back(v->ptr());
      len = v->len();
    }
    return run(bases, ptrs.size(), [&](uint64_t* xy, uint8_t* inf) {
      return amsm_msm_batch_device(bases.ctx().get(), bases.get(), 0, ptrs.data(), ptrs.size(), len, 1, xy, inf);
    }, "amsm_msm_batch_device");
  }
  // windows of one key: job = (first generator, scalars)
  static std::vector<Affine> windows(const CommitterKey& bases, const std::vector<std::pair<size_t, const FrVector*>>& jobs) {
    std::vector<size_t> offs, ns;
    std::vector<const void*> ptrs;
    for (auto& j : jobs) {
      offs.push_back(j.first);
      ns.push_back(j.second->len());
      ptrs.push_back(j.second->ptr());
    }
    return run(bases, jobs.size(), [&](uint64_t* xy, uint8_t* inf) {
      return amsm_msm_multi_device(bases.ctx().get(), bases.get(), jobs.size(), offs.data(), ptrs.data(), ns.data(), 1, xy, inf);
    }, "amsm_msm_multi_device");
  }
  // two sums over the index classes ((i >> group_shift) & 1) of ONE scalar vector, in one pass
  static std::vector<Affine> grouped(const CommitterKey& bases, const FrVector& scalars, unsigned group_shift) {
    return grouped(bases, scalars, scalars.len(), group_shift);
  }
  // ... over the first n entries of `scalars`
  static std::vector<Affine> grouped(const CommitterKey& bases, const FrVector& scalars, size_t n, unsigned group_shift) {
    return run(bases, 2, [&](uint64_t* xy, uint8_t* inf) {
      return amsm_msm_grouped_device(bases.ctx().get(), bases.get(), 0, scalars.ptr(), n, 1, group_shift, xy, inf);
    }, "amsm_msm_grouped_device");
  }

 private:
  template <class F>
  static std::vector<Affine> run(const CommitterKey& bases, size_t k, F&& call, const char* where) {
    size_t w = 2 * (size_t)bases.ctx().fq_limbs();
    std::vector<uint64_t> xy(k * w + 1);
    std::vector<uint8_t> inf(k + 1);
    check(call(xy.data(), inf.data()), where);
    std::vector<Affine> out(k);
    for (size_t i = 0; i < k; i++) {
      out[i].xy.assign(xy.begin() + (long)(i * w), xy.begin() + (long)((i + 1) * w));
      out[i].infinity = inf[i] != 0;
    }
    return out;
  }
};

struct PedersenCommitment {
  // setup(n): n generators + a hiding generator (synthetic stream; see amsm_bases_generate)
  static CommitterKey setup(Context& ctx, size_t n, uint64_t seed = 0x5EED1001ull, unsigned flags = AMSM_BASES_DEFAULT) {
    amsm_bases* tmp = nullptr;
    check(amsm_bases_generate(ctx.get(), seed, n + 1, AMSM_BASES_NO_PRECOMPUTE, &tmp), "amsm_bases_generate");
    size_t w = 2 * (size_t)ctx.fq_limbs();
    std::vector<uint64_t> xy((n + 1) * w);
    int s = amsm_bases_read(ctx.get(), tmp, 0, n + 1, xy.data(), nullptr);
    amsm_bases_free(tmp);
    check(s, "amsm_bases_read");
    CommitterKey k(ctx);
    check(amsm_bases_load(ctx.get(), xy.data(), nullptr, n, flags, &k.h_), "amsm_bases_load");
    k.hiding_generator.assign(xy.begin() + (long)(n * w), xy.end());
    return k;
  }
  // commit(ck, elems, randomizer): elems Montgomery; randomizer == nullptr <=> None
  static Affine commit(const CommitterKey& ck, const FrVector& elems, const Fr* randomizer = nullptr) {
    Affine out;
    out.xy.assign(2 * (size_t)ck.ctx().fq_limbs(), 0);
    uint8_t inf = 0;
    if (randomizer && ck.hiding_generator.empty()) throw Error(AMSM_E_INVALID_ARG, "commit: key has no hiding generator");
    check(amsm_pedersen_commit_device(ck.ctx().get(), ck.get(), elems.ptr(), elems.len(),
                                      randomizer ? randomizer->data() : nullptr,
                                      randomizer ? ck.hiding_generator.data() : nullptr, out.xy.data(), &inf),
          "amsm_pedersen_commit_device");
    out.infinity = inf != 0;
    return out;
  }
  // commit(ck, elems[v], randomizers[v]) for HOST vectors (`&[Fr]` memory, lengths may differ) in one call: the upload of
  // vector v + 1 overlaps the MSM of vector v (amsm_pedersen_commit_batch) -- what the reference's back-to-back commits map
  // to behind a patched PedersenCommitment (src/hp_as/mod.rs:372-385, src/r1cs_nark_as/r1cs_nark/mod.rs:216-218).
  static std::vector<Affine> commit_batch_host(const CommitterKey& ck, const std::vector<const std::vector<Fr>*>& elems,
                                               const std::vector<const Fr*>& randomizers = {}) {
    Context& ctx = ck.ctx();
    const size_t k = elems.size(), w = 2 * (size_t)ctx.fq_limbs();
    std::vector<const uint64_t*> ptrs(k), rptrs(k, nullptr);
    std::vector<size_t> ns(k);
    bool any = false;
    for (size_t v = 0; v < k; v++) {
      ptrs[v] = reinterpret_cast<const uint64_t*>(elems[v]->data());
      ns[v] = elems[v]->size();
      if (v < randomizers.size() && randomizers[v]) {
        rptrs[v] = randomizers[v]->data();
        any = true;
      }
    }
    if (any && ck.hiding_generator.empty()) throw Error(AMSM_E_INVALID_ARG, "commit_batch_host: key has no hiding generator");
    std::vector<uint64_t> xy(k * w, 0);
    std::vector<uint8_t> inf(k, 0);
    check(amsm_pedersen_commit_batch(ctx.get(), ck.get(), ptrs.data(), ns.data(), k, any ? rptrs.data() : nullptr,
                                     any ? ck.hiding_generator.data() : nullptr, xy.data(), inf.data()),
          "amsm_pedersen_commit_batch");
    std::vector<Affine> out(k);
    for (size_t v = 0; v < k; v++) {
      out[v].xy.assign(xy.begin() + (long)(v * w), xy.begin() + (long)((v + 1) * w));
      out[v].infinity = inf[v] != 0;
    }
    return out;
  }
  // Several independent commitments to vectors of one length as ONE pipelined MSM batch; the hiding terms
  // randomizer * hiding_generator are added on the host.  Same points as vecs.size() calls of commit().
  static std::vector<Affine> commit_batch(const CommitterKey& ck, const std::vector<const FrVector*>& vecs,
                                          const std::vector<const Fr*>& randomizers) {
    Context& ctx = ck.ctx();
    std::vector<Affine> out = MsmBatch::same_bases(ck, vecs);
    const size_t w = 2 * (size_t)ctx.fq_limbs();
    const int curve = amsm_ctx_curve(ctx.get());
    for (size_t i = 0; i < out.size(); i++) {
      const Fr* r = i < randomizers.size() ? randomizers[i] : nullptr;
      if (!r) continue;
      if (ck.hiding_generator.empty()) throw Error(AMSM_E_INVALID_ARG, "commit_batch: key has no hiding generator");
      std::vector<uint64_t> xy(out[i].xy);
      xy.insert(xy.end(), ck.hiding_generator.begin(), ck.hiding_generator.end());
      uint8_t infs[2] = {(uint8_t)(out[i].infinity ? 1 : 0), 0};
      Fr sc[2] = {Fr{1, 0, 0, 0}, *r};
      check(amsm_fr_to_mont(curve, sc[0].data(), 1, sc[0].data()), "amsm_fr_to_mont");
      Affine sum;
      sum.xy.assign(w, 0);
      uint8_t inf = 0;
      check(amsm_host_lincomb(curve, xy.data(), infs, reinterpret_cast<const uint64_t*>(sc), 2, sum.xy.data(), &inf),
            "amsm_host_lincomb");
      sum.infinity = inf != 0;
      out[i] = sum;
    }
    return out;
  }
};

// The scalar-field vector loops of ASForHadamardProducts (src/hp_as/mod.rs).
namespace hp_as {

inline FrVector compute_hp(const FrVector& a, const FrVector& b) {  // :278-285 (zip truncates)
  size_t n = a.len() < b.len() ? a.len() : b.len();
  FrVector out(a.ctx(), n);
  check(amsm_vec_hadamard(a.ctx().get(), a.ptr(), b.ptr(), out.ptr(), n), "amsm_vec_hadamard");
  return out;
}

inline FrVector combine_vectors(Context& ctx, const std::vector<const FrVector*>& vectors, const std::vector<Fr>& challenges,
                                const FrVector* hiding = nullptr) {  // :492-512
  size_t n = hiding ? hiding->len() : 0;
  std::vector<const void*> ptrs;
  std::vector<size_t> lens;
  for (auto* v : vectors) {
    ptrs.push_back(v->ptr());
    lens.push_back(v->len());
    if (v->len() > n) n = v->len();
  }
  FrVector out(ctx, n);
  check(amsm_vec_combine(ctx.get(), ptrs.data(), lens.data(), vectors.size(),
                         reinterpret_cast<const uint64_t*>(challenges.data()), hiding ? hiding->ptr() : nullptr,
                         hiding ? hiding->len() : 0, out.ptr(), n),
        "amsm_vec_combine");
  return out;
}

inline FrVector scale_vector(const FrVector& v, const Fr& coeff) {  // :482-489
  return combine_vectors(v.ctx(), {&v}, {coeff});
}

// :288-349.  Returns the 2n-1 coefficient vectors; index n-1 (never committed, :373-375) is computed only when
// `with_uncommitted` is set.
inline std::vector<std::unique_ptr<FrVector>> compute_t_vecs(Context& ctx, const std::vector<const FrVector*>& a_vecs,
                                                             const std::vector<const FrVector*>& b_vecs,
                                                             const std::vector<Fr>& mu, size_t hp_vec_len,
                                                             const FrVector* hiding_a = nullptr,
                                                             const FrVector* hiding_b = nullptr,
                                                             bool with_uncommitted = true) {
  size_t n = a_vecs.size();
  std::vector<const void*> pa, pb;
  std::vector<size_t> la, lb;
  for (size_t j = 0; j < n; j++) {
    pa.push_back(a_vecs[j]->ptr());
    la.push_back(a_vecs[j]->len());
    pb.push_back(b_vecs[j]->ptr());
    lb.push_back(b_vecs[j]->len());
  }
  std::vector<std::unique_ptr<FrVector>> out;
  std::vector<void*> pt;
  for (size_t k = 0; k + 1 < 2 * n; k++) {
    if (k == n - 1 && !with_uncommitted) {
      out.emplace_back(nullptr);
      pt.push_back(nullptr);
    } else {
      out.emplace_back(new FrVector(ctx, hp_vec_len));
      pt.push_back(out.back()->ptr());
    }
  }
  check(amsm_hp_t_vecs(ctx.get(), pa.data(), la.data(), pb.data(), lb.data(), n,
                       reinterpret_cast<const uint64_t*>(mu.data()), mu.size(), hiding_a ? hiding_a->ptr() : nullptr,
                       hiding_a ? hiding_a->len() : 0, hiding_b ? hiding_b->ptr() : nullptr,
                       hiding_b ? hiding_b->len() : 0, pt.data(), hp_vec_len),
        "amsm_hp_t_vecs");
  return out;
}

// :354-388: commitments to every t_vec except index n-1, as (low, high); one batched call (3 MSMs in flight).
inline std::pair<std::vector<Affine>, std::vector<Affine>> compute_product_poly_comm(
    const CommitterKey& ck, const std::vector<std::unique_ptr<FrVector>>& t_vecs) {
  std::pair<std::vector<Affine>, std::vector<Affine>> out;
  if (t_vecs.empty()) return out;
  size_t n = (t_vecs.size() + 1) / 2;
  std::vector<const void*> ptrs;
  size_t len = 0;
  for (size_t i = 0; i < t_vecs.size(); i++)
    if (i != n - 1) {
      ptrs.push_back(t_vecs[i]->ptr());
      len = t_vecs[i]->len();
    }
  size_t w = 2 * (size_t)ck.ctx().fq_limbs();
  std::vector<uint64_t> xy(ptrs.size() * w);
  std::vector<uint8_t> inf(ptrs.size());
  check(amsm_msm_batch_device(ck.ctx().get(), ck.get(), 0, ptrs.data(), ptrs.size(), len, 1, xy.data(), inf.data()),
        "amsm_msm_batch_device");
  for (size_t i = 0; i < ptrs.size(); i++) {
    Affine p;
    p.xy.assign(xy.begin() + (long)(i * w), xy.begin() + (long)((i + 1) * w));
    p.infinity = inf[i] != 0;
    (i < n - 1 ? out.first : out.second).push_back(std::move(p));
  }
  return out;
}

}  // namespace hp_as
}  // namespace amsm
