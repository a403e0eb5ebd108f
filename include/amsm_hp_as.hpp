// C++17 restatement of `ASForHadamardProducts` (reference: src/hp_as/mod.rs -- index :620-644, prove :646-813,
// verify :815-892, decide :894-925; data structures src/hp_as/data_structures.rs) above the C ABI of include/amsm.h.
//
// the reference is compiled (Rust) code and no Rust toolchain exists here, so the
// scheme driver is restated in C++ next to the Python mirror (accumulation_amd/hp_as.py, which the tests use as the
// cross-check: same sponge, same transcript, byte-identical accumulators).  Every O(len) loop and every MSM runs on
// the GPU (amsm::hp_as::{compute_hp, combine_vectors, compute_t_vecs, compute_product_poly_comm} of amsm.hpp); the
// O(#inputs) work -- challenges, their products, the linear combinations of a handful of commitments -- stays on the
// host through the ABI's host helpers (amsm_fr_*, amsm_host_lincomb), like rows a11 of the scope table.
//
// Sponge: the reference is generic over `S: CryptographicSponge` and its tests use ark-sponge's Poseidon, whose
// parameters are not in the reference tree.  The driver is generic over the sponge type too; `Sha256Sponge` below is the
// same stand-in as accumulation_amd/sponge.py (NOT Poseidon: transcripts are self-consistent, not comparable with a
// Rust run).
#pragma once
#include <cstring>
#include <functional>
#include <optional>

#include "amsm.hpp"

namespace amsm {
namespace hp_as {

// ---- SHA-256 (FIPS 180-4), only what the stand-in sponge needs ------------------------------------------------
class Sha256 {
 public:
  static std::array<uint8_t, 32> digest(const std::vector<uint8_t>& msg) {
    uint32_t h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    std::vector<uint8_t> m(msg);
    uint64_t bits = (uint64_t)msg.size() * 8;
    m.push_back(0x80);
    while (m.size() % 64 != 56) m.push_back(0);
    for (int i = 7; i >= 0; i--) m.push_back((uint8_t)(bits >> (8 * i)));
    for (size_t off = 0; off < m.size(); off += 64) compress(h, m.data() + off);
    std::array<uint8_t, 32> out;
    for (int i = 0; i < 8; i++)
      for (int j = 0; j < 4; j++) out[4 * i + j] = (uint8_t)(h[i] >> (24 - 8 * j));
    return out;
  }

 private:
  static uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
  static void compress(uint32_t h[8], const uint8_t* p) {
    static const uint32_t K[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u,
        0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u,
        0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u,
        0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u,
        0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u,
        0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au,
        0x5b9cca4fu, 0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u,
        0xc67178f2u};
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
      w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
      uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
      uint32_t ch = (e & f) ^ (~e & g);
      uint32_t t1 = hh + S1 + ch + K[i] + w[i];
      uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
      uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
      uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
};

// The `Sponge` template argument of every scheme driver (the reference's `S: CryptographicSponge<ConstraintF<G>>`,
// src/hp_as/mod.rs:98-103) is any type with these members; the absorb calls of the drivers follow the reference's `absorb!`
// item lists one to one:
//   void absorb_bytes(const std::vector<uint8_t>&)   a byte string item (hashes, `to_bytes!(..)` of scalars, Option tags)
//   void absorb_u64(uint64_t)                        a usize item
//   void absorb_len(uint64_t)                        the length of a Vec that is about to be absorbed item by item: FRAMING for
//                                                    hash-based sponges only -- the reference's field-element encoding of a
//                                                    Vec has no length (include/amsm_poseidon.hpp ignores it)
//   void absorb_point(const Affine&), absorb_points(const std::vector<Affine>&)
//   Fr squeeze_bits(unsigned n_bits)                 `squeeze_nonnative_field_elements_with_sizes(&[Truncated(n_bits)])`
//   std::vector<Fr> squeeze_field_elements(size_t count, unsigned n_bits)    the same for `[Truncated(n_bits); count]`: ONE
//                                                    squeeze (a Poseidon sponge cuts the windows out of one bit stream)
//   S fork(const char* domain)
//   void for_curve(int curve)                        OPTIONAL: a sponge whose field is the curve's base field (the reference's
//                                                    `PoseidonSponge<ConstraintF<G>>`) re-creates itself for the context's curve; the
//                                                    drivers call it on every sponge they are handed (a default-constructed
//                                                    argument knows no context) or create, before the first absorb
// Sha256Sponge: same construction as accumulation_amd/sponge.py:Sha256Sponge (byte-identical challenges).
template <class S>
auto sponge_for_curve(S& s, int curve, int) -> decltype(s.for_curve(curve), void()) {
  s.for_curve(curve);
}
template <class S>
void sponge_for_curve(S&, int, long) {}
template <class S>
S fresh_sponge(int curve) {
  S s;
  sponge_for_curve(s, curve, 0);
  return s;
}

class Sha256Sponge {
 public:
  Sha256Sponge() {
    const char* tag = "amsm-sha256-sponge";
    state_ = Sha256::digest(std::vector<uint8_t>(tag, tag + strlen(tag)));
  }
  void absorb_bytes(const std::vector<uint8_t>& b) {
    std::vector<uint8_t> m(state_.begin(), state_.end());
    m.push_back('A');
    push_u64(m, b.size());
    m.insert(m.end(), b.begin(), b.end());
    state_ = Sha256::digest(m);
    ctr_ = 0;
  }
  void absorb_u64(uint64_t x) {
    std::vector<uint8_t> b;
    push_u64(b, x);
    absorb_bytes(b);
  }
  void absorb_point(const Affine& p) {
    std::vector<uint8_t> b;
    for (uint64_t w : p.xy) push_u64(b, w);
    b.push_back(p.infinity ? 1 : 0);
    absorb_bytes(b);
  }
  void absorb_len(uint64_t n) { absorb_u64(n); }
  void absorb_points(const std::vector<Affine>& pts) {
    absorb_len(pts.size());
    for (auto& p : pts) absorb_point(p);
  }
  std::vector<Fr> squeeze_field_elements(size_t count, unsigned n_bits) {
    std::vector<Fr> out;
    for (size_t i = 0; i < count; i++) out.push_back(squeeze_bits(n_bits));
    return out;
  }
  // `squeeze_nonnative_field_elements_with_sizes(Truncated(n_bits))`, n_bits <= 256: canonical limbs
  Fr squeeze_bits(unsigned n_bits) {
    std::vector<uint8_t> out;
    while (out.size() * 8 < n_bits) {
      std::vector<uint8_t> m(state_.begin(), state_.end());
      m.push_back('S');
      push_u64(m, ctr_++);
      auto d = Sha256::digest(m);
      out.insert(out.end(), d.begin(), d.end());
    }
    Fr r = {0, 0, 0, 0};
    for (unsigned bit = 0; bit < n_bits && bit < 256; bit++)
      if ((out[bit / 8] >> (bit % 8)) & 1) r[bit / 64] |= 1ull << (bit % 64);
    return r;
  }

  // domain-separated child sponge (the reference forks per protocol, src/r1cs_nark_as/mod.rs:112-125)
  Sha256Sponge fork(const char* domain) const {
    std::vector<uint8_t> m(state_.begin(), state_.end());
    m.push_back('F');
    m.insert(m.end(), domain, domain + strlen(domain));
    Sha256Sponge c;
    c.state_ = Sha256::digest(m);
    c.ctr_ = 0;
    return c;
  }

 private:
  static void push_u64(std::vector<uint8_t>& v, uint64_t x) {
    for (int i = 0; i < 8; i++) v.push_back((uint8_t)(x >> (8 * i)));
  }
  std::array<uint8_t, 32> state_;
  uint64_t ctr_ = 0;
};

// ---- errors (src/error.rs:8-20) ------------------------------------------------------------------------------
struct ASError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
struct MalformedAccumulator : ASError {
  using ASError::ASError;
};
struct MalformedInput : ASError {
  using ASError::ASError;
};
struct MissingRng : ASError {
  using ASError::ASError;
};

// ---- host scalar helpers: Montgomery-form Fr through the ABI ----------------------------------------------------
struct FrOps {
  int curve;
  Fr mul(const Fr& a, const Fr& b) const {
    Fr r;
    check(amsm_fr_mul(curve, a.data(), b.data(), 1, r.data()), "amsm_fr_mul");
    return r;
  }
  Fr add(const Fr& a, const Fr& b) const {
    Fr r;
    check(amsm_fr_add(curve, a.data(), b.data(), 1, r.data()), "amsm_fr_add");
    return r;
  }
  Fr to_mont(const Fr& canonical) const {
    Fr r;
    check(amsm_fr_to_mont(curve, canonical.data(), 1, r.data()), "amsm_fr_to_mont");
    return r;
  }
  Fr one() const { return to_mont(Fr{1, 0, 0, 0}); }
  Fr zero() const { return Fr{0, 0, 0, 0}; }
};

// host: sum_i scalars[i] * points[i] -> affine (scalars in Montgomery form); the O(#inputs) point arithmetic of the
// verifiers (`combine_commitments`, src/hp_as/mod.rs:391-406)
inline Affine host_lincomb(Context& ctx, const std::vector<const Affine*>& points, const std::vector<Fr>& scalars) {
  size_t k = points.size(), w = 2 * (size_t)ctx.fq_limbs();
  Affine out;
  out.xy.assign(w, 0);
  out.infinity = true;
  if (k == 0) return out;
  std::vector<uint64_t> xy(k * w);
  std::vector<uint8_t> inf(k);
  for (size_t i = 0; i < k; i++) {
    std::copy(points[i]->xy.begin(), points[i]->xy.end(), xy.begin() + (long)(i * w));
    inf[i] = points[i]->infinity ? 1 : 0;
  }
  uint8_t oinf = 0;
  check(amsm_host_lincomb(amsm_ctx_curve(ctx.get()), xy.data(), inf.data(), reinterpret_cast<const uint64_t*>(scalars.data()), k,
                          out.xy.data(), &oinf),
        "amsm_host_lincomb");
  out.infinity = oinf != 0;
  if (out.infinity) std::fill(out.xy.begin(), out.xy.end(), 0);
  return out;
}

// host: independent combinations in one library call (amsm_host_lincomb_batch: a small pool of host threads, one
// normalisation); jobs[j] = (points, Montgomery scalars).  Same results as host_lincomb job by job.
using LincombJob = std::pair<std::vector<const Affine*>, std::vector<Fr>>;
inline std::vector<Affine> host_lincomb_batch(Context& ctx, const std::vector<LincombJob>& jobs) {
  const size_t nj = jobs.size(), w = 2 * (size_t)ctx.fq_limbs();
  std::vector<Affine> out(nj);
  if (nj == 0) return out;
  std::vector<std::vector<uint64_t>> xy(nj);
  std::vector<std::vector<uint8_t>> inf(nj);
  std::vector<size_t> n_terms(nj);
  std::vector<const uint64_t*> xy_p(nj), sc_p(nj);
  std::vector<const uint8_t*> inf_p(nj);
  for (size_t j = 0; j < nj; j++) {
    const size_t k = jobs[j].first.size();
    if (jobs[j].second.size() < k) throw Error(AMSM_E_INVALID_ARG, "host_lincomb_batch: fewer scalars than points");
    n_terms[j] = k;
    xy[j].resize(k * w);
    inf[j].resize(k);
    for (size_t i = 0; i < k; i++) {
      std::copy(jobs[j].first[i]->xy.begin(), jobs[j].first[i]->xy.end(), xy[j].begin() + (long)(i * w));
      inf[j][i] = jobs[j].first[i]->infinity ? 1 : 0;
    }
    xy_p[j] = xy[j].data();
    inf_p[j] = inf[j].data();
    sc_p[j] = reinterpret_cast<const uint64_t*>(jobs[j].second.data());
  }
  std::vector<uint64_t> oxy(nj * w);
  std::vector<uint8_t> oinf(nj, 0);
  check(amsm_host_lincomb_batch(amsm_ctx_curve(ctx.get()), nj, n_terms.data(), xy_p.data(), inf_p.data(), sc_p.data(), oxy.data(),
                                oinf.data()),
        "amsm_host_lincomb_batch");
  for (size_t j = 0; j < nj; j++) {
    out[j].infinity = oinf[j] != 0 || n_terms[j] == 0;
    if (out[j].infinity) out[j].xy.assign(w, 0);
    else out[j].xy.assign(oxy.begin() + (long)(j * w), oxy.begin() + (long)((j + 1) * w));
  }
  return out;
}

// ---- data structures (src/hp_as/data_structures.rs) ------------------------------------------------------------
struct InputInstance {  // :14-33
  Affine comm_1, comm_2, comm_3;
  static InputInstance zero(Context& ctx) {
    Affine z;
    z.xy.assign(2 * (size_t)ctx.fq_limbs(), 0);
    z.infinity = true;
    return InputInstance{z, z, z};
  }
  bool operator==(const InputInstance& o) const { return comm_1 == o.comm_1 && comm_2 == o.comm_2 && comm_3 == o.comm_3; }
  template <class S>
  void absorb_into(S& sponge) const {
    sponge.absorb_point(comm_1);
    sponge.absorb_point(comm_2);
    sponge.absorb_point(comm_3);
  }
};
struct InputWitnessRandomness {  // :77-90, Montgomery form
  Fr rand_1, rand_2, rand_3;
};
struct InputWitness {  // :54-74
  std::shared_ptr<FrVector> a_vec, b_vec;
  std::optional<InputWitnessRandomness> randomness;
};
struct ProductPolynomialCommitment {  // :95-114
  std::vector<Affine> low, high;
};
struct ProofHidingCommitments {
  Affine comm_1, comm_2, comm_3;
};
struct Proof {
  ProductPolynomialCommitment product_poly_comm;
  std::optional<ProofHidingCommitments> hiding_comms;
};
struct Accumulator {  // also the shape of an Input: {instance, witness}
  InputInstance instance;
  InputWitness witness;
};

constexpr unsigned CHALLENGE_SIZE = 128;  // src/hp_as/mod.rs:29

inline std::shared_ptr<FrVector> filled(Context& ctx, const Fr& value_mont, size_t n) {  // vec![value; n]
  auto v = std::make_shared<FrVector>(ctx, n);
  check(amsm_vec_fill(ctx.get(), value_mont.data(), n, v->ptr()), "amsm_vec_fill");
  return v;
}

// `rng`: empty = MakeZK::Disabled, else returns CANONICAL scalars < r (MakeZK::Enabled(rng)).
using Rng = std::function<Fr()>;

template <class Sponge = Sha256Sponge>
class ASForHadamardProducts {
 public:
  struct Keys {
    const CommitterKey* prover_key;
    size_t verifier_key;
    const CommitterKey* decider_key;
  };
  static Keys index(const CommitterKey& ck) { return Keys{&ck, ck.supported_num_elems(), &ck}; }  // :620-644

  // ---- prove (:646-813) ----------------------------------------------------------------------------------------
  static std::pair<Accumulator, Proof> prove(const CommitterKey& pk, std::vector<Accumulator> inputs,
                                             std::vector<Accumulator> old_accumulators, const Rng& rng = Rng(),
                                             Sponge sponge = Sponge()) {
    Context& ctx = pk.ctx();
    FrOps fr{amsm_ctx_curve(ctx.get())};
    sponge_for_curve(sponge, amsm_ctx_curve(ctx.get()), 0);
    const bool make_zk = (bool)rng;
    size_t num_all = inputs.size() + old_accumulators.size();
    if (!make_zk)  // :664-673
      for (auto* group : {&inputs, &old_accumulators})
        for (auto& x : *group)
          if (x.witness.randomness) throw MissingRng("Accumulating inputs with hiding requires rng.");
    size_t hp_vec_len = !old_accumulators.empty() ? old_accumulators[0].witness.a_vec->len()
                        : !inputs.empty()         ? inputs[0].witness.a_vec->len()
                                                  : pk.supported_num_elems();  // :676-682
    auto zero_input = [&]() {
      return Accumulator{InputInstance::zero(ctx), InputWitness{filled(ctx, fr.zero(), hp_vec_len), filled(ctx, fr.zero(), hp_vec_len), {}}};
    };
    if (num_all == 0) {  // default input :685-696
      inputs.push_back(zero_input());
      num_all++;
    }
    if (make_zk && num_all == 1) {  // placeholder for hiding :698-710
      inputs.push_back(zero_input());
      num_all++;
    }
    std::vector<const InputInstance*> instances;
    std::vector<const InputWitness*> witnesses;
    for (auto& x : inputs) {
      instances.push_back(&x.instance);
      witnesses.push_back(&check_witness(x.witness, pk, hp_vec_len, false));
    }
    for (auto& x : old_accumulators) {
      instances.push_back(&x.instance);
      witnesses.push_back(&check_witness(x.witness, pk, hp_vec_len, true));
    }
    // step 3: prover randomness (:179-230) -- the hiding vectors are CONSTANT vectors (`vec![rand; len]`)
    std::shared_ptr<FrVector> hid_a, hid_b;
    InputWitnessRandomness hid_r{};
    std::optional<ProofHidingCommitments> hiding_comms;
    if (make_zk) {
      Fr a_val = fr.to_mont(rng()), b_val = fr.to_mont(rng());
      hid_a = filled(ctx, a_val, hp_vec_len);
      hid_b = filled(ctx, b_val, hp_vec_len);
      hid_r = InputWitnessRandomness{fr.to_mont(rng()), fr.to_mont(rng()), fr.to_mont(rng())};
      FrVector p1 = compute_hp(*hid_a, *witnesses.front()->b_vec);
      FrVector p2 = compute_hp(*witnesses.back()->a_vec, *hid_b);
      FrVector sum = combine_vectors(ctx, {&p1, &p2}, {fr.one(), fr.one()});
      // three independent commitments: one pipelined batch (same points as three commit() calls, :196-214)
      auto c = PedersenCommitment::commit_batch(pk, {hid_a.get(), hid_b.get(), &sum}, {&hid_r.rand_1, &hid_r.rand_2, &hid_r.rand_3});
      hiding_comms = ProofHidingCommitments{c[0], c[1], c[2]};
    }
    absorb_statement(sponge, pk.supported_num_elems(), instances, hiding_comms);  // step 4
    std::vector<Fr> mu = squeeze_mu(sponge, fr, num_all, make_zk);
    // steps 5-8: t-vectors on the device (the uncommitted middle one is never materialised), batched commits
    std::vector<const FrVector*> av, bv;
    for (auto* w : witnesses) {
      av.push_back(w->a_vec.get());
      bv.push_back(w->b_vec.get());
    }
    auto t_vecs = compute_t_vecs(ctx, av, bv, mu, hp_vec_len, hid_a.get(), hid_b.get(), false);
    auto comm = compute_product_poly_comm(pk, t_vecs);
    Proof proof{ProductPolynomialCommitment{comm.first, comm.second}, hiding_comms};
    sponge.absorb_points(proof.product_poly_comm.low);  // step 9
    sponge.absorb_points(proof.product_poly_comm.high);
    std::vector<Fr> nu = squeeze_nu(sponge, fr, num_all);
    std::vector<Fr> chi;
    for (size_t i = 0; i < mu.size() && i < nu.size(); i++) chi.push_back(fr.mul(mu[i], nu[i]));
    InputInstance acc_instance = combined_commitments(ctx, fr, instances, proof, mu, nu, chi);  // steps 10-12
    // steps 13-15: combined openings (:535-607)
    std::vector<Fr> chi_n(chi.begin(), chi.begin() + (long)num_all), nu_n(nu.begin(), nu.begin() + (long)num_all);
    std::vector<const FrVector*> bv_rev(bv.rbegin(), bv.rend());
    std::shared_ptr<FrVector> a_open, b_open;
    if (make_zk) {
      FrVector add1 = scale_vector(*hid_a, mu[num_all]);
      FrVector add2 = scale_vector(*hid_b, mu[1]);
      a_open = std::make_shared<FrVector>(combine_vectors(ctx, av, chi_n, &add1));
      b_open = std::make_shared<FrVector>(combine_vectors(ctx, bv_rev, nu_n, &add2));
    } else {
      a_open = std::make_shared<FrVector>(combine_vectors(ctx, av, chi_n));
      b_open = std::make_shared<FrVector>(combine_vectors(ctx, bv_rev, nu_n));
    }
    std::optional<InputWitnessRandomness> randomness;
    if (make_zk) {  // :515-532
      auto comb = [&](auto get, const std::vector<Fr>& ch, bool reversed, const Fr& hiding) {
        Fr acc = fr.zero();
        for (size_t i = 0; i < witnesses.size(); i++) {
          const InputWitness* w = witnesses[reversed ? witnesses.size() - 1 - i : i];
          if (w->randomness) acc = fr.add(acc, fr.mul(get(*w->randomness), ch[i]));
        }
        return fr.add(acc, hiding);
      };
      Fr a_r = comb([](const InputWitnessRandomness& r) { return r.rand_1; }, chi, false, fr.mul(hid_r.rand_1, mu[num_all]));
      Fr b_r = comb([](const InputWitnessRandomness& r) { return r.rand_2; }, nu, true, fr.mul(hid_r.rand_2, mu[1]));
      Fr p_r = fr.mul(comb([](const InputWitnessRandomness& r) { return r.rand_3; }, mu, false, fr.mul(hid_r.rand_3, mu[num_all])),
                      nu[num_all - 1]);
      randomness = InputWitnessRandomness{a_r, b_r, p_r};
    }
    return {Accumulator{acc_instance, InputWitness{a_open, b_open, randomness}}, proof};
  }

  // ---- verify (:815-892; host only, no MSM) -------------------------------------------------------------------
  static bool verify(Context& ctx, size_t verifier_key, std::vector<InputInstance> input_instances,
                     const std::vector<InputInstance>& old_accumulator_instances, const InputInstance& new_accumulator_instance,
                     const Proof& proof, Sponge sponge = Sponge()) {
    FrOps fr{amsm_ctx_curve(ctx.get())};
    sponge_for_curve(sponge, amsm_ctx_curve(ctx.get()), 0);
    size_t num_all = input_instances.size() + old_accumulator_instances.size();
    const bool make_zk = proof.hiding_comms.has_value();
    if (num_all == 0) {
      input_instances.push_back(InputInstance::zero(ctx));
      num_all++;
    }
    if (make_zk && num_all == 1) {
      input_instances.push_back(InputInstance::zero(ctx));
      num_all++;
    }
    if (!check_proof_structure(proof, num_all)) return false;
    std::vector<const InputInstance*> instances;
    for (auto& i : input_instances) instances.push_back(&i);
    for (auto& i : old_accumulator_instances) instances.push_back(&i);
    absorb_statement(sponge, verifier_key, instances, proof.hiding_comms);
    std::vector<Fr> mu = squeeze_mu(sponge, fr, num_all, make_zk);
    sponge.absorb_points(proof.product_poly_comm.low);
    sponge.absorb_points(proof.product_poly_comm.high);
    std::vector<Fr> nu = squeeze_nu(sponge, fr, num_all);
    std::vector<Fr> chi;
    for (size_t i = 0; i < mu.size() && i < nu.size(); i++) chi.push_back(fr.mul(mu[i], nu[i]));
    return combined_commitments(ctx, fr, instances, proof, mu, nu, chi) == new_accumulator_instance;
  }

  // ---- decide (:894-925): three Pedersen commitments of full-length vectors, then equality ---------------------
  static bool decide(const CommitterKey& dk, const Accumulator& acc, Sponge = Sponge()) {
    const InputWitness& w = acc.witness;
    FrVector product = compute_hp(*w.a_vec, *w.b_vec);
    Affine c1, c2, c3;
    if (!w.randomness) {
      auto c = MsmBatch::same_bases(dk, {w.a_vec.get(), w.b_vec.get(), &product});
      c1 = c[0];
      c2 = c[1];
      c3 = c[2];
    } else {
      auto c = PedersenCommitment::commit_batch(dk, {w.a_vec.get(), w.b_vec.get(), &product},
                                                {&w.randomness->rand_1, &w.randomness->rand_2, &w.randomness->rand_3});
      c1 = c[0];
      c2 = c[1];
      c3 = c[2];
    }
    return c1 == acc.instance.comm_1 && c2 == acc.instance.comm_2 && c3 == acc.instance.comm_3;
  }

 private:
  static const InputWitness& check_witness(const InputWitness& w, const CommitterKey& pk, size_t vec_len, bool is_acc) {
    auto fail = [&](const char* msg) -> void {
      if (is_acc) throw MalformedAccumulator(msg);
      throw MalformedInput(msg);
    };
    if (w.a_vec->len() == 0 || w.b_vec->len() == 0)  // :117-126
      fail("A vector of the Hadamard Product relation with a length of 0 is unsupported.");
    if (w.a_vec->len() > pk.supported_num_elems() || w.b_vec->len() > pk.supported_num_elems())  // :129-141
      fail("A vector of the Hadamard Product relation has a length that exceeds the prover key's supported length.");
    if (w.a_vec->len() != w.b_vec->len() || w.a_vec->len() != vec_len)  // :144-154
      fail("All of the vectors of the Hadamard Product relation that have or will be accumulated must have equal lengths");
    return w;
  }
  static bool check_proof_structure(const Proof& proof, size_t num_inputs) {  // :160-176
    auto& p = proof.product_poly_comm;
    return p.low.size() == p.high.size() && p.low.size() == num_inputs - 1;
  }
  static std::vector<Fr> squeeze_mu(Sponge& sponge, const FrOps& fr, size_t num_inputs, bool make_zk) {  // :233-253
    std::vector<Fr> mu{fr.one()};
    if (num_inputs > 1)
      for (const Fr& c : sponge.squeeze_field_elements(num_inputs - 1, CHALLENGE_SIZE)) mu.push_back(fr.to_mont(c));
    if (make_zk) mu.push_back(fr.mul(mu[1], mu[num_inputs - 1]));
    return mu;
  }
  static std::vector<Fr> squeeze_nu(Sponge& sponge, const FrOps& fr, size_t num_inputs) {  // :256-275
    Fr nu1 = fr.to_mont(sponge.squeeze_bits(CHALLENGE_SIZE));
    std::vector<Fr> out;
    Fr cur = fr.one();
    for (size_t i = 0; i + 1 < 2 * num_inputs; i++) {
      out.push_back(cur);
      cur = fr.mul(cur, nu1);
    }
    return out;
  }
  static void absorb_statement(Sponge& sponge, size_t num_elems, const std::vector<const InputInstance*>& instances,
                               const std::optional<ProofHidingCommitments>& hiding) {  // absorb!(...) :753-758, :863-868
    sponge.absorb_u64(num_elems);
    sponge.absorb_len(instances.size());
    for (auto* i : instances) i->absorb_into(sponge);
    if (!hiding) {
      sponge.absorb_bytes({0});
    } else {
      sponge.absorb_bytes({1});
      sponge.absorb_point(hiding->comm_1);
      sponge.absorb_point(hiding->comm_2);
      sponge.absorb_point(hiding->comm_3);
    }
  }
  // host: sum_i scalars[i] * points[i] -> affine (`combine_commitments`, :391-406)
  static Affine lincomb(Context& ctx, const std::vector<const Affine*>& points, const std::vector<Fr>& scalars) {
    size_t k = points.size(), w = 2 * (size_t)ctx.fq_limbs();
    Affine out;
    out.xy.assign(w, 0);
    out.infinity = true;
    if (k == 0) return out;
    std::vector<uint64_t> xy(k * w);
    std::vector<uint8_t> inf(k);
    for (size_t i = 0; i < k; i++) {
      std::copy(points[i]->xy.begin(), points[i]->xy.end(), xy.begin() + (long)(i * w));
      inf[i] = points[i]->infinity ? 1 : 0;
    }
    uint8_t oinf = 0;
    check(amsm_host_lincomb(amsm_ctx_curve(ctx.get()), xy.data(), inf.data(), reinterpret_cast<const uint64_t*>(scalars.data()), k,
                            out.xy.data(), &oinf),
          "amsm_host_lincomb");
    out.infinity = oinf != 0;
    if (out.infinity) std::fill(out.xy.begin(), out.xy.end(), 0);
    return out;
  }
  static InputInstance combined_commitments(Context& ctx, const FrOps& fr, const std::vector<const InputInstance*>& instances,
                                            const Proof& proof, const std::vector<Fr>& mu, const std::vector<Fr>& nu,
                                            const std::vector<Fr>& chi) {  // :409-479
    size_t n = instances.size();
    std::vector<const Affine*> p1, p2, p3;
    std::vector<Fr> s1, s2, s3;
    for (size_t i = 0; i < n; i++) {
      p1.push_back(&instances[i]->comm_1);
      s1.push_back(chi[i]);
      p2.push_back(&instances[n - 1 - i]->comm_2);
      s2.push_back(nu[i]);
    }
    auto& low = proof.product_poly_comm.low;
    auto& high = proof.product_poly_comm.high;
    for (size_t i = 0; i < low.size(); i++) {
      p3.push_back(&low[i]);
      s3.push_back(nu[i]);
    }
    for (size_t i = 0; i < high.size(); i++) {
      p3.push_back(&high[i]);
      s3.push_back(nu[n + i]);
    }
    for (size_t i = 0; i < n; i++) {
      p3.push_back(&instances[i]->comm_3);
      s3.push_back(fr.mul(mu[i], nu[n - 1]));
    }
    if (proof.hiding_comms) {
      p1.push_back(&proof.hiding_comms->comm_1);
      s1.push_back(mu[n]);
      p2.push_back(&proof.hiding_comms->comm_2);
      s2.push_back(mu[1]);
      p3.push_back(&proof.hiding_comms->comm_3);
      s3.push_back(fr.mul(mu[n], nu[n - 1]));
    }
    std::vector<Affine> c = host_lincomb_batch(ctx, {{p1, s1}, {p2, s2}, {p3, s3}});  // three independent combinations
    return InputInstance{c[0], c[1], c[2]};
  }
};

}  // namespace hp_as
}  // namespace amsm
