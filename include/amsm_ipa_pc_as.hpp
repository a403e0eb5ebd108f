// C++17 restatement of `ark_poly_commit::ipa_pc::InnerProductArgPC` (ext: setup / trim / commit / open /
// succinct_check / check, the calls made at src/ipa_pc_as/mod.rs:155,198,400,418,454,525,836) and of
// `AtomicASForInnerProductArgPC` (reference: src/ipa_pc_as/mod.rs -- index :502-553, prove :555-676, verify :678-818,
// decide :820-848) above the C ABI of include/amsm.h.  SURVEY.md section 8(a) row a9 / section 8(f) rank 1: the opening
// runs on the device -- every round's two cross commitments are ONE grouped MSM over the original precomputed key
// (amsm_ipa_round_scalars + amsm_msm_grouped_device, no key folding), coefficient / evaluation-vector folds and inner
// products are vector kernels, the final key is one MSM with the check polynomial's coefficients; only the O(log d)
// challenge / point algebra stays on the host.  Same structure and stand-in sponge as accumulation_amd/ipa_pc.py and
// ipa_pc_as.py; tests compare the two byte for byte.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <exception>
#include <future>
#include <mutex>

#include "amsm_hp_as.hpp"

namespace amsm {
namespace ipa_pc {

using hp_as::FrOps;
using hp_as::host_lincomb;
using hp_as::Sha256Sponge;

constexpr unsigned CHALLENGE_SIZE = 128;

struct FrX : FrOps {  // the few extra host operations the IPA needs
  explicit FrX(int c) : FrOps{c} {}
  Fr sub(const Fr& a, const Fr& b) const {
    Fr r;
    check(amsm_fr_sub(curve, a.data(), b.data(), 1, r.data()), "amsm_fr_sub");
    return r;
  }
  Fr neg(const Fr& a) const { return sub(zero(), a); }
  Fr inv(const Fr& a) const {
    Fr r;
    check(amsm_fr_inv(curve, a.data(), 1, r.data()), "amsm_fr_inv");
    return r;
  }
  Fr canon(const Fr& mont) const {
    Fr r;
    check(amsm_fr_from_mont(curve, mont.data(), 1, r.data()), "amsm_fr_from_mont");
    return r;
  }
};
inline std::vector<uint8_t> le_bytes(const Fr& canonical, size_t n = 32) {
  std::vector<uint8_t> b;
  for (size_t i = 0; i < n; i++) b.push_back((uint8_t)(canonical[i / 8] >> (8 * (i % 8))));
  return b;
}
inline Affine zero_point(Context& ctx) {
  Affine z;
  z.xy.assign(2 * (size_t)ctx.fq_limbs(), 0);
  z.infinity = true;
  return z;
}
inline size_t next_pow2(size_t x) {
  size_t n = 1;
  while (n < x) n <<= 1;
  return n;
}

struct SuccinctVerifierKey {
  Affine h, s;
  size_t supported_degree;
};
struct CommitterKey {  // ipa_pc::CommitterKey{comm_key, h, s, max_degree}; VerifierKey is the same type
  std::shared_ptr<amsm::CommitterKey> comm_key;
  Affine h, s;
  size_t max_degree;
  size_t supported_degree() const { return comm_key->supported_num_elems() - 1; }
  SuccinctVerifierKey svk() const { return SuccinctVerifierKey{h, s, supported_degree()}; }
};
struct Commitment {
  Affine comm;
  std::optional<Affine> shifted_comm;
};
struct Proof {  // scalars in Montgomery form
  std::vector<Affine> l_vec, r_vec;
  Affine final_comm_key;
  Fr c;
  std::optional<Affine> hiding_comm;
  std::optional<Fr> rand;
};

struct SuccinctCheckPolynomial {
  std::vector<Fr> challenges;  // Montgomery form
  FrVector compute_coeffs(Context& ctx) const {
    size_t k = challenges.size();
    FrVector out(ctx, (size_t)1 << k);
    check(amsm_ipa_check_poly_coeffs(ctx.get(), k ? reinterpret_cast<const uint64_t*>(challenges.data()) : nullptr, k, out.ptr()),
          "amsm_ipa_check_poly_coeffs");
    return out;
  }
  Fr evaluate(const FrX& fr, const Fr& point) const {  // prod_i (1 + xi_i point^(2^(k-i)))
    size_t k = challenges.size();
    std::vector<Fr> pw(k);  // pw[j] = point^(2^j)
    Fr cur = point;
    for (size_t j = 0; j < k; j++) {
      pw[j] = cur;
      cur = fr.mul(cur, cur);
    }
    Fr prod = fr.one(), one = fr.one();
    for (size_t i = 1; i <= k; i++) prod = fr.mul(prod, fr.add(one, fr.mul(pw[k - i], challenges[i - 1])));
    return prod;
  }
  std::vector<uint8_t> to_bytes(const FrX& fr) const {
    std::vector<uint8_t> b;
    for (auto& c : challenges) {
      auto x = le_bytes(fr.canon(c));
      b.insert(b.end(), x.begin(), x.end());
    }
    return b;
  }
};

template <class Sponge = Sha256Sponge>
struct InnerProductArgPC {
  // builder of one Fiat-Shamir challenge: the sponge of domain `IpaPCDomain` = "IPA-PC-2020" (src/ipa_pc_as/data_structures.rs:88-94;
  // the reference's IpaPC is `InnerProductArgPC<.., DomainSeparatedSponge<CF, S, IpaPCDomain>>`, src/ipa_pc_as/mod.rs:33-39), absorb
  // the parts, squeeze 128 bits
  struct Challenge {
    const FrX& fr;
    Sponge sp;
    explicit Challenge(const FrX& f) : fr(f), sp(hp_as::fresh_sponge<Sponge>(f.curve).fork("IPA-PC-2020")) {}
    Challenge& point(const Affine& p) {
      sp.absorb_point(p);
      return *this;
    }
    Challenge& scalar(const Fr& mont) {
      sp.absorb_bytes(le_bytes(fr.canon(mont)));
      return *this;
    }
    Challenge& bytes(const std::vector<uint8_t>& b) {
      sp.absorb_bytes(b);
      return *this;
    }
    Fr squeeze_canonical() { return sp.squeeze_bits(CHALLENGE_SIZE); }
  };

  // UniversalParams: next_pow2(max_degree + 1) generators + h + s (synthetic stream; ark-poly-commit hashes to the curve)
  static CommitterKey setup(Context& ctx, size_t max_degree, uint64_t seed = 0x1BA5EED) {
    size_t n = next_pow2(max_degree + 1), w = 2 * (size_t)ctx.fq_limbs();
    amsm_bases* tmp = nullptr;
    amsm::check(amsm_bases_generate(ctx.get(), seed, n + 2, AMSM_BASES_NO_PRECOMPUTE, &tmp), "amsm_bases_generate");
    std::vector<uint64_t> xy((n + 2) * w);
    int st = amsm_bases_read(ctx.get(), tmp, 0, n + 2, xy.data(), nullptr);
    amsm_bases_free(tmp);
    amsm::check(st, "amsm_bases_read");
    CommitterKey k;
    k.comm_key = std::make_shared<amsm::CommitterKey>(
        amsm::CommitterKey::load(ctx, std::vector<uint64_t>(xy.begin(), xy.begin() + (long)(n * w)), nullptr));
    k.h.xy.assign(xy.begin() + (long)(n * w), xy.begin() + (long)((n + 1) * w));
    k.h.infinity = false;
    k.s.xy.assign(xy.begin() + (long)((n + 1) * w), xy.end());
    k.s.infinity = false;
    k.max_degree = n - 1;
    return k;
  }
  static CommitterKey trim(const CommitterKey& pp, size_t supported_degree) {  // ck == vk
    size_t n = next_pow2(supported_degree + 1);
    if (n == pp.comm_key->supported_num_elems()) return pp;
    CommitterKey k = pp;
    k.comm_key = std::make_shared<amsm::CommitterKey>(amsm::CommitterKey::load(pp.comm_key->ctx(), pp.comm_key->read(0, n), nullptr));
    return k;
  }

  // msm(key[..len], scalars) (+ randomizer * hiding_generator)
  static Affine cm_commit(const amsm::CommitterKey& key, const FrVector& scalars, const Affine* hiding_generator = nullptr,
                          const Fr* randomizer = nullptr) {
    Affine out = VariableBaseMSM::multi_scalar_mul(key, scalars);
    if (randomizer) {
      FrX fr(amsm_ctx_curve(key.ctx().get()));
      out = host_lincomb(key.ctx(), {&out, hiding_generator}, {fr.one(), *randomizer});
    }
    return out;
  }
  // -> (Commitment, rand); hiding <=> LabeledPolynomial::hiding_bound().is_some().  rng returns canonical scalars.
  static std::pair<Commitment, Fr> commit(const CommitterKey& ck, const FrVector& polynomial, bool hiding, const hp_as::Rng& rng) {
    FrX fr(amsm_ctx_curve(ck.comm_key->ctx().get()));
    Fr rand = hiding ? fr.to_mont(rng()) : fr.zero();
    Affine c = cm_commit(*ck.comm_key, polynomial, &ck.s, hiding ? &rand : nullptr);
    return {Commitment{c, {}}, rand};
  }

  // Leading rounds that fold the key physically: (smallest log2(d+1) that folds at all, fold while the key has more than
  // 2^T generators), measured on MI355X -- the same table as accumulation_amd/ipa_pc.py:IPA_FOLD.
  static size_t fold_rounds(int curve, size_t log_n) {
    size_t min_log = curve == AMSM_PALLAS ? 18 : 16, t = 15;
    if (const char* e = getenv("AMSM_IPA_FOLD_ABOVE")) {
      int v = atoi(e);
      if (v > 0) return log_n > (size_t)v ? log_n - (size_t)v : 0;
    }
    return log_n >= min_log && log_n > t ? log_n - t : 0;
  }
  static Fr inner_product(Context& ctx, const void* a, const void* b, size_t n) {
    Fr out;
    amsm::check(amsm_vec_inner_product(ctx.get(), a, b, n, out.data()), "amsm_vec_inner_product");
    return out;
  }
  static const void* at(const FrVector& v, size_t off) { return static_cast<const char*>(v.ptr()) + off * 32; }
  // out = a[0..n) * ca + b[0..n) * cb over raw device ranges
  static FrVector combine2(Context& ctx, const void* a, size_t na, const Fr& ca, const void* b, size_t nb, const Fr& cb, size_t n) {
    const void* ptrs[2] = {a, b};
    size_t lens[2] = {na, nb};
    Fr co[2] = {ca, cb};
    FrVector out(ctx, n);
    amsm::check(amsm_vec_combine(ctx.get(), ptrs, lens, 2, reinterpret_cast<const uint64_t*>(co), nullptr, 0, out.ptr(), n), "amsm_vec_combine");
    return out;
  }

  // vector length at which an opening jumps to the host (AMSM_IPA_JUMP=0: never; =M: at M entries, a power of two >= 64)
  static size_t jump_m() {
    const char* e = std::getenv("AMSM_IPA_JUMP");
    return e ? (size_t)std::atol(e) : 64;
  }
  // The last log2(cur) rounds of `open` on the host over the `cur` generators amsm_ipa_jump_fold returned (the key folded by the
  // challenges so far): the reference's own loop (ark_poly_commit::ipa_pc::open ext) -- l = <c_r, key_l> + <c_r, z_l> h',
  // r = <c_l, key_r> + <c_l, z_r> h', then c_l += x^-1 c_r, z_l += x z_r, key_l += x key_r.  The fold of c and z by the last
  // challenge is still pending on the device (amsm_ipa_round_fused applies it lazily): applied here first.
  static void host_rounds(Context& ctx, const FrX& fr, const std::vector<uint64_t>& bxy, const std::vector<uint8_t>& binf, size_t cur,
                          const FrVector& coeffs, const FrVector& z, const Affine& h_prime, Fr rc_canon, std::vector<Fr>& xs, Proof& proof) {
    const size_t w = 2 * (size_t)ctx.fq_limbs();
    std::vector<Affine> B(cur);
    for (size_t k = 0; k < cur; k++) {
      B[k].xy.assign(bxy.begin() + (long)(k * w), bxy.begin() + (long)((k + 1) * w));
      B[k].infinity = binf[k] != 0;
    }
    std::vector<Fr> cv(2 * cur), zv(2 * cur);
    amsm::check(amsm_dev_download(ctx.get(), cv.data(), coeffs.ptr(), 2 * cur * sizeof(Fr)), "amsm_dev_download");
    amsm::check(amsm_dev_download(ctx.get(), zv.data(), z.ptr(), 2 * cur * sizeof(Fr)), "amsm_dev_download");
    Fr x = xs.back(), xinv = fr.inv(x);
    std::vector<Fr> c(cur), zz(cur);
    for (size_t i = 0; i < cur; i++) {
      c[i] = fr.add(cv[i], fr.mul(xinv, cv[cur + i]));
      zz[i] = fr.add(zv[i], fr.mul(x, zv[cur + i]));
    }
    const Fr one = fr.one();
    while (cur > 1) {
      const size_t half = cur / 2;
      Fr ip_l = fr.zero(), ip_r = fr.zero();
      for (size_t i = 0; i < half; i++) {
        ip_l = fr.add(ip_l, fr.mul(c[half + i], zz[i]));  // <c_r, z_l>
        ip_r = fr.add(ip_r, fr.mul(c[i], zz[half + i]));  // <c_l, z_r>
      }
      hp_as::LincombJob jl, jr;
      for (size_t i = 0; i < half; i++) {
        jl.first.push_back(&B[i]);
        jl.second.push_back(c[half + i]);
        jr.first.push_back(&B[half + i]);
        jr.second.push_back(c[i]);
      }
      if (!h_prime.infinity) {
        jl.first.push_back(&h_prime);
        jl.second.push_back(ip_l);
        jr.first.push_back(&h_prime);
        jr.second.push_back(ip_r);
      }
      std::vector<Affine> lr = hp_as::host_lincomb_batch(ctx, {jl, jr});
      proof.l_vec.push_back(lr[0]);
      proof.r_vec.push_back(lr[1]);
      rc_canon = Challenge(fr).bytes(le_bytes(rc_canon, 16)).point(lr[0]).point(lr[1]).squeeze_canonical();
      x = fr.to_mont(rc_canon);
      xinv = fr.inv(x);
      xs.push_back(x);
      std::vector<hp_as::LincombJob> folds(half);
      for (size_t i = 0; i < half; i++) {
        c[i] = fr.add(c[i], fr.mul(xinv, c[half + i]));
        zz[i] = fr.add(zz[i], fr.mul(x, zz[half + i]));
        folds[i] = {{&B[i], &B[half + i]}, {one, x}};
      }
      std::vector<Affine> nb = hp_as::host_lincomb_batch(ctx, folds);
      for (size_t i = 0; i < half; i++) B[i] = nb[i];
      cur = half;
    }
    proof.final_comm_key = B[0];
    proof.c = c[0];
  }

  // open_individual_opening_challenges for ONE polynomial with opening challenge 1
  static Proof open(const CommitterKey& ck, const FrVector& polynomial, const Commitment& commitment, const Fr& point, const Fr& rand,
                    bool hiding, const hp_as::Rng& rng) {
    const amsm::CommitterKey& key = *ck.comm_key;
    Context& ctx = key.ctx();
    FrX fr(amsm_ctx_curve(ctx.get()));
    size_t n = ck.supported_degree() + 1;
    if (polynomial.len() > n) throw Error(AMSM_E_INVALID_ARG, "ipa open: polynomial longer than the key");
    const Fr one = fr.one(), zero = fr.zero();
    FrVector z(ctx, n);
    amsm::check(amsm_vec_powers(ctx.get(), point.data(), n, z.ptr()), "amsm_vec_powers");
    // coefficient vector padded to d + 1
    FrVector coeffs = combine2(ctx, polynomial.ptr(), polynomial.len(), one, polynomial.ptr(), 0, zero, n);
    Affine combined_comm = commitment.comm;
    Fr combined_v = inner_product(ctx, coeffs.ptr(), z.ptr(), n);
    std::optional<Affine> hiding_comm;
    std::optional<Fr> proof_rand;
    if (hiding) {
      // d + 1 random coefficients: drawn in canonical form, uploaded as they are and taken to Montgomery form by ONE device kernel
      // (a Montgomery product with R^2: x R^2 R^-1 = x R) -- one amsm_fr_to_mont call per coefficient was 1.6 ms of a 2^16 prove
      std::vector<Fr> hp_canon(n);
      for (size_t i = 0; i < n; i++) hp_canon[i] = rng();
      FrVector hp_raw(ctx, hp_canon);
      FrVector hp0 = combine2(ctx, hp_raw.ptr(), n, fr.to_mont(one), hp_raw.ptr(), 0, zero, n);
      Fr hv = inner_product(ctx, hp0.ptr(), z.ptr(), n);
      FrVector shift(ctx, std::vector<Fr>{fr.neg(hv)});  // random polynomial that vanishes at `point`
      FrVector hp = combine2(ctx, hp0.ptr(), n, one, shift.ptr(), 1, one, n);
      Fr hiding_rand = fr.to_mont(rng());
      hiding_comm = cm_commit(key, hp, &ck.s, &hiding_rand);
      Fr hch = fr.to_mont(Challenge(fr).point(combined_comm).scalar(point).scalar(combined_v).point(*hiding_comm).squeeze_canonical());
      coeffs = combine2(ctx, coeffs.ptr(), n, one, hp.ptr(), n, hch, n);
      proof_rand = fr.add(rand, fr.mul(hch, hiding_rand));
      combined_comm = host_lincomb(ctx, {&combined_comm, &*hiding_comm, &ck.s}, {one, hch, fr.neg(*proof_rand)});
    }
    Fr rc_canon = Challenge(fr).point(combined_comm).scalar(point).scalar(combined_v).squeeze_canonical();
    Fr round_challenge = fr.to_mont(rc_canon);
    Affine h_prime = host_lincomb(ctx, {&ck.h}, {round_challenge});
    // Small and medium openings never fold the key: round j's cross commitments are expressed over the ORIGINAL key (see
    // the header).  Large ones fold it physically in their first rounds (amsm_bases_fold, the reference's
    // `key_l += key_r * xi`): a fold costs about nine MSM rounds but halves every later one (fold_rounds).  The proof
    // does not depend on the choice.
    size_t log_n = 0;
    while (((size_t)1 << log_n) < n) log_n++;
    // (a key sharded over the devices of a multi-device context is never folded -- a fold pairs generator i with i + n/2, which
    // live on different devices: every round is one grouped MSM over the original sharded key, amsm.h amsm_ctx_create_multi)
    const size_t n_fold = amsm_bases_num_shards(key.get()) > 1 ? 0 : fold_rounds(amsm_ctx_curve(ctx.get()), log_n);
    FrVector u(ctx, n);
    std::vector<Fr> xs;
    Proof proof;
    std::unique_ptr<amsm::CommitterKey> folded;  // the key the current round's scalars are expressed over, once folded
    const amsm::CommitterKey* cur_key = &key;
    size_t log_key = log_n;
    size_t cur = n;
    // One library call per round (amsm_ipa_round_fused): the previous round's fold of c and z (in place), the scalar
    // expansion, the grouped MSM, both inner products, their h' multiples and one normalisation of L and R.
    const size_t w = 2 * (size_t)ctx.fq_limbs();
    bool jumped = false;
    while (cur > 1) {
      size_t half = cur / 2, j = xs.size() - (log_n - log_key);  // challenges since cur_key was formed
      // Round 6 -- the JUMP FOLD: once the vectors are down to jump_m() entries, the key folded by the j challenges so far comes
      // from ONE pass over the key's window table (amsm_ipa_jump_fold) and the remaining rounds -- ~0.36 ms each on the device
      // whatever their logical size -- run on the host over those few generators (host_rounds below); the final folded key falls
      // out of the last fold instead of one more full-size MSM.  A key that does not qualify (AMSM_E_UNSUPPORTED: folded / plain
      // keys, 20-bit tables, sharded keys) keeps its rounds on the device.  The proof does not depend on the choice.
      if (cur == jump_m() && j >= 1 && amsm_bases_num_shards(cur_key->get()) == 1) {
        std::vector<uint64_t> bxy(cur * w);
        std::vector<uint8_t> binf(cur);
        const int rc = amsm_ipa_jump_fold(ctx.get(), cur_key->get(), log_key, reinterpret_cast<const uint64_t*>(xs.data() + (xs.size() - j)), j,
                                          bxy.data(), binf.data());
        if (rc == AMSM_OK) {
          host_rounds(ctx, fr, bxy, binf, cur, coeffs, z, h_prime, rc_canon, xs, proof);
          jumped = true;
          break;
        }
        if (rc != AMSM_E_UNSUPPORTED) amsm::check(rc, "amsm_ipa_jump_fold");
      }
      std::vector<uint64_t> lr_xy(2 * w);
      uint8_t lr_inf[2] = {0, 0};
      Fr ips[2];  // <c_r, z_l>, <c_l, z_r>
      amsm::check(amsm_ipa_round_fused(ctx.get(), cur_key->get(), j ? reinterpret_cast<const uint64_t*>(xs.data() + (xs.size() - j)) : nullptr,
                                       j, log_key, coeffs.ptr(), z.ptr(), xs.empty() ? nullptr : xs.back().data(),
                                       h_prime.infinity ? nullptr : h_prime.xy.data(), u.ptr(), lr_xy.data(), lr_inf,
                                       reinterpret_cast<uint64_t*>(ips)),
                  "amsm_ipa_round_fused");
      Affine lr[2];
      for (int g = 0; g < 2; g++) {
        lr[g].xy.assign(lr_xy.begin() + (long)(g * w), lr_xy.begin() + (long)((g + 1) * w));
        lr[g].infinity = lr_inf[g] != 0;
      }
      proof.l_vec.push_back(lr[0]);
      proof.r_vec.push_back(lr[1]);
      rc_canon = Challenge(fr).bytes(le_bytes(rc_canon, 16)).point(lr[0]).point(lr[1]).squeeze_canonical();
      round_challenge = fr.to_mont(rc_canon);
      xs.push_back(round_challenge);
      if (xs.size() <= n_fold) {  // physical fold: the next round sees a key of `half` generators
        folded.reset(new amsm::CommitterKey(cur_key->fold(half, round_challenge, CHALLENGE_SIZE)));
        cur_key = folded.get();
        log_key--;
      }
      cur = half;
    }
    if (jumped) {
      proof.hiding_comm = hiding_comm;
      proof.rand = proof_rand;
      return proof;
    }
    std::vector<Fr> since(xs.begin() + (long)(log_n - log_key), xs.end());
    if (!since.empty()) {
      FrVector s_vec = SuccinctCheckPolynomial{since}.compute_coeffs(ctx);
      proof.final_comm_key = VariableBaseMSM::multi_scalar_mul(*cur_key, s_vec);
    } else {
      proof.final_comm_key.xy = cur_key->read(0, 1);
      proof.final_comm_key.infinity = false;
    }
    // the last fold happens here: c = c_0 + x^-1 c_1 over the two coefficients the last round left
    {
      std::vector<Fr> head(std::min<size_t>(2, n));
      amsm::check(amsm_dev_download(ctx.get(), head.data(), coeffs.ptr(), head.size() * sizeof(Fr)), "amsm_dev_download");
      proof.c = xs.empty() ? head.at(0) : fr.add(head.at(0), fr.mul(fr.inv(xs.back()), head.at(1)));
    }
    proof.hiding_comm = hiding_comm;
    proof.rand = proof_rand;
    return proof;
  }

  // The succinct check in two halves, so that several checks can put their point algebra into ONE batched call: `prepare` derives the
  // round challenges (sponge only) and lists the two combinations the check compares -- the round commitment
  // C + v h' + sum_j (x_j^-1 L_j + x_j R_j) (the inverses are full-size scalars: 2 log2(d+1) + 2 terms) and c U + v' h' --,
  // `finish` compares their values.  Rejections that need no group arithmetic come back from `prepare` as an empty optional.
  struct PendingCheck {
    SuccinctCheckPolynomial poly;
    Affine combined_comm, h_prime;
    std::vector<const Affine*> round_pts, check_pts;  // (into this object, the proof and the key: all outlive the batch call)
    std::vector<Fr> round_scs, check_scs;
  };
  static std::optional<std::unique_ptr<PendingCheck>> succinct_check_prepare(Context& ctx, const SuccinctVerifierKey& svk,
                                                                             const Commitment& commitment, const Fr& point,
                                                                             const Fr& value, const Proof& proof) {
    FrX fr(amsm_ctx_curve(ctx.get()));
    size_t log_d = 0;
    while (((size_t)1 << (log_d + 1)) <= svk.supported_degree + 1) log_d++;
    if (commitment.shifted_comm) return {};
    if (proof.l_vec.size() != proof.r_vec.size() || proof.l_vec.size() != log_d) return {};
    if (proof.hiding_comm.has_value() != proof.rand.has_value()) return {};
    const Fr one = fr.one();
    std::unique_ptr<PendingCheck> pc(new PendingCheck());
    pc->combined_comm = commitment.comm;
    if (proof.hiding_comm) {
      Fr hch = fr.to_mont(Challenge(fr).point(pc->combined_comm).scalar(point).scalar(value).point(*proof.hiding_comm).squeeze_canonical());
      pc->combined_comm = host_lincomb(ctx, {&commitment.comm, &*proof.hiding_comm, &svk.s}, {one, hch, fr.neg(*proof.rand)});
    }
    Fr rc_canon = Challenge(fr).point(pc->combined_comm).scalar(point).scalar(value).squeeze_canonical();
    Fr round_challenge = fr.to_mont(rc_canon);
    pc->h_prime = host_lincomb(ctx, {&svk.h}, {round_challenge});
    pc->round_pts = {&pc->combined_comm, &pc->h_prime};
    pc->round_scs = {one, value};
    for (size_t k = 0; k < proof.l_vec.size(); k++) {
      rc_canon = Challenge(fr).bytes(le_bytes(rc_canon, 16)).point(proof.l_vec[k]).point(proof.r_vec[k]).squeeze_canonical();
      if (rc_canon == Fr{0, 0, 0, 0}) return {};
      round_challenge = fr.to_mont(rc_canon);
      pc->poly.challenges.push_back(round_challenge);
      pc->round_pts.push_back(&proof.l_vec[k]);
      pc->round_pts.push_back(&proof.r_vec[k]);
      pc->round_scs.push_back(fr.inv(round_challenge));
      pc->round_scs.push_back(round_challenge);
    }
    Fr v_prime = fr.mul(pc->poly.evaluate(fr, point), proof.c);
    pc->check_pts = {&proof.final_comm_key, &pc->h_prime};
    pc->check_scs = {proof.c, v_prime};
    return std::optional<std::unique_ptr<PendingCheck>>(std::move(pc));
  }
  // the combinations of all pending checks in one amsm_host_lincomb_batch (their terms are shared out over the host pool);
  // ok[k] = check k holds
  static std::vector<bool> succinct_check_finish(Context& ctx, const std::vector<const PendingCheck*>& pending) {
    std::vector<hp_as::LincombJob> jobs;
    for (const PendingCheck* pc : pending) {
      jobs.push_back({pc->round_pts, pc->round_scs});
      jobs.push_back({pc->check_pts, pc->check_scs});
    }
    std::vector<Affine> res = jobs.empty() ? std::vector<Affine>() : hp_as::host_lincomb_batch(ctx, jobs);
    std::vector<bool> ok(pending.size());
    for (size_t k = 0; k < pending.size(); k++) ok[k] = res[2 * k] == res[2 * k + 1];
    return ok;
  }
  static std::optional<SuccinctCheckPolynomial> succinct_check(Context& ctx, const SuccinctVerifierKey& svk, const Commitment& commitment,
                                                              const Fr& point, const Fr& value, const Proof& proof) {
    auto pc = succinct_check_prepare(ctx, svk, commitment, point, value, proof);
    if (!pc) return {};
    if (!succinct_check_finish(ctx, {pc->get()})[0]) return {};
    return (*pc)->poly;
  }

  static bool check(const CommitterKey& vk, const Commitment& commitment, const Fr& point, const Fr& value, const Proof& proof) {
    Context& ctx = vk.comm_key->ctx();
    auto cp = succinct_check(ctx, vk.svk(), commitment, point, value, proof);
    if (!cp) return false;
    FrVector coeffs = cp->compute_coeffs(ctx);                 // 2^k field multiplications on the device
    Affine final_key = cm_commit(*vk.comm_key, coeffs);        // THE (d+1)-point MSM of the decider
    return final_key == proof.final_comm_key;
  }
};

}  // namespace ipa_pc

namespace ipa_pc_as {

using hp_as::host_lincomb;
using hp_as::MalformedAccumulator;
using hp_as::MalformedInput;
using hp_as::MissingRng;
using hp_as::Sha256Sponge;
using ipa_pc::Commitment;
using ipa_pc::CommitterKey;
using ipa_pc::FrX;
using ipa_pc::le_bytes;
using ipa_pc::SuccinctCheckPolynomial;

constexpr unsigned LINEAR_COMBINATION_CHALLENGE_SIZE = 128;  // :42
constexpr unsigned CHALLENGE_POINT_SIZE = 184;               // :43

struct InputInstance {  // data_structures.rs:56-68; point / evaluation in Montgomery form
  Commitment ipa_commitment;
  Fr point, evaluation;
  ipa_pc::Proof ipa_proof;
};
struct Randomness {  // :71-86 (the scheme's Proof is Option<Randomness>); Montgomery form
  std::vector<Fr> random_linear_polynomial;  // degree <= 1
  Affine random_linear_polynomial_commitment;
  Fr commitment_randomness;
};
struct VerifierKey {  // :37-49
  ipa_pc::SuccinctVerifierKey ipa_svk;
  CommitterKey ipa_ck_linear;
  ipa_pc::Proof default_proof;
};
struct ProverKey {  // :27-34
  CommitterKey ipa_ck;
  VerifierKey verifier_key;
};
using Accumulator = InputInstance;  // the witness is ()
using Proof = std::optional<Randomness>;

template <class Sponge = Sha256Sponge>
class AtomicASForInnerProductArgPC {
  using Ipa = ipa_pc::InnerProductArgPC<Sponge>;

 public:
  struct Keys {
    ProverKey pk;
    VerifierKey vk;
    CommitterKey dk;
  };
  static Keys index(const CommitterKey& pp, size_t supported_degree_bound) {  // :502-553
    Context& ctx = pp.comm_key->ctx();
    FrX fr(amsm_ctx_curve(ctx.get()));
    CommitterKey ipa_ck = Ipa::trim(pp, supported_degree_bound);
    FrVector zero_poly(ctx, std::vector<Fr>{fr.zero()});
    ipa_pc::Proof default_proof = Ipa::open(ipa_ck, zero_poly, Commitment{ipa_pc::zero_point(ctx), {}}, fr.zero(), fr.zero(), false, {});
    VerifierKey vk{ipa_ck.svk(), Ipa::trim(pp, 1), default_proof};
    return Keys{ProverKey{ipa_ck, vk}, vk, ipa_ck};
  }

  // ---- prove (:555-676) ----------------------------------------------------------------------------------------
  static std::pair<Accumulator, Proof> prove(const ProverKey& pk, std::vector<InputInstance> ins, const std::vector<InputInstance>& olds,
                                             const hp_as::Rng& rng = hp_as::Rng()) {
    const CommitterKey& ipa_ck = pk.ipa_ck;
    Context& ctx = ipa_ck.comm_key->ctx();
    FrX fr(amsm_ctx_curve(ctx.get()));
    for (auto& x : ins)  // check_input_instance_structure :112-128
      if (x.ipa_commitment.shifted_comm) throw MalformedInput("Explicit degree bounds not supported.");
    for (auto& x : olds)
      if (x.ipa_commitment.shifted_comm) throw MalformedAccumulator("Explicit degree bounds not supported.");
    const bool make_zk = (bool)rng;
    if (!make_zk) {
      for (const std::vector<InputInstance>* group : {(const std::vector<InputInstance>*)&ins, &olds})
        for (auto& x : *group)
          if (x.ipa_proof.hiding_comm || x.ipa_proof.rand) throw MissingRng("Accumulating inputs with hiding requires rng.");
      if (ins.empty() && olds.empty())  // default instance :599-609
        ins.push_back(InputInstance{Commitment{ipa_pc::zero_point(ctx), {}}, fr.zero(), fr.zero(), pk.verifier_key.default_proof});
    }
    Proof proof;
    if (make_zk) {  // generate_prover_randomness :165-187
      std::vector<Fr> lin{fr.to_mont(rng()), fr.to_mont(rng())};
      Affine comm = deterministic_commit(pk.verifier_key.ipa_ck_linear, lin);
      proof = Randomness{lin, comm, fr.to_mont(rng())};
    }
    // AMSM_IPA_TRACE=1: wall time of the prover's stages on stderr (where a prove's host time goes)
    static const bool trace = [] { const char* e = getenv("AMSM_IPA_TRACE"); return e && atoi(e) != 0; }();
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
      if (!trace) return;
      auto t = std::chrono::steady_clock::now();
      fprintf(stderr, "[ipa_pc_as prove] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
      t_prev = t;
    };
    std::vector<Check> checks;
    succinct_checks(ctx, pk.verifier_key.ipa_svk, ins, olds, checks);
    mark("succinct checks");
    Sponge as_sponge = hp_as::fresh_sponge<Sponge>(amsm_ctx_curve(ctx.get())).fork("AS-FOR-IPA-PC-2020");
    Combined comb = combine(ctx, fr, pk.verifier_key.ipa_svk, checks, proof, as_sponge);
    mark("combine");
    // combined check polynomial on the device: sum_i alpha_i * h_i (+ random linear polynomial)  :391-404
    std::vector<std::unique_ptr<FrVector>> vecs;
    std::vector<const void*> ptrs;
    std::vector<size_t> lens;
    size_t n_poly = proof ? proof->random_linear_polynomial.size() : 0;
    for (auto& c : checks) {
      vecs.emplace_back(new FrVector(c.poly.compute_coeffs(ctx)));
      ptrs.push_back(vecs.back()->ptr());
      lens.push_back(vecs.back()->len());
      n_poly = std::max(n_poly, vecs.back()->len());
    }
    std::unique_ptr<FrVector> lin;
    if (proof) lin.reset(new FrVector(ctx, proof->random_linear_polynomial));
    std::unique_ptr<FrVector> poly;
    if (!vecs.empty()) {
      poly.reset(new FrVector(ctx, n_poly));
      check(amsm_vec_combine(ctx.get(), ptrs.data(), lens.data(), ptrs.size(), reinterpret_cast<const uint64_t*>(comb.alphas.data()),
                             lin ? lin->ptr() : nullptr, lin ? lin->len() : 0, poly->ptr(), n_poly),
            "amsm_vec_combine");
    } else if (lin) {
      poly = std::move(lin);
    } else {
      poly.reset(new FrVector(ctx, std::vector<Fr>{fr.zero()}));
    }
    mark("check polynomials (device)");
    Fr challenge_canon = new_challenge(fr, as_sponge, comb.combined, comb.alphas_canon, checks,
                                       proof ? &proof->random_linear_polynomial : nullptr);
    Fr challenge = fr.to_mont(challenge_canon);
    mark("new challenge");
    // compute_new_accumulator :424-472: evaluate, then ONE IPA opening of the combined polynomial
    FrVector z(ctx, poly->len());
    check(amsm_vec_powers(ctx.get(), challenge.data(), poly->len(), z.ptr()), "amsm_vec_powers");
    Fr evaluation = Ipa::inner_product(ctx, poly->ptr(), z.ptr(), poly->len());
    mark("evaluation");
    ipa_pc::Proof ipa_proof = Ipa::open(ipa_ck, *poly, comb.randomized, challenge, proof ? proof->commitment_randomness : fr.zero(),
                                        proof.has_value(), rng);
    mark("open");
    return {InputInstance{comb.randomized, challenge, evaluation, ipa_proof}, proof};
  }

  // ---- verify (:678-818; host only) ----------------------------------------------------------------------------
  static bool verify(Context& ctx, const VerifierKey& vk, std::vector<InputInstance> ins, const std::vector<InputInstance>& olds,
                     const InputInstance& new_acc, const Proof& proof) {
    FrX fr(amsm_ctx_curve(ctx.get()));
    for (const std::vector<InputInstance>* group : {(const std::vector<InputInstance>*)&ins, &olds})
      for (auto& x : *group)
        if (x.ipa_commitment.shifted_comm) return false;
    if (!check_proof_structure(proof)) return false;
    const bool make_zk = proof.has_value();
    if (!make_zk && ins.empty() && olds.empty())
      ins.push_back(InputInstance{Commitment{ipa_pc::zero_point(ctx), {}}, fr.zero(), fr.zero(), vk.default_proof});
    std::vector<Check> checks;
    try {
      succinct_checks(ctx, vk.ipa_svk, ins, olds, checks);
    } catch (const hp_as::ASError&) {
      return false;
    }
    if (proof) {
      Affine lc = deterministic_commit(vk.ipa_ck_linear, proof->random_linear_polynomial);
      if (!(lc == proof->random_linear_polynomial_commitment)) return false;
    }
    Sponge as_sponge = hp_as::fresh_sponge<Sponge>(amsm_ctx_curve(ctx.get())).fork("AS-FOR-IPA-PC-2020");
    Combined comb = combine(ctx, fr, vk.ipa_svk, checks, proof, as_sponge);
    if (!(comb.randomized.comm == new_acc.ipa_commitment.comm)) return false;
    Fr challenge_canon = new_challenge(fr, as_sponge, comb.combined, comb.alphas_canon, checks,
                                       proof ? &proof->random_linear_polynomial : nullptr);
    Fr challenge = fr.to_mont(challenge_canon);
    if (challenge != new_acc.point) return false;
    Fr ev = fr.zero();
    if (proof) {
      std::vector<Fr> co = proof->random_linear_polynomial;
      co.resize(2, fr.zero());
      ev = fr.add(co[0], fr.mul(co[1], challenge));
    }
    for (size_t k = 0; k < checks.size(); k++)  // evaluate_combined_succinct_check_polynomials :407-421
      ev = fr.add(ev, fr.mul(checks[k].poly.evaluate(fr, challenge), comb.alphas[k]));
    return ev == new_acc.evaluation;
  }

  // ---- decide (:820-848) ---------------------------------------------------------------------------------------
  static bool decide(const CommitterKey& dk, const Accumulator& acc) {
    return Ipa::check(dk, acc.ipa_commitment, acc.point, acc.evaluation, acc.ipa_proof);
  }

 private:
  struct Check {
    SuccinctCheckPolynomial poly;
    Affine final_comm_key;
  };
  struct Combined {
    Affine combined;
    Commitment randomized;
    std::vector<Fr> alphas, alphas_canon;  // linear-combination challenges (Montgomery / canonical)
  };
  static bool check_proof_structure(const Proof& proof) {  // :130-137
    if (!proof) return true;
    std::vector<Fr> c = proof->random_linear_polynomial;
    while (!c.empty() && c.back() == Fr{0, 0, 0, 0}) c.pop_back();
    return c.size() <= 2;
  }
  static Affine deterministic_commit(const CommitterKey& ck_linear, const std::vector<Fr>& coeffs) {  // :147-162 (an MSM of size 2)
    Context& ctx = ck_linear.comm_key->ctx();
    size_t w = 2 * (size_t)ctx.fq_limbs();
    std::vector<uint64_t> xy = ck_linear.comm_key->read(0, 2);
    Affine g0, g1;
    g0.xy.assign(xy.begin(), xy.begin() + (long)w);
    g1.xy.assign(xy.begin() + (long)w, xy.end());
    g0.infinity = g1.infinity = false;
    std::vector<Fr> sc = coeffs;
    sc.resize(2, Fr{0, 0, 0, 0});
    return host_lincomb(ctx, {&g0, &g1}, sc);
  }
  // :190-221, inputs first, then accumulators.  Every check is host work -- a chain of log d sponge challenges and one 2 log d + 2
  // point combination (0.5-0.8 ms at d + 1 = 2^16) -- and independent of the others: one thread each (the reference runs them in
  // sequence; same results, same order, and the error of the FIRST failing instance)
  static void succinct_checks(Context& ctx, const ipa_pc::SuccinctVerifierKey& svk, const std::vector<InputInstance>& ins,
                              const std::vector<InputInstance>& olds, std::vector<Check>& out) {
    std::vector<const InputInstance*> all;
    for (auto& x : ins) all.push_back(&x);
    for (auto& x : olds) all.push_back(&x);
    // Ipa::succinct_check is host-only (sponge + amsm_host_lincomb: no device call, nothing of the context is written), so the
    // checks may run side by side -- on at most as many threads as the host pool would use (one per instance was unbounded and
    // oversubscribed a node's ranks: ADVICE r4); an exception of a worker surfaces only after every worker has finished
    auto one = [&ctx, &svk](const InputInstance* inst) {
      return Ipa::succinct_check_prepare(ctx, svk, inst->ipa_commitment, inst->point, inst->evaluation, inst->ipa_proof);
    };
    std::vector<std::optional<std::unique_ptr<typename Ipa::PendingCheck>>> pend(all.size());
    const size_t n_workers = std::min<size_t>(all.size(), (size_t)amsm_host_threads() + 1);
    std::atomic<size_t> next{0};
    std::exception_ptr failure;
    std::mutex failure_mu;
    auto loop = [&] {
      for (size_t k; (k = next.fetch_add(1)) < all.size();) {
        try {
          pend[k] = one(all[k]);
        } catch (...) {
          std::lock_guard<std::mutex> lk(failure_mu);
          if (!failure) failure = std::current_exception();
        }
      }
    };
    std::vector<std::future<void>> futs;
    for (size_t t = 1; t < n_workers; t++) futs.push_back(std::async(std::launch::async, loop));
    loop();  // (the caller's thread works too)
    for (auto& f : futs) f.get();
    if (failure) std::rethrow_exception(failure);
    // the point algebra of every check that got this far, in ONE batched call (round 5: three checks side by side used to run two
    // of their 34-term combinations on one thread each, the pool being taken by the third)
    std::vector<const typename Ipa::PendingCheck*> live;
    for (auto& p : pend)
      if (p) live.push_back(p->get());
    std::vector<bool> ok = Ipa::succinct_check_finish(ctx, live);
    size_t at = 0;
    for (size_t k = 0; k < all.size(); k++) {
      const bool good = pend[k].has_value() && ok[at];
      if (pend[k].has_value()) at++;
      if (!good) {
        if (k >= ins.size()) throw MalformedAccumulator("Succinct check failed on accumulator.");
        throw MalformedInput("Succinct check failed on input.");
      }
      out.push_back(Check{(*pend[k])->poly, all[k]->ipa_proof.final_comm_key});
    }
  }
  // combine_succinct_check_polynomials_and_commitments :254-346
  static Combined combine(Context& ctx, const FrX& fr, const ipa_pc::SuccinctVerifierKey& svk, const std::vector<Check>& checks,
                          const Proof& proof, const Sponge& as_sponge) {
    Sponge sp = as_sponge;
    if (proof) {
      std::vector<Fr> co = proof->random_linear_polynomial;
      co.resize(2, fr.zero());
      for (int i = 0; i < 2; i++) sp.absorb_bytes(le_bytes(fr.canon(co[i])));
      sp.absorb_point(proof->random_linear_polynomial_commitment);
    }
    for (auto& c : checks) {
      sp.absorb_bytes(c.poly.to_bytes(fr));
      sp.absorb_point(c.final_comm_key);
    }
    Combined out;
    std::vector<const Affine*> pts;
    std::vector<Fr> scs;
    std::vector<Fr> squeezed = sp.squeeze_field_elements(checks.size(), LINEAR_COMBINATION_CHALLENGE_SIZE);
    size_t k_chal = 0;
    for (auto& c : checks) {
      Fr a = squeezed[k_chal++];
      out.alphas_canon.push_back(a);
      out.alphas.push_back(fr.to_mont(a));
      pts.push_back(&c.final_comm_key);
      scs.push_back(out.alphas.back());
    }
    if (proof) {
      pts.push_back(&proof->random_linear_polynomial_commitment);
      scs.push_back(fr.one());
    }
    out.combined = host_lincomb(ctx, pts, scs);
    Affine randomized = out.combined;
    if (proof) randomized = host_lincomb(ctx, {&out.combined, &svk.s}, {fr.one(), proof->commitment_randomness});
    out.randomized = Commitment{randomized, {}};
    return out;
  }
  // :349-388; returns the canonical challenge point
  static Fr new_challenge(const FrX& fr, const Sponge& as_sponge, const Affine& combined, const std::vector<Fr>& alphas_canon,
                          const std::vector<Check>& checks, const std::vector<Fr>* lin) {
    Sponge sp = as_sponge;
    sp.absorb_point(combined);
    if (!lin) {
      sp.absorb_bytes({0});
    } else {
      std::vector<Fr> co = *lin;
      co.resize(2, fr.zero());
      sp.absorb_bytes({1});  // `Option<Vec<u8>>`: the tag is an item of its own (one sponge element), then the byte string by itself
      std::vector<uint8_t> b;
      for (int i = 0; i < 2; i++) {
        auto x = le_bytes(fr.canon(co[i]));
        b.insert(b.end(), x.begin(), x.end());
      }
      sp.absorb_bytes(b);
    }
    for (size_t k = 0; k < checks.size(); k++) {
      sp.absorb_bytes(le_bytes(alphas_canon[k], (LINEAR_COMBINATION_CHALLENGE_SIZE + 7) / 8));
      sp.absorb_bytes(checks[k].poly.to_bytes(fr));
    }
    return sp.squeeze_bits(CHALLENGE_POINT_SIZE);
  }
};

}  // namespace ipa_pc_as
}  // namespace amsm
