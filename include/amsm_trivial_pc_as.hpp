// C++17 restatement of `ASForTrivialPC` (reference: src/trivial_pc_as/mod.rs -- index :310-330, prove :332-468,
// verify :470-609, decide :611-632) and of the `ark_poly_commit::trivial_pc::TrivialPC` calls it makes (ext), above
// the C ABI of include/amsm.h.  SURVEY.md section 8(a) row a10 / BASELINE config 1: every commitment is a Pedersen
// commitment of a coefficient vector (an MSM of at most d+1 pairs through amsm_pedersen_commit_device); the O(d)
// polynomial work per claim -- the quotient (p(X) - v) / (X - z), evaluations at the challenge point, the linear
// combination of the witness polynomials -- is sequential host arithmetic in the reference and here (amsm_fr_*).
// Same structure and same stand-in sponge as accumulation_amd/trivial_pc_as.py; tests compare the two byte for byte.
#pragma once
#include <memory>

#include "amsm_hp_as.hpp"

namespace amsm {
namespace trivial_pc_as {

using hp_as::FrOps;
using hp_as::MalformedAccumulator;
using hp_as::MalformedInput;
using hp_as::Sha256Sponge;

constexpr unsigned LINEAR_COMBINATION_CHALLENGE_SIZE = 126;  // :31
constexpr unsigned CHALLENGE_POINT_SIZE = 184;               // :32

struct LabeledPolynomial {  // ark_poly_commit::LabeledPolynomial over DensePolynomial (ext); Montgomery coefficients
  std::vector<Fr> coeffs;   // little-endian, trailing zeros allowed
  std::optional<size_t> degree_bound, hiding_bound;
  size_t degree() const {
    size_t d = coeffs.empty() ? 0 : coeffs.size() - 1;
    while (d > 0 && coeffs[d] == Fr{0, 0, 0, 0}) d--;
    return d;
  }
  Fr evaluate(const FrOps& fr, const Fr& x) const {  // Horner
    Fr acc = fr.zero();
    for (size_t i = coeffs.size(); i-- > 0;) acc = fr.add(fr.mul(acc, x), coeffs[i]);
    return acc;
  }
};
struct LabeledCommitment {  // LabeledCommitment<trivial_pc::Commitment<G>> (ext)
  Affine elem;
  std::optional<size_t> degree_bound;
};
struct InputInstance {  // data_structures.rs:11-21; point / eval in Montgomery form
  LabeledCommitment commitment;
  Fr point, eval;
  static InputInstance zero(Context& ctx) {  // :24-35
    Affine z;
    z.xy.assign(2 * (size_t)ctx.fq_limbs(), 0);
    z.infinity = true;
    return InputInstance{LabeledCommitment{z, {}}, Fr{0, 0, 0, 0}, Fr{0, 0, 0, 0}};
  }
};
struct SingleProof {
  LabeledCommitment witness_commitment;
  Fr witness_eval, eval;
};
using Proof = std::vector<SingleProof>;
struct Input {  // also the shape of an Accumulator
  InputInstance instance;
  LabeledPolynomial witness;
};
using Accumulator = Input;

struct TrivialPC {  // setup / trim / commit / check (ext)
  static CommitterKey setup(Context& ctx, size_t max_degree, uint64_t seed = 0x7121A1) {
    return PedersenCommitment::setup(ctx, max_degree + 1, seed);
  }
  static CommitterKey trim(const CommitterKey& pp, size_t supported_degree) {
    std::vector<uint64_t> xy = pp.read(0, supported_degree + 1);
    CommitterKey ck = CommitterKey::load(pp.ctx(), xy, nullptr);
    ck.hiding_generator = pp.hiding_generator;
    return ck;
  }
  static size_t supported_degree(const CommitterKey& ck) { return ck.supported_num_elems() - 1; }
  static LabeledCommitment commit(const CommitterKey& ck, const LabeledPolynomial& poly) {
    size_t n = std::min(poly.coeffs.size(), ck.supported_num_elems());
    if (n == 0) return InputInstance::zero(ck.ctx()).commitment;
    FrVector v(ck.ctx(), std::vector<Fr>(poly.coeffs.begin(), poly.coeffs.begin() + (long)n));
    return LabeledCommitment{PedersenCommitment::commit(ck, v), {}};
  }
  // several commitments under one key in ONE library call (amsm_msm_multi_device: small keys sum them in a single launch) --
  // the same points as polys.size() calls of commit()
  static std::vector<LabeledCommitment> commit_many(const CommitterKey& ck, const std::vector<const LabeledPolynomial*>& polys) {
    std::vector<LabeledCommitment> out(polys.size(), InputInstance::zero(ck.ctx()).commitment);
    std::vector<std::unique_ptr<FrVector>> vecs;
    std::vector<std::pair<size_t, const FrVector*>> jobs;
    std::vector<size_t> which;
    for (size_t k = 0; k < polys.size(); k++) {
      size_t n = std::min(polys[k]->coeffs.size(), ck.supported_num_elems());
      if (n == 0) continue;
      vecs.push_back(std::make_unique<FrVector>(ck.ctx(), std::vector<Fr>(polys[k]->coeffs.begin(), polys[k]->coeffs.begin() + (long)n)));
      jobs.push_back({0, vecs.back().get()});
      which.push_back(k);
    }
    if (jobs.empty()) return out;
    std::vector<Affine> pts = MsmBatch::windows(ck, jobs);
    for (size_t j = 0; j < which.size(); j++) out[which[j]] = LabeledCommitment{pts[j], {}};
    return out;
  }
  // check_individual_opening_challenges with one commitment and opening challenge 1: the proof IS the polynomial
  static bool check(const CommitterKey& vk, const LabeledCommitment& c, const Fr& point, const Fr& value,
                    const LabeledPolynomial& polynomial) {
    FrOps fr{amsm_ctx_curve(vk.ctx().get())};
    if (polynomial.degree() > supported_degree(vk)) return false;
    return commit(vk, polynomial).elem == c.elem && polynomial.evaluate(fr, point) == value;
  }
};

// (p(X) - v) / (X - z) by synthetic division; the remainder p(z) - v is dropped, as `Div` does
inline std::vector<Fr> poly_div_linear(const FrOps& fr, const std::vector<Fr>& coeffs, const Fr& z) {
  size_t n = coeffs.size();
  if (n <= 1) return {fr.zero()};
  std::vector<Fr> q(n - 1);
  Fr carry = fr.zero();
  for (size_t i = n - 1; i >= 1; i--) {
    carry = fr.add(coeffs[i], fr.mul(carry, z));
    q[i - 1] = carry;
  }
  return q;
}

template <class Sponge = Sha256Sponge>
class ASForTrivialPC {
 public:
  struct Keys {
    CommitterKey prover_key;
    size_t verifier_key;
  };
  static Keys index(const CommitterKey& pp, size_t predicate_index) {  // :310-330 (decider key == prover key)
    return Keys{TrivialPC::trim(pp, predicate_index), predicate_index};
  }

  // ---- prove (:332-468) ----------------------------------------------------------------------------------------
  static std::pair<Accumulator, Proof> prove(const CommitterKey& pk, std::vector<Input> inputs, const std::vector<Accumulator>& accs,
                                             Sponge sponge = Sponge()) {
    Context& ctx = pk.ctx();
    FrOps fr{amsm_ctx_curve(ctx.get())};
    hp_as::sponge_for_curve(sponge, amsm_ctx_curve(ctx.get()), 0);
    if (inputs.empty() && accs.empty())  // default input :349-364
      inputs.push_back(Input{InputInstance::zero(ctx), LabeledPolynomial{{fr.zero()}, {}, {}}});
    std::vector<const InputInstance*> instances;
    std::vector<const LabeledPolynomial*> witnesses;
    for (auto& i : inputs) {
      instances.push_back(&check_instance(i.instance, false));
      witnesses.push_back(&check_witness(i.witness, pk, false));
    }
    for (auto& a : accs) {
      instances.push_back(&check_instance(a.instance, true));
      witnesses.push_back(&check_witness(a.witness, pk, true));
    }
    // steps 1c-1d: witness polynomials w = (p - v) / (X - z) and their commitments (:181-222)
    std::vector<LabeledPolynomial> wit_polys;
    for (size_t k = 0; k < instances.size(); k++)
      wit_polys.push_back(LabeledPolynomial{poly_div_linear(fr, witnesses[k]->coeffs, instances[k]->point), {}, {}});
    std::vector<const LabeledPolynomial*> wit_ptrs;
    for (auto& w : wit_polys) wit_ptrs.push_back(&w);
    std::vector<LabeledCommitment> wit_comms = TrivialPC::commit_many(pk, wit_ptrs);  // (the reference commits one by one, :198)
    Fr z = challenge_point(fr, sponge, TrivialPC::supported_degree(pk), instances, wit_comms);  // step 2
    Proof proof;  // steps 3-4
    for (size_t k = 0; k < instances.size(); k++)
      proof.push_back(SingleProof{wit_comms[k], wit_polys[k].evaluate(fr, z), witnesses[k]->evaluate(fr, z)});
    std::vector<Fr> ch = lc_challenges(fr, sponge, z, proof);
    // steps 5-7: combined polynomial / evaluation / commitment
    std::vector<const LabeledPolynomial*> polys(witnesses);
    for (auto& w : wit_polys) polys.push_back(&w);
    size_t width = 0;
    for (auto* p : polys) width = std::max(width, p->coeffs.size());
    LabeledPolynomial combined{std::vector<Fr>(width, fr.zero()), {}, {}};
    for (size_t k = 0; k < polys.size(); k++)
      for (size_t i = 0; i < polys[k]->coeffs.size(); i++)
        combined.coeffs[i] = fr.add(combined.coeffs[i], fr.mul(ch[k], polys[k]->coeffs[i]));
    Fr combined_eval = combined.evaluate(fr, z);
    std::vector<const Affine*> comms;
    for (auto* i : instances) comms.push_back(&i->commitment.elem);
    for (auto& w : wit_comms) comms.push_back(&w.elem);
    Affine cc = lincomb(ctx, comms, ch);
    return {Accumulator{InputInstance{LabeledCommitment{cc, {}}, z, combined_eval}, combined}, proof};
  }

  // ---- verify (:470-609) ---------------------------------------------------------------------------------------
  static bool verify(Context& ctx, size_t vk, const std::vector<InputInstance>& input_instances,
                     const std::vector<InputInstance>& old_accumulator_instances, const InputInstance& new_acc, const Proof& proof,
                     Sponge sponge = Sponge()) {
    FrOps fr{amsm_ctx_curve(ctx.get())};
    hp_as::sponge_for_curve(sponge, amsm_ctx_curve(ctx.get()), 0);
    std::vector<const InputInstance*> instances;
    InputInstance dflt = InputInstance::zero(ctx);
    try {
      for (auto& i : input_instances) instances.push_back(&check_instance(i, false));
      for (auto& a : old_accumulator_instances) instances.push_back(&check_instance(a, true));
      if (instances.empty()) instances.push_back(&dflt);
      check_instance(new_acc, true);
    } catch (const hp_as::ASError&) {
      return false;
    }
    if (proof.size() != instances.size()) return false;
    // step 4: eval - v == w(z') (z' - z), written without a subtraction
    for (size_t k = 0; k < instances.size(); k++) {
      Fr lhs = fr.add(proof[k].eval, fr.mul(proof[k].witness_eval, instances[k]->point));
      Fr rhs = fr.add(instances[k]->eval, fr.mul(proof[k].witness_eval, new_acc.point));
      if (lhs != rhs) return false;
    }
    std::vector<LabeledCommitment> wc;  // step 3: the challenge point
    for (auto& p : proof) wc.push_back(p.witness_commitment);
    Fr z = challenge_point(fr, sponge, vk, instances, wc);
    if (z != new_acc.point) return false;
    std::vector<Fr> ch = lc_challenges(fr, sponge, z, proof);  // steps 5-7
    Fr ev = fr.zero();
    for (size_t k = 0; k < proof.size(); k++) ev = fr.add(ev, fr.mul(ch[k], proof[k].eval));
    for (size_t k = 0; k < proof.size(); k++) ev = fr.add(ev, fr.mul(ch[proof.size() + k], proof[k].witness_eval));
    if (ev != new_acc.eval) return false;
    std::vector<const Affine*> comms;
    for (auto* i : instances) comms.push_back(&i->commitment.elem);
    for (auto& p : proof) comms.push_back(&p.witness_commitment.elem);
    return lincomb(ctx, comms, ch) == new_acc.commitment.elem;
  }

  // ---- decide (:611-632) ---------------------------------------------------------------------------------------
  static bool decide(const CommitterKey& dk, const Accumulator& acc, Sponge = Sponge()) {
    return TrivialPC::check(dk, acc.instance.commitment, acc.instance.point, acc.instance.eval, acc.witness);
  }

 private:
  static const InputInstance& check_instance(const InputInstance& i, bool is_acc) {  // :101-120
    if (i.commitment.degree_bound) {
      if (is_acc) throw MalformedAccumulator("Degree bounds on accumulator instances are unsupported.");
      throw MalformedInput("Degree bounds on input instances are unsupported.");
    }
    return i;
  }
  static const LabeledPolynomial& check_witness(const LabeledPolynomial& w, const CommitterKey& pk, bool is_acc) {  // :122-172
    auto fail = [&](const char* m) {
      if (is_acc) throw MalformedAccumulator(m);
      throw MalformedInput(m);
    };
    if (w.degree_bound) fail("Degree bounds on witnesses are unsupported.");
    if (w.hiding_bound) fail("Hiding bounds on witnesses are unsupported.");
    if (w.degree() > TrivialPC::supported_degree(pk)) fail("A witness of this degree is unsupported for this prover key");
    return w;
  }
  static std::vector<uint8_t> canonical_bytes(const FrOps& fr, const Fr& mont, size_t n_bytes = 32) {
    Fr c;
    check(amsm_fr_from_mont(fr.curve, mont.data(), 1, c.data()), "amsm_fr_from_mont");
    std::vector<uint8_t> b;
    for (size_t i = 0; i < n_bytes; i++) b.push_back((uint8_t)(c[i / 8] >> (8 * (i % 8))));
    return b;
  }
  static void absorb_instance(Sponge& sp, const FrOps& fr, const InputInstance& i) {  // data_structures.rs:38-55
    sp.absorb_point(i.commitment.elem);
    sp.absorb_bytes(canonical_bytes(fr, i.point));
    sp.absorb_bytes(canonical_bytes(fr, i.eval));
  }
  static Fr challenge_point(const FrOps& fr, const Sponge& sponge, size_t supported_degree,
                            const std::vector<const InputInstance*>& instances, const std::vector<LabeledCommitment>& wit_comms) {
    Sponge sp = sponge;  // `sponge.clone()`
    sp.absorb_u64(supported_degree);
    for (size_t k = 0; k < instances.size(); k++) {
      absorb_instance(sp, fr, *instances[k]);
      sp.absorb_point(wit_comms[k].elem);
    }
    return fr.to_mont(sp.squeeze_bits(CHALLENGE_POINT_SIZE));
  }
  static std::vector<Fr> lc_challenges(const FrOps& fr, Sponge& sp, const Fr& z, const Proof& proof) {
    sp.absorb_bytes(canonical_bytes(fr, z, (CHALLENGE_POINT_SIZE + 7) / 8));
    for (auto& p : proof) {
      sp.absorb_bytes(canonical_bytes(fr, p.eval));
      sp.absorb_bytes(canonical_bytes(fr, p.witness_eval));
    }
    std::vector<Fr> out;
    for (const Fr& c : sp.squeeze_field_elements(2 * proof.size(), LINEAR_COMBINATION_CHALLENGE_SIZE)) out.push_back(fr.to_mont(c));
    return out;
  }
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
};

}  // namespace trivial_pc_as
}  // namespace amsm
