// C++17 restatement of `ASForR1CSNark` (reference: src/r1cs_nark_as/mod.rs -- index :664-711, prove :713-926, verify
// :928-1029, decide :1031-1112; data structures src/r1cs_nark_as/data_structures.rs) above the C ABI of include/amsm.h:
// the SpMVs, the witness linear combinations, the nested Hadamard-product accumulation (amsm_hp_as.hpp) and every Pedersen
// commitment on the GPU; the O(#inputs) point / challenge algebra on the host.  SURVEY.md section 8(a) row a8.
// Same structure, sponge forks and hashes as accumulation_amd/r1cs_nark_as.py; tests compare the two byte for byte.
#pragma once
#include "amsm_r1cs_nark.hpp"

namespace amsm {
namespace r1cs_nark_as {

using hp_as::FrOps;
using hp_as::host_lincomb;
using hp_as::MalformedAccumulator;
using hp_as::MalformedInput;
using hp_as::MissingRng;
using hp_as::Sha256Sponge;
using r1cs_nark::FirstRoundMessage;
using r1cs_nark::FirstRoundMessageRandomness;
using r1cs_nark::IndexProverKey;
using r1cs_nark::SecondRoundMessage;

inline const char* protocol_name() { return "AS-FOR-R1CS-NARK-2020"; }  // :37
inline const char* hp_as_protocol_name() { return "AS-FOR-HP-2020"; }
inline const char* nark_protocol_name() { return "R1CS-NARK-2020"; }
constexpr unsigned CHALLENGE_SIZE = 128;  // :41

inline std::vector<uint8_t> canonical_bytes(const std::vector<Fr>& xs) {  // canonical scalars, 32 bytes LE each
  std::vector<uint8_t> b;
  for (auto& x : xs)
    for (uint64_t w : x)
      for (int i = 0; i < 8; i++) b.push_back((uint8_t)(w >> (8 * i)));
  return b;
}
inline Affine zero_point(Context& ctx) {
  Affine z;
  z.xy.assign(2 * (size_t)ctx.fq_limbs(), 0);
  z.infinity = true;
  return z;
}

struct InputInstance {  // data_structures.rs:78-99; r1cs_input canonical
  std::vector<Fr> r1cs_input;
  FirstRoundMessage first_round_message;
  static InputInstance zero(Context& ctx, size_t input_len, bool make_zk) {
    Affine z = zero_point(ctx);
    FirstRoundMessage m{z, z, z, {}};
    if (make_zk) m.randomness = FirstRoundMessageRandomness{z, z, z, z, z};
    return InputInstance{std::vector<Fr>(input_len, Fr{0, 0, 0, 0}), m};
  }
  template <class S>
  void absorb_into(S& sponge) const {
    sponge.absorb_bytes(canonical_bytes(r1cs_input));
    first_round_message.absorb_into(sponge);
  }
};
using InputWitness = SecondRoundMessage;  // :125
struct AccumulatorInstance {              // :156-171
  std::vector<Fr> r1cs_input;
  Affine comm_a, comm_b, comm_c;
  hp_as::InputInstance hp_instance;
  template <class S>
  void absorb_into(S& sponge) const {
    sponge.absorb_bytes(canonical_bytes(r1cs_input));
    sponge.absorb_point(comm_a);
    sponge.absorb_point(comm_b);
    sponge.absorb_point(comm_c);
    hp_instance.absorb_into(sponge);
  }
};
struct AccumulatorWitnessRandomness {  // Montgomery form
  Fr sigma_a, sigma_b, sigma_c;
};
struct AccumulatorWitness {  // :218-227
  std::shared_ptr<FrVector> r1cs_blinded_witness;
  hp_as::InputWitness hp_witness;
  std::optional<AccumulatorWitnessRandomness> randomness;
};
struct ProofRandomness {  // :250-262; r1cs_r_input canonical
  std::vector<Fr> r1cs_r_input;
  Affine comm_r_a, comm_r_b, comm_r_c;
};
struct Proof {
  hp_as::Proof hp_proof;
  std::optional<ProofRandomness> randomness;
};
struct Accumulator {
  AccumulatorInstance instance;
  AccumulatorWitness witness;
};
struct Input {
  InputInstance instance;
  SecondRoundMessage witness;
};
struct ProverKey {  // :27-34
  const IndexProverKey* nark_pk;
  std::array<uint8_t, 32> as_matrices_hash;
};
struct VerifierKey {  // :37-50
  size_t num_instance_variables, num_constraints;
  std::array<uint8_t, 32> nark_matrices_hash, as_matrices_hash;
};

template <class Sponge = Sha256Sponge>
class ASForR1CSNark {
  using HP = hp_as::ASForHadamardProducts<Sponge>;
  using Nark = r1cs_nark::R1CSNark<Sponge>;

 public:
  struct Keys {
    ProverKey pk;
    VerifierKey vk;
    const IndexProverKey* dk;
  };
  static Keys index(const IndexProverKey& ipk) {  // :664-711
    auto as_hash = r1cs_nark::hash_matrices(protocol_name(), *ipk.a, *ipk.b, *ipk.c);
    const auto& info = ipk.index_info;
    return Keys{ProverKey{&ipk, as_hash}, VerifierKey{info.num_instance_variables, info.num_constraints, info.matrices_hash, as_hash},
                &ipk};
  }
  struct Sponges {  // :112-125
    Sponge nark, as, hp;
  };
  static Sponges sponges(const Sponge& s) {
    return Sponges{s.fork(nark_protocol_name()), s.fork(protocol_name()), s.fork(hp_as_protocol_name())};
  }

  // ---- prove (:713-926) ----------------------------------------------------------------------------------------
  static std::pair<Accumulator, Proof> prove(const ProverKey& pk, std::vector<Input> inputs, const std::vector<Accumulator>& old_accumulators,
                                             const hp_as::Rng& rng = hp_as::Rng(), Sponge sponge = Sponge()) {
    const IndexProverKey& ipk = *pk.nark_pk;
    const CommitterKey& ck = *ipk.ck;
    Context& ctx = ck.ctx();
    FrOps fr{amsm_ctx_curve(ctx.get())};
    hp_as::sponge_for_curve(sponge, amsm_ctx_curve(ctx.get()), 0);
    Sponges sp = sponges(sponge);
    const auto& info = ipk.index_info;
    const size_t in_len = info.num_instance_variables, wit_len = info.num_variables - in_len;
    for (auto& a : old_accumulators) {
      check_acc_instance(a.instance, in_len);
      check_acc_witness(a.witness, wit_len);
    }
    for (auto& i : inputs) check_input(i, in_len, wit_len);
    if (inputs.empty() && old_accumulators.empty())  // default input :761-768
      inputs.push_back(Input{InputInstance::zero(ctx, in_len, false), SecondRoundMessage{hp_as::filled(ctx, fr.zero(), wit_len), {}}});
    const bool make_zk = (bool)rng;
    if (!make_zk) {
      for (auto& i : inputs)
        if (i.witness.randomness) throw MissingRng("Accumulating inputs with hiding requires rng.");
      for (auto& a : old_accumulators)
        if (a.witness.randomness) throw MissingRng("Accumulating accumulators with hiding requires rng.");
    }
    // step 4 (:793-811, generate_prover_randomness :366-420): constant vectors, 3 SpMV + 3 commits
    std::optional<ProofRandomness> proof_randomness;
    std::shared_ptr<FrVector> d_rwit;
    Fr r1{}, r2{}, r3{};
    if (make_zk) {
      Fr r_in_val = rng(), r_wit_val = rng();
      r1 = fr.to_mont(rng());
      r2 = fr.to_mont(rng());
      r3 = fr.to_mont(rng());
      auto d_rin = hp_as::filled(ctx, fr.to_mont(r_in_val), in_len);
      d_rwit = hp_as::filled(ctx, fr.to_mont(r_wit_val), wit_len);
      FrVector za = ipk.a->vec_mul(*d_rin, *d_rwit), zb = ipk.b->vec_mul(*d_rin, *d_rwit), zc = ipk.c->vec_mul(*d_rin, *d_rwit);
      auto cr = PedersenCommitment::commit_batch(ck, {&za, &zb, &zc}, {&r1, &r2, &r3});
      proof_randomness = ProofRandomness{std::vector<Fr>(in_len, r_in_val), cr[0], cr[1], cr[2]};
    }
    std::vector<const InputInstance*> input_instances;
    for (auto& i : inputs) input_instances.push_back(&i.instance);
    std::vector<const AccumulatorInstance*> acc_instances;
    for (auto& a : old_accumulators) acc_instances.push_back(&a.instance);
    // steps 1-2
    Blinded bl = blinded_commitments(ctx, fr, info.matrices_hash, input_instances, sp.nark);
    std::vector<hp_as::Accumulator> hp_inputs, hp_accs;
    for (size_t k = 0; k < inputs.size(); k++) {  // compute_hp_input_witnesses :316-363
      std::vector<Fr> in_m;
      for (auto& x : inputs[k].instance.r1cs_input) in_m.push_back(fr.to_mont(x));
      FrVector d_in(ctx, in_m);
      auto a_vec = std::make_shared<FrVector>(ipk.a->vec_mul(d_in, *inputs[k].witness.blinded_witness));
      auto b_vec = std::make_shared<FrVector>(ipk.b->vec_mul(d_in, *inputs[k].witness.blinded_witness));
      std::optional<hp_as::InputWitnessRandomness> hp_rnd;
      if (auto& rnd = inputs[k].witness.randomness) hp_rnd = hp_as::InputWitnessRandomness{rnd->sigma_a, rnd->sigma_b, rnd->sigma_o};
      hp_inputs.push_back(hp_as::Accumulator{hp_as::InputInstance{bl.a[k], bl.b[k], bl.prod[k]}, hp_as::InputWitness{a_vec, b_vec, hp_rnd}});
    }
    for (auto& a : old_accumulators) hp_accs.push_back(hp_as::Accumulator{a.instance.hp_instance, a.witness.hp_witness});
    // step 3: nested Hadamard-product accumulation over the NARK's committer key
    auto hp_res = HP::prove(ck, hp_inputs, hp_accs, rng, sp.hp);
    // step 5
    size_t num_addends = inputs.size() + old_accumulators.size() + (make_zk ? 1 : 0);
    std::vector<Fr> beta = beta_challenges(fr, num_addends, pk.as_matrices_hash, acc_instances, input_instances, proof_randomness, sp.as);
    // step 6
    Components comp = instance_components(ctx, fr, input_instances, bl, acc_instances, beta, proof_randomness);
    AccumulatorInstance acc_instance{comp.r1cs_input, comp.ca, comp.cb, comp.cc, hp_res.first.instance};
    // step 7 (:546-658): accumulators first, then inputs, then the prover's randomness
    std::vector<const FrVector*> wits;
    std::vector<const Fr*> sa, sb, sc;
    for (auto& a : old_accumulators) {
      wits.push_back(a.witness.r1cs_blinded_witness.get());
      sa.push_back(a.witness.randomness ? &a.witness.randomness->sigma_a : nullptr);
      sb.push_back(a.witness.randomness ? &a.witness.randomness->sigma_b : nullptr);
      sc.push_back(a.witness.randomness ? &a.witness.randomness->sigma_c : nullptr);
    }
    for (auto& i : inputs) {
      wits.push_back(i.witness.blinded_witness.get());
      sa.push_back(i.witness.randomness ? &i.witness.randomness->sigma_a : nullptr);
      sb.push_back(i.witness.randomness ? &i.witness.randomness->sigma_b : nullptr);
      sc.push_back(i.witness.randomness ? &i.witness.randomness->sigma_c : nullptr);
    }
    if (make_zk) {
      wits.push_back(d_rwit.get());
      sa.push_back(&r1);
      sb.push_back(&r2);
      sc.push_back(&r3);
    }
    auto blinded = std::make_shared<FrVector>(
        hp_as::combine_vectors(ctx, wits, std::vector<Fr>(beta.begin(), beta.begin() + (long)wits.size())));
    std::optional<AccumulatorWitnessRandomness> randomness;
    if (make_zk) {
      auto comb = [&](const std::vector<const Fr*>& r) {
        Fr acc = fr.zero();
        for (size_t i = 0; i < r.size(); i++)
          if (r[i]) acc = fr.add(acc, fr.mul(*r[i], beta[i]));
        return acc;
      };
      randomness = AccumulatorWitnessRandomness{comb(sa), comb(sb), comb(sc)};
    }
    return {Accumulator{acc_instance, AccumulatorWitness{blinded, hp_res.first.witness, randomness}}, Proof{hp_res.second, proof_randomness}};
  }

  // ---- verify (:928-1029) --------------------------------------------------------------------------------------
  static bool verify(Context& ctx, const VerifierKey& vk, std::vector<InputInstance> ins, const std::vector<AccumulatorInstance>& olds,
                     const AccumulatorInstance& new_acc, const Proof& proof, Sponge sponge = Sponge()) {
    FrOps fr{amsm_ctx_curve(ctx.get())};
    hp_as::sponge_for_curve(sponge, amsm_ctx_curve(ctx.get()), 0);
    Sponges sp = sponges(sponge);
    const bool make_zk = proof.randomness.has_value();
    const size_t in_len = vk.num_instance_variables;
    try {
      for (auto& i : ins) check_input_instance(i, in_len);
      for (auto& a : olds) check_acc_instance(a, in_len);
    } catch (const hp_as::ASError&) {
      return false;
    }
    if (ins.empty() && olds.empty()) ins.push_back(InputInstance::zero(ctx, in_len, false));
    std::vector<const InputInstance*> input_instances;
    for (auto& i : ins) input_instances.push_back(&i);
    std::vector<const AccumulatorInstance*> acc_instances;
    for (auto& a : olds) acc_instances.push_back(&a);
    Blinded bl = blinded_commitments(ctx, fr, vk.nark_matrices_hash, input_instances, sp.nark);
    std::vector<hp_as::InputInstance> hp_ins, hp_olds;
    for (size_t k = 0; k < ins.size(); k++) hp_ins.push_back(hp_as::InputInstance{bl.a[k], bl.b[k], bl.prod[k]});
    for (auto& a : olds) hp_olds.push_back(a.hp_instance);
    bool hp_ok = HP::verify(ctx, vk.num_constraints, hp_ins, hp_olds, new_acc.hp_instance, proof.hp_proof, sp.hp);
    size_t num_addends = ins.size() + olds.size() + (make_zk ? 1 : 0);
    std::vector<Fr> beta = beta_challenges(fr, num_addends, vk.as_matrices_hash, acc_instances, input_instances, proof.randomness, sp.as);
    Components comp = instance_components(ctx, fr, input_instances, bl, acc_instances, beta, proof.randomness);
    return hp_ok && comp.r1cs_input == new_acc.r1cs_input && comp.ca == new_acc.comm_a && comp.cb == new_acc.comm_b &&
           comp.cc == new_acc.comm_c;
  }

  // ---- decide (:1031-1112) -------------------------------------------------------------------------------------
  static bool decide(const IndexProverKey& dk, const Accumulator& acc, Sponge = Sponge()) {
    const CommitterKey& ck = *dk.ck;
    Context& ctx = ck.ctx();
    FrOps fr{amsm_ctx_curve(ctx.get())};
    const size_t in_len = dk.index_info.num_instance_variables, wit_len = dk.index_info.num_variables - in_len;
    try {
      check_acc_instance(acc.instance, in_len);
      check_acc_witness(acc.witness, wit_len);
    } catch (const hp_as::ASError&) {
      return false;
    }
    std::vector<Fr> in_m;
    for (auto& x : acc.instance.r1cs_input) in_m.push_back(fr.to_mont(x));
    FrVector d_in(ctx, in_m);
    const FrVector& w = *acc.witness.r1cs_blinded_witness;
    FrVector za = dk.a->vec_mul(d_in, w), zb = dk.b->vec_mul(d_in, w), zc = dk.c->vec_mul(d_in, w);
    const AccumulatorWitnessRandomness* rnd = acc.witness.randomness ? &*acc.witness.randomness : nullptr;
    // The three commitments of this check and the three of the nested hp_as decision (src/hp_as/mod.rs:894-925) are six MSMs over
    // the same key: ONE batch (one fill and one drain of the device pipeline instead of two) when the vectors are of one length --
    // the same verdict as `comm_check && HP::decide(..)`.
    const hp_as::InputWitness& hw = acc.witness.hp_witness;
    if (hw.a_vec && hw.b_vec && hw.a_vec->len() == za.len() && hw.b_vec->len() == za.len()) {
      FrVector product = hp_as::compute_hp(*hw.a_vec, *hw.b_vec);
      const auto* hr = hw.randomness ? &*hw.randomness : nullptr;
      auto cd = PedersenCommitment::commit_batch(ck, {&za, &zb, &zc, hw.a_vec.get(), hw.b_vec.get(), &product},
                                                 {rnd ? &rnd->sigma_a : nullptr, rnd ? &rnd->sigma_b : nullptr, rnd ? &rnd->sigma_c : nullptr,
                                                  hr ? &hr->rand_1 : nullptr, hr ? &hr->rand_2 : nullptr, hr ? &hr->rand_3 : nullptr});
      const hp_as::InputInstance& hi = acc.instance.hp_instance;
      return cd[0] == acc.instance.comm_a && cd[1] == acc.instance.comm_b && cd[2] == acc.instance.comm_c && cd[3] == hi.comm_1 &&
             cd[4] == hi.comm_2 && cd[5] == hi.comm_3;
    }
    auto cd = PedersenCommitment::commit_batch(
        ck, {&za, &zb, &zc}, {rnd ? &rnd->sigma_a : nullptr, rnd ? &rnd->sigma_b : nullptr, rnd ? &rnd->sigma_c : nullptr});
    const Affine &ca = cd[0], &cb = cd[1], &cc = cd[2];
    bool comm_check = ca == acc.instance.comm_a && cb == acc.instance.comm_b && cc == acc.instance.comm_c;
    return comm_check && HP::decide(ck, hp_as::Accumulator{acc.instance.hp_instance, acc.witness.hp_witness});
  }

 private:
  // ---- structure checks :128-217 ----------------------------------------------------------------------------------
  static void check_input_instance(const InputInstance& i, size_t in_len) {
    if (i.r1cs_input.size() != in_len) throw MalformedInput("All R1CS input lengths must be equal and supported by the index prover key.");
  }
  static void check_input(const Input& i, size_t in_len, size_t wit_len) {
    check_input_instance(i.instance, in_len);
    if (i.witness.blinded_witness->len() != wit_len)
      throw MalformedInput("All R1CS witness lengths must be equal and supported by the index prover key.");
    if (i.instance.first_round_message.randomness.has_value() != i.witness.randomness.has_value())
      throw MalformedInput("The existence of the first round message randomness and the second round message randomness must be equal.");
  }
  static void check_acc_instance(const AccumulatorInstance& i, size_t in_len) {
    if (i.r1cs_input.size() != in_len)
      throw MalformedAccumulator("All R1CS input lengths must be equal and supported by the index prover key.");
  }
  static void check_acc_witness(const AccumulatorWitness& w, size_t wit_len) {
    if (w.r1cs_blinded_witness->len() != wit_len)
      throw MalformedAccumulator("All R1CS witness lengths must be equal and supported by the index prover key.");
  }

  // ---- shared by prover and verifier ---------------------------------------------------------------------------------
  struct Blinded {
    std::vector<Affine> a, b, c, prod;
  };
  static Blinded blinded_commitments(Context& ctx, const FrOps& fr, const std::array<uint8_t, 32>& nark_hash,
                                     const std::vector<const InputInstance*>& instances, const Sponge& nark_sponge) {  // :220-286
    Blinded out;
    // the four combinations of every input that carries randomness are independent of each other and of the other
    // inputs': one batched library call for all of them
    std::vector<hp_as::LincombJob> jobs;
    std::vector<size_t> job_of(instances.size(), (size_t)-1);
    for (size_t k = 0; k < instances.size(); k++) {
      const FirstRoundMessage& m = instances[k]->first_round_message;
      if (!m.randomness) continue;
      Sponge s = nark_sponge;  // `nark_sponge.clone()`
      Fr g = Nark::compute_challenge(fr, nark_hash, instances[k]->r1cs_input, m, s);
      const FirstRoundMessageRandomness& r = *m.randomness;
      Fr one = fr.one(), g2 = fr.mul(g, g);
      job_of[k] = jobs.size();
      jobs.push_back({{&m.comm_a, &r.comm_r_a}, {one, g}});
      jobs.push_back({{&m.comm_b, &r.comm_r_b}, {one, g}});
      jobs.push_back({{&m.comm_c, &r.comm_r_c}, {one, g}});
      jobs.push_back({{&m.comm_c, &r.comm_1, &r.comm_2}, {one, g, g2}});
    }
    std::vector<Affine> res = hp_as::host_lincomb_batch(ctx, jobs);
    for (size_t k = 0; k < instances.size(); k++) {
      const FirstRoundMessage& m = instances[k]->first_round_message;
      const bool blinded = job_of[k] != (size_t)-1;
      out.a.push_back(blinded ? res[job_of[k]] : m.comm_a);
      out.b.push_back(blinded ? res[job_of[k] + 1] : m.comm_b);
      out.c.push_back(blinded ? res[job_of[k] + 2] : m.comm_c);
      out.prod.push_back(blinded ? res[job_of[k] + 3] : m.comm_c);
    }
    return out;
  }
  // :423-448; Montgomery form
  static std::vector<Fr> beta_challenges(const FrOps& fr, size_t num, const std::array<uint8_t, 32>& as_hash,
                                         const std::vector<const AccumulatorInstance*>& accs, const std::vector<const InputInstance*>& ins,
                                         const std::optional<ProofRandomness>& pr, Sponge& s) {
    s.absorb_bytes(std::vector<uint8_t>(as_hash.begin(), as_hash.end()));
    s.absorb_len(accs.size());
    for (auto* a : accs) a->absorb_into(s);
    s.absorb_len(ins.size());
    for (auto* i : ins) i->absorb_into(s);
    if (!pr) {
      s.absorb_bytes({0});
    } else {
      s.absorb_bytes({1});  // `Option<ProofRandomness>`: the tag is an item of its own (one sponge element) ...
      s.absorb_bytes(canonical_bytes(pr->r1cs_r_input));  // ... then `to_bytes!(r1cs_r_input)` packed by itself (data_structures.rs:342-348)
      s.absorb_point(pr->comm_r_a);
      s.absorb_point(pr->comm_r_b);
      s.absorb_point(pr->comm_r_c);
    }
    std::vector<Fr> beta{fr.one()};
    if (num > 1)
      for (const Fr& c : s.squeeze_field_elements(num - 1, CHALLENGE_SIZE)) beta.push_back(fr.to_mont(c));
    return beta;
  }
  struct Components {
    std::vector<Fr> r1cs_input;  // canonical
    Affine ca, cb, cc;
  };
  // :452-542: accumulators first, then (blinded) inputs, then the prover's randomness
  static Components instance_components(Context& ctx, const FrOps& fr, const std::vector<const InputInstance*>& ins, const Blinded& bl,
                                        const std::vector<const AccumulatorInstance*>& accs, const std::vector<Fr>& beta,
                                        const std::optional<ProofRandomness>& pr) {
    std::vector<const std::vector<Fr>*> r1cs_inputs;
    std::vector<const Affine*> ca, cb, cc;
    for (auto* a : accs) {
      r1cs_inputs.push_back(&a->r1cs_input);
      ca.push_back(&a->comm_a);
      cb.push_back(&a->comm_b);
      cc.push_back(&a->comm_c);
    }
    for (size_t k = 0; k < ins.size(); k++) {
      r1cs_inputs.push_back(&ins[k]->r1cs_input);
      ca.push_back(&bl.a[k]);
      cb.push_back(&bl.b[k]);
      cc.push_back(&bl.c[k]);
    }
    if (pr) {
      r1cs_inputs.push_back(&pr->r1cs_r_input);
      ca.push_back(&pr->comm_r_a);
      cb.push_back(&pr->comm_r_b);
      cc.push_back(&pr->comm_r_c);
    }
    size_t n_in = 0;
    for (auto* v : r1cs_inputs) n_in = std::max(n_in, v->size());
    std::vector<Fr> acc(n_in, fr.zero());  // Montgomery
    for (size_t j = 0; j < r1cs_inputs.size(); j++)
      for (size_t i = 0; i < r1cs_inputs[j]->size(); i++)
        acc[i] = fr.add(acc[i], fr.mul(beta[j], fr.to_mont((*r1cs_inputs[j])[i])));
    Components out;
    out.r1cs_input.resize(n_in);
    if (n_in) check(amsm_fr_from_mont(fr.curve, acc[0].data(), n_in, out.r1cs_input[0].data()), "amsm_fr_from_mont");
    std::vector<Fr> b(beta.begin(), beta.begin() + (long)ca.size());
    std::vector<Affine> comb = hp_as::host_lincomb_batch(ctx, {{ca, b}, {cb, b}, {cc, b}});
    out.ca = comb[0];
    out.cb = comb[1];
    out.cc = comb[2];
    return out;
  }
};

}  // namespace r1cs_nark_as
}  // namespace amsm
