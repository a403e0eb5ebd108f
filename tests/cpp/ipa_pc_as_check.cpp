// The reference's accumulation-scheme test template (src/lib.rs:334-395, six scenarios) for AtomicASForInnerProductArgPC as
// instantiated at src/ipa_pc_as/mod.rs:851-1034 (random degree-11 polynomials, commit, random point, open; zk and no-zk) on
// the C++ driver include/amsm_ipa_pc_as.hpp, plus the IPA's own open / check round trip; prints one deterministic run's
// accumulator for the byte-for-byte comparison with accumulation_amd/ipa_pc_as.py.
#include <cstdio>

#include "amsm_ipa_pc_as.hpp"
#include "amsm_poseidon.hpp"

#include "check_device.hpp"

// -DAMSM_TEST_POSEIDON: the same template runs with the reference's sponge (ark-sponge Poseidon, include/amsm_poseidon.hpp)
// as the Sponge argument instead of the SHA-256 stand-in
#ifdef AMSM_TEST_POSEIDON
using TestSponge = amsm::poseidon::PoseidonSponge;
#else
using TestSponge = amsm::hp_as::Sha256Sponge;
#endif

using namespace amsm;
using namespace amsm::ipa_pc_as;
using AS = AtomicASForInnerProductArgPC<TestSponge>;
using Ipa = ipa_pc::InnerProductArgPC<TestSponge>;

static const size_t DEGREE = 11;

struct SchemeRng {  // tests/test_hp_as_scheme_gpu.py:SchemeRng
  uint64_t seed, i = 0;
  explicit SchemeRng(uint64_t s) : seed(s) {}
  Fr field() {
    Fr x;
    for (uint64_t k = 0; k < 4; k++) {
      uint64_t z = seed * 0xD1342543DE82EF95ull + (4 * i + k) * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      x[k] = z;
    }
    i++;
    x[3] &= (1ull << 62) - 1;
    return x;
  }
};

static std::vector<InputInstance> generate_inputs(Context& ctx, const ProverKey& pk, size_t num, bool make_zk, SchemeRng& rng) {
  ipa_pc::FrX fr(amsm_ctx_curve(ctx.get()));
  hp_as::Rng prng([&rng]() { return rng.field(); });
  std::vector<InputInstance> out;
  for (size_t t = 0; t < num; t++) {
    std::vector<Fr> coeffs;
    for (size_t k = 0; k <= DEGREE; k++) coeffs.push_back(fr.to_mont(rng.field()));
    FrVector poly(ctx, coeffs);
    auto cr = Ipa::commit(pk.ipa_ck, poly, make_zk, prng);
    Fr point = fr.to_mont(rng.field());
    Fr value = fr.zero();
    for (size_t k = coeffs.size(); k-- > 0;) value = fr.add(fr.mul(value, point), coeffs[k]);
    ipa_pc::Proof proof = Ipa::open(pk.ipa_ck, poly, cr.first, point, cr.second, make_zk, prng);
    out.push_back(InputInstance{cr.first, point, value, proof});
  }
  return out;
}

static Accumulator run_template(Context& ctx, const ipa_pc::CommitterKey& pp, const std::vector<size_t>& per_iteration, bool make_zk,
                                size_t iterations) {
  auto keys = AS::index(pp, DEGREE);
  SchemeRng rng(4096);
  size_t total = 0;
  for (size_t k : per_iteration) total += k;
  std::vector<InputInstance> inputs = generate_inputs(ctx, keys.pk, total * iterations, make_zk, rng);
  hp_as::Rng prng = make_zk ? hp_as::Rng([&rng]() { return rng.field(); }) : hp_as::Rng();
  size_t start = 0;
  Accumulator last;
  for (size_t it = 0; it < iterations; it++) {
    std::vector<Accumulator> old;
    for (size_t k : per_iteration) {
      std::vector<InputInstance> step(inputs.begin() + (long)start, inputs.begin() + (long)(start + k));
      start += k;
      auto res = AS::prove(keys.pk, step, old, prng);
      if (!AS::verify(ctx, keys.vk, step, old, res.first, res.second)) throw std::runtime_error("Verify failed");
      old.push_back(res.first);
    }
    if (!AS::decide(keys.dk, old.back())) throw std::runtime_error("Decide failed");
    last = old.back();
  }
  return last;
}

static void print_point(const char* name, const Affine& p) {
  printf("%s %d", name, p.infinity ? 1 : 0);
  for (uint64_t w : p.xy) printf(" %016llx", (unsigned long long)w);
  printf("\n");
}
static void print_fr(const char* name, const Fr& canonical) {
  printf("%s 0", name);
  for (uint64_t w : canonical) printf(" %016llx", (unsigned long long)w);
  printf("\n");
}

int main() {
  try {
    Context ctx = check_context(AMSM_PALLAS);
    ipa_pc::FrX fr(amsm_ctx_curve(ctx.get()));
    ipa_pc::CommitterKey pp = Ipa::setup(ctx, DEGREE, 0xABCDEF);
    // the polynomial commitment alone: open / check round trip and its two rejections (tests/test_ipa_gpu.py)
    for (int zk = 0; zk < 2; zk++) {
      SchemeRng rng(77);
      hp_as::Rng prng([&rng]() { return rng.field(); });
      ipa_pc::CommitterKey ck = Ipa::trim(pp, DEGREE);
      std::vector<Fr> coeffs;
      for (size_t k = 0; k <= DEGREE; k++) coeffs.push_back(fr.to_mont(rng.field()));
      FrVector poly(ctx, coeffs);
      auto cr = Ipa::commit(ck, poly, zk != 0, prng);
      Fr point = fr.to_mont(rng.field()), value = fr.zero();
      for (size_t k = coeffs.size(); k-- > 0;) value = fr.add(fr.mul(value, point), coeffs[k]);
      ipa_pc::Proof proof = Ipa::open(ck, poly, cr.first, point, cr.second, zk != 0, prng);
      bool good = Ipa::check(ck, cr.first, point, value, proof);
      bool bad_v = Ipa::check(ck, cr.first, point, fr.add(value, fr.one()), proof);
      bool bad_p = Ipa::check(ck, cr.first, fr.add(point, fr.one()), value, proof);
      printf("ipa_pc %s %d %d %d\n", zk ? "zk" : "no_zk", good, bad_v, bad_p);
    }
    // the proof does not depend on how many leading rounds fold the key physically (forced at d + 1 = 64)
    {
      ipa_pc::CommitterKey pp64 = Ipa::setup(ctx, 63, 0xF01D);
      bool same = true;
      std::vector<ipa_pc::Proof> proofs;
      for (const char* t : {"99", "5", "2", "1"}) {
        setenv("AMSM_IPA_FOLD_ABOVE", t, 1);
        SchemeRng rng(31);
        hp_as::Rng prng([&rng]() { return rng.field(); });
        std::vector<Fr> coeffs;
        for (size_t k = 0; k < 50; k++) coeffs.push_back(fr.to_mont(rng.field()));
        FrVector poly(ctx, coeffs);
        auto cr = Ipa::commit(pp64, poly, true, prng);
        Fr point = fr.to_mont(rng.field()), value = fr.zero();
        for (size_t k = coeffs.size(); k-- > 0;) value = fr.add(fr.mul(value, point), coeffs[k]);
        proofs.push_back(Ipa::open(pp64, poly, cr.first, point, cr.second, true, prng));
        same = same && Ipa::check(pp64, cr.first, point, value, proofs.back());
      }
      unsetenv("AMSM_IPA_FOLD_ABOVE");
      for (auto& p : proofs)
        same = same && p.l_vec == proofs[0].l_vec && p.r_vec == proofs[0].r_vec && p.final_comm_key == proofs[0].final_comm_key &&
               p.c == proofs[0].c && p.rand == proofs[0].rand;
      printf("fold_invariance %d\n", same ? 1 : 0);
    }
    struct Scenario {
      const char* name;
      std::vector<size_t> per_iteration;
      size_t iterations;
    } scenarios[] = {{"single_input_init", {1}, 2},          {"multiple_inputs_init", {3}, 2},
                     {"simple_accumulation", {1, 1}, 2},     {"multiple_inputs_accumulation", {1, 1, 2, 3}, 2},
                     {"accumulators_only", {1, 0, 0, 0}, 2}, {"no_inputs_init", {0}, 1}};
    for (int zk = 0; zk < 2; zk++)
      for (auto& s : scenarios) {
        run_template(ctx, pp, s.per_iteration, zk != 0, check_iterations(s.iterations));
        printf("scenario %s %s ok %zu\n", s.name, zk ? "zk" : "no_zk", check_iterations(s.iterations));
      }
    // error behaviour (src/ipa_pc_as/mod.rs:587-597): hiding inputs without an rng
    {
      auto keys = AS::index(pp, DEGREE);
      SchemeRng rng(5);
      auto ins = generate_inputs(ctx, keys.pk, 1, true, rng);
      try {
        AS::prove(keys.pk, ins, {});
        printf("missing_rng not_raised\n");
      } catch (const hp_as::MissingRng&) {
        printf("missing_rng raised\n");
      }
      // a tampered input evaluation fails the succinct check
      auto bad = generate_inputs(ctx, keys.pk, 1, false, rng);
      bad[0].evaluation = fr.add(bad[0].evaluation, fr.one());
      try {
        AS::prove(keys.pk, bad, {});
        printf("malformed_input not_raised\n");
      } catch (const hp_as::MalformedInput&) {
        printf("malformed_input raised\n");
      }
    }
    for (int zk = 0; zk < 2; zk++) {
      Accumulator acc = run_template(ctx, pp, {1, 1, 2, 3}, zk != 0, 1);
      const char* t = zk ? "zk" : "nozk";
      char name[64];
      snprintf(name, sizeof name, "%s_comm", t); print_point(name, acc.ipa_commitment.comm);
      snprintf(name, sizeof name, "%s_final_comm_key", t); print_point(name, acc.ipa_proof.final_comm_key);
      snprintf(name, sizeof name, "%s_l_last", t); print_point(name, acc.ipa_proof.l_vec.back());
      snprintf(name, sizeof name, "%s_r_first", t); print_point(name, acc.ipa_proof.r_vec.front());
      snprintf(name, sizeof name, "%s_point", t); print_fr(name, fr.canon(acc.point));
      snprintf(name, sizeof name, "%s_evaluation", t); print_fr(name, fr.canon(acc.evaluation));
      snprintf(name, sizeof name, "%s_c", t); print_fr(name, fr.canon(acc.ipa_proof.c));
    }
    printf("done\n");
    return 0;
  } catch (const std::exception& e) {
    printf("exception %s\n", e.what());
    return 1;
  }
}
