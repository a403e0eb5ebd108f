// The reference's accumulation-scheme test template (src/lib.rs:334-395, six scenarios) for ASForTrivialPC like
// src/trivial_pc_as/mod.rs:634-815 (degree 11, no zk) on the C++ driver include/amsm_trivial_pc_as.hpp; prints one
// deterministic run's accumulator for the byte-for-byte comparison with accumulation_amd/trivial_pc_as.py.
#include <cstdio>

#include "amsm_trivial_pc_as.hpp"
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
using namespace amsm::trivial_pc_as;
using AS = ASForTrivialPC<TestSponge>;

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

static std::vector<Input> generate_inputs(Context& ctx, const CommitterKey& ck, size_t num, SchemeRng& rng) {
  hp_as::FrOps fr{amsm_ctx_curve(ctx.get())};
  std::vector<Input> out;
  for (size_t t = 0; t < num; t++) {
    LabeledPolynomial poly;
    for (size_t i = 0; i <= TrivialPC::supported_degree(ck); i++) poly.coeffs.push_back(fr.to_mont(rng.field()));
    LabeledCommitment comm = TrivialPC::commit(ck, poly);
    Fr point = fr.to_mont(rng.field());
    out.push_back(Input{InputInstance{comm, point, poly.evaluate(fr, point)}, poly});
  }
  return out;
}

static Accumulator run_template(Context& ctx, const CommitterKey& pp, const std::vector<size_t>& per_iteration, size_t iterations) {
  CommitterKey ck = TrivialPC::trim(pp, DEGREE);
  auto keys = AS::index(pp, DEGREE);
  size_t total = 0;
  for (size_t k : per_iteration) total += k;
  SchemeRng rng(777);
  std::vector<Input> inputs = generate_inputs(ctx, ck, total * iterations, rng);
  size_t start = 0;
  Accumulator last;
  for (size_t it = 0; it < iterations; it++) {
    std::vector<Accumulator> old;
    for (size_t k : per_iteration) {
      std::vector<Input> step(inputs.begin() + (long)start, inputs.begin() + (long)(start + k));
      start += k;
      auto res = AS::prove(keys.prover_key, step, old);
      std::vector<InputInstance> ii, oi;
      for (auto& x : step) ii.push_back(x.instance);
      for (auto& x : old) oi.push_back(x.instance);
      if (!AS::verify(ctx, keys.verifier_key, ii, oi, res.first.instance, res.second)) throw std::runtime_error("Verify failed");
      old.push_back(res.first);
    }
    if (!AS::decide(keys.prover_key, old.back())) throw std::runtime_error("Decide failed");
    last = old.back();
  }
  return last;
}

static void print_words(const char* name, const uint64_t* w, size_t n, int flag) {
  printf("%s %d", name, flag);
  for (size_t i = 0; i < n; i++) printf(" %016llx", (unsigned long long)w[i]);
  printf("\n");
}

int main() {
  try {
    Context ctx = check_context(AMSM_PALLAS);
    CommitterKey pp = TrivialPC::setup(ctx, DEGREE, 0x7121A1);
    struct Scenario {
      const char* name;
      std::vector<size_t> per_iteration;
      size_t iterations;
    } scenarios[] = {{"single_input_init", {1}, 3},          {"multiple_inputs_init", {3}, 3},
                     {"simple_accumulation", {1, 1}, 3},     {"multiple_inputs_accumulation", {1, 1, 2, 3}, 2},
                     {"accumulators_only", {1, 0, 0, 0}, 3}, {"no_inputs_init", {0}, 1}};
    for (auto& s : scenarios) {
      run_template(ctx, pp, s.per_iteration, check_iterations(s.iterations));
      printf("scenario %s ok %zu\n", s.name, check_iterations(s.iterations));
    }
    Accumulator acc = run_template(ctx, pp, {1, 1, 2, 3}, 1);
    print_words("acc_comm", acc.instance.commitment.elem.xy.data(), acc.instance.commitment.elem.xy.size(),
                acc.instance.commitment.elem.infinity ? 1 : 0);
    print_words("acc_point", acc.instance.point.data(), 4, 0);
    print_words("acc_eval", acc.instance.eval.data(), 4, 0);
    // a tampered accumulator must be rejected
    acc.witness.coeffs[0][0] ^= 1;
    printf("tampered_decide %d\n", AS::decide(TrivialPC::trim(pp, DEGREE), acc) ? 1 : 0);
    printf("done\n");
    return 0;
  } catch (const std::exception& e) {
    printf("exception %s\n", e.what());
    return 1;
  }
}
