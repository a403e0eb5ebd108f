// Runs the reference's accumulation-scheme test template (src/lib.rs:334-395, six scenarios :398-459, instantiated like
// src/hp_as/mod.rs:957-1151: vector_len 11, zk and no-zk) on the C++ driver include/amsm_hp_as.hpp, and prints one
// deterministic run's accumulator so tests/test_cpp_hp_as.py can compare it byte for byte with the Python mirror.
#include <cstdio>

#include "amsm_hp_as.hpp"
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
using namespace amsm::hp_as;
using AS = ASForHadamardProducts<TestSponge>;

static const size_t VECTOR_LEN = 11;

// the same stream as tests/test_hp_as_scheme_gpu.py:SchemeRng
struct SchemeRng {
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
    x[3] &= (1ull << 62) - 1;  // 254 bits
    return x;
  }
};

static std::vector<Accumulator> generate_inputs(Context& ctx, const CommitterKey& ck, size_t num, bool make_zk) {
  FrOps fr{amsm_ctx_curve(ctx.get())};
  SchemeRng rng(0xC0FFEE);
  std::vector<Accumulator> out;
  for (size_t t = 0; t < num; t++) {
    auto a = filled(ctx, fr.to_mont(rng.field()), VECTOR_LEN);
    auto b = filled(ctx, fr.to_mont(rng.field()), VECTOR_LEN);
    FrVector prod = compute_hp(*a, *b);
    std::optional<InputWitnessRandomness> rnd;
    if (make_zk) {
      Fr r1 = fr.to_mont(rng.field());
      Fr r2 = fr.to_mont(rng.field());
      Fr r3 = fr.to_mont(rng.field());
      rnd = InputWitnessRandomness{r1, r2, r3};
    }
    Affine c1 = PedersenCommitment::commit(ck, *a, rnd ? &rnd->rand_1 : nullptr);
    Affine c2 = PedersenCommitment::commit(ck, *b, rnd ? &rnd->rand_2 : nullptr);
    Affine c3 = PedersenCommitment::commit(ck, prod, rnd ? &rnd->rand_3 : nullptr);
    out.push_back(Accumulator{InputInstance{c1, c2, c3}, InputWitness{a, b, rnd}});
  }
  return out;
}

static void print_point(const char* name, const Affine& p) {
  printf("%s %d", name, p.infinity ? 1 : 0);
  for (uint64_t w : p.xy) printf(" %016llx", (unsigned long long)w);
  printf("\n");
}

// src/lib.rs:334-395; returns the last accumulator
static Accumulator run_template(Context& ctx, const CommitterKey& ck, const std::vector<size_t>& per_iteration, bool make_zk,
                                size_t iterations) {
  auto keys = AS::index(ck);
  size_t total = 0;
  for (size_t k : per_iteration) total += k;
  total *= iterations;
  std::vector<Accumulator> inputs = generate_inputs(ctx, ck, total, make_zk);
  SchemeRng prng(7);
  Rng rng = make_zk ? Rng([&prng]() { return prng.field(); }) : Rng();
  size_t start = 0;
  Accumulator last;
  for (size_t it = 0; it < iterations; it++) {
    std::vector<Accumulator> old;
    for (size_t k : per_iteration) {
      std::vector<Accumulator> step(inputs.begin() + (long)start, inputs.begin() + (long)(start + k));
      start += k;
      auto res = AS::prove(*keys.prover_key, step, old, rng);
      std::vector<InputInstance> ii, oi;
      for (auto& x : step) ii.push_back(x.instance);
      for (auto& x : old) oi.push_back(x.instance);
      if (!AS::verify(ctx, keys.verifier_key, ii, oi, res.first.instance, res.second)) throw std::runtime_error("Verify failed");
      old.push_back(res.first);
    }
    if (!AS::decide(*keys.decider_key, old.back())) throw std::runtime_error("Decide failed");
    last = old.back();
  }
  return last;
}

int main() {
  try {
    Context ctx = check_context(AMSM_PALLAS);
    CommitterKey ck = PedersenCommitment::setup(ctx, VECTOR_LEN, 4242);
    struct Scenario {
      const char* name;
      std::vector<size_t> per_iteration;
      size_t iterations;
    } scenarios[] = {{"single_input_init", {1}, 3},          {"multiple_inputs_init", {3}, 3},
                     {"simple_accumulation", {1, 1}, 3},     {"multiple_inputs_accumulation", {1, 1, 2, 3}, 2},
                     {"accumulators_only", {1, 0, 0, 0}, 3}, {"no_inputs_init", {0}, 1}};
    for (int zk = 0; zk < 2; zk++)
      for (auto& s : scenarios) {
        run_template(ctx, ck, s.per_iteration, zk != 0, check_iterations(s.iterations));
        printf("scenario %s %s ok %zu\n", s.name, zk ? "zk" : "no_zk", check_iterations(s.iterations));
      }
    // deterministic run for the cross-check with the Python mirror
    for (int zk = 0; zk < 2; zk++) {
      Accumulator acc = run_template(ctx, ck, {1, 1, 2, 3}, zk != 0, 1);
      print_point(zk ? "zk_comm_1" : "nozk_comm_1", acc.instance.comm_1);
      print_point(zk ? "zk_comm_2" : "nozk_comm_2", acc.instance.comm_2);
      print_point(zk ? "zk_comm_3" : "nozk_comm_3", acc.instance.comm_3);
    }
    // error behaviour (src/hp_as/mod.rs:664-673): hiding inputs without an rng
    {
      auto in = generate_inputs(ctx, ck, 1, true);
      try {
        AS::prove(ck, in, {});
        printf("missing_rng not_raised\n");
      } catch (const MissingRng&) {
        printf("missing_rng raised\n");
      }
    }
    printf("done\n");
    return 0;
  } catch (const std::exception& e) {
    printf("exception %s\n", e.what());
    return 1;
  }
}
