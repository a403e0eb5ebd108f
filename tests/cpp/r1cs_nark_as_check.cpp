// The reference's accumulation-scheme test template (src/lib.rs:334-395, six scenarios) for ASForR1CSNark as instantiated
// at src/r1cs_nark_as/mod.rs:1190-1395 (DummyCircuit, 5 inputs, 10 constraints, zk and no-zk; every input a fresh NARK
// proof) on the C++ driver include/amsm_r1cs_nark_as.hpp; prints one deterministic run's accumulator for the byte-for-byte
// comparison with accumulation_amd/r1cs_nark_as.py.
#include <cstdio>

#include "amsm_r1cs_nark_as.hpp"
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
using namespace amsm::r1cs_nark_as;
using AS = ASForR1CSNark<TestSponge>;
using Nark = r1cs_nark::R1CSNark<TestSponge>;

static const size_t NUM_INPUTS = 5, NUM_CONSTRAINTS = 10;

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

static std::vector<Input> generate_inputs(Context& ctx, const r1cs_nark::IndexProverKey& ipk, size_t num, bool make_zk, SchemeRng& rng) {
  hp_as::FrOps fr{amsm_ctx_curve(ctx.get())};
  const Fr one = {1, 0, 0, 0};
  hp_as::Rng prng = make_zk ? hp_as::Rng([&rng]() { return rng.field(); }) : hp_as::Rng();
  std::vector<Input> out;
  for (size_t t = 0; t < num; t++) {
    Fr a = rng.field(), b = rng.field();
    Fr am = fr.to_mont(a), bm = fr.to_mont(b), abm = fr.mul(am, bm), ab;
    check(amsm_fr_from_mont(amsm_ctx_curve(ctx.get()), abm.data(), 1, ab.data()), "from_mont");
    std::vector<Fr> inst{one, ab};
    for (size_t k = 1; k < NUM_INPUTS; k++) inst.push_back(a);
    auto wit = std::make_shared<FrVector>(ctx, std::vector<Fr>{am, bm});
    auto sp = AS::sponges(hp_as::fresh_sponge<TestSponge>(amsm_ctx_curve(ctx.get())));
    r1cs_nark::Proof proof = Nark::prove(ipk, inst, wit, prng, sp.nark);
    out.push_back(Input{InputInstance{inst, proof.first_msg}, proof.second_msg});
  }
  return out;
}

static Accumulator run_template(Context& ctx, const r1cs_nark::IndexProverKey& ipk, const std::vector<size_t>& per_iteration, bool make_zk,
                                size_t iterations) {
  auto keys = AS::index(ipk);
  SchemeRng rng(2024);
  size_t total = 0;
  for (size_t k : per_iteration) total += k;
  std::vector<Input> inputs = generate_inputs(ctx, ipk, total * iterations, make_zk, rng);
  hp_as::Rng prng = make_zk ? hp_as::Rng([&rng]() { return rng.field(); }) : hp_as::Rng();
  size_t start = 0;
  Accumulator last;
  for (size_t it = 0; it < iterations; it++) {
    std::vector<Accumulator> old;
    for (size_t k : per_iteration) {
      std::vector<Input> step(inputs.begin() + (long)start, inputs.begin() + (long)(start + k));
      start += k;
      auto res = AS::prove(keys.pk, step, old, prng);
      std::vector<InputInstance> ii;
      std::vector<AccumulatorInstance> oi;
      for (auto& x : step) ii.push_back(x.instance);
      for (auto& x : old) oi.push_back(x.instance);
      if (!AS::verify(ctx, keys.vk, ii, oi, res.first.instance, res.second)) throw std::runtime_error("Verify failed");
      old.push_back(res.first);
    }
    if (!AS::decide(*keys.dk, old.back())) throw std::runtime_error("Decide failed");
    last = old.back();
  }
  return last;
}

static void print_point(const char* name, const Affine& p) {
  printf("%s %d", name, p.infinity ? 1 : 0);
  for (uint64_t w : p.xy) printf(" %016llx", (unsigned long long)w);
  printf("\n");
}

int main() {
  try {
    Context ctx = check_context(AMSM_PALLAS);
    const size_t n_inst = NUM_INPUTS + 1;
    const Fr one = {1, 0, 0, 0};
    std::vector<r1cs_nark::Matrix::Row> A, B, C;
    for (size_t k = 0; k + 1 < NUM_CONSTRAINTS; k++) {
      A.push_back({{one, n_inst + 0}});
      B.push_back({{one, n_inst + 1}});
      C.push_back({{one, 1}});
    }
    A.push_back({});
    B.push_back({});
    C.push_back({});
    r1cs_nark::IndexProverKey ipk = Nark::index(ctx, A, B, C, n_inst, n_inst + 2, 31337);
    struct Scenario {
      const char* name;
      std::vector<size_t> per_iteration;
      size_t iterations;
    } scenarios[] = {{"single_input_init", {1}, 2},          {"multiple_inputs_init", {3}, 2},
                     {"simple_accumulation", {1, 1}, 2},     {"multiple_inputs_accumulation", {1, 1, 2, 3}, 2},
                     {"accumulators_only", {1, 0, 0, 0}, 2}, {"no_inputs_init", {0}, 1}};
    for (int zk = 0; zk < 2; zk++)
      for (auto& s : scenarios) {
        run_template(ctx, ipk, s.per_iteration, zk != 0, check_iterations(s.iterations));
        printf("scenario %s %s ok %zu\n", s.name, zk ? "zk" : "no_zk", check_iterations(s.iterations));
      }
    for (int zk = 0; zk < 2; zk++) {
      Accumulator acc = run_template(ctx, ipk, {1, 1, 2, 3}, zk != 0, 1);
      const char* t = zk ? "zk" : "nozk";
      char name[64];
      snprintf(name, sizeof name, "%s_comm_a", t); print_point(name, acc.instance.comm_a);
      snprintf(name, sizeof name, "%s_comm_b", t); print_point(name, acc.instance.comm_b);
      snprintf(name, sizeof name, "%s_comm_c", t); print_point(name, acc.instance.comm_c);
      snprintf(name, sizeof name, "%s_hp_comm_3", t); print_point(name, acc.instance.hp_instance.comm_3);
      printf("%s_r1cs_input 0", t);
      for (auto& x : acc.instance.r1cs_input)
        for (uint64_t w : x) printf(" %016llx", (unsigned long long)w);
      printf("\n");
    }
    printf("done\n");
    return 0;
  } catch (const std::exception& e) {
    printf("exception %s\n", e.what());
    return 1;
  }
}
