// The reference's `test_simple_circuit` (src/r1cs_nark_as/r1cs_nark/mod.rs:509-556: honest proofs verify, a wrong public
// input does not) on the C++ driver include/amsm_r1cs_nark.hpp with the DummyCircuit of src/r1cs_nark_as/mod.rs:1159-1188;
// prints the last proof of each mode for the byte-for-byte comparison with accumulation_amd/r1cs_nark.py.
#include <cstdio>

#include "amsm_r1cs_nark.hpp"
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
using namespace amsm::r1cs_nark;
using Nark = R1CSNark<TestSponge>;

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

static void print_point(const char* name, const Affine& p) {
  printf("%s %d", name, p.infinity ? 1 : 0);
  for (uint64_t w : p.xy) printf(" %016llx", (unsigned long long)w);
  printf("\n");
}
static void print_fr(const char* name, const Fr& x) {
  printf("%s 0", name);
  for (uint64_t w : x) printf(" %016llx", (unsigned long long)w);
  printf("\n");
}

int main() {
  try {
    Context ctx = check_context(AMSM_PALLAS);
    hp_as::FrOps fr{amsm_ctx_curve(ctx.get())};
    const size_t num_inputs = 5, num_constraints = 100, n_inst = num_inputs + 1;
    const Fr one = {1, 0, 0, 0};
    // DummyCircuit: instance = [1, a*b, a, ..., a], witness = [a, b]; num_constraints - 1 copies of a * b = c, one empty
    std::vector<Matrix::Row> A, B, C;
    for (size_t k = 0; k + 1 < num_constraints; k++) {
      A.push_back({{one, n_inst + 0}});
      B.push_back({{one, n_inst + 1}});
      C.push_back({{one, 1}});
    }
    A.push_back({});
    B.push_back({});
    C.push_back({});
    IndexProverKey ipk = Nark::index(ctx, A, B, C, n_inst, n_inst + 2, 7);
    printf("matrices_hash 0");
    for (uint8_t b : ipk.index_info.matrices_hash) printf(" %02x", b);
    printf("\n");
    for (int zk = 0; zk < 2; zk++) {
      SchemeRng rng(9);
      hp_as::Rng prover_rng = zk ? hp_as::Rng([&rng]() { return rng.field(); }) : hp_as::Rng();
      Proof last;
      for (int round = 0; round < 3; round++) {
        Fr a = rng.field(), b = rng.field();  // canonical (< 2^254 < r)
        Fr am = fr.to_mont(a), bm = fr.to_mont(b), abm = fr.mul(am, bm), ab;
        check(amsm_fr_from_mont(amsm_ctx_curve(ctx.get()), abm.data(), 1, ab.data()), "from_mont");
        std::vector<Fr> inst{one, ab};
        for (size_t k = 1; k < num_inputs; k++) inst.push_back(a);
        auto wit = std::make_shared<FrVector>(ctx, std::vector<Fr>{am, bm});
        Proof proof = Nark::prove(ipk, inst, wit, prover_rng);
        if (!Nark::verify(ipk, inst, proof)) throw std::runtime_error("honest proof rejected");
        std::vector<Fr> bad = inst;
        bad[1][0] ^= 1;  // a*b +- 1
        if (Nark::verify(ipk, bad, proof)) throw std::runtime_error("wrong public input accepted");
        last = proof;
      }
      const char* t = zk ? "zk" : "nozk";
      char name[64];
      snprintf(name, sizeof name, "%s_comm_a", t); print_point(name, last.first_msg.comm_a);
      snprintf(name, sizeof name, "%s_comm_b", t); print_point(name, last.first_msg.comm_b);
      snprintf(name, sizeof name, "%s_comm_c", t); print_point(name, last.first_msg.comm_c);
      if (zk) {
        print_point("zk_comm_r_a", last.first_msg.randomness->comm_r_a);
        print_point("zk_comm_1", last.first_msg.randomness->comm_1);
        print_point("zk_comm_2", last.first_msg.randomness->comm_2);
        print_fr("zk_sigma_a", last.second_msg.randomness->sigma_a);
        print_fr("zk_sigma_o", last.second_msg.randomness->sigma_o);
      }
      printf("mode %s ok\n", t);
    }
    printf("done\n");
    return 0;
  } catch (const std::exception& e) {
    printf("exception %s\n", e.what());
    return 1;
  }
}
