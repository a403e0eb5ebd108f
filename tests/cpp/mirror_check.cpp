// Exercises include/amsm.hpp (the C++ host-side mirror) on a GPU and prints results as `name hex...` lines;
// tests/test_cpp_mirror.py builds it with g++ and checks every line against the oracle.
#include <cstdio>

#include "amsm.hpp"

#include "check_device.hpp"

using namespace amsm;

static void print_point(const char* name, const Affine& p) {
  printf("%s %d", name, p.infinity ? 1 : 0);
  for (uint64_t w : p.xy) printf(" %016llx", (unsigned long long)w);
  printf("\n");
}

int main() {
  try {
    Context ctx = check_context(AMSM_PALLAS);
    const size_t n = 1000;
    CommitterKey ck = PedersenCommitment::setup(ctx, n, 0x5EED1001ull);
    printf("supported_num_elems %zu\n", ck.supported_num_elems());
    FrVector a = FrVector::random(ctx, 11, n, true), b = FrVector::random(ctx, 12, n, true);
    print_point("commit_a", VariableBaseMSM::multi_scalar_mul(ck, a));
    print_point("commit_b", PedersenCommitment::commit(ck, b));
    // canonical host scalars through the BigInt entry point
    FrVector a_canon = FrVector::random(ctx, 11, n, false);
    print_point("msm_a_bigint", VariableBaseMSM::multi_scalar_mul(ck, a_canon.to_host()));
    // the ark-ec call shape itself (amsm_msm_oneshot): bases AND scalars are host slices of this call; an identity base by flag
    {
      std::vector<uint64_t> gens = ck.read(0, n);
      print_point("oneshot_a", VariableBaseMSM::multi_scalar_mul(ctx, gens.data(), nullptr, n, a_canon.to_host()));
      std::vector<uint8_t> inf(n, 0);
      inf[7] = 1;
      print_point("oneshot_a_without_7", VariableBaseMSM::multi_scalar_mul(ctx, gens.data(), inf.data(), n, a_canon.to_host()));
      print_point("oneshot_a_first_300", VariableBaseMSM::multi_scalar_mul(ctx, gens.data(), nullptr, 300, a_canon.to_host()));
    }
    // a + 3 b  (combine_vectors), a o b (compute_hp)
    Fr one = {0x5b2b3e9cfffffffdull, 0x992c350be3420567ull, 0xffffffffffffffffull, 0x3fffffffffffffffull};  // R mod r
    Fr three;
    {
      // 3 in Montgomery form = one + one + one computed on the device to avoid host field code here
      FrVector ones(ctx, std::vector<Fr>{one});
      FrVector t = hp_as::combine_vectors(ctx, {&ones, &ones, &ones}, {one, one, one});
      three = t.to_host()[0];
    }
    FrVector lin = hp_as::combine_vectors(ctx, {&a, &b}, {one, three});
    print_point("commit_a_plus_3b", PedersenCommitment::commit(ck, lin));
    FrVector prod = hp_as::compute_hp(a, b);
    print_point("commit_a_had_b", PedersenCommitment::commit(ck, prod));
    Fr rnd = three;
    print_point("commit_a_hiding_3", PedersenCommitment::commit(ck, a, &rnd));
    // t-vectors of two inputs and their commitments (compute_product_poly_comm)
    FrVector a2 = FrVector::random(ctx, 13, n, true), b2 = FrVector::random(ctx, 14, n, true);
    auto t = hp_as::compute_t_vecs(ctx, {&a, &a2}, {&b, &b2}, {one, three}, n, nullptr, nullptr, false);
    printf("t_vecs %zu middle_skipped %d\n", t.size(), t[1] == nullptr ? 1 : 0);
    auto comm = hp_as::compute_product_poly_comm(ck, t);
    print_point("ppc_low0", comm.first.at(0));
    print_point("ppc_high0", comm.second.at(0));
    // several MSMs per call: same bases, windows of the key, index classes of one vector; key-to-key fold
    {
      auto same = MsmBatch::same_bases(ck, {&a, &b});
      print_point("batch_a", same.at(0));
      print_point("batch_b", same.at(1));
      auto grp = MsmBatch::grouped(ck, a, 3);
      print_point("grouped_0", grp.at(0));
      print_point("grouped_1", grp.at(1));
      CommitterKey folded = ck.fold(500, three, 8);  // key[i] + 3 key[500 + i]
      FrVector a500 = FrVector::random(ctx, 11, 500, true);
      print_point("fold_commit", VariableBaseMSM::multi_scalar_mul(folded, a500));
      auto win = MsmBatch::windows(ck, {{0, &a500}, {500, &a500}});
      print_point("win_lo", win.at(0));
      print_point("win_hi", win.at(1));
    }
    // host slices, several per call (amsm_msm_batch / amsm_pedersen_commit_batch): the same points as the device forms
    {
      std::vector<Fr> ha = a_canon.to_host(), hb = FrVector::random(ctx, 12, n, false).to_host();
      auto hb2 = MsmBatch::same_bases_host(ck, {&ha, &hb, &ha});
      print_point("hostbatch_a", hb2.at(0));
      print_point("hostbatch_b", hb2.at(1));
      print_point("hostbatch_a2", hb2.at(2));
      std::vector<Fr> ma = a.to_host(), mb = b.to_host();
      mb.resize(777);  // lengths may differ
      auto cb = PedersenCommitment::commit_batch_host(ck, {&ma, &mb, &ma}, {nullptr, nullptr, &rnd});
      print_point("hostcommit_a", cb.at(0));
      print_point("hostcommit_b777", cb.at(1));
      print_point("hostcommit_a_hiding_3", cb.at(2));
      FrVector b777(ctx, mb);
      print_point("commit_b777", PedersenCommitment::commit(ck, b777));
    }
    // error behaviour: a key without hiding generator cannot take a randomizer
    std::vector<uint64_t> xy = ck.read(0, 4);
    CommitterKey bare = CommitterKey::load(ctx, xy, nullptr);
    try {
      PedersenCommitment::commit(bare, a, &rnd);
      printf("error_check missing\n");
    } catch (const Error& e) {
      printf("error_check %d\n", e.status);
    }
    printf("done\n");
    return 0;
  } catch (const std::exception& e) {
    printf("exception %s\n", e.what());
    return 1;
  }
}
