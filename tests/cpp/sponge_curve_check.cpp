// The Poseidon sponge of a driver call is the sponge over the CONTEXT's curve's base field (the reference's
// `PoseidonSponge<ConstraintF<G>>`: src/hp_as/mod.rs:98-103), whatever the caller default-constructed: a prove over BLS12-381 with
// the `Sponge sponge = Sponge()` default argument must give the proof of the same call with an explicit
// PoseidonSponge(AMSM_BLS12_381_G1), and a sponge that was created for another curve AND already used must be refused, not run as
// a transcript over the wrong field.  (Round 6: the default argument used to stay a Pallas-field sponge on any curve.)
#include <cstdio>

#include "amsm_hp_as.hpp"
#include "amsm_ipa_pc_as.hpp"
#include "amsm_poseidon.hpp"

#include "check_device.hpp"

using namespace amsm;
using namespace amsm::hp_as;
using P = amsm::poseidon::PoseidonSponge;
using AS = ASForHadamardProducts<P>;

static std::vector<Accumulator> inputs(Context& ctx, const CommitterKey& ck, size_t n, size_t num) {
  FrOps fr{amsm_ctx_curve(ctx.get())};
  std::vector<Accumulator> out;
  for (size_t t = 0; t < num; t++) {
    Fr va{{3 + t, 0, 0, 0}}, vb{{11 + 2 * t, 0, 0, 0}};
    auto a = filled(ctx, fr.to_mont(va), n);
    auto b = filled(ctx, fr.to_mont(vb), n);
    FrVector prod = compute_hp(*a, *b);
    out.push_back(Accumulator{InputInstance{PedersenCommitment::commit(ck, *a, nullptr), PedersenCommitment::commit(ck, *b, nullptr),
                                            PedersenCommitment::commit(ck, prod, nullptr)},
                              InputWitness{a, b, std::nullopt}});
  }
  return out;
}

static bool same(const Affine& x, const Affine& y) { return x.infinity == y.infinity && x.xy == y.xy; }

int main() {
  try {
    for (int curve : {AMSM_PALLAS, AMSM_BLS12_381_G1}) {
      Context ctx = check_context(curve);
      CommitterKey ck = PedersenCommitment::setup(ctx, 9, 4242);
      auto in = inputs(ctx, ck, 9, 2);
      auto dflt = AS::prove(ck, in, {});                    // Sponge sponge = Sponge(): knows no context
      auto expl = AS::prove(ck, in, {}, Rng(), P(curve));   // the sponge over this curve's base field
      if (!same(dflt.first.instance.comm_1, expl.first.instance.comm_1) || !same(dflt.first.instance.comm_3, expl.first.instance.comm_3) ||
          dflt.second.product_poly_comm.low.size() != expl.second.product_poly_comm.low.size())
        throw std::runtime_error("default-argument sponge and explicit sponge disagree");
      std::vector<InputInstance> ii{in[0].instance, in[1].instance};
      if (!AS::verify(ctx, ck.supported_num_elems(), ii, {}, dflt.first.instance, dflt.second)) throw std::runtime_error("verify failed");
      if (!AS::verify(ctx, ck.supported_num_elems(), ii, {}, dflt.first.instance, dflt.second, P(curve))) throw std::runtime_error("verify (explicit) failed");
      // a sponge of the OTHER curve: still pristine -> re-created for this one; already used -> refused
      const int other = curve == AMSM_PALLAS ? AMSM_BLS12_381_G1 : AMSM_PALLAS;
      auto rebound = AS::prove(ck, in, {}, Rng(), P(other));
      if (!same(rebound.first.instance.comm_1, expl.first.instance.comm_1)) throw std::runtime_error("pristine sponge of the other curve was not re-created");
      P used(other);
      used.absorb_u64(7);
      bool refused = false;
      try {
        AS::prove(ck, in, {}, Rng(), used);
      } catch (const std::runtime_error&) {
        refused = true;
      }
      if (!refused) throw std::runtime_error("a used sponge over another curve's field was accepted");
      // a fork keeps its parent's curve and counts as used
      P child = P(curve).fork("X");
      if (child.curve() != curve) throw std::runtime_error("fork lost the curve");
      AS::prove(ck, in, {}, Rng(), child);  // same curve: fine
      printf("curve %d ok\n", curve);
    }
    printf("SPONGE_CURVE_OK\n");
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "FAILED: %s\n", e.what());
    return 1;
  }
}
