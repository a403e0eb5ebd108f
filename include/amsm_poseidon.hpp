// amsm_poseidon.hpp -- the reference's sponge as the `Sponge` template argument of the scheme drivers
// (include/amsm_hp_as.hpp: the Sponge concept): ark-sponge `PoseidonSponge<ConstraintF<G>>` over the curve's base field
// (ext, git branch `accumulation-experimental`, Cargo.toml:18), instantiated by every scheme test of the reference
// (src/hp_as/mod.rs:1047-1055, src/r1cs_nark_as/mod.rs:1279-1287, src/ipa_pc_as/mod.rs:1006-1014) and by its benchmark harness
// (examples/scaling-as.rs:27,36).  A thin RAII wrapper of the host-side C ABI (include/amsm.h: amsm_poseidon_*); parameters,
// duplex rules and `Absorbable` encodings are stated in accumulation_amd/csrc/host_poseidon.h (PARITY UNPINNED).
//
//   using AS = amsm::hp_as::ASForHadamardProducts<amsm::poseidon::PoseidonSponge>;
#pragma once
#include <cstring>
#include <utility>
#include <vector>

#include "amsm.hpp"

namespace amsm {
namespace poseidon {

class PoseidonSponge {
 public:
  explicit PoseidonSponge(int curve = AMSM_PALLAS) { check(amsm_poseidon_new(curve, &h_), "amsm_poseidon_new"); }
  PoseidonSponge(const PoseidonSponge& o) { check(amsm_poseidon_clone(o.h_, &h_), "amsm_poseidon_clone"); }
  PoseidonSponge(PoseidonSponge&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
  PoseidonSponge& operator=(PoseidonSponge o) noexcept {
    std::swap(h_, o.h_);
    return *this;
  }
  ~PoseidonSponge() { amsm_poseidon_free(h_); }

  void absorb_bytes(const std::vector<uint8_t>& b) {  // `Absorbable for [u8]`: 31-byte (47 for BLS12-381) LE chunks
    check(amsm_poseidon_absorb_bytes(h_, b.data(), b.size()), "amsm_poseidon_absorb_bytes");
  }
  void absorb_u64(uint64_t x) { check(amsm_poseidon_absorb_u64(h_, x), "amsm_poseidon_absorb_u64"); }  // a usize
  void absorb_len(uint64_t) {}  // a Vec is absorbed item by item, without its length
  void absorb_point(const Affine& p) {  // (x, y, infinity)
    uint8_t inf = (p.infinity || p.xy.empty()) ? 1 : 0;
    static const uint64_t zeros[12] = {0};
    check(amsm_poseidon_absorb_points(h_, p.xy.empty() ? zeros : p.xy.data(), &inf, 1), "amsm_poseidon_absorb_points");
  }
  void absorb_points(const std::vector<Affine>& pts) {
    for (const Affine& p : pts) absorb_point(p);
  }
  // `squeeze_nonnative_field_elements_with_sizes(&[Truncated(n_bits); count])`: canonical limbs
  std::vector<Fr> squeeze_field_elements(size_t count, unsigned n_bits) {
    std::vector<Fr> out(count);
    if (count) check(amsm_poseidon_squeeze_nonnative(h_, n_bits, count, out[0].data()), "amsm_poseidon_squeeze_nonnative");
    return out;
  }
  Fr squeeze_bits(unsigned n_bits) { return squeeze_field_elements(1, n_bits)[0]; }
  PoseidonSponge fork(const char* domain) const {
    PoseidonSponge c(nullptr);
    check(amsm_poseidon_fork(h_, reinterpret_cast<const uint8_t*>(domain), strlen(domain), &c.h_), "amsm_poseidon_fork");
    return c;
  }

 private:
  explicit PoseidonSponge(std::nullptr_t) {}
  amsm_sponge* h_ = nullptr;
};

}  // namespace poseidon
}  // namespace amsm
