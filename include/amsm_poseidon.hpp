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
#include <stdexcept>
#include <utility>
#include <vector>

#include "amsm.hpp"

namespace amsm {
namespace poseidon {

class PoseidonSponge {
 public:
  explicit PoseidonSponge(int curve = AMSM_PALLAS) : curve_(curve) { check(amsm_poseidon_new(curve, &h_), "amsm_poseidon_new"); }
  PoseidonSponge(const PoseidonSponge& o) : curve_(o.curve_), pristine_(o.pristine_) { check(amsm_poseidon_clone(o.h_, &h_), "amsm_poseidon_clone"); }
  PoseidonSponge(PoseidonSponge&& o) noexcept : h_(o.h_), curve_(o.curve_), pristine_(o.pristine_) { o.h_ = nullptr; }
  PoseidonSponge& operator=(PoseidonSponge o) noexcept {
    std::swap(h_, o.h_);
    curve_ = o.curve_;
    pristine_ = o.pristine_;
    return *this;
  }
  ~PoseidonSponge() { amsm_poseidon_free(h_); }
  // The sponge field is the CURVE's base field (`ConstraintF<G>`): a sponge the caller default-constructed (the drivers' `Sponge
  // sponge = Sponge()` arguments cannot see the context) becomes a sponge for the context's curve here, before its first absorb.
  // One that was created for another curve and already used is an error, never a silent transcript over the wrong field.
  void for_curve(int curve) {
    if (curve == curve_) return;
    if (!pristine_) throw std::runtime_error("PoseidonSponge: created for another curve and already absorbed into");
    amsm_sponge* h = nullptr;
    check(amsm_poseidon_new(curve, &h), "amsm_poseidon_new");
    amsm_poseidon_free(h_);
    h_ = h;
    curve_ = curve;
  }
  int curve() const { return curve_; }

  void absorb_bytes(const std::vector<uint8_t>& b) {  // `Absorbable for [u8]`: 31-byte (47 for BLS12-381) LE chunks
    pristine_ = false;
    check(amsm_poseidon_absorb_bytes(h_, b.data(), b.size()), "amsm_poseidon_absorb_bytes");
  }
  void absorb_u64(uint64_t x) {  // a usize
    pristine_ = false;
    check(amsm_poseidon_absorb_u64(h_, x), "amsm_poseidon_absorb_u64");
  }
  void absorb_len(uint64_t) {}  // a Vec is absorbed item by item, without its length
  void absorb_point(const Affine& p) {  // (x, y, infinity)
    pristine_ = false;
    uint8_t inf = (p.infinity || p.xy.empty()) ? 1 : 0;
    static const uint64_t zeros[12] = {0};
    check(amsm_poseidon_absorb_points(h_, p.xy.empty() ? zeros : p.xy.data(), &inf, 1), "amsm_poseidon_absorb_points");
  }
  void absorb_points(const std::vector<Affine>& pts) {
    for (const Affine& p : pts) absorb_point(p);
  }
  // `squeeze_nonnative_field_elements_with_sizes(&[Truncated(n_bits); count])`: canonical limbs
  std::vector<Fr> squeeze_field_elements(size_t count, unsigned n_bits) {
    pristine_ = false;
    std::vector<Fr> out(count);
    if (count) check(amsm_poseidon_squeeze_nonnative(h_, n_bits, count, out[0].data()), "amsm_poseidon_squeeze_nonnative");
    return out;
  }
  Fr squeeze_bits(unsigned n_bits) { return squeeze_field_elements(1, n_bits)[0]; }
  PoseidonSponge fork(const char* domain) const {
    PoseidonSponge c(nullptr);
    check(amsm_poseidon_fork(h_, reinterpret_cast<const uint8_t*>(domain), strlen(domain), &c.h_), "amsm_poseidon_fork");
    c.curve_ = curve_;
    c.pristine_ = false;
    return c;
  }

 private:
  explicit PoseidonSponge(std::nullptr_t) {}
  amsm_sponge* h_ = nullptr;
  int curve_ = AMSM_PALLAS;
  bool pristine_ = true;  // nothing absorbed or squeezed yet: for_curve() may still re-create it
};

}  // namespace poseidon
}  // namespace amsm
