// Synthetic input streams shared by device kernels and the host backend (counter-based splitmix64).
//   rng_scalar      the MULTIPLIER stream: 254-bit integers k_i, only used for G_i = k_i G (any fixed set of distinct
//                   subgroup points serves as a committer key)
//   rng_scalar_fr   the SCALAR stream of amsm_vec_random: uniform in [0, r) of the curve's scalar field (round 6: the
//                   254-bit stream never produced the 45 % of BLS12-381 scalars that have bit 254 set)
#pragma once
#include "fp.h"

namespace amsm {

// ---------------------------------------------------------------------------------------------
// Synthetic scalar stream (identical to oracle/pyref.py:rng_word / rng_scalar).
// ---------------------------------------------------------------------------------------------
AMSM_HD u64 rng_mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
AMSM_HD u64 rng_word(u64 seed, u64 j) {
  return rng_mix64(seed * 0xD1342543DE82EF95ull + j * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull);
}
AMSM_HD void rng_scalar(u64 seed, u64 i, u32 out[8]) {
  for (int k = 0; k < 4; k++) {
    u64 w = rng_word(seed, 4 * i + k);
    if (k == 3) w &= (1ull << 62) - 1;
    out[2 * k] = (u32)w;
    out[2 * k + 1] = (u32)(w >> 32);
  }
}

// Uniform in [0, r): candidate t of scalar i = words (t << 40) + 4 i .. + 3 masked to 255 bits (both scalar fields have 255-bit
// moduli), the first candidate below r wins (rejection sampling: exactly uniform; acceptance 0.50 Pallas, 0.906 BLS12-381).
// After 64 rejections (probability < 2^-64) candidate 63 with bit 254 cleared (< 2^254 < r for both fields).
// The checkers restate it (pyref.py: rng_fr; the C restatement: ark_rng_scalars_fr; tools/ark_vectors/src/main.rs: rng_frs).
template <class Fr>
AMSM_HD void rng_scalar_fr(u64 seed, u64 i, u32 out[8]) {
  static_assert(Fr::W == 8, "255-bit scalar fields");
  for (u64 t = 0; t < 64; t++) {
    for (int k = 0; k < 4; k++) {
      u64 w = rng_word(seed, (t << 40) + 4 * i + k);
      if (k == 3) w &= (1ull << 63) - 1;
      out[2 * k] = (u32)w;
      out[2 * k + 1] = (u32)(w >> 32);
    }
    bool below = false;  // out < r ?
#pragma unroll
    for (int k = 7; k >= 0; k--) {
      if (out[k] != Fr::mod(k)) {
        below = out[k] < Fr::mod(k);
        break;
      }
    }
    if (below) return;
  }
  out[7] &= 0x3fffffffu;
}

}  // namespace amsm
