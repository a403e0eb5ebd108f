// Synthetic input stream shared by device kernels (counter-based splitmix64).
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

}  // namespace amsm
