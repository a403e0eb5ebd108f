// Scalar-field kernels (digit extraction, synthetic vectors, Hadamard / linear-combination / t-vector
// loops) instantiated for the scalar fields of both curves, plus the curve-independent bounds kernel.
#include "launch.h"
#include "vec_kernels.h"

#include <cstring>

namespace amsm {

static inline u32 cdiv_(u32 a, u32 b) { return (a + b - 1) / b; }

void launch_bounds(hipStream_t st, const void* keys_sorted, bool keys16, u32* vals_sorted, MsmGeom g, u32* start,
                   u32* items) {
  if (keys16)
    hipLaunchKernelGGL((k_bounds<uint16_t>), dim3(cdiv_(g.B, 256)), dim3(256), 0, st, (const uint16_t*)keys_sorted,
                       vals_sorted, g, start, items);
  else
    hipLaunchKernelGGL((k_bounds<u32>), dim3(cdiv_(g.B, 256)), dim3(256), 0, st, (const u32*)keys_sorted, vals_sorted,
                       g, start, items);
}

void launch_vec_fill(hipStream_t st, u32* out, const u32 v[8], u32 n) {
  hipLaunchKernelGGL(k_vec_fill, dim3(cdiv_(n, 256)), dim3(256), 0, st, out, make_uint4(v[0], v[1], v[2], v[3]),
                     make_uint4(v[4], v[5], v[6], v[7]), n);
}

#define AMSM_FR_LAUNCHERS(FR)                                                                                        \
  template <>                                                                                                        \
  void launch_digits<FR>(hipStream_t st, const u32* scalars, int mont, MsmGeom g, void* keys, bool keys16, u32* vals, \
                         u32* err) {                                                                                 \
    if (keys16)                                                                                                      \
      hipLaunchKernelGGL((k_digits<FR, uint16_t>), dim3(cdiv_(g.n, 256)), dim3(256), 0, st, scalars, mont, g,         \
                         (uint16_t*)keys, vals, err);                                                                \
    else                                                                                                             \
      hipLaunchKernelGGL((k_digits<FR, u32>), dim3(cdiv_(g.n, 256)), dim3(256), 0, st, scalars, mont, g, (u32*)keys,  \
                         vals, err);                                                                                 \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_vec_random<FR>(hipStream_t st, u32* out, u64 seed, u32 n, int mont) {                                  \
    hipLaunchKernelGGL((k_vec_random<FR>), dim3(cdiv_(n, 256)), dim3(256), 0, st, out, seed, n, mont);               \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_vec_hadamard<FR>(hipStream_t st, const u32* a, const u32* b, u32* out, u32 n) {                        \
    hipLaunchKernelGGL((k_vec_hadamard<FR>), dim3(cdiv_(n, 256)), dim3(256), 0, st, a, b, out, n);                   \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_vec_combine<FR>(hipStream_t st, const CombineArgs& a, u32* out) {                                      \
    hipLaunchKernelGGL((k_vec_combine<FR>), dim3(cdiv_(a.n, 256)), dim3(256), 0, st, a, out);                        \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_vec_powers<FR>(hipStream_t st, const u32 point[8], u32 n, u32* out) {                                  \
    CombineArgs a;                                                                                                   \
    memset(&a, 0, sizeof(a));                                                                                        \
    memcpy(a.coeff[0], point, 32);                                                                                   \
    a.n = n;                                                                                                         \
    hipLaunchKernelGGL((k_vec_powers<FR>), dim3(cdiv_(n, 256)), dim3(256), 0, st, a, out);                           \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_vec_inner_product<FR>(hipStream_t st, const u32* a, const u32* b, u32 n, u32 blocks, u32* out) {       \
    hipLaunchKernelGGL((k_vec_inner_product<FR>), dim3(blocks), dim3(256), 0, st, a, b, n, out);                     \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_check_poly_coeffs<FR>(hipStream_t st, const u32* xi, u32 k, u32* out) {                                \
    CheckPolyArgs a;                                                                                                 \
    memset(&a, 0, sizeof(a));                                                                                        \
    memcpy(a.xi, xi, (size_t)k * 32);                                                                                \
    a.k = k;                                                                                                         \
    hipLaunchKernelGGL((k_check_poly_coeffs<FR>), dim3(cdiv_(1u << k, 256)), dim3(256), 0, st, a, out);              \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_spmv<FR>(hipStream_t st, const u32* row_ptr, const u32* col, const u32* val, const u32* input,         \
                       u32 n_input, const u32* witness, u32 n_witness, u32* out, u32 n_rows) {                       \
    hipLaunchKernelGGL((k_spmv<FR>), dim3(cdiv_(n_rows, 256)), dim3(256), 0, st, row_ptr, col, val, input, n_input,   \
                       witness, n_witness, out, n_rows);                                                             \
  }                                                                                                                  \
  template <>                                                                                                        \
  void launch_hp_t_vecs<FR>(hipStream_t st, const TVecArgs& a, int n_inputs) {                                       \
    dim3 grid(cdiv_(a.len, 256)), block(256);                                                                        \
    switch (n_inputs) {                                                                                              \
      case 1: hipLaunchKernelGGL((k_hp_t_vecs<FR, 1>), grid, block, 0, st, a); break;                                \
      case 2: hipLaunchKernelGGL((k_hp_t_vecs<FR, 2>), grid, block, 0, st, a); break;                                \
      case 3: hipLaunchKernelGGL((k_hp_t_vecs<FR, 3>), grid, block, 0, st, a); break;                                \
      case 4: hipLaunchKernelGGL((k_hp_t_vecs<FR, 4>), grid, block, 0, st, a); break;                                \
      case 5: hipLaunchKernelGGL((k_hp_t_vecs<FR, 5>), grid, block, 0, st, a); break;                                \
      case 6: hipLaunchKernelGGL((k_hp_t_vecs<FR, 6>), grid, block, 0, st, a); break;                                \
      case 7: hipLaunchKernelGGL((k_hp_t_vecs<FR, 7>), grid, block, 0, st, a); break;                                \
      default: hipLaunchKernelGGL((k_hp_t_vecs<FR, 8>), grid, block, 0, st, a); break;                               \
    }                                                                                                                \
  }

AMSM_FR_LAUNCHERS(PallasFr)
AMSM_FR_LAUNCHERS(Bls12381Fr)

}  // namespace amsm
