// Throughput of the field multiplication / XYZZ mixed addition in isolation (no memory traffic):
// cycles per operation per SIMD at several occupancies.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I accumulation_amd/csrc tools/fp_bench.hip -o build/fp_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "ec.h"
using namespace amsm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class P, int MODE, int MINW = 1>
__global__ void __launch_bounds__(256, MINW) kern(u32* out, int iters) {
  Fe<P> x, y;
  for (int i = 0; i < P::L; i++) { x.v[i] = threadIdx.x * 2654435761u + i * 40503u + blockIdx.x; y.v[i] = threadIdx.x * 97u + i * 7919u + 12345u; }
  x.v[P::L - 1] &= 0x0fffffffu; y.v[P::L - 1] &= 0x0fffffffu;
  if constexpr (P::UNSAT) {  // tight limbs, value < 2^254
    for (int i = 0; i < P::L; i++) { x.v[i] &= u_mask<P>(); y.v[i] &= u_mask<P>(); }
    x.v[P::L - 1] &= (P::L == 9 ? 0x003fffffu : 0x00000fffu); y.v[P::L - 1] &= (P::L == 9 ? 0x003fffffu : 0x00000fffu);
  }
  if constexpr (MODE == 0) {  // dependent multiplication chain
    for (int i = 0; i < iters; i++) x = fe_mul<P>(x, y);
  } else if constexpr (MODE == 1) {  // portable reference multiplication
    for (int i = 0; i < iters; i++) x = fe_mul_ref<P>(x, y);
  } else if constexpr (MODE == 2) {  // add/sub chain
    for (int i = 0; i < iters; i++) { x = fe_add<P>(x, y); y = fe_sub<P>(y, x); }
  } else if constexpr (MODE == 3) {  // mixed addition chain
    XYZZ<P> acc; acc.x = x; acc.y = y; acc.zz = fe_one<P>(); acc.zzz = fe_one<P>();
    Affine<P> q; q.x = y; q.y = x;
    for (int i = 0; i < iters; i++) { xyzz_madd<P>(acc, q); q.x.v[0] += 1; }
    for (int i = 0; i < P::L; i++) x.v[i] = acc.x.v[i] ^ acc.y.v[i] ^ acc.zz.v[i] ^ acc.zzz.v[i];
  } else if constexpr (MODE == 4) {  // squaring chain
    for (int i = 0; i < iters; i++) x = fe_sqr<P>(x);
  }
  u32 o = 0;
  for (int i = 0; i < P::L; i++) o ^= x.v[i] ^ y.v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = o;
}

template <class P, int MODE, int MINW = 1>
int run(const char* name, int waves_per_simd, int iters, double ops_per_iter, u32* d_out) {
  int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<P, MODE, MINW><<<blocks, 256>>>(d_out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  kern<P, MODE, MINW><<<blocks, 256>>>(d_out, iters);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double wave_ops = (double)blocks * 4 * iters * ops_per_iter;
  double per_simd_s = wave_ops / 1024.0 / (ms * 1e-3);
  printf("%-30s w/SIMD=%d  %8.3f ms  %8.1f cycles/op/SIMD @2.4GHz   %7.2f Gops/s (lane ops)\n", name, waves_per_simd, ms,
         2.4e9 / per_simd_s, wave_ops * 64 / (ms * 1e-3) * 1e-9);
  return 0;
}

int main() {
  u32* d_out;
  CK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
  for (int w : {1, 2, 4, 5, 8}) {
    run<PallasFq, 0>("pallas fe_mul (asm)", w, 2000, 1, d_out);
    run<PallasFq, 1>("pallas fe_mul_ref (hipcc)", w, 2000, 1, d_out);
    run<PallasFq, 2>("pallas fe_add+fe_sub", w, 4000, 2, d_out);
    run<PallasFq, 3>("pallas xyzz_madd", w, 400, 1, d_out);
    run<PallasFqU, 0>("pallas 9x29 fe_mul", w, 2000, 1, d_out);
    run<PallasFqU, 4>("pallas 9x29 fe_sqr", w, 2000, 1, d_out);
    run<PallasFqU, 3>("pallas 9x29 xyzz_madd", w, 400, 1, d_out);
    run<PallasFqU, 3, 4>("pallas 9x29 xyzz_madd (<=128 VGPR)", w, 400, 1, d_out);
    run<Bls12381Fq, 0>("bls12-381 fe_mul (asm)", w, 1000, 1, d_out);
    run<Bls12381Fq, 3>("bls12-381 xyzz_madd", w, 200, 1, d_out);
    run<Bls12381FqU, 0>("bls12-381 14x28 fe_mul", w, 1000, 1, d_out);
    run<Bls12381FqU, 4>("bls12-381 14x28 fe_sqr", w, 1000, 1, d_out);
    run<Bls12381FqU, 3>("bls12-381 14x28 xyzz_madd", w, 200, 1, d_out);
  }
  return 0;
}
