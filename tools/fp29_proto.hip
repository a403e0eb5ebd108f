// Prototype: Pallas Fq Montgomery multiplication with 9 x 29-bit unsaturated limbs (R' = 2^261): every column sum
// fits a 64-bit accumulator, so there is NO carry handling per product (one v_mad_u64_u32 each).  Compared with the
// saturated 8 x 32-bit asm schedule (fp_mul_gfx950.h).  Correctness is checked against fe_mul via conversions.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ec.h"
using namespace amsm;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr u32 M29 = (1u << 29) - 1;
struct F29 { u32 l[9]; };

// modulus limbs radix 2^29
__host__ __device__ constexpr u32 p29(int i) {
  constexpr u32 t[9] = {0x00000001u, 0x0c96987du /*filled below*/, 0, 0, 0, 0, 0, 0, 0};
  return t[i];
}

struct P29 {
  // computed on host, passed as kernel arg
  u32 p[9];
};

__device__ __forceinline__ F29 mul29(const F29& a, const F29& b, const P29& P) {
  u64 acc = 0;
  u32 m[9];
  F29 r;
#pragma unroll
  for (int k = 0; k < 18; k++) {
#pragma unroll
    for (int i = 0; i < 9; i++) {
      int j = k - i;
      if (j >= 0 && j < 9) acc += (u64)a.l[i] * b.l[j];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) {
      int j = k - i;
      if (i < k && j >= 1 && j < 9 && (j <= 4 || j == 8)) acc += (u64)m[i] * P.p[j];
    }
    if (k < 9) {
      u32 lo = (u32)acc & M29;
      m[k] = (0u - lo) & M29;
      acc += m[k];  // p0 = 1
      acc >>= 29;
    } else {
      r.l[k - 9] = (k == 17) ? (u32)acc : ((u32)acc & M29);
      acc >>= 29;
    }
  }
  return r;
}

__device__ __forceinline__ F29 to29(const Fe<PallasFq>& x) {
  F29 r;
  u64 lo = 0; int bits = 0; int wi = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    while (bits < 29 && wi < 8) { lo |= (u64)x.v[wi++] << bits; bits += 32; }
    r.l[i] = (u32)lo & M29; lo >>= 29; bits -= 29;
  }
  return r;
}
__device__ __forceinline__ Fe<PallasFq> from29(const F29& x) {  // assumes value < 2^256, limbs tight
  Fe<PallasFq> r; u64 lo = 0; int bits = 0; int wi = 0;
#pragma unroll
  for (int i = 0; i < 9; i++) {
    lo |= (u64)x.l[i] << bits; bits += 29;
    while (bits >= 32 && wi < 8) { r.v[wi++] = (u32)lo; lo >>= 32; bits -= 32; }
  }
  if (wi < 8) r.v[wi] = (u32)lo;
  return r;
}

template <int MODE>
__global__ void __launch_bounds__(256) kern(u32* out, int iters, P29 P, const u32* in) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  Fe<PallasFq> x = fe_load<PallasFq>(in + (t % 1024) * 16), y = fe_load<PallasFq>(in + (t % 1024) * 16 + 8);
  if (MODE == 0) {
    for (int i = 0; i < iters; i++) x = fe_mul<PallasFq>(x, y);
    fe_store<PallasFq>(out + t * 8, x);
  } else {
    F29 a = to29(x), b = to29(y);
    for (int i = 0; i < iters; i++) a = mul29(a, b, P);
    // final: result limbs tight, value < 2p: conditional subtract via saturated form
    Fe<PallasFq> r = from29(a);
    fe_cond_sub<PallasFq>(r, 0);
    fe_store<PallasFq>(out + t * 8, r);
  }
}

int main() {
  // modulus limbs
  P29 P;
  {
    unsigned __int128 t = ((unsigned __int128)0x224698fc094cf91bull << 64) | 0x992d30ed00000001ull;
    for (int i = 0; i < 9; i++) P.p[i] = 0;
    for (int i = 0; i < 5; i++) { P.p[i] = (u32)(t & M29); t >>= 29; }
    P.p[8] = 1u << 22;
  }
  const int N = 256 * 256 * 8;
  u32 *d_in, *d_o0, *d_o1;
  CK(hipMalloc(&d_in, 1024 * 16 * 4)); CK(hipMalloc(&d_o0, N * 8 * 4)); CK(hipMalloc(&d_o1, N * 8 * 4));
  std::vector<u32> h(1024 * 16);
  unsigned long long s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (u32)s; }
  for (int i = 0; i < 1024 * 2; i++) h[i * 8 + 7] &= 0x3fffffffu;  // < 2^254 < p
  CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  // correctness with 1 iteration: mul29 computes a*b/2^261 vs fe_mul a*b/2^256: compare after scaling: check (29 result)*2^5 == (32 result) mod p
  // simpler: iterate both 1x and compare r32 == r29 * 32 mod p on the host
  kern<0><<<4, 256>>>(d_o0, 1, P, d_in); kern<1><<<4, 256>>>(d_o1, 1, P, d_in);
  CK(hipDeviceSynchronize());
  std::vector<u32> o0(1024 * 8), o1(1024 * 8);
  CK(hipMemcpy(o0.data(), d_o0, o0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), d_o1, o1.size() * 4, hipMemcpyDeviceToHost));
  // host check using __int128 big arithmetic: (o1 * 32) mod p == o0
  int bad = 0;
  for (int t = 0; t < 1024; t++) {
    // multiply o1 by 32 mod p via 5 doublings
    unsigned long long a[4], pm[4] = {0x992d30ed00000001ull, 0x224698fc094cf91bull, 0, 0x4000000000000000ull};
    for (int k = 0; k < 4; k++) a[k] = (unsigned long long)o1[t * 8 + 2 * k] | ((unsigned long long)o1[t * 8 + 2 * k + 1] << 32);
    for (int d = 0; d < 5; d++) {
      unsigned long long c = 0, r[4];
      for (int k = 0; k < 4; k++) { unsigned __int128 x = (unsigned __int128)a[k] + a[k] + c; r[k] = (unsigned long long)x; c = (unsigned long long)(x >> 64); }
      // subtract p if >= p
      unsigned long long br = 0, q[4];
      for (int k = 0; k < 4; k++) { unsigned __int128 x = (unsigned __int128)r[k] - pm[k] - br; q[k] = (unsigned long long)x; br = (unsigned long long)(x >> 64) & 1; }
      bool ge = c || !br;
      for (int k = 0; k < 4; k++) a[k] = ge ? q[k] : r[k];
    }
    for (int k = 0; k < 4; k++) {
      unsigned long long e = (unsigned long long)o0[t * 8 + 2 * k] | ((unsigned long long)o0[t * 8 + 2 * k + 1] << 32);
      if (e != a[k]) bad++;
    }
  }
  printf("correctness: %s (%d limb mismatches)\n", bad ? "FAIL" : "ok", bad);
  for (int w : {2, 4, 8}) {
    for (int mode = 0; mode < 2; mode++) {
      int blocks = 256 * w, iters = 2000;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      if (mode == 0) kern<0><<<blocks, 256>>>(d_o0, iters, P, d_in); else kern<1><<<blocks, 256>>>(d_o0, iters, P, d_in);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (mode == 0) kern<0><<<blocks, 256>>>(d_o0, iters, P, d_in); else kern<1><<<blocks, 256>>>(d_o0, iters, P, d_in);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      double per = (double)blocks * 4 * iters / 1024.0 / (ms * 1e-3);
      printf("%s w/SIMD=%d  %.3f ms  %.1f cycles/mul/SIMD @2.4GHz\n", mode ? "9x29 unsaturated (hipcc)" : "8x32 asm schedule       ", w, ms, 2.4e9 / per);
    }
  }
  return 0;
}
