// Calibration of rocprofv3 FETCH_SIZE for the access pattern of k_accum_l0 (MI355X_MICROARCH.md section HBM:
// "calibrate on a known byte count in your own access pattern before trusting an absolute").
// Kernel A: random 64-byte records of a 1 GiB table, 4 adjacent lanes x 16 B per record (the LDS-DMA gather).
// Kernel B: streaming 16 B/lane read of the same table (the guide's calibrated case: FETCH_SIZE = bytes / 2).
// Run under:  rocprofv3 --pmc FETCH_SIZE -- ./build/calib_gather   and compare with the printed byte counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

__global__ void __launch_bounds__(256) k_gather64(const uint4* __restrict__ table, uint32_t n_records, uint32_t iters, uint32_t* out) {
  uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t quad = t >> 2, piece = t & 3;
  uint32_t acc = 0;
  for (uint32_t it = 0; it < iters; it++) {
    uint32_t rec = (uint32_t)(mix((uint64_t)quad * 0x9E3779B97F4A7C15ull + it) % n_records);
    uint4 v = table[(size_t)rec * 4 + piece];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  out[t] = acc;
}

__global__ void __launch_bounds__(256) k_stream(const uint4* __restrict__ table, size_t n16, uint32_t* out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  for (size_t i = t; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = table[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  out[t] = acc;
}

int main() {
  const size_t bytes = 1ull << 30;
  uint4* table; uint32_t* out;
  CK(hipMalloc(&table, bytes)); CK(hipMalloc(&out, 4096 * 256 * 4));
  CK(hipMemset(table, 1, bytes));
  const uint32_t n_records = bytes / 64, iters = 64, blocks = 4096;
  k_gather64<<<blocks, 256>>>(table, n_records, iters, out);
  CK(hipDeviceSynchronize());
  k_stream<<<blocks, 256>>>(table, bytes / 16, out);
  CK(hipDeviceSynchronize());
  double gathered = (double)blocks * 256 / 4 * iters * 64;
  printf("k_gather64: %u records gathered, useful bytes = %.0f (%.3f GB); k_stream: bytes = %zu (%.3f GB)\n",
         (unsigned)(blocks * 256 / 4 * iters), gathered, gathered / 1e9, bytes, bytes / 1e9);
  return 0;
}
