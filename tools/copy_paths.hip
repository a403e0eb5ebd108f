// Host-to-device copy paths for a 32 MiB scalar vector (2^20 scalars), GB/s each: hipMemcpyAsync from pageable / registered /
// hipHostMalloc'ed memory, and a copy KERNEL that reads the page-locked memory through its device pointer.  Why page-locked host
// slices ran SLOWER than pageable ones on some boxes (profiles/r05_host_slices.md).
//   hipcc --offload-arch=gfx950 -O2 tools/copy_paths.hip -o build/copy_paths && build/copy_paths; HSA_ENABLE_SDMA=0 build/copy_paths
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t bytes = 32u << 20;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  void* d;
  CK(hipMalloc(&d, bytes));
  void* pageable = aligned_alloc(4096, bytes);
  void* registered = aligned_alloc(4096, bytes);
  void* pinned;
  memset(pageable, 1, bytes);
  memset(registered, 2, bytes);
  CK(hipHostRegister(registered, bytes, hipHostRegisterPortable));
  CK(hipHostMalloc(&pinned, bytes, hipHostMallocDefault));
  memset(pinned, 3, bytes);
  const char* sdma = getenv("HSA_ENABLE_SDMA");
  printf("HSA_ENABLE_SDMA=%s\n", sdma ? sdma : "(unset)");
  struct { const char* name; void* p; } srcs[] = {{"pageable", pageable}, {"registered", registered}, {"hipHostMalloc", pinned}};
  for (auto& s : srcs) {
    CK(hipMemcpyAsync(d, s.p, bytes, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    double t = now();
    for (int i = 0; i < 10; i++) CK(hipMemcpyAsync(d, s.p, bytes, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    double dt = (now() - t) / 10;
    printf("hipMemcpyAsync %-14s %.3f ms  %.1f GB/s\n", s.name, dt * 1e3, bytes / dt / 1e9);
  }
  for (int w = 0; w < 2; w++) {
    void* dp = nullptr;
    CK(hipHostGetDevicePointer(&dp, w ? pinned : registered, 0));
    for (int blocks : {32, 64, 128, 256, 512}) {
      hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, (const uint4*)dp, (uint4*)d, bytes / 16);
      CK(hipStreamSynchronize(st));
      double t = now();
      for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, st, (const uint4*)dp, (uint4*)d, bytes / 16);
      CK(hipStreamSynchronize(st));
      double dt = (now() - t) / 10;
      printf("copy kernel %-14s %3d workgroups  %.3f ms  %.1f GB/s\n", w ? "hipHostMalloc" : "registered", blocks, dt * 1e3, bytes / dt / 1e9);
    }
  }
  return 0;
}
