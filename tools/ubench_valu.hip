// Micro-benchmark: issue rate of the integer / fp64 VALU instructions that bound
// big-integer Montgomery arithmetic on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 ubench_valu.hip -o ubench_valu
// Prints cycles per wave-instruction per SIMD (assuming every SIMD busy) at several occupancies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

#define ITERS 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) kern(uint32_t* out, uint32_t seed, unsigned long long* clk) {
  uint32_t a = seed * (threadIdx.x + 1) | 1, b = seed ^ (blockIdx.x * 977u) | 3;
  unsigned long long r0 = a, r1 = b, r2 = a + 1, r3 = b + 1, r4 = a + 2, r5 = b + 2, r6 = a + 3, r7 = b + 3;
  uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  double d0 = a, d1 = b, d2 = a + 1, d3 = b + 1, d4 = a + 2, d5 = b + 2, d6 = a + 3, d7 = b + 3, da = 1.0000001, db = 0.5;
  uint32_t x0 = a, x1 = b, x2 = a + 1, x3 = b + 1, x4 = a + 2, x5 = b + 2, x6 = a + 3, x7 = b + 3;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; it++) {
    if (OP == 0) {        // v_mad_u64_u32, 8 independent accumulators, carry-out ignored
#define X(i) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(r##i) : "v"(a), "v"(b) : "s10", "s11");
      REP8(X)
#undef X
    } else if (OP == 1) { // v_mad_u64_u32 + v_addc consuming its carry (2 wait states padded by interleaving)
      asm volatile(
          "v_mad_u64_u32 %0, s[10:11], %16, %17, %0\n\t"
          "v_mad_u64_u32 %1, s[12:13], %16, %17, %1\n\t"
          "v_mad_u64_u32 %2, s[14:15], %16, %17, %2\n\t"
          "v_mad_u64_u32 %3, s[16:17], %16, %17, %3\n\t"
          "v_addc_co_u32 %8, s[10:11], 0, %8, s[10:11]\n\t"
          "v_addc_co_u32 %9, s[12:13], 0, %9, s[12:13]\n\t"
          "v_addc_co_u32 %10, s[14:15], 0, %10, s[14:15]\n\t"
          "v_addc_co_u32 %11, s[16:17], 0, %11, s[16:17]\n\t"
          "v_mad_u64_u32 %4, s[10:11], %16, %17, %4\n\t"
          "v_mad_u64_u32 %5, s[12:13], %16, %17, %5\n\t"
          "v_mad_u64_u32 %6, s[14:15], %16, %17, %6\n\t"
          "v_mad_u64_u32 %7, s[16:17], %16, %17, %7\n\t"
          "v_addc_co_u32 %12, s[10:11], 0, %12, s[10:11]\n\t"
          "v_addc_co_u32 %13, s[12:13], 0, %13, s[12:13]\n\t"
          "v_addc_co_u32 %14, s[14:15], 0, %14, s[14:15]\n\t"
          "v_addc_co_u32 %15, s[16:17], 0, %15, s[16:17]\n\t"
          : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7),
            "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7)
          : "v"(a), "v"(b) : "s10", "s11", "s12", "s13", "s14", "s15", "s16", "s17");
    } else if (OP == 2) { // v_mul_lo_u32
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x##i) : "v"(a));
      REP8(X)
#undef X
    } else if (OP == 3) { // v_mul_hi_u32
#define X(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x##i) : "v"(a));
      REP8(X)
#undef X
    } else if (OP == 4) { // v_add_u32 (full-rate reference)
#define X(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x##i) : "v"(a));
      REP8(X)
#undef X
    } else if (OP == 5) { // v_fma_f64
#define X(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##i) : "v"(da), "v"(db));
      REP8(X)
#undef X
    } else if (OP == 6) { // v_mad_u32_u24
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x##i) : "v"(a), "v"(b));
      REP8(X)
#undef X
    } else if (OP == 7) { // v_mul_hi_u32_u24
#define X(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x##i) : "v"(a));
      REP8(X)
#undef X
    } else if (OP == 8) { // dependent carry chain: add_co + 7 addc with the compiler's own hazard padding
      uint32_t y0 = x0, y1 = x1, y2 = x2, y3 = x3, y4 = x4, y5 = x5, y6 = x6, y7 = x7;
      unsigned __int128 lo = ((unsigned __int128)y1 << 32 | y0) | ((unsigned __int128)y3 << 96) | ((unsigned __int128)y2 << 64);
      unsigned __int128 hi = ((unsigned __int128)y5 << 32 | y4) | ((unsigned __int128)y7 << 96) | ((unsigned __int128)y6 << 64);
      unsigned __int128 al = ((unsigned __int128)b << 64) | a;
      lo += al; hi += al + (lo < al);
      x0 = (uint32_t)lo; x1 = (uint32_t)(lo >> 32); x2 = (uint32_t)(lo >> 64); x3 = (uint32_t)(lo >> 96);
      x4 = (uint32_t)hi; x5 = (uint32_t)(hi >> 32); x6 = (uint32_t)(hi >> 64); x7 = (uint32_t)(hi >> 96);
    } else if (OP == 9) { // v_mad_u64_u32 single dependent chain (latency)
      asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b) : "s10", "s11");
      asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b) : "s10", "s11");
      asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b) : "s10", "s11");
      asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b) : "s10", "s11");
      asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b) : "s10", "s11");
      asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b) : "s10", "s11");
      asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b) : "s10", "s11");
      asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(r0) : "v"(a), "v"(b) : "s10", "s11");
    } else if (OP == 10) { // v_mad_u64_u32 with 32-bit (zero-hi) addend pattern: mul_lo+mul_hi replacement check -> v_mul_lo + v_mul_hi pair
#define X(i) asm volatile("v_mul_lo_u32 %0, %1, %2\n\tv_mul_hi_u32 %0, %0, %2" : "+v"(x##i) : "v"(a), "v"(b));
      REP8(X)
#undef X
    } else if (OP == 11) { // 64-bit right shift
#define X(i) asm volatile("v_lshrrev_b64 %0, 29, %0" : "+v"(r##i));
      REP8(X)
#undef X
    } else if (OP == 12) { // 64-bit shift-add
#define X(i) asm volatile("v_lshl_add_u64 %0, %1, 22, %0" : "+v"(r##i) : "v"(r7));
      REP8(X)
#undef X
    } else if (OP == 13) { // funnel shift
#define X(i) asm volatile("v_alignbit_b32 %0, %0, %1, 29" : "+v"(x##i) : "v"(a));
      REP8(X)
#undef X
    } else if (OP == 14) { // 64-bit left shift
#define X(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(r##i));
      REP8(X)
#undef X
    } else if (OP == 15) { // v_add3_u32
#define X(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x##i) : "v"(a), "v"(b));
      REP8(X)
#undef X
    } else if (OP == 16) { // v_and_b32 with literal
#define X(i) asm volatile("v_and_b32 %0, 0x1fffffff, %0" : "+v"(x##i));
      REP8(X)
#undef X
    } else if (OP == 17) { // v_mad_u64_u32 with SGPR multiplier
#define X(i) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, s20, %0" : "+v"(r##i) : "v"(a) : "s10", "s11");
      REP8(X)
#undef X
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t acc = (uint32_t)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) ^ (uint32_t)((r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7) >> 32);
  acc ^= c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
  acc ^= (uint32_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

struct Op { const char* name; int ops_per_iter; };

template <int OP>
int run(const Op& op, int waves_per_simd, uint32_t* d_out, unsigned long long* d_clk) {
  int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = one per SIMD
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  kern<OP><<<blocks, 256>>>(d_out, 12345u, d_clk);  // warm-up
  CK(hipDeviceSynchronize());
  const int reps = 5;
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) kern<OP><<<blocks, 256>>>(d_out, 12345u + r, d_clk);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::vector<unsigned long long> clk(blocks);
  CK(hipMemcpy(clk.data(), d_clk, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double avgclk = 0; for (auto c : clk) avgclk += c; avgclk /= blocks;
  double wave_instr = (double)ITERS * op.ops_per_iter;              // per wave
  double waves = (double)blocks * 4;
  double instr_per_s = wave_instr * waves / (ms * 1e-3);
  double per_simd_per_s = instr_per_s / 1024.0;
  // s_memtime counts at a fixed 100 MHz-derived "shader clock"; report both
  printf("%-34s w/SIMD=%d  %.3f ms  %.2f Ginstr/s/SIMD  => %.2f cyc/instr/SIMD @2.4GHz  (in-kernel %.2f ticks/instr/wave)\n",
         op.name, waves_per_simd, ms, per_simd_per_s * 1e-9, 2.4e9 / per_simd_per_s, avgclk / wave_instr);
  return 0;
}

int main() {
  uint32_t* d_out; unsigned long long* d_clk;
  CK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(uint32_t)));
  CK(hipMalloc(&d_clk, 256 * 8 * sizeof(unsigned long long)));
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs=%d  clock=%d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  for (int w : {1, 2, 4, 8}) {
    run<0>({"v_mad_u64_u32 (8 indep)", 8}, w, d_out, d_clk);
    run<1>({"v_mad_u64_u32+v_addc (4-way ilv)", 16}, w, d_out, d_clk);
    run<2>({"v_mul_lo_u32", 8}, w, d_out, d_clk);
    run<3>({"v_mul_hi_u32", 8}, w, d_out, d_clk);
    run<4>({"v_add_u32", 8}, w, d_out, d_clk);
    run<5>({"v_fma_f64", 8}, w, d_out, d_clk);
    run<6>({"v_mad_u32_u24", 8}, w, d_out, d_clk);
    run<7>({"v_mul_hi_u32_u24", 8}, w, d_out, d_clk);
    run<8>({"256-bit add (compiler carry chain)", 8}, w, d_out, d_clk);
    run<9>({"v_mad_u64_u32 dependent chain", 8}, w, d_out, d_clk);
    run<10>({"v_mul_lo+v_mul_hi pair", 16}, w, d_out, d_clk);
    run<11>({"v_lshrrev_b64", 8}, w, d_out, d_clk);
    run<12>({"v_lshl_add_u64", 8}, w, d_out, d_clk);
    run<13>({"v_alignbit_b32", 8}, w, d_out, d_clk);
    run<14>({"v_lshlrev_b64", 8}, w, d_out, d_clk);
    run<15>({"v_add3_u32", 8}, w, d_out, d_clk);
    run<16>({"v_and_b32 literal", 8}, w, d_out, d_clk);
    run<17>({"v_mad_u64_u32 sgpr operand", 8}, w, d_out, d_clk);
  }
  return 0;
}
