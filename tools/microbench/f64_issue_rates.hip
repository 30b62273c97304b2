// Microbenchmark (round 4): how many cycles does a SIMD spend on one wave64 f64 instruction - v_fma_f64, v_mul_f64, v_add_f64 - and on the
// pairs a division by a launch constant is made of?  Written after the W&C pair kernel (wc_pair_totals_biallelic_kernel) ran at twice the
// time 4 cycles per instruction would give while every issue / wait counter said the VALU was the only thing busy.
// Every wave runs CHAINS independent dependency chains of ITER x UNROLL instructions in inline assembly (nothing for the compiler to fuse or
// drop); waves per SIMD is a launch parameter, so the latency of a dependent instruction separates from the issue rate:
//   ./f64_issue_rates            -> one JSON line per (instruction, chains, waves per SIMD): cycles per wave-instruction per SIMD
// Build: hipcc --offload-arch=gfx950 -O3 -o f64_issue_rates f64_issue_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { kFma = 0, kMul = 1, kAdd = 2, kFmaAsMul = 3, kFmaAsAdd = 4, kMixMulAdd = 5, kFma32 = 6, kFma3 = 7, kMul2 = 8, kAdd2 = 9, kFmaSgpr = 10 };

template <int OP, int CHAINS>
__global__ __launch_bounds__(256) void rate_kernel(int iters, double seed, double* __restrict__ sink) {
  double v[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) v[c] = seed + (double)(threadIdx.x + c) * 1e-9;
  const double m = 1.0000000001, a = 1e-12, one = 1.0, nzero = -0.0;
  float f[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) f[c] = (float)v[c];
  double w[2 * CHAINS];  // per-chain multipliers near 1 and addends near 0 in their own registers
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) { w[c] = 1.0 + 1e-10 * (double)(threadIdx.x % 7 + c); w[c + CHAINS] = 1e-12 * (double)(threadIdx.x % 5 + c); }
#pragma unroll
  for (int c = 0; c < 2 * CHAINS; ++c) asm volatile("" : "+v"(w[c]));
  const float mf = 1.0000001f, af = 1e-7f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        if (OP == kFma) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[c]) : "v"(m), "v"(a));
        if (OP == kMul) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[c]) : "v"(m));
        if (OP == kAdd) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(a));
        if (OP == kFmaAsMul) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[c]) : "v"(m), "v"(nzero));  // x * m, the same bits as v_mul_f64
        if (OP == kFmaAsAdd) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[c]) : "v"(one), "v"(a));    // x + a, the same bits as v_add_f64
        if (OP == kMixMulAdd) {
          if (u & 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[c]) : "v"(m));
          else asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(a));
        }
        if (OP == kFma32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(mf), "v"(af));
        // three / two DIFFERENT vector sources, none of them the destination's chain neighbour: what real arithmetic looks like
        if (OP == kFma3) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(v[c]) : "v"(w[c]), "v"(w[(c + 1) % CHAINS + CHAINS]));
        if (OP == kMul2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[c]) : "v"(w[c]));
        if (OP == kAdd2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(w[c + CHAINS]));
        if (OP == kFmaSgpr) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[c]) : "s"(m), "v"(w[c + CHAINS]));
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += v[c] + (double)f[c];
  if (s == 123.456) sink[0] = s;
}

template <int OP, int CHAINS>
void run(const char* name, int cus, double clock_ghz, double* sink) {
  const int iters = 4096;
  for (int waves_per_simd : {1, 2, 4, 8}) {
    const int blocks = cus * waves_per_simd;  // 256 threads = one wave per SIMD of a CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((rate_kernel<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, iters, 1.0, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CHECK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL((rate_kernel<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, iters, 1.0, sink);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    const double per_wave = (double)iters * 8 * CHAINS;          // instructions of one wave
    const double per_simd = per_wave * waves_per_simd;           // the waves of one SIMD run them one after another
    const double cycles = best * 1e-3 * clock_ghz * 1e9;
    printf("{\"instruction\": \"%s\", \"independent_chains\": %d, \"waves_per_simd\": %d, \"ms\": %.4f, \"cycles_per_wave_instruction_per_simd\": %.2f, \"clock_ghz_assumed\": %.2f}\n",
           name, CHAINS, waves_per_simd, best, cycles / per_simd, clock_ghz);
    fflush(stdout);
  }
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clock_ghz = argc > 1 ? atof(argv[1]) : prop.clockRate * 1e-6;
  double* sink;
  CHECK(hipMalloc(&sink, 8));
  run<kFma, 1>("v_fma_f64", cus, clock_ghz, sink);
  run<kFma, 4>("v_fma_f64", cus, clock_ghz, sink);
  run<kMul, 1>("v_mul_f64", cus, clock_ghz, sink);
  run<kMul, 4>("v_mul_f64", cus, clock_ghz, sink);
  run<kAdd, 1>("v_add_f64", cus, clock_ghz, sink);
  run<kAdd, 4>("v_add_f64", cus, clock_ghz, sink);
  run<kFmaAsMul, 4>("v_fma_f64 x, m, -0.0", cus, clock_ghz, sink);
  run<kFmaAsAdd, 4>("v_fma_f64 x, 1.0, a", cus, clock_ghz, sink);
  run<kMixMulAdd, 4>("v_add_f64 / v_mul_f64 alternating", cus, clock_ghz, sink);
  run<kFma32, 4>("v_fma_f32", cus, clock_ghz, sink);
  run<kFma3, 4>("v_fma_f64, three different VGPR pairs", cus, clock_ghz, sink);
  run<kMul2, 4>("v_mul_f64, two different VGPR pairs", cus, clock_ghz, sink);
  run<kAdd2, 4>("v_add_f64, two different VGPR pairs", cus, clock_ghz, sink);
  run<kFmaSgpr, 4>("v_fma_f64 v, s, v", cus, clock_ghz, sink);
  return 0;
}
