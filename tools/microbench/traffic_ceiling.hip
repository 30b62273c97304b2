// Microbenchmark (round 4): what does HBM give a kernel with exactly a sweep's TRAFFIC and none of its arithmetic?  One wave per 64-row
// tile; the tile (64 x pitch contiguous bytes) is read as whole-KiB wave loads, all of them in flight before the first is consumed
// (the best case for the read side: flat, coalesced, nothing dependent), reduced to one dummy value, and the configuration's per-site
// tracks are written per tile the way site_epilogue writes them (64 lanes x 8 / 4 / 1 bytes per track, non-temporal).
//   ./traffic_ceiling rows pitch n_f64 n_u32 n_u8 [workgroups per CU (256 threads) ...]
//   C2 / C2x10 Hudson: pitch 128, 5 f64 + 4 u32 (+ fst: 6 f64)      C3 W&C: pitch 320, 14 f64 + 4 u32 + 7 u8      C3 summaries: pitch 320, 8 u32
//   C4 Hudson: pitch 640, 6 f64 + 4 u32
// Modes: read only / read + tracks / tracks only.  The table goes to DESIGN.md section 3 as the ceiling the real kernels are held against.
// Build: hipcc --offload-arch=gfx950 -O3 -o traffic_ceiling traffic_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// DEFER > 1: a wave reads DEFER of its tiles back to back and only then writes their tracks (what deferring the epilogues of a real kernel does to
// its traffic: the same bytes, bursts of DEFER x tracks stores per wave instead of `tracks` stores after every tile)
template <int NV, bool READ, bool WRITE, int DEFER = 1>
__global__ __launch_bounds__(256) void ceiling_kernel(const uint8_t* __restrict__ data, size_t rows, int n_f64, int n_u32, int n_u8,
                                                      double* __restrict__ o64, uint32_t* __restrict__ o32, uint8_t* __restrict__ o8,
                                                      unsigned long long* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t ntiles = rows / 64;
  const size_t wave_id = (size_t)blockIdx.x * 4 + wave, nwaves = (size_t)gridDim.x * 4;
  unsigned acc = 0;
  for (size_t tile0 = wave_id; tile0 < ntiles; tile0 += nwaves * DEFER) {
    if (READ) {
#pragma unroll 1
      for (int b = 0; b < DEFER; ++b) {
        const size_t tile = tile0 + (size_t)b * nwaves;
        if (tile >= ntiles) break;
        const uint8_t* base = data + tile * (size_t)(NV * 1024) + (size_t)lane * 16;
        uint4 x[NV];
#pragma unroll
        for (int c = 0; c < NV; ++c) x[c] = *reinterpret_cast<const uint4*>(base + (size_t)c * 1024);
#pragma unroll
        for (int c = 0; c < NV; ++c) acc += x[c].x ^ x[c].y ^ x[c].z ^ x[c].w;
      }
    }
    if (WRITE) {
#pragma unroll 1
      for (int b = 0; b < DEFER; ++b) {
        const size_t tile = tile0 + (size_t)b * nwaves;
        if (tile >= ntiles) break;
        const size_t site = tile * 64 + lane;
        for (int k = 0; k < n_u32; ++k) __builtin_nontemporal_store(acc, o32 + (size_t)k * rows + site);
        for (int k = 0; k < n_f64; ++k) __builtin_nontemporal_store((double)acc, o64 + (size_t)k * rows + site);
        for (int k = 0; k < n_u8; ++k) __builtin_nontemporal_store((uint8_t)acc, o8 + (size_t)k * rows + site);
      }
    }
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// Phase-aligned deferral: every wave of the chip looks at the same constant-rate clock (s_memrealtime, 100 MHz) and keeps reading tiles (parking
// their results) until that clock enters the write window of the current period, then writes everything it has parked - so that, without any
// barrier, most of the chip reads at the same time and writes at the same time.  period = 8 << shift ticks of 10 ns, write window = wnum / 8 of it.
template <int NV, int CAP>
__global__ __launch_bounds__(256) void aligned_kernel(const uint8_t* __restrict__ data, size_t rows, int n_f64, int n_u32, int n_u8, double* __restrict__ o64,
                                                      uint32_t* __restrict__ o32, uint8_t* __restrict__ o8, unsigned long long* __restrict__ sink, unsigned shift,
                                                      unsigned wnum) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t ntiles = rows / 64;
  const size_t wave_id = (size_t)blockIdx.x * 4 + wave, nwaves = (size_t)gridDim.x * 4;
  unsigned acc = 0;
  size_t tile = wave_id;
  while (tile < ntiles) {
    const size_t first = tile;
    int parked = 0;
    while (tile < ntiles && parked < CAP) {
      if (parked > 0 && ((unsigned)(__builtin_amdgcn_s_memrealtime() >> shift) & 7u) < wnum) break;
      const uint8_t* base = data + tile * (size_t)(NV * 1024) + (size_t)lane * 16;
      uint4 x[NV];
#pragma unroll
      for (int c = 0; c < NV; ++c) x[c] = *reinterpret_cast<const uint4*>(base + (size_t)c * 1024);
#pragma unroll
      for (int c = 0; c < NV; ++c) acc += x[c].x ^ x[c].y ^ x[c].z ^ x[c].w;
      tile += nwaves;
      ++parked;
    }
#pragma unroll 1
    for (int b = 0; b < parked; ++b) {
      const size_t site = (first + (size_t)b * nwaves) * 64 + lane;
      for (int k = 0; k < n_u32; ++k) __builtin_nontemporal_store(acc, o32 + (size_t)k * rows + site);
      for (int k = 0; k < n_f64; ++k) __builtin_nontemporal_store((double)acc, o64 + (size_t)k * rows + site);
      for (int k = 0; k < n_u8; ++k) __builtin_nontemporal_store((uint8_t)acc, o8 + (size_t)k * rows + site);
    }
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <int NV>
void run_all(const uint8_t* data, size_t rows, int n_f64, int n_u32, int n_u8, double* o64, uint32_t* o32, uint8_t* o8, unsigned long long* sink, int cus,
             int argc, char** argv) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const double rd = (double)rows * NV * 16, wr = (double)rows * (8.0 * n_f64 + 4.0 * n_u32 + n_u8);
  for (int a = 6; a < argc || a == 6; ++a) {
    const int per_cu = a < argc ? atoi(argv[a]) : 4;
    const int grid = cus * per_cu;
    for (int mode = 0; mode < 9; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 30; ++rep) {
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL((ceiling_kernel<NV, true, false>), dim3(grid), dim3(256), 0, 0, data, rows, n_f64, n_u32, n_u8, o64, o32, o8, sink);
        if (mode == 1) hipLaunchKernelGGL((ceiling_kernel<NV, true, true>), dim3(grid), dim3(256), 0, 0, data, rows, n_f64, n_u32, n_u8, o64, o32, o8, sink);
        if (mode == 2) hipLaunchKernelGGL((ceiling_kernel<NV, false, true>), dim3(grid), dim3(256), 0, 0, data, rows, n_f64, n_u32, n_u8, o64, o32, o8, sink);
        if (mode == 3) hipLaunchKernelGGL((ceiling_kernel<NV, true, true, 2>), dim3(grid), dim3(256), 0, 0, data, rows, n_f64, n_u32, n_u8, o64, o32, o8, sink);
        if (mode == 4) hipLaunchKernelGGL((ceiling_kernel<NV, true, true, 4>), dim3(grid), dim3(256), 0, 0, data, rows, n_f64, n_u32, n_u8, o64, o32, o8, sink);
        if (mode == 5) hipLaunchKernelGGL((ceiling_kernel<NV, true, true, 8>), dim3(grid), dim3(256), 0, 0, data, rows, n_f64, n_u32, n_u8, o64, o32, o8, sink);
        if (mode == 6) hipLaunchKernelGGL((ceiling_kernel<NV, true, true, 16>), dim3(grid), dim3(256), 0, 0, data, rows, n_f64, n_u32, n_u8, o64, o32, o8, sink);
        if (mode == 7) hipLaunchKernelGGL((ceiling_kernel<NV, true, true, 32>), dim3(grid), dim3(256), 0, 0, data, rows, n_f64, n_u32, n_u8, o64, o32, o8, sink);
        if (mode == 8) hipLaunchKernelGGL((ceiling_kernel<NV, true, true, 64>), dim3(grid), dim3(256), 0, 0, data, rows, n_f64, n_u32, n_u8, o64, o32, o8, sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 10 && ms < best) best = ms;
      }
      const double bytes = (mode != 2 ? rd : 0.0) + (mode != 0 ? wr : 0.0);
      printf("{\"mode\": \"%s\", \"rows\": %zu, \"pitch\": %d, \"f64_tracks\": %d, \"u32_tracks\": %d, \"u8_tracks\": %d, \"workgroups_per_cu\": %d, \"best_ms\": %.4f, \"GBs\": %.0f, "
             "\"frac_of_8TBs\": %.3f}\n",
             mode == 0 ? "read" : mode == 1 ? "read+tracks" : mode == 2 ? "tracks" : mode == 3 ? "read+tracks defer 2" : mode == 4 ? "read+tracks defer 4" : mode == 5 ? "read+tracks defer 8" : mode == 6 ? "read+tracks defer 16" : mode == 7 ? "read+tracks defer 32" : "read+tracks defer 64", rows, NV * 16, n_f64, n_u32, n_u8, per_cu, best, bytes / best / 1e6, bytes / best / 1e6 / 8000.0);
      fflush(stdout);
    }
    if (getenv("CEILING_ALIGNED")) {
      for (unsigned shift = 6; shift <= 11; ++shift)
        for (unsigned wnum = 2; wnum <= 3; ++wnum) {
          float best = 1e9f;
          for (int rep = 0; rep < 30; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL((aligned_kernel<NV, 64>), dim3(grid), dim3(256), 0, 0, data, rows, n_f64, n_u32, n_u8, o64, o32, o8, sink, shift, wnum);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 10 && ms < best) best = ms;
          }
          printf("{\"mode\": \"read+tracks clock-aligned\", \"period_us\": %.2f, \"write_window\": \"%u/8\", \"rows\": %zu, \"pitch\": %d, \"workgroups_per_cu\": %d, \"best_ms\": %.4f, "
                 "\"GBs\": %.0f, \"frac_of_8TBs\": %.3f}\n", (8u << shift) * 0.01, wnum, rows, NV * 16, per_cu, best, (rd + wr) / best / 1e6, (rd + wr) / best / 1e6 / 8000.0);
          fflush(stdout);
        }
    }
  }
}

int main(int argc, char** argv) {
  if (argc < 6) { printf("usage: traffic_ceiling rows pitch n_f64 n_u32 n_u8 [workgroups per CU ...]\n"); return 1; }
  const size_t rows = atoll(argv[1]) / 64 * 64, pitch = atoll(argv[2]);
  const int n_f64 = atoi(argv[3]), n_u32 = atoi(argv[4]), n_u8 = atoi(argv[5]);
  uint8_t* data; double* o64; uint32_t* o32; uint8_t* o8; unsigned long long* sink;
  CHECK(hipMalloc(&data, rows * pitch));
  CHECK(hipMemset(data, 1, rows * pitch));
  CHECK(hipMalloc(&o64, (size_t)(n_f64 + 1) * rows * 8));
  CHECK(hipMalloc(&o32, (size_t)(n_u32 + 1) * rows * 4));
  CHECK(hipMalloc(&o8, (size_t)(n_u8 + 1) * rows));
  CHECK(hipMalloc(&sink, 8));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
#define RUN(NV) run_all<NV>(data, rows, n_f64, n_u32, n_u8, o64, o32, o8, sink, cus, argc, argv)
  if (pitch == 128) RUN(8);
  else if (pitch == 320) RUN(20);
  else if (pitch == 640) RUN(40);
  else { printf("pitch must be 128, 320 or 640\n"); return 1; }
  return 0;
}
