// Microbenchmark: what do the per-site result tracks cost next to a 50 GB read stream, as a function of how the
// same bytes are written?  A persistent grid streams `rows` x `pitch` bytes (the sweep's read pattern: 16-lane groups,
// one row per group per step, 16-byte vectors) and writes `streams` output tracks of 8 bytes per row.
//   mode 0: no stores.   mode 1: per 64-row tile, 512 B per stream (what sweep_kernel does).
//   mode 2: the same bytes, but a wave writes only every 8th tile: 4 KiB contiguous per stream.
//   mode 3: every 64th tile: 32 KiB contiguous per stream.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_bursts store_bursts.hip ; run: ./store_bursts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void stream_kernel(const uint8_t* __restrict__ data, size_t pitch, size_t rows, int streams,
                                                     double* __restrict__ out, unsigned long long* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4, gl = lane & 15;
  const size_t ntiles = rows / 64, nvec = pitch / 16;
  unsigned acc = 0;
  const size_t wave_id = (size_t)blockIdx.x * 4 + wave, nwaves = (size_t)gridDim.x * 4;
  constexpr int BURST = MODE == 2 ? 8 : (MODE == 3 ? 64 : 1);
  // a wave owns BURST consecutive tiles at a time so that its deferred stores are contiguous
  for (size_t t0 = wave_id * BURST; t0 < ntiles; t0 += nwaves * BURST) {
    for (int b = 0; b < BURST && t0 + b < ntiles; ++b) {
      const size_t tile = t0 + b;
      for (int s = 0; s < 16; ++s) {
        const uint8_t* row = data + (tile * 64 + grp * 16 + s) * pitch;
        for (size_t v = gl; v < nvec; v += 64) {
          uint4 g0 = *reinterpret_cast<const uint4*>(row + v * 16);
          uint4 g1 = v + 16 < nvec ? *reinterpret_cast<const uint4*>(row + (v + 16) * 16) : make_uint4(0, 0, 0, 0);
          uint4 g2 = v + 32 < nvec ? *reinterpret_cast<const uint4*>(row + (v + 32) * 16) : make_uint4(0, 0, 0, 0);
          uint4 g3 = v + 48 < nvec ? *reinterpret_cast<const uint4*>(row + (v + 48) * 16) : make_uint4(0, 0, 0, 0);
          acc += g0.x + g0.y + g0.z + g0.w + g1.x + g1.y + g1.z + g1.w + g2.x + g2.y + g2.z + g2.w + g3.x + g3.y + g3.z + g3.w;
        }
      }
      if (MODE == 1) {
        for (int k = 0; k < streams; ++k) __builtin_nontemporal_store((double)acc, out + (size_t)k * rows + tile * 64 + lane);
      }
    }
    if (MODE >= 2) {
      const size_t first = t0 * 64, count = (t0 + BURST <= ntiles ? (size_t)BURST : ntiles - t0) * 64;
      for (int k = 0; k < streams; ++k)
        for (size_t i = lane; i < count; i += 64) __builtin_nontemporal_store((double)acc, out + (size_t)k * rows + first + i);
    }
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main(int argc, char** argv) {
  const size_t rows = argc > 1 ? atoll(argv[1]) : 10000000, pitch = 5008;
  const int streams = argc > 2 ? atoi(argv[2]) : 7;  // 7 x 8 B = 56 B per row, the Hudson sweep's output volume
  uint8_t* data; double* out; unsigned long long* sink;
  CHECK(hipMalloc(&data, rows * pitch));
  CHECK(hipMemset(data, 1, rows * pitch));
  CHECK(hipMalloc(&out, (size_t)streams * rows * 8));
  CHECK(hipMalloc(&sink, 8));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount * 4;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto run = [&](int mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CHECK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 3) hipLaunchKernelGGL(stream_kernel<3>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)rows * pitch + (mode ? (double)streams * rows * 8 : 0.0);
    printf("{\"mode\": %d, \"rows\": %zu, \"streams\": %d, \"best_ms\": %.3f, \"GBs\": %.0f}\n", mode, rows, streams, best, bytes / best / 1e6);
  };
  for (int mode = 0; mode < 4; ++mode) run(mode);
  return 0;
}
