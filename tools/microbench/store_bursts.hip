// Microbenchmark: what do the per-site result tracks cost next to a 50 GB read stream, as a function of how the
// same bytes are written?  A persistent grid streams `rows` x `pitch` bytes (the sweep's read pattern: 16-lane groups,
// one row per group per step, 16-byte vectors) and writes `streams` output tracks of 8 bytes per row.
//   mode 0: no stores.   mode 1: per 64-row tile, 512 B per stream (what sweep_kernel does).
//   mode 2: the same bytes, but a wave writes only every 8th tile: 4 KiB contiguous per stream.
//   mode 3: every 64th tile: 32 KiB contiguous per stream.
//   mode 4: as mode 1 with ordinary (cached) stores instead of non-temporal ones.
//   mode 5: as mode 1, but the streams interleaved per row (one streams x 8-byte record per row: a tile writes ONE burst).
//   mode 6: a wave owns 8 consecutive tiles (as mode 2) but stores after every tile (spatial locality only, no deferral).
//   mode 7 / 8 / 9: the WORKGROUP owns 16 / 8 / 32 consecutive tiles (4 / 2 / 8 per wave), results staged in LDS, one cooperative flush per chunk:
//           8 / 4 / 16 KiB contiguous per stream written by the 256 threads with 16-byte stores (what a real kernel can afford: 56 B per row of LDS).
//   mode 10..17: mode 1's stores with the cache-policy bits spelled out: none, nt, sc0, sc1, sc0 sc1, sc0 nt, sc1 nt, sc0 sc1 nt.
// Build: hipcc --offload-arch=gfx950 -O3 -o store_bursts store_bursts.hip
// Run: ./store_bursts [rows [streams [pitch [workgroups per CU]]]]   (pitch 5008 = C4 as u8 rows, 640 = C4 as bit planes)
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
  constexpr int BURST = (MODE == 2 || MODE == 6) ? 8 : (MODE == 3 ? 64 : 1);
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
      if (MODE == 1 || MODE == 6) {
        for (int k = 0; k < streams; ++k) __builtin_nontemporal_store((double)acc, out + (size_t)k * rows + tile * 64 + lane);
      }
      if (MODE == 4) {
        for (int k = 0; k < streams; ++k) out[(size_t)k * rows + tile * 64 + lane] = (double)acc;
      }
      if (MODE == 5) {
        for (int k = 0; k < streams; ++k) __builtin_nontemporal_store((double)acc, out + (tile * 64) * streams + (size_t)k * 64 + lane);
      }
    }
    if (MODE == 2 || MODE == 3) {
      const size_t first = t0 * 64, count = (t0 + BURST <= ntiles ? (size_t)BURST : ntiles - t0) * 64;
      for (int k = 0; k < streams; ++k)
        for (size_t i = lane; i < count; i += 64) __builtin_nontemporal_store((double)acc, out + (size_t)k * rows + first + i);
    }
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// mode 18: a wave owns 8 consecutive tiles, defers its stores to the end of the chunk and writes them TILE-major (tile 0: all tracks, tile 1: ...) -
//          what a kernel that parks the chunk's counts in LDS and runs its epilogues at the end would do.  mode 19: the same deferral with round-robin
//          tiles (eight 512-byte pieces per track, 64 tiles apart): temporal clustering without contiguity.
template <int CONTIG>
__global__ __launch_bounds__(256) void deferred_kernel(const uint8_t* __restrict__ data, size_t pitch, size_t rows, int streams,
                                                       double* __restrict__ out, unsigned long long* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4, gl = lane & 15;
  const size_t ntiles = rows / 64, nvec = pitch / 16;
  unsigned acc = 0;
  const size_t wave_id = (size_t)blockIdx.x * 4 + wave, nwaves = (size_t)gridDim.x * 4;
  const size_t nchunks = (ntiles + 7) / 8;
  for (size_t chunk = wave_id; chunk < nchunks; chunk += nwaves) {
    size_t tiles[8];
    for (int b = 0; b < 8; ++b) tiles[b] = CONTIG ? chunk * 8 + b : (chunk / nwaves * 8 + b) * nwaves + wave_id;
    for (int b = 0; b < 8; ++b) {
      const size_t tile = tiles[b];
      if (tile >= ntiles) continue;
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
    }
    for (int b = 0; b < 8; ++b) {
      const size_t tile = tiles[b];
      if (tile >= ntiles) continue;
      for (int k = 0; k < streams; ++k) __builtin_nontemporal_store((double)acc, out + (size_t)k * rows + tile * 64 + lane);
    }
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// mode 20 / 21: modes 1 / 19 with the real sweep's dependency structure: every row's loads are waited for and reduced across lanes before the
//          next row's loads are issued (one row in flight per lane group), so the only difference between the two is WHEN a wave issues its stores.
template <int DEFER>
__global__ __launch_bounds__(256) void rowdrain_kernel(const uint8_t* __restrict__ data, size_t pitch, size_t rows, int streams,
                                                       double* __restrict__ out, unsigned long long* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4, gl = lane & 15;
  const size_t ntiles = rows / 64, nvec = pitch / 16;
  unsigned acc = 0;
  const size_t wave_id = (size_t)blockIdx.x * 4 + wave, nwaves = (size_t)gridDim.x * 4;
  constexpr int CH = DEFER ? 8 : 1;
  const size_t nchunks = (ntiles + CH - 1) / CH;
  for (size_t chunk = wave_id; chunk < nchunks; chunk += nwaves) {
    size_t tiles[CH];
    for (int b = 0; b < CH; ++b) tiles[b] = (chunk / nwaves * CH + b) * nwaves + wave_id;
    for (int b = 0; b < CH; ++b) {
      const size_t tile = tiles[b];
      if (tile >= ntiles) continue;
      for (int s = 0; s < 16; ++s) {
        const uint8_t* row = data + (tile * 64 + grp * 16 + s) * pitch;
        unsigned t = 0;
        for (size_t v = gl; v < nvec; v += 64) {
          uint4 g0 = *reinterpret_cast<const uint4*>(row + v * 16);
          uint4 g1 = v + 16 < nvec ? *reinterpret_cast<const uint4*>(row + (v + 16) * 16) : make_uint4(0, 0, 0, 0);
          uint4 g2 = v + 32 < nvec ? *reinterpret_cast<const uint4*>(row + (v + 32) * 16) : make_uint4(0, 0, 0, 0);
          t += __builtin_popcount(g0.x) + __builtin_popcount(g0.y) + __builtin_popcount(g0.z) + __builtin_popcount(g0.w) + __builtin_popcount(g1.x) + __builtin_popcount(g1.y) +
               __builtin_popcount(g1.z) + __builtin_popcount(g1.w) + __builtin_popcount(g2.x) + __builtin_popcount(g2.y) + __builtin_popcount(g2.z) + __builtin_popcount(g2.w);
        }
        for (int off = 1; off < 16; off <<= 1) t += __shfl_xor(t, off, 64);
        acc += t;
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!DEFER) for (int k = 0; k < streams; ++k) __builtin_nontemporal_store((double)acc, out + (size_t)k * rows + tile * 64 + lane);
    }
    if (DEFER)
      for (int b = 0; b < CH; ++b) {
        const size_t tile = tiles[b];
        if (tile >= ntiles) continue;
        for (int k = 0; k < streams; ++k) __builtin_nontemporal_store((double)acc, out + (size_t)k * rows + tile * 64 + lane);
      }
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// mode 1's stores with an explicit cache policy on the store instruction (gfx950: sc0 / sc1 / nt bits)
#define POLICY_STORE(SUFFIX) asm volatile("global_store_dwordx2 %0, %1, off " SUFFIX :: "v"(ptr), "v"(val) : "memory")
template <int POLICY>
__device__ __forceinline__ void policy_store(double* ptr, double val) {
  if constexpr (POLICY == 0) POLICY_STORE("");
  else if constexpr (POLICY == 1) POLICY_STORE("nt");
  else if constexpr (POLICY == 2) POLICY_STORE("sc0");
  else if constexpr (POLICY == 3) POLICY_STORE("sc1");
  else if constexpr (POLICY == 4) POLICY_STORE("sc0 sc1");
  else if constexpr (POLICY == 5) POLICY_STORE("sc0 nt");
  else if constexpr (POLICY == 6) POLICY_STORE("sc1 nt");
  else POLICY_STORE("sc0 sc1 nt");
}
template <int POLICY>
__global__ __launch_bounds__(256) void policy_kernel(const uint8_t* __restrict__ data, size_t pitch, size_t rows, int streams,
                                                     double* __restrict__ out, unsigned long long* __restrict__ sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4, gl = lane & 15;
  const size_t ntiles = rows / 64, nvec = pitch / 16;
  unsigned acc = 0;
  const size_t wave_id = (size_t)blockIdx.x * 4 + wave, nwaves = (size_t)gridDim.x * 4;
  for (size_t tile = wave_id; tile < ntiles; tile += nwaves) {
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
    for (int k = 0; k < streams; ++k) policy_store<POLICY>(out + (size_t)k * rows + tile * 64 + lane, (double)acc);
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <int TB>
__global__ __launch_bounds__(256) void staged_kernel(const uint8_t* __restrict__ data, size_t pitch, size_t rows, int streams,
                                                     double* __restrict__ out, unsigned long long* __restrict__ sink) {
  extern __shared__ __align__(16) double stage[];  // [stream][TB * 64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4, gl = lane & 15;
  const size_t ntiles = rows / 64, nvec = pitch / 16;
  constexpr int PER_WAVE = TB / 4;
  unsigned acc = 0;
  for (size_t c0 = (size_t)blockIdx.x * TB; c0 < ntiles; c0 += (size_t)gridDim.x * TB) {
    for (int b = 0; b < PER_WAVE; ++b) {
      const size_t tile = c0 + wave * PER_WAVE + b;
      if (tile < ntiles) {
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
      }
      for (int k = 0; k < streams; ++k) stage[(size_t)k * TB * 64 + (wave * PER_WAVE + b) * 64 + lane] = (double)acc;
    }
    __syncthreads();
    const size_t first = c0 * 64, count = (c0 + TB <= ntiles ? (size_t)TB : ntiles - c0) * 64;  // rows of this chunk
    for (int k = 0; k < streams; ++k)
      for (size_t i = threadIdx.x * 2; i < count; i += 512) {
        const double2 v = *reinterpret_cast<const double2*>(stage + (size_t)k * TB * 64 + i);
        __builtin_nontemporal_store(v.x, out + (size_t)k * rows + first + i);
        __builtin_nontemporal_store(v.y, out + (size_t)k * rows + first + i + 1);
      }
    __syncthreads();
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

int main(int argc, char** argv) {
  const size_t rows = argc > 1 ? atoll(argv[1]) : 10000000, pitch = argc > 3 ? atoll(argv[3]) : 5008;
  const int wg_per_cu = argc > 4 ? atoi(argv[4]) : 4;
  const int streams = argc > 2 ? atoi(argv[2]) : 7;  // 7 x 8 B = 56 B per row, the Hudson sweep's output volume
  uint8_t* data; double* out; unsigned long long* sink;
  CHECK(hipMalloc(&data, rows * pitch));
  CHECK(hipMemset(data, 1, rows * pitch));
  CHECK(hipMalloc(&out, (size_t)streams * rows * 8));
  CHECK(hipMalloc(&sink, 8));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount * wg_per_cu;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto run = [&](int mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CHECK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 3) hipLaunchKernelGGL(stream_kernel<3>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 5) hipLaunchKernelGGL(stream_kernel<5>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 6) hipLaunchKernelGGL(stream_kernel<6>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 7) hipLaunchKernelGGL(staged_kernel<16>, dim3(grid), dim3(256), (size_t)streams * 16 * 512, 0, data, pitch, rows, streams, out, sink);
      if (mode == 8) hipLaunchKernelGGL(staged_kernel<8>, dim3(grid), dim3(256), (size_t)streams * 8 * 512, 0, data, pitch, rows, streams, out, sink);
      if (mode == 9) hipLaunchKernelGGL(staged_kernel<32>, dim3(grid), dim3(256), (size_t)streams * 32 * 512, 0, data, pitch, rows, streams, out, sink);
      if (mode == 10) hipLaunchKernelGGL(policy_kernel<0>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 11) hipLaunchKernelGGL(policy_kernel<1>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 12) hipLaunchKernelGGL(policy_kernel<2>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 13) hipLaunchKernelGGL(policy_kernel<3>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 14) hipLaunchKernelGGL(policy_kernel<4>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 15) hipLaunchKernelGGL(policy_kernel<5>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 16) hipLaunchKernelGGL(policy_kernel<6>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 17) hipLaunchKernelGGL(policy_kernel<7>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 18) hipLaunchKernelGGL(deferred_kernel<1>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 19) hipLaunchKernelGGL(deferred_kernel<0>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 20) hipLaunchKernelGGL(rowdrain_kernel<0>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      if (mode == 21) hipLaunchKernelGGL(rowdrain_kernel<1>, dim3(grid), dim3(256), 0, 0, data, pitch, rows, streams, out, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)rows * pitch + (mode ? (double)streams * rows * 8 : 0.0);
    printf("{\"mode\": %d, \"rows\": %zu, \"streams\": %d, \"pitch\": %zu, \"workgroups_per_cu\": %d, \"best_ms\": %.3f, \"GBs\": %.0f}\n", mode, rows, streams, pitch, wg_per_cu, best, bytes / best / 1e6);
  };
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&staged_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int mode = 0; mode < 22; ++mode) run(mode);
  return 0;
}
