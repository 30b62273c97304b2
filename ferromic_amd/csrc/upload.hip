// upload.hip — host buffers -> the resident bit-plane image (fmh_matrix_create's packed route, fmh_matrix_create_packed).
//
// The boundary hands over the reference's host layout: one u8 per allele (stats.rs:250-331) and a linear missing bitset
// (1298-1302).  A cohort with alleles 0..7 is resident as one to three bit planes plus a called plane, so only
// ceil(H / 8) bytes per site and plane have to cross PCIe: the rows are packed ON THE HOST by a few threads (SSE2 movemask:
// 16 columns per instruction and plane, i.e. at memory bandwidth) into two pinned staging slabs, and the H2D copy of one slab
// runs on its own stream while the threads pack the next.  Round 1 shipped the u8 rows (pageable, slab by slab, synchronous)
// and packed on the device: 8x the bytes on the wire.  No kernel is involved; nothing here computes a statistic.
#include <hip/hip_runtime.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <thread>

#include "abi_internal.hpp"
#include "host_cpus.hpp"
#include "host_pack.hpp"

using namespace fmhi;

namespace {

constexpr size_t kStageBytes = (size_t)24 << 20;  // largest pinned slab (two per device, kept between calls and grown on demand: pinning the full
                                                   // 48 MB took 6 ms, which a process that only ever uploads a 1 MB cohort paid on its first statistic)

struct Staging {
  std::mutex mu;  // one upload at a time per device
  uint8_t* pinned[2] = {nullptr, nullptr};
  hipStream_t stream[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  size_t cap = 0;  // bytes per slab
  bool ready = false;
};
Staging g_staging[64];

// Called with s.mu held (the uploader's lock, which upload_release takes too): `ready` is only ever read or written under it, so a
// release cannot free the slabs between the check and their use.  A partial failure frees what was created.
int ensure_staging(Staging& s, size_t want) {
  want = std::min(kStageBytes, std::max<size_t>(want, (size_t)1 << 20));
  if (s.ready && s.cap >= want) return FMH_OK;
  if (s.ready) want = std::min(kStageBytes, std::max(want, 2 * s.cap));  // grow geometrically
  hipError_t e = hipSuccess;
  for (int k = 0; k < 2 && e == hipSuccess; ++k) {
    if (s.pinned[k]) { (void)hipHostFree(s.pinned[k]); s.pinned[k] = nullptr; }  // idle: the previous upload waited for its copies under this lock
    e = hipHostMalloc((void**)&s.pinned[k], want, hipHostMallocDefault);
    if (e == hipSuccess && !s.stream[k]) e = hipStreamCreateWithFlags(&s.stream[k], hipStreamNonBlocking);
    if (e == hipSuccess && !s.done[k]) e = hipEventCreateWithFlags(&s.done[k], hipEventDisableTiming);
  }
  if (e != hipSuccess) {
    for (int k = 0; k < 2; ++k) {
      if (s.pinned[k]) (void)hipHostFree(s.pinned[k]);
      if (s.stream[k]) (void)hipStreamDestroy(s.stream[k]);
      if (s.done[k]) (void)hipEventDestroy(s.done[k]);
      s.pinned[k] = nullptr; s.stream[k] = nullptr; s.done[k] = nullptr;
    }
    (void)hipGetLastError();
    s.ready = false;
    s.cap = 0;
    return fail(e == hipErrorNoDevice ? FMH_ERR_NO_DEVICE : FMH_ERR_HIP, "pinned staging for uploads: %s", hipGetErrorString(e));
  }
  s.ready = true;
  s.cap = want;
  return FMH_OK;
}

unsigned host_threads(size_t bytes) {
  if (bytes < ((size_t)2 << 20)) return 1;  // small matrices (run_vcf's many small regions): no thread start-up
  unsigned n = fmh_host::usable_cpus();
  if (const long long e = options().upload_threads.load(); e > 0) n = (unsigned)e;
  return std::max(1u, std::min(n, 16u));
}

}  // namespace

// Fills the (already allocated) planes of `m` from the reference host layout.  *overflow: a called entry exceeds max_allele.
int fmhi::upload_planes_from_bytes(fmh_matrix* m, const uint8_t* h_data, const uint64_t* h_missing, bool* overflow) {
  *overflow = false;
  if (m->variants == 0) return FMH_OK;
  FMH_TRY(use_device(m->device));
  if (m->device < 0 || m->device >= 64) return fail(FMH_ERR_INVALID, "device index %d unsupported", m->device);
  Staging* st = &g_staging[m->device];
  std::lock_guard<std::mutex> lock(st->mu);
  const int nplanes = m->p2 ? 3 : (m->p1 ? 2 : 1);
  const bool with_called = m->pc != nullptr;
  const size_t pitch = m->plane_pitch, per_row = (size_t)(nplanes + (with_called ? 1 : 0)) * pitch;
  FMH_TRY(ensure_staging(*st, std::max(per_row, m->variants * per_row)));
  const size_t slab_rows = std::max<size_t>(1, std::min(m->variants, st->cap / per_row));
  const size_t total_bits = m->variants * (size_t)m->columns;
  const unsigned T = host_threads(m->variants * (size_t)m->columns);
  uint8_t* d_planes[4] = {m->p0, m->p1, m->p2, m->pc};
  std::atomic<bool> any_overflow{false};
  bool used[2] = {false, false};
  size_t k = 0;
  for (size_t r0 = 0; r0 < m->variants; r0 += slab_rows, ++k) {
    const int b = (int)(k & 1);
    const size_t rows = std::min(slab_rows, m->variants - r0);
    if (used[b]) HIP_TRY(hipEventSynchronize(st->done[b]));  // the copy that last read this staging slab has finished
    uint8_t* dst = st->pinned[b];
    auto work = [&](unsigned t, unsigned n) {
      const size_t a = r0 + rows * t / n, e = r0 + rows * (t + 1) / n;
      if (fmh_host::pack_rows_host(h_data, h_missing, m->columns, total_bits, a, e, nplanes, with_called, dst, r0, rows, pitch)) any_overflow = true;
    };
    const unsigned n = (unsigned)std::min<size_t>(T, rows);
    if (n <= 1) {
      work(0, 1);
    } else {
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < n; ++t) pool.emplace_back(work, t, n);
      work(0, n);
      for (auto& th : pool) th.join();
    }
    for (int p = 0; p < nplanes; ++p)
      HIP_TRY(hipMemcpyAsync(d_planes[p] + r0 * pitch, dst + (size_t)p * rows * pitch, rows * pitch, hipMemcpyHostToDevice, st->stream[b]));
    if (with_called)
      HIP_TRY(hipMemcpyAsync(m->pc + r0 * pitch, dst + (size_t)nplanes * rows * pitch, rows * pitch, hipMemcpyHostToDevice, st->stream[b]));
    HIP_TRY(hipEventRecord(st->done[b], st->stream[b]));
    used[b] = true;
  }
  for (int b = 0; b < 2; ++b)
    if (used[b]) HIP_TRY(hipEventSynchronize(st->done[b]));
  *overflow = any_overflow.load();
  return FMH_OK;
}

// Host planes in the device's own bit order -> device planes (pitched copies, no packing).
int fmhi::upload_planes_from_planes(fmh_matrix* m, const uint8_t* const h_planes[4], size_t h_pitch) {
  if (m->variants == 0) return FMH_OK;
  FMH_TRY(use_device(m->device));
  uint8_t* d_planes[4] = {m->p0, m->p1, m->p2, m->pc};
  const size_t row_bytes = ((size_t)m->columns + 7) / 8;
  // On the calling thread's own stream, not the legacy default stream: that one is ordered against every blocking stream of the device, so
  // each region worker of run_vcf waited here for the sweeps of all the others (and held them up meanwhile).  The planes are a fresh
  // allocation: nothing can be pending on them.
  hipStream_t st = hipStreamPerThread;
  for (int p = 0; p < 4; ++p) {
    if (!d_planes[p]) continue;
    if (h_pitch == m->plane_pitch) {
      // the caller's rows have the device's pitch (their bytes past the last column are zero, include/ferromic_hip.h): ONE copy of the whole
      // plane.  A pitched copy is a descriptor per row: a million 125-byte rows took seconds.
      HIP_TRY(hipMemcpyAsync(d_planes[p], h_planes[p], m->variants * m->plane_pitch, hipMemcpyHostToDevice, st));
      continue;
    }
    HIP_TRY(hipMemsetAsync(d_planes[p], 0, m->variants * m->plane_pitch, st));  // padding bytes stay zero
    HIP_TRY(hipMemcpy2DAsync(d_planes[p], m->plane_pitch, h_planes[p], h_pitch, row_bytes, m->variants, hipMemcpyHostToDevice, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  return FMH_OK;
}

void fmhi::upload_release(int device) {
  if (device < 0 || device >= 64) return;
  Staging& s = g_staging[device];
  std::lock_guard<std::mutex> lock(s.mu);
  if (!s.ready) return;
  for (int k = 0; k < 2; ++k) {
    (void)hipHostFree(s.pinned[k]);
    (void)hipStreamDestroy(s.stream[k]);
    (void)hipEventDestroy(s.done[k]);
    s.pinned[k] = nullptr; s.stream[k] = nullptr; s.done[k] = nullptr;
  }
  s.ready = false;
  s.cap = 0;
}
