// upload.hip — host buffers -> the resident bit-plane image (fmh_matrix_create's packed route, fmh_matrix_create_packed).
//
// The boundary hands over the reference's host layout: one u8 per allele (stats.rs:250-331) and a linear missing bitset
// (1298-1302).  A cohort with alleles 0..7 is resident as one to three bit planes plus a called plane, so only
// ceil(H / 8) bytes per site and plane have to cross PCIe: the rows are packed ON THE HOST by a few threads (SSE2 movemask:
// 16 columns per instruction and plane, i.e. at memory bandwidth) into two pinned staging slabs, and the H2D copy of one slab
// runs on its own stream while the threads pack the next.  Round 1 shipped the u8 rows (pageable, slab by slab, synchronous)
// and packed on the device: 8x the bytes on the wire.  No kernel is involved; nothing here computes a statistic.
#include <hip/hip_runtime.h>

#include <emmintrin.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <thread>

#include "abi_internal.hpp"
#include "host_cpus.hpp"

using namespace fmhi;

namespace {

constexpr size_t kStageBytes = (size_t)24 << 20;  // per pinned slab (two per device, kept between calls)

struct Staging {
  std::mutex mu;  // one upload at a time per device
  uint8_t* pinned[2] = {nullptr, nullptr};
  hipStream_t stream[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  bool ready = false;
};
Staging g_staging[64];

std::mutex g_staging_init;

int staging(int device, Staging** out) {
  if (device < 0 || device >= 64) return fail(FMH_ERR_INVALID, "device index %d unsupported", device);
  Staging& s = g_staging[device];
  std::lock_guard<std::mutex> init(g_staging_init);
  if (!s.ready) {
    for (int k = 0; k < 2; ++k) {
      HIP_TRY(hipHostMalloc((void**)&s.pinned[k], kStageBytes, hipHostMallocDefault));
      HIP_TRY(hipStreamCreateWithFlags(&s.stream[k], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&s.done[k], hipEventDisableTiming));
    }
    s.ready = true;
  }
  *out = &s;
  return FMH_OK;
}

unsigned host_threads(size_t bytes) {
  if (bytes < ((size_t)2 << 20)) return 1;  // small matrices (run_vcf's many small regions): no thread start-up
  unsigned n = fmh_host::usable_cpus();
  if (const char* e = getenv("FMH_UPLOAD_THREADS")) n = (unsigned)atoi(e);
  return std::max(1u, std::min(n, 16u));
}

// 8 bits of the missing bitset starting at bit `b` (LSB-first u64 words, stats.rs:1298-1302); bits past `total` read as 0
inline uint32_t missing_bits16(const uint64_t* words, size_t b, size_t total) {
  if (b >= total) return 0;
  const size_t w = b >> 6, sh = b & 63, last = (total - 1) >> 6;
  uint64_t v = words[w] >> sh;
  if (sh > 48 && w < last) v |= words[w + 1] << (64 - sh);
  const size_t left = total - b;
  return (uint32_t)(left >= 16 ? (v & 0xFFFF) : (v & ((1ull << left) - 1)));
}

// rows [r0, r1) of the host matrix -> plane rows in `dst` (plane k at dst + k * rows_in_slab * pitch); returns true when a
// CALLED entry carries a bit above the planes (a max_allele below the data)
bool pack_rows_host(const uint8_t* data, const uint64_t* missing, size_t columns, size_t total_bits, size_t r0, size_t r1, int nplanes,
                    bool with_called, uint8_t* dst, size_t slab_row0, size_t slab_rows, size_t pitch) {
  const __m128i himask = _mm_set1_epi8((char)(nplanes >= 3 ? 0xF8 : (nplanes == 2 ? 0xFC : 0xFE)));
  const __m128i zero = _mm_setzero_si128();
  bool overflow = false;
  uint8_t* planes[4] = {dst, dst + slab_rows * pitch, dst + 2 * slab_rows * pitch, dst + (size_t)nplanes * slab_rows * pitch};  // [nplanes] = called
  for (size_t r = r0; r < r1; ++r) {
    const uint8_t* row = data + r * columns;
    const size_t o = (r - slab_row0) * pitch;
    for (int k = 0; k < nplanes; ++k) memset(planes[k] + o, 0, pitch);
    if (with_called) memset(planes[3] + o, 0, pitch);
    for (size_t c = 0; c < columns; c += 16) {
      __m128i v;
      if (c + 16 <= columns) {
        v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(row + c));
      } else {  // ragged tail
        alignas(16) uint8_t tmp[16] = {0};
        memcpy(tmp, row + c, columns - c);
        v = _mm_load_si128(reinterpret_cast<const __m128i*>(tmp));
      }
      uint32_t called = c + 16 <= columns ? 0xFFFFu : ((1u << (columns - c)) - 1u);
      if (missing) called &= ~missing_bits16(missing, r * columns + c, total_bits);
      const uint32_t high = 0xFFFFu ^ (uint32_t)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_and_si128(v, himask), zero));
      overflow |= (high & called) != 0;
      const uint16_t b0 = (uint16_t)_mm_movemask_epi8(_mm_slli_epi16(v, 7));
      memcpy(planes[0] + o + (c >> 3), &b0, 2);
      if (nplanes >= 2) { const uint16_t b1 = (uint16_t)_mm_movemask_epi8(_mm_slli_epi16(v, 6)); memcpy(planes[1] + o + (c >> 3), &b1, 2); }
      if (nplanes >= 3) { const uint16_t b2 = (uint16_t)_mm_movemask_epi8(_mm_slli_epi16(v, 5)); memcpy(planes[2] + o + (c >> 3), &b2, 2); }
      if (with_called) { const uint16_t bc = (uint16_t)called; memcpy(planes[3] + o + (c >> 3), &bc, 2); }
    }
  }
  return overflow;
}

}  // namespace

// Fills the (already allocated) planes of `m` from the reference host layout.  *overflow: a called entry exceeds max_allele.
int fmhi::upload_planes_from_bytes(fmh_matrix* m, const uint8_t* h_data, const uint64_t* h_missing, bool* overflow) {
  *overflow = false;
  if (m->variants == 0) return FMH_OK;
  FMH_TRY(use_device(m->device));
  Staging* st = nullptr;
  FMH_TRY(staging(m->device, &st));
  std::lock_guard<std::mutex> lock(st->mu);
  const int nplanes = m->p2 ? 3 : (m->p1 ? 2 : 1);
  const bool with_called = m->pc != nullptr;
  const size_t pitch = m->plane_pitch, per_row = (size_t)(nplanes + (with_called ? 1 : 0)) * pitch;
  const size_t slab_rows = std::max<size_t>(1, std::min(m->variants, kStageBytes / per_row));
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
      if (pack_rows_host(h_data, h_missing, m->columns, total_bits, a, e, nplanes, with_called, dst, r0, rows, pitch)) any_overflow = true;
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
  for (int p = 0; p < 4; ++p) {
    if (!d_planes[p]) continue;
    if (h_pitch != m->plane_pitch || row_bytes != m->plane_pitch) HIP_TRY(hipMemsetAsync(d_planes[p], 0, m->variants * m->plane_pitch, nullptr));  // padding bytes stay zero
    HIP_TRY(hipMemcpy2DAsync(d_planes[p], m->plane_pitch, h_planes[p], h_pitch, row_bytes, m->variants, hipMemcpyHostToDevice, nullptr));
  }
  HIP_TRY(hipStreamSynchronize(nullptr));
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
    s.pinned[k] = nullptr;
  }
  s.ready = false;
}
