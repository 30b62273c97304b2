// host_pack.hpp — the host-side packer of upload.hip: the reference's u8 rows (stats.rs:250-331) and linear missing bitset
// (1298-1302) -> bit-plane rows.  Plain C++ + SSE2, no HIP: upload.hip includes it, and so does the sanitizer build
// (`make asan`, tests/test_sanitizers_cpu.py), which compiles it with g++ -fsanitize=address,undefined and checks it
// against a bit-by-bit restatement.
#pragma once

#include <emmintrin.h>

#include <cstddef>
#include <cstdint>
#include <cstring>

namespace fmh_host {

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
inline bool pack_rows_host(const uint8_t* data, const uint64_t* missing, size_t columns, size_t total_bits, size_t r0, size_t r1, int nplanes,
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

}  // namespace fmh_host
