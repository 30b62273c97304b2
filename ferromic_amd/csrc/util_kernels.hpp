// util_kernels.hpp — the non-template kernels of libferromic_hip.so: partials finalisation, layout conversion, the
// synthetic-cohort generator, the bit-plane packer.  Included by abi.hip only (one definition per library).
#pragma once

#include "sweep_kernels.hpp"

namespace fmh {

// Sum the per-block partials into one vector each.  One workgroup per slot (128 slots); thread t adds
// blocks t, t+256, ... in ascending order and the 256 thread sums are combined by a fixed LDS tree,
// so the result is deterministic for a given grid size.
__global__ __launch_bounds__(256) void finalize_kernel(const double* __restrict__ part_f64,
                                                       const unsigned long long* __restrict__ part_u64,
                                                       int nblocks, double* __restrict__ out_f64,
                                                       unsigned long long* __restrict__ out_u64) {
  __shared__ double s_f[256];
  __shared__ unsigned long long s_u[256];
  const int slot = blockIdx.x;
  const int t = threadIdx.x;
  if (slot < kMaxF64) {
    double v = 0.0;
    for (int b = t; b < nblocks; b += 256) v += part_f64[(size_t)b * kMaxF64 + slot];
    s_f[t] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (t < w) s_f[t] += s_f[t + w];
      __syncthreads();
    }
    if (t == 0) out_f64[slot] = s_f[0];
  } else {
    const int j = slot - kMaxF64;
    unsigned long long v = 0;
    for (int b = t; b < nblocks; b += 256) v += part_u64[(size_t)b * kMaxU64 + j];
    s_u[t] = v;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (t < w) s_u[t] += s_u[t + w];
      __syncthreads();
    }
    if (t == 0) out_u64[j] = s_u[0];
  }
}

// ------------------------------------------------------------------------------------------------
// layout / generator / utility kernels
// ------------------------------------------------------------------------------------------------

// reference missing bitset (bit per linear entry, set = missing) -> called bit-rows
__global__ void missing_to_called_rows(const unsigned long long* __restrict__ missing, size_t variants,
                                       uint32_t columns, uint8_t* __restrict__ bits, size_t bits_pitch) {
  const size_t total = variants * bits_pitch;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t s = i / bits_pitch;
    const uint32_t j = (uint32_t)(i - s * bits_pitch);
    uint32_t out = 0;
    for (int b = 0; b < 8; ++b) {
      const uint32_t h = j * 8 + b;
      if (h < columns) {
        const size_t idx = s * columns + h;
        const uint32_t miss = (uint32_t)((missing[idx >> 6] >> (idx & 63)) & 1ull);
        out |= (miss ^ 1u) << b;
      }
    }
    bits[i] = (uint8_t)out;
  }
}

// called bit-rows -> reference missing bitset (one thread per output word)
__global__ void called_rows_to_missing(const uint8_t* __restrict__ bits, size_t bits_pitch, size_t variants,
                                       uint32_t columns, unsigned long long* __restrict__ missing, size_t words) {
  for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += (size_t)gridDim.x * blockDim.x) {
    unsigned long long out = 0;
    for (int b = 0; b < 64; ++b) {
      const size_t idx = w * 64 + b;
      if (idx < variants * (size_t)columns) {
        const size_t s = idx / columns;
        const uint32_t h = (uint32_t)(idx - s * columns);
        const uint32_t called = (bits[s * bits_pitch + (h >> 3)] >> (h & 7)) & 1u;
        out |= (unsigned long long)(called ^ 1u) << b;
      }
    }
    missing[w] = out;
  }
}

// splitmix64 finaliser: the counter-based stream shared with oracle/dense_oracle.c
__host__ __device__ __forceinline__ uint32_t hash24(uint64_t seed, uint64_t site, uint64_t column) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (site * 0x100000001B3ull + column + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 40);
}

// one thread writes 16 columns (one vector) of one site
__global__ void generate_kernel(uint8_t* __restrict__ data, size_t pitch, uint8_t* __restrict__ bits,
                                size_t bits_pitch, size_t variants, uint32_t columns, uint32_t nvec,
                                uint64_t seed, uint64_t first_site, const uint32_t* __restrict__ thresholds,
                                const uint8_t* __restrict__ pop_of_column, uint32_t missing_thr) {
  const size_t total = variants * (size_t)(pitch / 16);
  const uint32_t vec_per_row = (uint32_t)(pitch / 16);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t s = i / vec_per_row;
    const uint32_t v = (uint32_t)(i - s * vec_per_row);
    uint32_t w[4] = {0, 0, 0, 0};
    uint32_t called = 0;
    if (v < nvec) {
      for (int b = 0; b < 16; ++b) {
        const uint32_t h = v * 16 + b;
        if (h < columns) {
          const uint32_t thr = thresholds[(size_t)pop_of_column[h] * variants + s];
          const uint32_t bit = hash24(seed, first_site + s, h) < thr ? 1u : 0u;
          bool miss = false;
          if (bits) miss = hash24(seed ^ 0xA5A5A5A5DEADBEEFull, first_site + s, h) < missing_thr;
          if (!miss) { w[b >> 2] |= bit << ((b & 3) * 8); called |= 1u << b; }
        }
      }
    }
    *reinterpret_cast<uint4*>(data + s * pitch + (size_t)v * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    if (bits && (size_t)v * 2 + 1 < bits_pitch) *reinterpret_cast<uint16_t*>(bits + s * bits_pitch + (size_t)v * 2) = (uint16_t)called;
  }
}

// ---- bit-packed image (fmh_matrix_pack) ---------------------------------------------------------------------------
// bytes -> planes: one thread per (row, 32 columns).  Bit c of a plane word = column 32 w + c; bits past the last column
// are zero whatever the padding bytes hold.  p1 / p2 / pc may be null (alleles <= 1 / <= 3 / nothing missing).
__device__ __forceinline__ uint32_t pack_nibble(uint32_t w, int shift) {  // bit `shift` of each of 4 bytes -> 4 bits, LSB = byte 0
  return ((((w >> shift) & 0x01010101u) * 0x01020408u) >> 24) & 0xFu;
}
__global__ void pack_rows_kernel(const uint8_t* __restrict__ data, size_t pitch, const uint8_t* __restrict__ bits, size_t bits_pitch,
                                 size_t rows, uint32_t columns, uint8_t* __restrict__ p0, uint8_t* __restrict__ p1, uint8_t* __restrict__ p2,
                                 uint8_t* __restrict__ pc, size_t plane_pitch, unsigned int* __restrict__ overflow_flag) {
  const size_t words = plane_pitch / 4, total = rows * words;
  // allele bits the planes do not store: a CALLED entry that carries one was handed over with a max_allele below the data
  const uint32_t himask = p2 ? 0xF8F8F8F8u : (p1 ? 0xFCFCFCFCu : 0xFEFEFEFEu);
  bool overflow = false;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t row = idx / words;
    const uint32_t w = (uint32_t)(idx - row * words), col0 = w * 32;
    uint32_t b0 = 0, b1 = 0, b2 = 0, bc = 0;
    if (col0 < columns) {
      const uint32_t valid = columns - col0 >= 32 ? 0xFFFFFFFFu : ((1u << (columns - col0)) - 1u);
      uint32_t live = valid;  // called entries inside the row
      if (bits) {
        const uint8_t* cb = bits + row * bits_pitch + (size_t)w * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if ((size_t)w * 4 + k < bits_pitch) bc |= (uint32_t)cb[k] << (8 * k);
        bc &= valid;
        live = bc;
      }
      const uint8_t* src = data + row * pitch + col0;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (col0 + 16 * q < columns) {  // pitch is a multiple of 16 and >= columns: the 16-byte piece is inside the row
          const uint4 g = *reinterpret_cast<const uint4*>(src + 16 * q);
          const uint32_t d[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            b0 |= pack_nibble(d[k], 0) << (16 * q + 4 * k);
            b1 |= pack_nibble(d[k], 1) << (16 * q + 4 * k);
            b2 |= pack_nibble(d[k], 2) << (16 * q + 4 * k);
            overflow |= (d[k] & himask & (nib_to_bytes(live >> (16 * q + 4 * k)) * 0xFFu)) != 0;
          }
        }
      }
      b0 &= valid;
      b1 &= valid;
      b2 &= valid;
    }
    *reinterpret_cast<uint32_t*>(p0 + row * plane_pitch + (size_t)w * 4) = b0;
    if (p1) *reinterpret_cast<uint32_t*>(p1 + row * plane_pitch + (size_t)w * 4) = b1;
    if (p2) *reinterpret_cast<uint32_t*>(p2 + row * plane_pitch + (size_t)w * 4) = b2;
    if (pc && bits) *reinterpret_cast<uint32_t*>(pc + row * plane_pitch + (size_t)w * 4) = bc;
  }
  if (overflow && overflow_flag) atomicOr(overflow_flag, 1u);
}

// row_gap[r] = 1 when some column of row r of a packed matrix with a called plane is NOT called (the plane's bits beyond the row are zero, so:
// fewer set bits than columns), else 0: the sweeps read the called plane of those rows only (MatrixView::row_gap).
__global__ __launch_bounds__(256) void row_gap_kernel(const uint8_t* __restrict__ pc, size_t plane_pitch, size_t rows, uint32_t columns,
                                                      uint8_t* __restrict__ row_gap) {
  const size_t group = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 16, groups = (size_t)gridDim.x * blockDim.x / 16;
  const uint32_t gl = threadIdx.x % 16, nvec = (uint32_t)(plane_pitch / 16);
  for (size_t r = group; r < rows; r += groups) {
    uint32_t called = 0;
    for (uint32_t v = gl; v < nvec; v += 16) {
      const uint4 a = *reinterpret_cast<const uint4*>(pc + r * plane_pitch + (size_t)v * 16);
      called += __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w);
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) called += __shfl_xor(called, off, 64);
    if (gl == 0) row_gap[r] = called != columns ? 1 : 0;
  }
}

// row_hi[r] = 1 when row r of a packed multi-allelic matrix has a bit in plane 1 or plane 2 (a called allele above 1), else 0: the sweeps
// read the upper planes of those rows only (MatrixView::row_hi).  Sixteen lanes per row, as many 16-byte vectors each as the row needs.
__global__ __launch_bounds__(256) void row_hi_kernel(const uint8_t* __restrict__ p1, const uint8_t* __restrict__ p2, size_t plane_pitch, size_t rows,
                                                     uint8_t* __restrict__ row_hi) {
  const size_t group = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 16, groups = (size_t)gridDim.x * blockDim.x / 16;
  const uint32_t gl = threadIdx.x % 16, nvec = (uint32_t)(plane_pitch / 16);
  for (size_t r = group; r < rows; r += groups) {
    uint32_t any = 0;
    for (uint32_t v = gl; v < nvec; v += 16) {
      const uint4 a = *reinterpret_cast<const uint4*>(p1 + r * plane_pitch + (size_t)v * 16);
      any |= a.x | a.y | a.z | a.w;
      if (p2) {
        const uint4 b = *reinterpret_cast<const uint4*>(p2 + r * plane_pitch + (size_t)v * 16);
        any |= b.x | b.y | b.z | b.w;
      }
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) any |= __shfl_xor(any, off, 64);
    if (gl == 0) row_hi[r] = any != 0 ? 1 : 0;
  }
}

// planes -> bytes: one thread per (row, 16 columns); padding columns come out zero
// (pc = called plane or null: the value bits of a missing entry are masked out, so that a download does not depend on which upload route
// built the matrix - fmh_matrix_create_packed takes the caller's planes as they are)
__global__ void unpack_rows_kernel(const uint8_t* __restrict__ p0, const uint8_t* __restrict__ p1, const uint8_t* __restrict__ p2, const uint8_t* __restrict__ pc,
                                   size_t plane_pitch, size_t rows, uint8_t* __restrict__ data, size_t pitch) {
  const size_t vecs = pitch / 16, total = rows * vecs;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const size_t row = idx / vecs;
    const uint32_t v = (uint32_t)(idx - row * vecs);
    uint32_t lo = 0, hi = 0, top = 0;
    if ((size_t)v * 2 + 2 <= plane_pitch) {
      lo = *reinterpret_cast<const uint16_t*>(p0 + row * plane_pitch + (size_t)v * 2);
      if (p1) hi = *reinterpret_cast<const uint16_t*>(p1 + row * plane_pitch + (size_t)v * 2);
      if (p2) top = *reinterpret_cast<const uint16_t*>(p2 + row * plane_pitch + (size_t)v * 2);
      if (pc) {
        const uint32_t c = *reinterpret_cast<const uint16_t*>(pc + row * plane_pitch + (size_t)v * 2);
        lo &= c; hi &= c; top &= c;
      }
    }
    uint4 a = called_bytes(lo);
    if (p1) {
      const uint4 b = called_bytes(hi);
      a.x |= b.x << 1; a.y |= b.y << 1; a.z |= b.z << 1; a.w |= b.w << 1;
    }
    if (p2) {
      const uint4 b = called_bytes(top);
      a.x |= b.x << 2; a.y |= b.y << 2; a.z |= b.z << 2; a.w |= b.w << 2;
    }
    *reinterpret_cast<uint4*>(data + row * pitch + (size_t)v * 16) = a;
  }
}

// largest called allele of a packed image: the largest bit pattern any column shows (exact: 7 needs all three plane bits in ONE column)
__global__ void packed_max_allele_kernel(const uint8_t* __restrict__ p0, const uint8_t* __restrict__ p1, const uint8_t* __restrict__ p2,
                                         const uint8_t* __restrict__ pc, size_t plane_pitch, size_t rows, unsigned int* __restrict__ out) {
  const size_t words = plane_pitch / 4, total = rows * words;
  unsigned int best = 0;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    uint32_t a = reinterpret_cast<const uint32_t*>(p0)[idx];
    uint32_t b = p1 ? reinterpret_cast<const uint32_t*>(p1)[idx] : 0u;
    uint32_t c = p2 ? reinterpret_cast<const uint32_t*>(p2)[idx] : 0u;
    if (pc) { const uint32_t k = reinterpret_cast<const uint32_t*>(pc)[idx]; a &= k; b &= k; c &= k; }
    unsigned int v;
    if (c) v = (c & b & a) ? 7u : ((c & b) ? 6u : ((c & a) ? 5u : 4u));
    else v = (a & b) ? 3u : (b ? 2u : (a ? 1u : 0u));
    best = v > best ? v : best;
  }
  for (int off = 32; off > 0; off >>= 1) { unsigned int o = __shfl_xor(best, off, 64); best = o > best ? o : best; }
  if ((threadIdx.x & 63) == 0) atomicMax(out, best);
}

// max over called entries
__global__ void max_allele_kernel(const uint8_t* __restrict__ data, size_t pitch, const uint8_t* __restrict__ bits,
                                  size_t bits_pitch, size_t variants, uint32_t columns, unsigned int* __restrict__ out) {
  unsigned int best = 0;
  const size_t total = variants * (size_t)columns;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t s = i / columns;
    const uint32_t h = (uint32_t)(i - s * columns);
    bool ok = true;
    if (bits) ok = ((bits[s * bits_pitch + (h >> 3)] >> (h & 7)) & 1u) != 0;
    if (ok) { unsigned int v = data[s * pitch + h]; best = v > best ? v : best; }
  }
  for (int off = 32; off > 0; off >>= 1) { unsigned int o = __shfl_xor(best, off, 64); best = o > best ? o : best; }
  if ((threadIdx.x & 63) == 0) atomicMax(out, best);
}

}  // namespace fmh
