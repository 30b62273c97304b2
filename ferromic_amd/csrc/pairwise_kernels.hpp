// pairwise_kernels.hpp — gfx950 kernels of fmh_pairwise_differences: sample-major int8 planes + the int8 MFMA Gram product.
// Include after sweep_kernels.hpp (MatrixView, load_vec).
#pragma once

#include <type_traits>

namespace fmh {

// ------------------------------------------------------------------------------------------------
// pairwise differences (calculate_pairwise_differences, stats.rs:4106-4231)
//   diff(i, j) = sum over sites where both genotypes are Some of  len_i*len_j - sum_a cnt_i(a)*cnt_j(a)
//   both(i, j) = number of sites where both genotypes are Some
// Step 1 turns the site-major matrix into sample-major int8 planes (K = sites contiguous); step 2 is a
// tiled Gram product on the int8 matrix cores (v_mfma_i32_16x16x64_i8) with split-K and exact integer atomics.
// Biallelic cohorts without missing calls need ONE plane: with a = cnt_i(1), b = cnt_j(1) and cnt(0) = ploidy - cnt(1),
//   len_i*len_j - cnt_i(0)*cnt_j(0) - cnt_i(1)*cnt_j(1) = ploidy*(a + b) - 2ab,
// so diff(i, j) = ploidy*(T_i + T_j) - 2*G(i, j) with G the Gram product of plane 1 and T_i = sum over sites of cnt_i(1),
// which the same product delivers as G(i, ones) against an all-ones row appended after the last sample.
// ------------------------------------------------------------------------------------------------
constexpr int kPdBlock = 128;   // samples per planes-kernel workgroup
constexpr int kPdStageK = 128;  // K BYTES per sample per Gram stage: 128 sites as int8, 256 sites as FP4 (two per byte)

// FP4 route (ploidy <= 4): the planes hold e2m1 codes of the counts 0..4 (0, 1, 2, 3, 4 are exact in e2m1) and the
// Gram product runs on v_mfma_scale_f32_16x16x128_f8f6f4 with unit scales - twice the MFMA rate of int8 at half the
// bytes per site, which is what counts for a kernel bound by its L2 -> LDS fill.  Products are integers <= 16 and the
// f32 accumulators hold sums <= 2^24 exactly (the host caps the K chunk of an item accordingly), so the result is as
// exact as the int8 route (checked on the device: tools/microbench/fp4_gram_probe.hip, the pairwise parity tests).
__device__ __forceinline__ uint32_t pd_fp4_code(uint32_t count) { return (0x65420u >> (4 * count)) & 0xFu; }

// Where the 16-byte chunk `chunk` (0..7 of a sample's 128 K bytes in K block kb) of plane p goes.
//   layout 0 (pd_gram256_kernel): planes[p][kb][sample][128], chunk positions XOR-swizzled by (sample >> 1) & 7;
//   layout 1 (pd_gram256p_kernel, round 4): planes[p][kb][K half][sample][64] - every (operand, K half) tile of 256 samples is ONE contiguous
//     16 KiB run - with the four chunk positions of a half XOR-swizzled by g((sample >> 2) & 3), g = {2, 0, 1, 3}: the 16 rows of an MFMA fragment
//     then hit 16 different 16-byte bank groups in every one of the hardware's ds_read_b128 lane groups ({0-3, 12-15, 20-27}, ...).
__device__ __forceinline__ size_t pd_plane_offset(int layout, size_t p, size_t k_blocks, size_t kb, size_t n_pad, size_t smp, uint32_t chunk) {
  if (layout == 0) return ((p * k_blocks + kb) * n_pad + smp) * (size_t)128 + ((chunk ^ (uint32_t)((smp >> 1) & 7)) << 4);
  const uint32_t half = chunk >> 2, c4 = chunk & 3, g = (0x3102u >> (4 * (uint32_t)((smp >> 2) & 3))) & 3u;
  return (((p * k_blocks + kb) * 2 + half) * n_pad + smp) * (size_t)64 + ((c4 ^ g) << 4);
}

// planes[p][site / 128][sample][site % 128]: p = 0..A-1 allele counts, then (only when calls can be missing)
// p = A genotype length and p = A+1 valid (length > 0).  K-blocked so that one Gram stage (128 samples x 128 K
// bytes) is one contiguous 16 KiB run.  Plane p holds allele value p + allele_base (single-plane mode: allele 1 only);
// with ones_row the row after the last sample is 1 at every real site.
// Workgroup = one K block (128 sites) x SB samples: the raw genotype bytes (and called bits) of the tile are staged
// in LDS with coalesced row reads, then every thread turns (sample, 16 consecutive sites) into one 16-byte store
// per plane; a sample's 128 bytes and the SB samples of the tile are contiguous in the output.
template <bool FP4>
__global__ __launch_bounds__(256) void pd_planes_kernel(const MatrixView mv, size_t row_count, uint32_t samples,
                                                        uint32_t ploidy, int n_alleles, int n_planes, int allele_base, int ones_row,
                                                        uint32_t sb, uint8_t* __restrict__ planes, size_t n_pad, size_t s_pad, int layout) {
  extern __shared__ __align__(16) unsigned char pd_smem[];
  constexpr uint32_t KS = FP4 ? 2 * kPdStageK : kPdStageK;  // sites per K block
  constexpr uint32_t PER = KS / 8;                          // sites per 16-byte chunk of the output row
  const uint32_t rowb = sb * ploidy;           // genotype bytes per site in the tile (multiple of 4)
  const uint32_t bitb = (rowb + 7) / 8 + 1;    // called-bit bytes per site in the tile (+1: unaligned start)
  uint8_t* raw = pd_smem;                      // [128][rowb]
  uint8_t* cbits = pd_smem + (size_t)KS * rowb;  // [KS][bitb]
  const size_t kb = blockIdx.x, site0 = kb * KS;
  const uint32_t samp0 = blockIdx.y * sb;
  const size_t col0 = (size_t)samp0 * ploidy;  // first column of the tile (multiple of 4)
  if ((rowb & 15) == 0 && (col0 & 15) == 0) {
    // 16-byte row pieces, eight loads in flight per thread before the first LDS store
    const uint32_t vecs = rowb / 16, total = KS * vecs;
    for (uint32_t base = 0; base < total; base += 256 * 8) {
      uint4 tmp[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t w = base + q * 256 + threadIdx.x;
        const uint32_t r = w / vecs, c = (w - r * vecs) * 16;
        tmp[q] = make_uint4(0, 0, 0, 0);
        if (w < total && site0 + r < row_count && col0 + c + 16 <= mv.pitch) tmp[q] = load_vec(mv.data + (site0 + r) * mv.pitch + col0 + c);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint32_t w = base + q * 256 + threadIdx.x;
        const uint32_t r = w / vecs, c = (w - r * vecs) * 16;
        if (w < total) *reinterpret_cast<uint4*>(raw + (size_t)r * rowb + c) = tmp[q];
      }
    }
  } else {
    const uint32_t words = rowb / 4;
    for (uint32_t w = threadIdx.x; w < KS * words; w += 256) {
      const uint32_t r = w / words, c = (w - r * words) * 4;
      uint32_t v = 0;
      if (site0 + r < row_count && col0 + c + 4 <= mv.pitch) v = *reinterpret_cast<const uint32_t*>(mv.data + (site0 + r) * mv.pitch + col0 + c);
      *reinterpret_cast<uint32_t*>(raw + (size_t)r * rowb + c) = v;
    }
  }
  const uint32_t bit0 = (uint32_t)(col0 & 7);
  if (mv.bits) {
    for (uint32_t w = threadIdx.x; w < KS * bitb; w += 256) {
      const uint32_t r = w / bitb, c = w - r * bitb;
      uint8_t v = 0;
      if (site0 + r < row_count && (col0 >> 3) + c < mv.bits_pitch) v = mv.bits[(site0 + r) * mv.bits_pitch + (col0 >> 3) + c];
      cbits[(size_t)r * bitb + c] = v;
    }
  }
  __syncthreads();
  const size_t k_blocks = s_pad / KS;
  auto put = [](uint32_t (&out)[4], int i, uint32_t val) {
    if constexpr (FP4) out[i >> 3] |= pd_fp4_code(val) << (4 * (i & 7));
    else out[i >> 2] |= val << (8 * (i & 3));
  };
  const bool diploid_complete = ploidy == 2 && !mv.bits;  // the common case: one 16-bit LDS read per genotype, no loops
  for (int p = 0; p < n_planes; ++p) {
    for (uint32_t v = threadIdx.x; v < sb * 8; v += 256) {
      const uint32_t s = v % sb, chunk = v / sb;
      const uint32_t smp = samp0 + s;
      uint32_t out[4] = {0, 0, 0, 0};
      const uint32_t pa = (uint32_t)(p + allele_base);
      if (ones_row && smp == samples) {
#pragma unroll
        for (int i = 0; i < (int)PER; ++i)
          if (site0 + chunk * PER + i < row_count) put(out, i, 1u);
      } else if (diploid_complete) {
        if (smp < samples) {
#pragma unroll
          for (int i = 0; i < (int)PER; ++i) {
            const uint32_t r = chunk * PER + i;
            const uint32_t g = *reinterpret_cast<const uint16_t*>(raw + (size_t)r * rowb + s * 2);
            uint32_t val = ((g & 0xFFu) == pa ? 1u : 0u) + ((g >> 8) == pa ? 1u : 0u);
            if (site0 + r >= row_count) val = 0;
            put(out, i, val);
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < (int)PER; ++i) {
          const uint32_t r = chunk * PER + i;
          // genotype length: CompressedGenotypes::get stops at the first missing allele (process.rs:479-496)
          uint32_t len = 0;
          if (smp < samples && site0 + r < row_count) {
            if (mv.bits) {
              for (uint32_t k = 0; k < ploidy; ++k) {
                const uint32_t h = bit0 + s * ploidy + k;
                if (((cbits[(size_t)r * bitb + (h >> 3)] >> (h & 7)) & 1u) == 0) break;
                ++len;
              }
            } else {
              len = ploidy;
            }
          }
          uint32_t val;
          if (pa < (uint32_t)n_alleles) {
            val = 0;
            const uint8_t* g = raw + (size_t)r * rowb + s * ploidy;
            for (uint32_t k = 0; k < len; ++k) val += g[k] == (uint8_t)pa ? 1u : 0u;
          } else {
            val = pa == (uint32_t)n_alleles ? len : (len > 0 ? 1u : 0u);
          }
          put(out, i, val);
        }
      }
      // 16-byte chunk positions are XOR-swizzled by (sample >> 1) & 7 for the Gram kernel's unpadded LDS image
      *reinterpret_cast<uint4*>(planes + pd_plane_offset(layout, (size_t)p, k_blocks, kb, n_pad, smp, chunk)) = make_uint4(out[0], out[1], out[2], out[3]);
    }
  }
}

// The same planes from a BIT-PACKED matrix (fmh_matrix_pack): p0 = allele & 1, p1 = allele >> 1 (null when biallelic),
// pc = called bits (null when nothing is missing), rows of plane_pitch bytes.  A workgroup stages the bit rows of its
// K block x SB samples in LDS (a few KiB) and emits the same 16-byte chunks as pd_planes_kernel; the matrix read is
// 1/8 of the u8 route's.
template <bool FP4>
__global__ __launch_bounds__(256) void pd_planes_packed_kernel(const uint8_t* __restrict__ p0, const uint8_t* __restrict__ p1,
                                                               const uint8_t* __restrict__ pc, size_t plane_pitch, size_t row_count,
                                                               uint32_t samples, uint32_t ploidy, int n_alleles, int n_planes, int allele_base,
                                                               int ones_row, uint32_t sb, uint8_t* __restrict__ planes, size_t n_pad, size_t s_pad, int layout) {
  extern __shared__ __align__(16) unsigned char pd_smem[];
  constexpr uint32_t KS = FP4 ? 2 * kPdStageK : kPdStageK;  // sites per K block
  constexpr uint32_t PER = KS / 8;                          // sites per 16-byte chunk of the output row
  const uint32_t bitb = (sb * ploidy + 7) / 8 + 1;          // bytes per site per plane in the tile (+1: unaligned start)
  uint8_t* l0 = pd_smem;                                    // [KS][bitb] each
  uint8_t* l1 = l0 + (size_t)KS * bitb;
  uint8_t* lc = l1 + (size_t)KS * bitb;
  const size_t kb = blockIdx.x, site0 = kb * KS;
  const uint32_t samp0 = blockIdx.y * sb;
  const size_t col0 = (size_t)samp0 * ploidy;
  const uint32_t bit0 = (uint32_t)(col0 & 7);
  const uint32_t body = bitb - 1;  // the tile's own bytes; the extra one only matters for an unaligned start
  if ((body & 3) == 0 && ((col0 >> 3) & 3) == 0 && bit0 == 0) {
    // dword pieces (diploid samples in blocks of 256: 64-byte row pieces); the spare byte of each LDS row is never read
    const uint32_t words = body / 4;
    for (uint32_t w = threadIdx.x; w < KS * words; w += 256) {
      const uint32_t r = w / words, c = (w - r * words) * 4;
      const bool ok = site0 + r < row_count && (col0 >> 3) + c + 4 <= plane_pitch;
      const size_t off = (site0 + r) * plane_pitch + (col0 >> 3) + c;
      const uint32_t v0 = ok ? *reinterpret_cast<const uint32_t*>(p0 + off) : 0u;
      const uint32_t v1 = ok && p1 ? *reinterpret_cast<const uint32_t*>(p1 + off) : 0u;
      const uint32_t vc = ok ? (pc ? *reinterpret_cast<const uint32_t*>(pc + off) : 0xFFFFFFFFu) : 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        l0[(size_t)r * bitb + c + k] = (uint8_t)(v0 >> (8 * k));
        l1[(size_t)r * bitb + c + k] = (uint8_t)(v1 >> (8 * k));
        lc[(size_t)r * bitb + c + k] = (uint8_t)(vc >> (8 * k));
      }
    }
  } else {
    for (uint32_t w = threadIdx.x; w < KS * bitb; w += 256) {
      const uint32_t r = w / bitb, c = w - r * bitb;
      const bool ok = site0 + r < row_count && (col0 >> 3) + c < plane_pitch;
      const size_t off = (site0 + r) * plane_pitch + (col0 >> 3) + c;
      l0[w] = ok ? p0[off] : (uint8_t)0;
      l1[w] = ok && p1 ? p1[off] : (uint8_t)0;
      lc[w] = ok ? (pc ? pc[off] : (uint8_t)0xFF) : (uint8_t)0;
    }
  }
  __syncthreads();
  const bool diploid_complete = ploidy == 2 && !pc;  // the common case: two bits per genotype from one LDS byte, no loops
  const size_t k_blocks = s_pad / KS;
  auto put = [](uint32_t (&out)[4], int i, uint32_t val) {
    if constexpr (FP4) out[i >> 3] |= pd_fp4_code(val) << (4 * (i & 7));
    else out[i >> 2] |= val << (8 * (i & 3));
  };
  auto bit_at = [&](const uint8_t* plane, uint32_t r, uint32_t h) { return (uint32_t)(plane[(size_t)r * bitb + (h >> 3)] >> (h & 7)) & 1u; };
  for (int p = 0; p < n_planes; ++p) {
    const uint32_t pa = (uint32_t)(p + allele_base);
    if (diploid_complete && bit0 == 0 && (sb & 3) == 0) {
      // one LDS byte = the genotypes of four neighbouring diploid samples at one site: a thread turns a column of PER such
      // bytes into the four samples' 16-byte chunks (a quarter of the LDS reads of the one-sample-per-thread form)
      for (uint32_t v = threadIdx.x; v < (sb / 4) * 8; v += 256) {
        const uint32_t q = v % (sb / 4), chunk = v / (sb / 4);
        uint32_t out4[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int d = 0; d < 4; ++d) {  // one output dword per sample at a time; the barrier keeps the live set small
#pragma unroll
          for (int j = 0; j < (int)PER / 4; ++j) {
            const int i = d * ((int)PER / 4) + j;
            const uint32_t r = chunk * PER + i;
            const uint32_t y0 = l0[(size_t)r * bitb + q];
            const uint32_t y1 = p1 ? (uint32_t)l1[(size_t)r * bitb + q] : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const uint32_t x0 = (y0 >> (2 * k)) & 3u, x1 = (y1 >> (2 * k)) & 3u;
              const uint32_t lo = (x0 & 1u) | ((x1 & 1u) << 1), hi = (x0 >> 1) | ((x1 >> 1) << 1);
              put(out4[k], i, (lo == pa ? 1u : 0u) + (hi == pa ? 1u : 0u));  // rows past the end were staged as zeros: count 0 unless pa == 0
            }
          }
          asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t smp = samp0 + q * 4 + k;
          uint32_t o[4] = {out4[k][0], out4[k][1], out4[k][2], out4[k][3]};
          if (smp >= samples || pa == 0) {
            // padding samples are zero rows; allele 0 must not count the zero-staged sites past the end of the matrix
            o[0] = o[1] = o[2] = o[3] = 0;
            if (smp < samples) {
#pragma unroll
              for (int i = 0; i < (int)PER; ++i) {
                const uint32_t r = chunk * PER + i;
                if (site0 + r < row_count) {
                  const uint32_t x0 = ((uint32_t)l0[(size_t)r * bitb + q] >> (2 * k)) & 3u;
                  const uint32_t x1 = p1 ? ((uint32_t)l1[(size_t)r * bitb + q] >> (2 * k)) & 3u : 0u;
                  const uint32_t lo = (x0 & 1u) | ((x1 & 1u) << 1), hi = (x0 >> 1) | ((x1 >> 1) << 1);
                  put(o, i, (lo == 0 ? 1u : 0u) + (hi == 0 ? 1u : 0u));
                }
              }
            }
            if (ones_row && smp == samples) {
#pragma unroll
              for (int i = 0; i < (int)PER; ++i)
                if (site0 + chunk * PER + i < row_count) put(o, i, 1u);
            }
          }
          *reinterpret_cast<uint4*>(planes + pd_plane_offset(layout, (size_t)p, k_blocks, kb, n_pad, smp, chunk)) = make_uint4(o[0], o[1], o[2], o[3]);
        }
      }
      continue;
    }
    for (uint32_t v = threadIdx.x; v < sb * 8; v += 256) {
      const uint32_t s = v % sb, chunk = v / sb;
      const uint32_t smp = samp0 + s;
      uint32_t out[4] = {0, 0, 0, 0};
      if (ones_row && smp == samples) {
#pragma unroll
        for (int i = 0; i < (int)PER; ++i)
          if (site0 + chunk * PER + i < row_count) put(out, i, 1u);
      } else if (diploid_complete) {
        if (smp < samples) {
          const uint32_t h = bit0 + s * 2, bi = h >> 3, sh = h & 7;  // h is even: both bits sit in one byte
#pragma unroll
          for (int i = 0; i < (int)PER; ++i) {
            const uint32_t r = chunk * PER + i;
            const uint32_t x0 = ((uint32_t)l0[(size_t)r * bitb + bi] >> sh) & 3u;
            const uint32_t x1 = ((uint32_t)l1[(size_t)r * bitb + bi] >> sh) & 3u;
            const uint32_t lo = (x0 & 1u) | ((x1 & 1u) << 1), hi = (x0 >> 1) | ((x1 >> 1) << 1);
            uint32_t val = (lo == pa ? 1u : 0u) + (hi == pa ? 1u : 0u);
            if (site0 + r >= row_count) val = 0;
            put(out, i, val);
          }
        }
      } else if (smp < samples) {
#pragma unroll 8
        for (int i = 0; i < (int)PER; ++i) {
          const uint32_t r = chunk * PER + i;
          uint32_t val = 0;
          if (site0 + r < row_count) {
            // genotype length: CompressedGenotypes::get stops at the first missing allele (process.rs:479-496)
            uint32_t len = 0, cnt = 0;
            for (uint32_t k = 0; k < ploidy; ++k) {
              const uint32_t h = bit0 + s * ploidy + k;
              if (!bit_at(lc, r, h)) break;
              ++len;
              cnt += (bit_at(l0, r, h) | (bit_at(l1, r, h) << 1)) == pa ? 1u : 0u;
            }
            val = pa < (uint32_t)n_alleles ? cnt : (pa == (uint32_t)n_alleles ? len : (len > 0 ? 1u : 0u));
          }
          put(out, i, val);
        }
      }
      *reinterpret_cast<uint4*>(planes + pd_plane_offset(layout, (size_t)p, k_blocks, kb, n_pad, smp, chunk)) = make_uint4(out[0], out[1], out[2], out[3]);
    }
  }
}

// Gram product of sample-major int8 planes on the matrix cores:
//   out(i, j) += sign * sum_{p in [plane_begin, plane_begin + plane_count)} sum_{k in chunk} planes[p][i][k] * planes[p][j][k]
// v_mfma_i32_16x16x64_i8: each lane feeds 16 consecutive K bytes of one row of A and of one row of B (row = lane & 15,
// K chunk = lane >> 4).  A and B fragments are cut from LDS images with the SAME (row, k) -> lane rule, so whatever
// order the instruction walks K inside a step, both operands agree and the sum over K is the plain dot product.
// C/D: column = lane & 15 (B row), row = 4 * (lane >> 4) + reg (A row).  (The 32x32x32 form measured 4 % slower.)
// LDS images are byte-for-byte copies of the stage tiles (global_load_lds writes wave-linear), whose 16-byte chunks
// the planes kernel stored XOR-swizzled by (row >> 1) & 7: 16 consecutive rows of one K chunk then sit in 16
// different 16-byte slots of the 256-byte bank row, so the fragment reads are conflict-free without padding.
// Persistent and XCD-aware: workgroups are dealt to the 8 XCDs round-robin, so the group blockIdx.x & 7 shares one L2.
// Each XCD owns `slices_per_xcd` K slices; its workgroups take (slice, tile pair) items tile-fastest, so at any moment
// they are walking the same K range over different tile pairs and every stage tile fetched from HBM by one of them
// is an L2 hit for the others that need it.
typedef int pd_v4i __attribute__((ext_vector_type(4)));
typedef int pd_v8i __attribute__((ext_vector_type(8)));
typedef float pd_v4f __attribute__((ext_vector_type(4)));

// Workgroup tile 256 x 256 samples (16 waves, each 64 x 64 = 4 x 4 MFMA tiles): half the operand bytes per MAC of a 128 x 128 tile, whose L2 -> LDS traffic per CU
// (not MFMA issue) bounded the first version of this kernel at 26 % of peak.  16 waves (4 x 4 of 64 x 64), one workgroup per CU;
// the two 64 KiB stage buffers alternate: the global_load_lds of stage s+1 are in flight while stage s feeds the MFMAs,
// one raw s_barrier per stage (a __syncthreads() would drain the loads before the MFMAs start).
constexpr int kPdBig = 256;
constexpr int kPdBigStageBytes = 2 * kPdBig * kPdStageK;  // A image + B image of one stage

// WM x WN waves per workgroup; a wave owns a (256 / WM) x (256 / WN) block of the tile = MT x NT MFMA tiles.  4 x 4 waves
// (64 x 64 each) read 256 KiB of LDS per stage per CU, 2 x 4 waves (128 x 64 each) 192 KiB; both run the 1 M x 2 500 case
// in 6.2 ms, so LDS reads are not what bounds the kernel - the L2 -> LDS fill is (64 KiB per stage per CU).
// s_pad and k_chunk are in K BYTES per sample (= sites for int8, sites / 2 for FP4).
template <int WM, int WN, bool FP4>
__global__ __launch_bounds__(WM * WN * 64) void pd_gram256_kernel(const uint8_t* __restrict__ planes, size_t n_pad, size_t s_pad, int plane_begin,
                                                                  int plane_count, size_t k_chunk, uint32_t slices_per_xcd, uint32_t n_samples,
                                                                  int negate, unsigned long long* __restrict__ out,
                                                                  unsigned long long* __restrict__ totals) {
  constexpr int THREADS = WM * WN * 64;
  constexpr int MT = kPdBig / WM / 16, NT = kPdBig / WN / 16;
  constexpr int kImageBytes = kPdBig * kPdStageK;          // one operand image of a stage: 32 KiB
  constexpr int kLoadsPerImage = kImageBytes / (THREADS * 16);
  extern __shared__ __align__(16) unsigned char pd_lds[];  // [2 buffers][A 32 KiB | B 32 KiB]
  const uint32_t nt = (uint32_t)(n_pad / kPdBig);
  const uint32_t tiles = nt * (nt + 1) / 2;
  const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave / WN, wc = wave % WN;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  uint32_t offa[MT], offb[NT], swza[MT], swzb[NT];
#pragma unroll
  for (int m = 0; m < MT; ++m) { const uint32_t r = wr * (MT * 16) + m * 16 + (lane & 15); offa[m] = r * kPdStageK; swza[m] = (r >> 1) & 7; }
#pragma unroll
  for (int n = 0; n < NT; ++n) { const uint32_t r = wc * (NT * 16) + n * 16 + (lane & 15); offb[n] = r * kPdStageK; swzb[n] = (r >> 1) & 7; }
  for (uint32_t item = slot; item < tiles * slices_per_xcd; item += slots) {
    uint32_t t = item % tiles, bi = 0;
    while (t >= nt - bi) { t -= nt - bi; ++bi; }
    const uint32_t bj = bi + t;
    const size_t k0 = ((size_t)xcd * slices_per_xcd + item / tiles) * k_chunk;
    if (k0 >= s_pad) continue;  // uniform for the workgroup
    const bool diagonal = bi == bj;
    const uint32_t b_image = diagonal ? 0u : (uint32_t)kImageBytes;  // where this item's B fragments are read from
    typedef typename std::conditional<FP4, pd_v4f, pd_v4i>::type acc_t;
    acc_t acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][r] = 0;
    const size_t k1 = k0 + k_chunk < s_pad ? k0 + k_chunk : s_pad;
    const size_t stages_per_plane = (k1 - k0) / kPdStageK;
    const size_t n_stages = stages_per_plane * (size_t)plane_count;
    const size_t k_blocks = s_pad / kPdStageK;
    const size_t kb0 = k0 / kPdStageK;
    const size_t tile_stride = n_pad * kPdStageK;
    const uint8_t* pa = planes + (((size_t)plane_begin * k_blocks + kb0) * n_pad + (size_t)bi * kPdBig) * kPdStageK + (size_t)threadIdx.x * 16;
    const uint8_t* pb = planes + (((size_t)plane_begin * k_blocks + kb0) * n_pad + (size_t)bj * kPdBig) * kPdStageK + (size_t)threadIdx.x * 16;
    const size_t plane_skip = (k_blocks - stages_per_plane) * tile_stride;
    size_t in_plane = 0;
    // THREADS x 16 B per instruction, kLoadsPerImage per operand image; wave-uniform LDS bases
    auto issue = [&](int buf) {
      if (in_plane == stages_per_plane) { pa += plane_skip; pb += plane_skip; in_plane = 0; }
      ++in_plane;
      unsigned char* la = pd_lds + (size_t)buf * kPdBigStageBytes + (size_t)wave * 1024;
      unsigned char* lb = la + kImageBytes;
#pragma unroll
      for (int q = 0; q < kLoadsPerImage; ++q) __builtin_amdgcn_global_load_lds((gptr_t)(pa + q * THREADS * 16), (lptr_t)(la + q * THREADS * 16), 16, 0, 0);
      if (!diagonal) {  // a diagonal tile pair's B operand IS its A operand: one image, half the fill
#pragma unroll
        for (int q = 0; q < kLoadsPerImage; ++q) __builtin_amdgcn_global_load_lds((gptr_t)(pb + q * THREADS * 16), (lptr_t)(lb + q * THREADS * 16), 16, 0, 0);
      }
      pa += tile_stride;
      pb += tile_stride;
    };
    if (n_stages) {
      issue(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    for (size_t stage = 0; stage < n_stages; ++stage) {
      if (stage + 1 < n_stages) issue((int)((stage + 1) & 1));
      const unsigned char* img = pd_lds + (stage & 1) * (size_t)kPdBigStageBytes;
#pragma unroll
      for (int ks = 0; ks < kPdStageK / 64; ++ks) {
        const uint32_t cl = (uint32_t)(ks * 4 + (lane >> 4));
        pd_v4i fa[MT], fb[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) fa[m] = *reinterpret_cast<const pd_v4i*>(&img[offa[m] + ((cl ^ swza[m]) << 4)]);
#pragma unroll
        for (int n = 0; n < NT; ++n) fb[n] = *reinterpret_cast<const pd_v4i*>(&img[b_image + offb[n] + ((cl ^ swzb[n]) << 4)]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            if constexpr (FP4) {
              // 32 e2m1 values per lane in the first four dwords (cbsz = blgp = 4), E8M0 scales 127 = 2^0
              const pd_v8i a8 = {fa[m][0], fa[m][1], fa[m][2], fa[m][3], 0, 0, 0, 0};
              const pd_v8i b8 = {fb[n][0], fb[n][1], fb[n][2], fb[n][3], 0, 0, 0, 0};
              acc[m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[m][n], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            } else {
              acc[m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[m], fb[n], acc[m][n], 0, 0, 0);
            }
          }
      }
      // every wave: my loads of the next stage have landed; everybody: done reading this stage's image
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t i = bi * kPdBig + wr * (MT * 16) + m * 16 + 4 * (lane >> 4) + r;
          const uint32_t j = bj * kPdBig + wc * (NT * 16) + n * 16 + (lane & 15);
          if (i < j && j < n_samples) {
            const long long v = acc[m][n][r];
            if (v != 0) atomicAdd(&out[(size_t)i * n_samples + j], (unsigned long long)(negate ? -v : v));
          } else if (totals && j == n_samples && i < n_samples) {  // the all-ones row: per-sample totals of the plane
            const long long v = acc[m][n][r];
            if (v != 0) atomicAdd(&totals[i], (unsigned long long)v);
          }
        }
  }
}


// ------------------------------------------------------------------------------------------------
// pd_gram256p_kernel (round 4): the same Gram product on a phase-interleaved schedule
// ------------------------------------------------------------------------------------------------
// pd_gram256_kernel above has two 64-KiB stage buffers and one barrier per stage behind `s_waitcnt vmcnt(0)`: every stage ends waiting for
// the slowest of its own global_load_lds, and LDS fragment reads, DMA issue and MFMAs of a wave run one after the other (46-47 % of the int8
// peak).  This kernel follows the 256-squared multi-phase recipe of cdna_hip_programming.md section 5 (counted vmcnt, raw s_barrier, two wave
// groups staggered by half a phase, s_setprio around the MFMA cluster), laid out for this operand format:
//   * 8 waves = 2 (M) x 4 (N); a wave owns 128 x 64 of the 256 x 256 tile = 8 x 4 MFMA tiles (128 accumulator registers);
//   * a K tile is 128 K bytes per sample = two K HALVES of 64 bytes; LDS holds two K tiles as 8 half-tile slots of 16 KiB
//     [buffer][A k0 | B k0 | A k1 | B k1], each the byte-for-byte copy of one contiguous run of the planes (layout 1 of pd_plane_offset);
//   * 4 phases per K tile: (rows 0-63, k0), (rows 64-127, k0), (rows 0-63, k1), (rows 64-127, k1) of the wave's 128 rows - 16 MFMAs each
//     (4 x 4 tiles x one 64-deep K step), 4 A fragment reads per phase and 4 B fragment reads in the first phase of each K half;
//   * every phase issues ONE half-tile of DMA (2 x global_load_lds_dwordx4 per wave), two K tiles ahead of the half it replaces:
//       phase 1: B k0 of tile t+1   phase 2: A k1 of t+1   phase 3: B k1 of t+1   phase 4: A k0 of t+2
//     A slot is refilled two phases after its last read (k0 slots are last read in phase 2, k1 slots in phase 4), and a half-tile is read
//     one phase after the counted wait that retires it: `vmcnt(6)` - three half-tiles stay in flight - in phases 4 (retires k0 of t+1)
//     and 2 (retires k1 of t); never vmcnt(0) inside the loop except in the last two K tiles of an item;
//   * wave group 1 (wr == 1) runs one barrier behind group 0, so that on every SIMD one wave's MFMA cluster runs beside the other's
//     fragment reads and DMA issue.
// Phase p of a wave:  fragment ds_reads, DMA issue, [counted vmcnt]  |barrier|  lgkmcnt(0), 16 MFMAs under s_setprio 1  |barrier|.
constexpr int kPdPhaseThreads = 512;
constexpr int kPdHalfTileBytes = kPdBig * 64;            // 256 samples x 64 K bytes
constexpr int kPdPhaseLdsBytes = 8 * kPdHalfTileBytes;   // two K tiles

template <bool FP4>
__global__ __launch_bounds__(kPdPhaseThreads) void pd_gram256p_kernel(const uint8_t* __restrict__ planes, size_t n_pad, size_t s_pad, int plane_begin,
                                                                      int plane_count, size_t k_chunk, uint32_t slices_per_xcd, uint32_t n_samples,
                                                                      int negate, unsigned long long* __restrict__ out,
                                                                      unsigned long long* __restrict__ totals, int* __restrict__ slabs) {
  extern __shared__ __align__(16) unsigned char pd_lds[];  // ALL of the kernel's LDS (a second __shared__ object makes hipcc wait vmcnt(0) before LDS reads)
  const uint32_t nt = (uint32_t)(n_pad / kPdBig);
  const uint32_t tiles = nt * (nt + 1) / 2;
  const uint32_t xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, slots = gridDim.x >> 3;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wr = wave >> 2, wc = wave & 3;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  typedef typename std::conditional<FP4, pd_v4f, pd_v4i>::type acc_t;
  // fragment addresses inside a half-tile slot: row r of the tile, 16-byte K chunk c = lane >> 4 at slot c ^ g((r >> 2) & 3)
  const uint32_t frow = (uint32_t)(lane & 15), fc = (uint32_t)(lane >> 4);
  const uint32_t fswz = (fc ^ ((0x3102u >> (4 * ((frow >> 2) & 3))) & 3u)) << 4;  // (tile rows are multiples of 16: the row's swizzle term is the lane's)
  const uint32_t a_off = ((uint32_t)wr * 128u + frow) * 64u + fswz;  // + m * 1024 for M tile m (16 rows x 64 bytes)
  const uint32_t b_off = ((uint32_t)wc * 64u + frow) * 64u + fswz;   // + n * 1024
  const size_t k_blocks = s_pad / kPdStageK;
  const size_t half_stride = n_pad * 64;  // bytes between the two K halves of a K block; a K block is 2 x half_stride
  for (uint32_t item = slot; item < tiles * slices_per_xcd; item += slots) {
    uint32_t t = item % tiles, bi = 0;
    while (t >= nt - bi) { t -= nt - bi; ++bi; }
    const uint32_t bj = bi + t;
    const size_t k0 = ((size_t)xcd * slices_per_xcd + item / tiles) * k_chunk;
    if (k0 >= s_pad) continue;  // uniform for the workgroup
    acc_t acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][n][r] = 0;
    const size_t k1 = k0 + k_chunk < s_pad ? k0 + k_chunk : s_pad;
    const uint32_t per_plane = (uint32_t)((k1 - k0) / kPdStageK);
    const uint32_t n_kt = per_plane * (uint32_t)plane_count;
    const size_t kb0 = k0 / kPdStageK;
    // one cursor per half-tile stream (A k0, B k0, A k1, B k1): each advances one K tile per use; a wave copies chunks 2 w and 2 w + 1 of a half-tile
    struct Cursor { const uint8_t* ptr; uint32_t in_plane; uint32_t kt; };
    const size_t kblock_bytes = 2 * half_stride;
    const size_t plane_skip = (k_blocks - per_plane) * kblock_bytes;
    const size_t my_chunks = (size_t)wave * 2048 + (size_t)lane * 16;
    auto cursor = [&](uint32_t tile_index, int half) {
      Cursor c;
      c.ptr = planes + ((size_t)plane_begin * k_blocks + kb0) * kblock_bytes + (size_t)half * half_stride + (size_t)tile_index * kPdBig * 64 + my_chunks;
      c.in_plane = 0;
      c.kt = 0;
      return c;
    };
    Cursor ca0 = cursor(bi, 0), cb0 = cursor(bj, 0), ca1 = cursor(bi, 1), cb1 = cursor(bj, 1);
    // (An L2 prefetch of the K block four tiles ahead - every workgroup touching its share of the lines with a 4-byte LDS-DMA into a landing pad -
    // was measured and changed nothing: 0.788 vs 0.780 of round 3's kernel on two boxes.  The loop is not waiting for HBM.)
    // issue the two DMA pieces of this wave for the cursor's K tile into half-tile slot `hs` of its buffer, then advance the cursor
    auto stage = [&](Cursor& c, int hs) {
      if (c.kt < n_kt) {
        unsigned char* dst = pd_lds + ((size_t)((c.kt & 1) * 4 + hs)) * kPdHalfTileBytes + (size_t)wave * 2048;
        __builtin_amdgcn_global_load_lds((gptr_t)c.ptr, (lptr_t)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)(c.ptr + 1024), (lptr_t)(dst + 1024), 16, 0, 0);
      }
      ++c.kt;
      c.ptr += kblock_bytes;
      if (++c.in_plane == per_plane) { c.ptr += plane_skip; c.in_plane = 0; }
    };
    // prologue: K tile 0 whole and A k0 of K tile 1 - the state every K tile of the loop starts from
    __builtin_amdgcn_s_barrier();  // the previous item's fragment reads are done in every wave before its slots are refilled
    stage(ca0, 0); stage(cb0, 1); stage(ca1, 2); stage(cb1, 3); stage(ca0, 0);
    if (n_kt > 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind
    pd_v4i fa[4], fb[4];
    for (uint32_t kt = 0; kt < n_kt; ++kt) {
      const unsigned char* buf = pd_lds + (size_t)(kt & 1) * 4 * kPdHalfTileBytes;
      const bool tail = kt + 2 >= n_kt;  // fewer than three half-tiles are issued behind the one waited for: drain instead of counting
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int mh = ph & 1, kh = ph >> 1;
        const unsigned char* ia = buf + (size_t)(kh * 2 + 0) * kPdHalfTileBytes;
        const unsigned char* ib = buf + (size_t)(kh * 2 + 1) * kPdHalfTileBytes;
        if (mh == 0) {
#pragma unroll
          for (int n = 0; n < 4; ++n) fb[n] = *reinterpret_cast<const pd_v4i*>(&ib[b_off + (uint32_t)n * 1024u]);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[m] = *reinterpret_cast<const pd_v4i*>(&ia[a_off + (uint32_t)(mh * 4 + m) * 1024u]);
        __builtin_amdgcn_sched_barrier(0);
        if (ph == 0) stage(cb0, 1);
        else if (ph == 1) stage(ca1, 2);
        else if (ph == 2) stage(cb1, 3);
        else stage(ca0, 0);
        if (ph == 1 || ph == 3) {
          if (tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // three half-tiles stay in flight
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            if constexpr (FP4) {
              const pd_v8i a8 = {fa[m][0], fa[m][1], fa[m][2], fa[m][3], 0, 0, 0, 0};
              const pd_v8i b8 = {fb[n][0], fb[n][1], fb[n][2], fb[n][3], 0, 0, 0, 0};
              acc[mh * 4 + m][n] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[mh * 4 + m][n], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            } else {
              acc[mh * 4 + m][n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[m], fb[n], acc[mh * 4 + m][n], 0, 0, 0);
            }
          }
        // the phase's sums are pinned here: without a use hipcc sinks the FP4 MFMAs of all four phases into the last one (the scaled-MFMA builtin
        // has no side effect and nothing reads an accumulator before the end of the item), which undoes the whole interleave
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(acc[mh * 4 + m][n]));
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
      }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // group 0 waits for group 1's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (slabs) {
      // The item's 256 x 256 partial sums as ONE 256-KiB slab of plain 16-byte stores (32 per lane, whole KiB per wave instruction), summed into
      // the 64-bit outputs by pd_slab_reduce_kernel afterwards.  As 64-bit atomics per element (128 per lane and item, 8.6 items per workgroup)
      // the epilogues were a quarter of the kernel's time with the matrix pipes idle.
      int* slab = slabs + ((size_t)xcd * tiles * slices_per_xcd + item) * (size_t)(kPdBig * kPdBig) + ((size_t)wave * 32 * 64 + (size_t)lane) * 4;
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          pd_v4i v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (int)acc[m][n][r];  // (FP4: integers up to 2^24 held exactly in f32)
          *reinterpret_cast<pd_v4i*>(slab + (size_t)(m * 4 + n) * 64 * 4) = v;
        }
      continue;
    }
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const uint32_t i = bi * kPdBig + (uint32_t)wr * 128u + (uint32_t)m * 16u + 4u * (uint32_t)(lane >> 4) + (uint32_t)r;
          const uint32_t j = bj * kPdBig + (uint32_t)wc * 64u + (uint32_t)n * 16u + (uint32_t)(lane & 15);
          if (i < j && j < n_samples) {
            const long long v = (long long)acc[m][n][r];
            if (v != 0) atomicAdd(&out[(size_t)i * n_samples + j], (unsigned long long)(negate ? -v : v));
          } else if (totals && j == n_samples && i < n_samples) {  // the all-ones row: per-sample totals of the plane
            const long long v = (long long)acc[m][n][r];
            if (v != 0) atomicAdd(&totals[i], (unsigned long long)v);
          }
        }
  }
}

// Sums the slabs pd_gram256p_kernel wrote for one launch into the 64-bit outputs: one thread per (tile pair, wave, MFMA tile, lane) = four
// elements, over every K slice that exists.  slab layout: [xcd][slice][tile pair][wave][m][n][lane][4] int32.
__global__ __launch_bounds__(256) void pd_slab_reduce_kernel(const int* __restrict__ slabs, uint32_t nt, uint32_t slices_per_xcd, size_t k_chunk, size_t s_pad,
                                                             uint32_t n_samples, int negate, unsigned long long* __restrict__ out,
                                                             unsigned long long* __restrict__ totals) {
  const uint32_t tiles = nt * (nt + 1) / 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // tile pair * 16384 + vector within the slab
  if (idx >= (size_t)tiles * 16384) return;
  const uint32_t tp = (uint32_t)(idx >> 14), e = (uint32_t)(idx & 16383);
  long long sum[4] = {0, 0, 0, 0};
  for (uint32_t x = 0; x < 8; ++x)
    for (uint32_t q = 0; q < slices_per_xcd; ++q) {
      if (((size_t)x * slices_per_xcd + q) * k_chunk >= s_pad) continue;  // the kernel skipped this slice
      const pd_v4i v = *reinterpret_cast<const pd_v4i*>(slabs + (((size_t)x * slices_per_xcd + q) * tiles + tp) * (size_t)(kPdBig * kPdBig) + (size_t)e * 4);
      sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3];
    }
  uint32_t t = tp, bi = 0;
  while (t >= nt - bi) { t -= nt - bi; ++bi; }
  const uint32_t bj = bi + t;
  const uint32_t lane = e & 63, mn = (e >> 6) & 31, wave = e >> 11, m = mn >> 2, n = mn & 3, wr = wave >> 2, wc = wave & 3;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t i = bi * kPdBig + wr * 128u + m * 16u + 4u * (lane >> 4) + (uint32_t)r;
    const uint32_t j = bj * kPdBig + wc * 64u + n * 16u + (lane & 15);
    const long long v = sum[r];
    if (v == 0) continue;
    if (i < j && j < n_samples) out[(size_t)i * n_samples + j] += (unsigned long long)(negate ? -v : v);  // one thread per element: no atomics
    else if (totals && j == n_samples && i < n_samples) totals[i] += (unsigned long long)v;
  }
}

// Without missing data every genotype has `ploidy` alleles: sum over sites of len_i * len_j = rows * ploidy^2 and every
// site counts for every pair, so those two Gram products collapse into constants.
__global__ void pd_constant_terms_kernel(unsigned long long* __restrict__ diff, unsigned long long* __restrict__ both,
                                         uint32_t n_samples, unsigned long long add_diff, unsigned long long add_both) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_samples * n_samples;
  if (idx >= total) return;
  const uint32_t i = (uint32_t)(idx / n_samples), j = (uint32_t)(idx % n_samples);
  if (i < j) { diff[idx] += add_diff; both[idx] += add_both; }
}

// Single-plane mode: diff(i, j) += ploidy*(T_i + T_j) - 2*G(i, j), both(i, j) += sites.
__global__ void pd_single_plane_finish_kernel(unsigned long long* __restrict__ diff, unsigned long long* __restrict__ both,
                                              const unsigned long long* __restrict__ gram, const unsigned long long* __restrict__ totals,
                                              uint32_t n_samples, unsigned long long ploidy, unsigned long long sites) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_samples * n_samples;
  if (idx >= total) return;
  const uint32_t i = (uint32_t)(idx / n_samples), j = (uint32_t)(idx % n_samples);
  if (i < j) { diff[idx] += ploidy * (totals[i] + totals[j]) - 2ull * gram[idx]; both[idx] += sites; }
}

}  // namespace fmh
