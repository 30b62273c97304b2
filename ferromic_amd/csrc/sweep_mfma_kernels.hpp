// sweep_mfma_kernels.hpp — the allele-count sweep as an int8 matrix-core contraction (BASELINE config C5, SURVEY.md 7
// step 7 / 8d): counts[P x S] = M[P x H] . G^T[H x S] on v_mfma_i32_16x16x64_i8, for u8-row matrices that are biallelic with
// nothing missing (then n_p is the group size and alt_p = sum of genotype bytes over the group's columns).
//
// It computes the same integers as count_row_biallelic (v_dot4) and shares everything after the counts with sweep_kernel
// (finish_biallelic_site, site_epilogue, reduce_block_totals), so every output is the same bits.  With at most four output
// rows the contraction stays HBM-bound - the matrix cores idle at a few percent - which is why it is an alternative route
// (FMH_COUNTS_MFMA=1), measured next to the dot4 and popcount routes, not the default.
//
// Mapping (wave64).  A wave owns a tile of 64 consecutive sites = four 16-site sub-tiles, one MFMA accumulator each.
//   A operand (16 x 64, "population slots" x columns): lane l supplies row (l & 15) -> population (l & 15) % P, K chunk
//     (l >> 4): 16 mask bytes, one ds_read_b128 from the LDS mask image, shared by the four sub-tiles of a K step.
//   B operand (64 x 16, columns x sites): lane l supplies site (l & 15) of the sub-tile, K chunk (l >> 4): 16 genotype
//     bytes, one global_load_dwordx4 (a K step reads 64 B of each of the tile's 64 rows; U steps are issued back to back,
//     so a row is read in runs of 64 U bytes).
//   Both operands cut K with the same (chunk = l >> 4) rule, so the instruction's internal K order does not matter.
//   C (16 x 16 i32): lane l holds rows 4 (l >> 4) + i, i = 0..3, of column l & 15.  Row r counts population r % P, so
//     register i of ANY lane group is population i % P, and lane l finds site l's counts in the accumulator of sub-tile
//     l >> 4: the epilogue's one-site-per-lane layout falls out with three v_cndmask per population, no shuffles, no LDS.
#pragma once

#include "sweep_kernels.hpp"

namespace fmh {

typedef int mfma_i32x4 __attribute__((ext_vector_type(4)));

// LDS stride between two populations' mask images, in 16-byte vectors: == 2 (mod 16), so that the eight distinct
// (population, chunk) addresses a ds_read_b128 lane group touches fall on eight different 4-bank slots
__host__ __device__ inline uint32_t mfma_mask_stride(uint32_t nvec, int unroll) {
  const uint32_t covered = (nvec + 4u * (uint32_t)unroll - 1u) / (4u * (uint32_t)unroll) * (4u * (uint32_t)unroll);
  return (covered + 15u) / 16u * 16u + 2u;
}

template <int P, int MODE, int U>
__global__ __launch_bounds__(kBlock) void sweep_mfma_kernel(const SweepArgs A) {
  static_assert(P == 1 || P == 2 || P == 4, "the population slots of one accumulator column hold at most four groups");
  extern __shared__ __align__(16) unsigned char smem[];
  const MatrixView mv = A.mv;
  const uint32_t nvec = mv.nvec;
  const uint32_t mstride = A.nvec_pad;  // mfma_mask_stride(nvec, U)
  uint4* staged = reinterpret_cast<uint4*>(smem);
  for (uint32_t i = threadIdx.x; i < (uint32_t)P * mstride; i += kBlock) {
    const uint32_t p = i / mstride, v = i - p * mstride;
    staged[i] = v < nvec ? load_vec(A.masks + (size_t)p * A.mask_pitch + (size_t)v * 16) : make_uint4(0, 0, 0, 0);
  }
  if constexpr ((MODE & kModeWc) != 0) { wc_rcp_init<P>(A); wc_shape_init<P>(A); }  // nothing is missing on this route: shared-denominator divisions
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col = lane & 15;    // A: population slot; B: site inside the sub-tile
  const int chunk = lane >> 4;  // K chunk of 16 columns inside a 64-column K step; also: the sub-tile whose results this lane keeps
  const uint4* my_mask = staged + (uint32_t)(col % P) * mstride + chunk;
  const uint32_t last = nvec - 1;
  const uint32_t ksteps = (nvec + 3) / 4;

  LaneTotals<P, MODE> T;
  T.clear();

  const size_t ntiles = (A.row_count + kTileRows - 1) / kTileRows;
  for (size_t tile = (size_t)blockIdx.x * kWavesPerBlock + wave; tile < ntiles; tile += (size_t)gridDim.x * kWavesPerBlock) {
    const size_t tile_row0 = tile * kTileRows;  // relative to row_begin
    const uint8_t* rp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const size_t rel = tile_row0 + 16 * t + col;
      rp[t] = mv.data + (A.row_begin + (rel < A.row_count ? rel : A.row_count - 1)) * mv.pitch;  // rows past the end re-read the last one
    }
    mfma_i32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = mfma_i32x4{0, 0, 0, 0};
    for (uint32_t k0 = 0; k0 < ksteps; k0 += U) {
      uint4 g[U][4];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t v = (k0 + u) * 4 + chunk;
        const size_t off = (size_t)(v < last ? v : last) * 16;  // vectors past the row re-read its last one; their mask vectors are zero
#pragma unroll
        for (int t = 0; t < 4; ++t) g[u][t] = load_stream(rp[t] + off);
      }
      // all 4 U loads of the batch are issued before the first MFMA (left alone, hipcc sinks them between the MFMAs and keeps four in flight)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint4 m = my_mask[(k0 + u) * 4];  // zero beyond the row (the image is padded to a multiple of 4 U vectors)
        const mfma_i32x4 a = mfma_i32x4{(int)m.x, (int)m.y, (int)m.z, (int)m.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const mfma_i32x4 b = mfma_i32x4{(int)g[u][t].x, (int)g[u][t].y, (int)g[u][t].z, (int)g[u][t].w};
          acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[t], 0, 0, 0);
        }
      }
    }
    SiteTally<P> mine;
    WcSite<P> wc;
    double hud_dot = 0.0;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int x0 = acc[0][p], x1 = acc[1][p], x2 = acc[2][p], x3 = acc[3][p];
      mine.alt[p] = (uint32_t)(chunk == 0 ? x0 : chunk == 1 ? x1 : chunk == 2 ? x2 : x3);
      mine.n[p] = A.group_size[p];
      mine.distinct[p] = 0;
      mine.ssq[p] = 0;
    }
    mine.n_all = mv.columns;
    if constexpr ((MODE & kModeWc) != 0) {
      constexpr int NW = 1 + (P * (P - 1)) / 2;
#pragma unroll
      for (int k = 0; k < NW; ++k) { wc.a[k] = 0.0; wc.b[k] = 0.0; }
    }
    finish_biallelic_site<P, MODE>(mine, hud_dot);
    const size_t my_rel = tile_row0 + lane;
    site_epilogue<P, MODE, false, false>(A, my_rel, my_rel < A.row_count, mine, hud_dot, wc, T, nullptr);
  }
  reduce_block_totals<P, MODE>(A, T);
}

}  // namespace fmh
