// sweep_flat_kernels.hpp — the LDS-staged flat-tile route of the sweep for SHORT packed rows (at most 32 vectors = 4 096 columns):
// biallelic matrices with nothing missing.  Included by sweep_flat.hip only.
//
// What it replaces (reference: the per-site gather over a population's columns, stats.rs:1665-1697, and the W&C / Hudson per-site code
// behind it, stats.rs:1814-2127): on the four-lane route of sweep_kernels.hpp a wave reads a 64-row tile as sixteen 64-byte pieces per load
// instruction, spreads every row over four lanes, reduces it with two DPP steps and keeps it by a `gl == s` select - per tile that is as
// many overhead instructions as counting instructions, and the counters show those kernels stalled on VALU issue (SQ_WAIT_INST_ANY 0.32-0.45
// of the wave-cycles against 0.04-0.05 on the sixteen-lane kernels; instruction cache, LDS and the TA queues are not involved:
// profiles/r04/wait_inst_split_*.csv).  Here
//   * a wave's tile - 64 rows x pitch bytes - is ONE contiguous run of nvec KiB in HBM (pitch = nvec x 16 exactly) and is fetched as
//     nvec whole-KiB LDS-DMA instructions (global_load_lds_dwordx4: no VGPR staging, no ds_write);
//   * LANE L OWNS ROW L: it reads its row's vectors back with ds_read_b128.  A row pitch of nvec x 16 bytes is a bank conflict of up to
//     16 ways for such a column read, so the DMA XOR-swizzles the vectors of a row on the way in: with t = min(ctz(nvec), 4), vector v of
//     row r sits in slot v ^ f(r), f(r) = (r >> (4 - t)) & (2^t - 1).  2^t divides nvec, so the permutation stays inside the row (the
//     same bytes of HBM per DMA instruction, only the lanes' source addresses are permuted), and for any of the hardware's 16-lane
//     ds_read_b128 groups the sixteen rows land in sixteen different 16-byte bank groups (nvec x r mod 16 takes 16 / 2^t values, f fills
//     the low t bits);
//   * the membership masks are WAVE-UNIFORM per step (every lane works on the same vector index of its own row), so they are scalar
//     operands: s_load from an interleaved image [vector][group][4 dwords] (fmh_groups' mask_flat), AND with an SGPR source, one
//     v_bcnt_u32_b32 chain per group.  No mask VGPRs, no DPP reduction, no select: the counting part is its floor of 8 x P VALU
//     instructions per vector;
//   * lane L then runs the SAME f64 epilogue on the same counts (finish_biallelic_site / site_epilogue of sweep_kernels.hpp), in the same
//     tile order per lane with the same grid rule, so per-site tracks are the same bits as on every other route; regional sums are the
//     same per-lane sequences reduced over workgroups of two waves instead of four (another grid: equal to 1e-12, as between any two grids).
// Pipeline: no barrier in the tile loop - every wave owns `slots` (1 or 2) tile images in LDS.  Two slots: the DMA of tile t + 2 is issued
// right after the counting of tile t freed its slot and lands under the epilogue of t and the counting of t + 1; one slot: the DMA of t + 1
// is issued after the counting of t and lands under the epilogue of t.  The DMA is inline assembly (the compiler would otherwise put a
// vmcnt(0) in front of EVERY LDS read that follows a DMA it cannot tell apart - the epilogue's table reads included), with explicit
// s_waitcnt where a slot is handed over.
#pragma once

#include "sweep_kernels.hpp"

namespace fmh {

#ifndef FMH_FLAT_WAVES
#define FMH_FLAT_WAVES 1
#endif
constexpr int kFlatWaves = FMH_FLAT_WAVES;  // waves per workgroup.  One: nothing is shared between the waves of this route but the W&C tables, and single-wave workgroups pack a CU's LDS with one more 20-KiB tile image than pairs do (same-box A/B, profiles/r04/ab_flat_variants.jsonl: four-group summaries 1.06-1.09 -> 1.00-1.03 of the four-lane route)
constexpr int kFlatBlock = kFlatWaves * kWave;
constexpr int kFlatMaxVec = 32;            // rows of up to 32 vectors (the four-lane route's range)
constexpr uint32_t kFlatLdsSlack = 256;    // reads of the padded last vectors of a slot's last row stay inside the allocation

// how the vectors of a row are permuted in LDS: slot = v ^ ((row >> shift) & mask)
__host__ __device__ inline void flat_swizzle(uint32_t nvec, uint32_t& shift, uint32_t& mask) {
  uint32_t t = 0;
  while (t < 4 && ((nvec >> t) & 1u) == 0) ++t;
  shift = 4 - t;
  mask = (1u << t) - 1;
}

typedef uint32_t flat_u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) uint32_t* flat_cptr_t;  // constant address space: uniform loads are s_load

// one LDS-DMA instruction: 64 lanes x 16 bytes, lane i -> LDS [lds_base + 16 i], from base + voff (per lane)
__device__ __forceinline__ void flat_dma16(const uint8_t* base, uint32_t voff, uint32_t lds_base) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %0, %1"
      :
      : "v"(voff), "s"(base), "s"(lds_base)
      : "memory");
}

template <int P, int MODE, int NVMAX>
__global__ __launch_bounds__(kFlatBlock) void sweep_kernel_flat(const SweepArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  const MatrixView mv = A.mv;
  const uint32_t nvec = mv.nvec;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if constexpr ((MODE & kModeWc) != 0) {
    wc_rcp_init<P, kFlatWaves>(A);
    wc_shape_init<P, kFlatWaves>(A);
    __syncthreads();
  }
  LaneTotals<P, MODE> T;
  T.clear();

  const size_t ntiles = (A.row_count + kTileRows - 1) / kTileRows;
  const size_t tile_stride = (size_t)gridDim.x * kFlatWaves;
  const uint32_t tile_bytes = nvec * 1024u;
  const int two = A.flat_slots == 2 ? 1 : 0;
  // LDS byte address of this wave's first slot (dynamic LDS starts behind the kernel's static tables)
  typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)((lds_ptr_t)smem) +
                        (uint32_t)wave * (uint32_t)(two + 1) * tile_bytes;

  uint32_t fshift, fmask;
  flat_swizzle(nvec, fshift, fmask);
  // DMA source offsets: LDS position j = 64 c + lane of chunk c holds row r = j / nvec, slot s = j % nvec, i.e. global vector s ^ f(r)
  uint32_t doff[NVMAX];
  {
    const uint32_t magic = 65536u / nvec + 1u;  // exact quotient for j < 2048, nvec <= 32
#pragma unroll
    for (int c = 0; c < NVMAX; ++c) {
      const uint32_t j = 64u * (uint32_t)c + (uint32_t)lane;
      const uint32_t r = (j * magic) >> 16;
      const uint32_t s = j - r * nvec;
      doff[c] = (r * nvec + (s ^ ((r >> fshift) & fmask))) * 16u;
    }
  }
  // my row in a slot, and my swizzle term
  const uint32_t my_row_off = (uint32_t)lane * nvec * 16u;
  const uint32_t my_f16 = (((uint32_t)lane >> fshift) & fmask) * 16u;
  const flat_cptr_t mk = (flat_cptr_t)(uintptr_t)A.mask_flat;  // [nvec4][P][4]
  const uint32_t nvec4 = (nvec + 3u) & ~3u;

  auto issue = [&](size_t t, uint32_t slot_base) {
    const size_t row0 = t * kTileRows;
    const uint8_t* base = mv.data + (A.row_begin + row0) * mv.pitch;
    // bytes of the matrix from this tile's first row to the end of the row range: lanes of rows past the end re-read its last vector
    const size_t left = (A.row_count - row0) * mv.pitch;
    const uint32_t lim = (uint32_t)(left < (size_t)tile_bytes ? left : (size_t)tile_bytes) - 16u;
#pragma unroll
    for (int c = 0; c < NVMAX; ++c) {
      if ((uint32_t)c < nvec) {
        const uint32_t o = doff[c] < lim ? doff[c] : lim;
        flat_dma16(base, o, slot_base + 1024u * (uint32_t)c);
      }
    }
  };

  auto count = [&](uint32_t slot_base, uint32_t (&alt)[P]) {
#pragma unroll
    for (int p = 0; p < P; ++p) alt[p] = 0;
    const lds_ptr_t row = (lds_ptr_t)(uintptr_t)(slot_base + my_row_off);
    for (uint32_t v0 = 0; v0 < nvec4; v0 += 4) {
      flat_u32x4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *(const __attribute__((address_space(3))) flat_u32x4*)(row + (((v0 + u) * 16u) ^ my_f16));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const flat_cptr_t m = mk + (size_t)(v0 + u) * (P * 4);
#pragma unroll
        for (int p = 0; p < P; ++p) {
          alt[p] = bcnt_add(x[u].x & m[4 * p + 0], alt[p]);
          alt[p] = bcnt_add(x[u].y & m[4 * p + 1], alt[p]);
          alt[p] = bcnt_add(x[u].z & m[4 * p + 2], alt[p]);
          alt[p] = bcnt_add(x[u].w & m[4 * p + 3], alt[p]);
        }
      }
    }
  };

  size_t tile = (size_t)blockIdx.x * kFlatWaves + (size_t)wave;
  uint32_t cur = lds0;
  const uint32_t other = lds0 + tile_bytes;  // the second slot (two-slot mode)
  if (tile < ntiles) {
    issue(tile, cur);
    if (two && tile + tile_stride < ntiles) issue(tile + tile_stride, other);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (first tile: both DMAs; nothing else is in flight yet)
  }
  while (tile < ntiles) {
    const size_t nxt = tile + tile_stride;
    const size_t nxt2 = nxt + tile_stride;
    uint32_t alt[P];
    count(cur, alt);
    // the slot is free once every read of it has returned; the data of the NEXT tile must have landed before the next count():
    //   two slots: the next tile's DMA was issued a whole epilogue + count ago, the stores of the previous tile a whole count ago - the
    //     wait below is for old requests - and tile t + 2 goes into the slot just freed;
    //   one slot: tile t + 1 goes into the slot just freed and is waited for at the top of the next round, behind this tile's epilogue.
    if (two) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      if (nxt2 < ntiles) issue(nxt2, cur);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (nxt < ntiles) issue(nxt, cur);
    }
    SiteTally<P> mine;
    WcSite<P> wc;
    double hud_dot = 0.0;
#pragma unroll
    for (int p = 0; p < P; ++p) { mine.n[p] = A.group_size[p]; mine.alt[p] = alt[p]; mine.distinct[p] = 0; mine.ssq[p] = 0; }
    mine.n_all = mv.columns;
    if constexpr ((MODE & kModeWc) != 0) {
      constexpr int NW = 1 + (P * (P - 1)) / 2;
#pragma unroll
      for (int k = 0; k < NW; ++k) { wc.a[k] = 0.0; wc.b[k] = 0.0; }
    }
    finish_biallelic_site<P, MODE>(mine, hud_dot);
    const size_t my_rel = tile * kTileRows + (size_t)lane;
    site_epilogue<P, MODE, false, false>(A, my_rel, my_rel < A.row_count, mine, hud_dot, wc, T);
    if (two) cur = cur == lds0 ? other : lds0;
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    tile = nxt;
  }
  reduce_block_totals<P, MODE, kFlatWaves>(A, T);
}

// ------------------------------------------------------------------------------------------------
// the register-staged variant
// ------------------------------------------------------------------------------------------------
// With LDS-DMA the bytes a CU has in flight are bounded by the tile images it can hold while others are being counted: 160 KiB of LDS is
// six or seven 20-KiB tiles (C3), half of them in flight at best - and the traffic-only microbenchmark (tools/microbench/traffic_ceiling.hip)
// needs about 96 KiB in flight per CU before reads reach 6.3 TB/s.  Here the NEXT tile travels in registers instead (nvec x 4 VGPRs per
// lane, whole-KiB global_load_dwordx4, issued right after the current tile has been copied to the wave's single LDS image) and is in flight
// through the whole counting and epilogue of the current one; LDS holds one tile per wave only for the write -> read-back transposition.
// The compiler places every s_waitcnt itself (plain loads, plain LDS accesses): the stores of the previous epilogue are YOUNGER than the
// loads of the next tile, so vmcnt(N) lets the loads through without waiting for the stores.
// Deferred epilogues (A.flat_defer > 1): a wave counts that many tiles back to back, parking the per-site counts in REGISTERS (a queue that
// is rotated, so the tile loop stays one copy of the code), then runs their epilogues and stores in one burst - the traffic-only
// microbenchmark puts 3 % (C3 W&C) to 8 % (C3 summaries) on bursting a wave's stores in time.  Tiles, their order per lane and every per-site
// operation are those of the undeferred loop.
constexpr int kFlatDeferMax = 8;
// tiles a wave can park: P registers each; four groups park four tiles (the W&C kernel of 20-vector rows must stay within 256 registers = two waves per SIMD)
template <int P> constexpr int flat_defer_max() { return P <= 2 ? kFlatDeferMax : 4; }

template <int P, int MODE, int NVMAX>
__global__ __launch_bounds__(kFlatBlock) void sweep_kernel_flat_rs(const SweepArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  const MatrixView mv = A.mv;
  const uint32_t nvec = mv.nvec;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if constexpr ((MODE & kModeWc) != 0) {
    wc_rcp_init<P, kFlatWaves>(A);
    wc_shape_init<P, kFlatWaves>(A);
    __syncthreads();
  }
  LaneTotals<P, MODE> T;
  T.clear();

  const size_t ntiles = (A.row_count + kTileRows - 1) / kTileRows;
  const size_t tile_stride = (size_t)gridDim.x * kFlatWaves;
  const uint32_t tile_bytes = nvec * 1024u;
  typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
  const lds_ptr_t slot = (lds_ptr_t)smem + (uint32_t)wave * tile_bytes;

  uint32_t fshift, fmask;
  flat_swizzle(nvec, fshift, fmask);
  uint32_t doff[NVMAX];  // see sweep_kernel_flat: LDS position 64 c + lane holds global vector (r, s ^ f(r))
  {
    const uint32_t magic = 65536u / nvec + 1u;
#pragma unroll
    for (int c = 0; c < NVMAX; ++c) {
      const uint32_t j = 64u * (uint32_t)c + (uint32_t)lane;
      const uint32_t r = (j * magic) >> 16;
      const uint32_t s2 = j - r * nvec;
#ifdef FMH_FLAT_SWZ_LDS
      doff[c] = (r * nvec + (s2 ^ ((r >> fshift) & fmask))) * 16u;  // here: the LDS byte offset this lane WRITES chunk c to (the global side stays linear)
#else
      doff[c] = (r * nvec + (s2 ^ ((r >> fshift) & fmask))) * 16u;
#endif
    }
  }
  const uint32_t my_row_off = (uint32_t)lane * nvec * 16u;
  const uint32_t my_f16 = (((uint32_t)lane >> fshift) & fmask) * 16u;
  const flat_cptr_t mk = (flat_cptr_t)(uintptr_t)A.mask_flat;
  const uint32_t nvec4 = (nvec + 3u) & ~3u;

  flat_u32x4 stage[NVMAX];
  auto issue = [&](size_t t) {
    const size_t row0 = t * kTileRows;
    const uint8_t* base = mv.data + (A.row_begin + row0) * mv.pitch;
    const size_t left = (A.row_count - row0) * mv.pitch;
    const uint32_t lim = (uint32_t)(left < (size_t)tile_bytes ? left : (size_t)tile_bytes) - 16u;
#pragma unroll
    for (int c = 0; c < NVMAX; ++c) {
      if ((uint32_t)c < nvec) {
#ifdef FMH_FLAT_SWZ_LDS
        const uint32_t lin = 1024u * (uint32_t)c + 16u * (uint32_t)lane;
        const uint32_t o = lin < lim ? lin : lim;
#else
        const uint32_t o = doff[c] < lim ? doff[c] : lim;
#endif
        stage[c] = *reinterpret_cast<const flat_u32x4*>(base + o);
      }
    }
  };
  auto to_lds = [&]() {
#pragma unroll
    for (int c = 0; c < NVMAX; ++c)
#ifdef FMH_FLAT_SWZ_LDS
      if ((uint32_t)c < nvec) *(__attribute__((address_space(3))) flat_u32x4*)(slot + doff[c]) = stage[c];
#else
      if ((uint32_t)c < nvec) *(__attribute__((address_space(3))) flat_u32x4*)(slot + 1024u * (uint32_t)c + 16u * (uint32_t)lane) = stage[c];
#endif
  };
  auto count = [&](uint32_t (&alt)[P]) {
#pragma unroll
    for (int p = 0; p < P; ++p) alt[p] = 0;
    const lds_ptr_t row = slot + my_row_off;
    for (uint32_t v0 = 0; v0 < nvec4; v0 += 4) {
      flat_u32x4 x[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = *(const __attribute__((address_space(3))) flat_u32x4*)(row + (((v0 + u) * 16u) ^ my_f16));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const flat_cptr_t m = mk + (size_t)(v0 + u) * (P * 4);
#pragma unroll
        for (int p = 0; p < P; ++p) {
          alt[p] = bcnt_add(x[u].x & m[4 * p + 0], alt[p]);
          alt[p] = bcnt_add(x[u].y & m[4 * p + 1], alt[p]);
          alt[p] = bcnt_add(x[u].z & m[4 * p + 2], alt[p]);
          alt[p] = bcnt_add(x[u].w & m[4 * p + 3], alt[p]);
        }
      }
    }
  };

  constexpr int DMAX = flat_defer_max<P>();
  const int depth = A.flat_defer < 1 ? 1 : (A.flat_defer > DMAX ? DMAX : A.flat_defer);
  uint32_t park[DMAX][P];
#pragma unroll
  for (int i = 0; i < DMAX; ++i)
#pragma unroll
    for (int p = 0; p < P; ++p) park[i][p] = 0;
  auto rotate_in = [&](const uint32_t (&alt)[P]) {  // the queue moves up by one, the new entry goes to the end
#pragma unroll
    for (int i = 0; i + 1 < DMAX; ++i)
#pragma unroll
      for (int p = 0; p < P; ++p) park[i][p] = park[i + 1][p];
#pragma unroll
    for (int p = 0; p < P; ++p) park[DMAX - 1][p] = alt[p];
  };

  size_t tile0 = (size_t)blockIdx.x * kFlatWaves + (size_t)wave;
  if (tile0 < ntiles) issue(tile0);
  while (tile0 < ntiles) {
    int nb = 0;
#pragma unroll 1
    for (int b = 0; b < depth; ++b) {  // counts of up to `depth` tiles
      const size_t tile = tile0 + (size_t)b * tile_stride;
      if (tile >= ntiles) break;
      to_lds();  // (the compiler waits for this tile's loads here, and for the previous tile's LDS reads before it overwrites the image)
      const size_t nxt = tile + tile_stride;
      if (nxt < ntiles) issue(nxt);  // in flight through this tile's counting (and, after the last count of the group, the epilogues)
      uint32_t alt[P];
      count(alt);
      rotate_in(alt);
      ++nb;
    }
    // the group's entries sit at the END of the queue: bring the first one to the front
    for (int k = nb; k < DMAX; ++k) {
      uint32_t zero[P];
#pragma unroll
      for (int p = 0; p < P; ++p) zero[p] = 0;
      rotate_in(zero);
    }
#pragma unroll 1
    for (int b = 0; b < nb; ++b) {  // their epilogues and stores, back to back
      const size_t tile = tile0 + (size_t)b * tile_stride;
      SiteTally<P> mine;
      WcSite<P> wc;
      double hud_dot = 0.0;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        uint32_t gs = A.group_size[p];
        asm volatile("" : "+s"(gs));  // not a visible loop invariant: what the epilogue derives from it must not be hoisted and kept alive across the counting loop
        mine.n[p] = gs;
        mine.alt[p] = park[0][p];
        mine.distinct[p] = 0;
        mine.ssq[p] = 0;
      }
      uint32_t cols = mv.columns;
      asm volatile("" : "+s"(cols));
      mine.n_all = cols;
      if constexpr ((MODE & kModeWc) != 0) {
        constexpr int NW = 1 + (P * (P - 1)) / 2;
#pragma unroll
        for (int k = 0; k < NW; ++k) { wc.a[k] = 0.0; wc.b[k] = 0.0; }
      }
      finish_biallelic_site<P, MODE>(mine, hud_dot);
      const size_t my_rel = tile * kTileRows + (size_t)lane;
      site_epilogue<P, MODE, false, false>(A, my_rel, my_rel < A.row_count, mine, hud_dot, wc, T);
      uint32_t zero[P];
#pragma unroll
      for (int p = 0; p < P; ++p) zero[p] = 0;
      rotate_in(zero);
    }
    tile0 += (size_t)depth * tile_stride;
  }
  reduce_block_totals<P, MODE, kFlatWaves>(A, T);
}

}  // namespace fmh
