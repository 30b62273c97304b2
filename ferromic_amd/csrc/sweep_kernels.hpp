// sweep_kernels.hpp — hand-written gfx950 (CDNA4, wave64) kernels for the per-site allele-count
// sweep and its fused statistics epilogues.  Included by abi.hip only.
//
// Geometry
//   * genotype matrix, default resident form: BIT PLANES (fmh_matrix_pack) - plane 0 = allele & 1, plane 1 = allele >> 1
//     (alleles 2..3), a "called" plane when calls can be missing; 128 columns per 16-byte vector.  Counting is AND +
//     v_bcnt against bit masks (count_row_packed); rows of up to 32 vectors are shared by 4 lanes, wider ones by 16.
//   * genotype matrix, u8 form (alleles >= 4, wrapped memory, FMH_LAYOUT=bytes): site-major u8, row pitch a multiple
//     of 16 B, padding bytes zero; optional "called" bit-row per site (bit h set = entry h is called).  The points
//     below describe this form; the packed cores keep the same tile / epilogue / reduction structure.
//   * one workgroup = 4 waves; one wave owns a tile of 64 consecutive sites.  A wave is split in
//     four 16-lane groups (one DPP row each).  Group g sweeps rows 16g .. 16g+15 of the tile, one
//     row per step; its 16 lanes read the row as consecutive 16-byte vectors (256 B contiguous per
//     group per load instruction, 16 B/lane), so every HBM byte is fetched exactly once, coalesced.
//   * membership masks (0/1 bytes per column, one per population) live in LDS and are shared by
//     all waves; counts are v_dot4_u32_u8 of genotype bytes against mask bytes.
//   * after the 16 steps lane L of the wave holds the integer counts of row L of the tile, so the
//     f64 statistics run on all 64 lanes at once and every per-site track is written with fully
//     coalesced 256/512-B stores.
//   * regional sums are kept per lane across tiles (persistent grid), reduced once per block and
//     combined in fixed block order by a one-block finalize kernel (deterministic for a given grid).
//
// All f64 arithmetic mirrors the reference expression by expression (file:line cited at each
// function) and is compiled with -ffp-contract=off so per-site values are bit-identical to the
// Rust code for equal counts.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fmh {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;
constexpr int kTileRows = 64;
constexpr double kFstEps = 1e-12;  // FST_EPSILON, stats.rs:26

enum Mode : int { kModeSummary = 1, kModeHudson = 2, kModeDiversity = 4, kModeWc = 8 };
constexpr int kDeferTiles = 16;  // tiles whose epilogues a wave defers (SweepArgs.defer_tiles; 4 bytes of LDS per site and group)
enum Formula : int { kFormulaSparse = 0, kFormulaDense = 1, kFormulaSummary = 2 };

struct MatrixView {
  const uint8_t* data;   // byte layout: genotype bytes; packed layout: bit plane 0 (allele & 1), one bit per column
  const uint8_t* data1;  // packed layout with alleles 2..7: bit plane 1 ((allele >> 1) & 1); else null
  const uint8_t* data2;  // packed layout with alleles 4..7: bit plane 2 (allele >> 2); else null
  const uint8_t* bits;  // called bits, may be null
  const uint8_t* row_gap; // packed layout with a called plane: one byte per row, non-zero when some column of that row is not called; may be null (= every row may)
  const uint8_t* row_hi;  // packed layout with alleles 2..7: one byte per row, non-zero when a plane above plane 0 has a bit set in that row; may be null (= every row may)
  size_t pitch;
  size_t bits_pitch;
  uint32_t columns;  // H = samples * ploidy
  uint32_t nvec;     // 16-byte vectors per row = ceil(H/16) (<= pitch/16); packed layout: ceil(H/128)
};

// The allele-independent part of calculate_variance_components (stats.rs:2034-2127) for one W&C slot
// (overall or one pair): everything that depends only on the called counts n_i of the groups taking part.
// Without missing data the n_i are the group sizes, so the host evaluates these once per launch with the same
// IEEE operations and hands them over as kernel arguments (scalar operands, no registers, no divisions).
struct WcShape {
  double s2_den;      // (r - 1) * n_bar
  double rm1_over_r;  // (r - 1) / r
  double nbar_m1;     // n_bar - 1
  double a_den;       // 1 - c_squared / (r - 1)
  double b_fac;       // n_bar / (n_bar - 1)
  int live;           // 0: r < 2 or n_bar - 1 < 1e-9 -> the slot contributes (0, 0)
  int s2_ok;          // (r - 1) > 1e-9 && n_bar > 1e-9
};
constexpr int kMaxWcSlots = 1 + (8 * 7) / 2;

struct SweepArgs {
  MatrixView mv;
  const uint8_t* masks;   // device [P][mask_pitch] 0/1 bytes, zero beyond the row
  size_t mask_pitch;      // bytes per mask: pitch rounded up to 2048 (>= nvec_pad * 16)
  const uint16_t* mask_bits;  // device [P][mask_pitch / 16]: bit b of word v = column 16 v + b is a member
  const uint32_t* mask_flat;  // flat-tile route (sweep_flat_kernels.hpp): the bit masks interleaved [vector, padded to a multiple of 4][P][4 dwords], zero beyond the row
  int flat_slots;             // flat-tile route: tile images per wave in LDS (1 or 2); 0 = the register-staged variant (one image, the next tile in VGPRs)
  int flat_defer;             // register-staged variant: tiles a wave counts before it runs their epilogues (1 .. kFlatDeferMax)
  uint32_t group_size[8]; // mask popcounts
  uint32_t nvec_pad;      // LDS mask stride: nvec rounded up to a multiple of 16*unroll
  int unroll;             // vectors per lane issued back to back (4 or 8)
  int single_trip;        // packed cores: nvec_pad == lanes-per-row * unroll, i.e. one batch of loads covers a row
  int defer_tiles;        // kDefer kernels: count this many of a wave's tiles, parking the counts in LDS, then run their epilogues and stores back to back
                          // (1 = the undeferred order; -1 on the host side = let launch_one choose by the launch size)
  uint32_t defer_offset;  // byte offset of the parked counts in dynamic LDS (behind the mask image, 16-byte aligned)
  int defer_cap;          // tiles a wave has LDS room to park (<= defer_depth; the host halves it until masks + parked counts leave three workgroups per CU)
  int n_groups;           // caller's group count (<= kernel P; padded groups are never reported)
  size_t row_begin;
  size_t row_count;
  int formula;
  int hudson_formula_p1;  // fused region sweep: formula of the HUDSON part + 1 when it differs from `formula` (which the population totals use); 0 = the same
  int max_allele;         // matrix max allele (general kernel loop bound upper limit)
  // per-site outputs (nullable)
  uint32_t* alt;          // [P][row_count]
  uint32_t* called;       // [P][row_count]
  double* fst; double* dxy; double* pi1; double* pi2; double* num; double* den;  // Hudson
  double* site_pi; double* site_theta; uint32_t* site_distinct;                  // diversity (pop 0)
  const double* harmonic;  // device table H_k, k = 0..H (stats.rs:4234-4240)
  double* wc_a; double* wc_b; uint8_t* wc_state;                                 // W&C [(1+npairs)][row_count]
  int8_t wc_slot[32];      // kernel slot k (padded-P pair order) -> caller slot, -1 = not reported
  WcShape wc_shape[kMaxWcSlots];  // kernel slot order; valid when the matrix has no missing data
  uint32_t wc_live_mask, wc_s2ok_mask;  // bit k = wc_shape[k].live / .s2_ok (the kernels read the doubles from LDS and the flags from here)
  // optional per-allele counts of the GENERAL path (many-group W&C): acounts[(a * acounts_groups + acounts_group0 + p) * row_count + site],
  // pre-zeroed by the host (alleles a row does not iterate stay 0)
  uint32_t* acounts;
  uint32_t acounts_groups;
  uint32_t acounts_group0;
  // block partials
  double* part_f64;        // [grid][kMaxF64]
  unsigned long long* part_u64;  // [grid][kMaxU64]
};

// slots of the per-block partial vectors
constexpr int kPopF64 = 1;  // pi_sum
constexpr int kPopU64 = 2;  // seg, uncallable
// Hudson f64: numerator_sum, denominator_sum, pi1_sum, pi2_sum, dxy_sum_all, site_num_sum, site_den_sum, site_dxy_sum
constexpr int kHudF64 = 8;
// Hudson u64: dxy_uncallable_sites, sites_with_components, site_dxy_skipped
constexpr int kHudU64 = 3;
constexpr int kMaxF64 = 64;
constexpr int kMaxU64 = 64;

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t dot4(uint32_t a, uint32_t b, uint32_t c) {
  return __builtin_amdgcn_udot4(a, b, c, false);  // v_dot4_u32_u8
}

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}

// all-reduce (sum) inside one 16-lane DPP row: xor-1, xor-2 via quad_perm, then row_half_mirror,
// then row_mirror.  Every lane of the row ends with the row total.
__device__ __forceinline__ uint32_t row16_sum(uint32_t x) {
  // gfx950 hazard: a v_dot4 result read by a DPP op needs 3 wait states.  hipcc (ROCm 7.2) leaves only the 2 of its
  // VALU->DPP rule when the last dot4 of a loop and the reduction sit in different basic blocks; the reduction then
  // misses that dot4 (found by the counting fuzz: one-group general sweeps lost bytes 12..15 of the last vectors).
  // Every value reduced here is a dot4 accumulator, so the wait states are spelled out, tied to the value.
  asm volatile("s_nop 3" : "+v"(x));
  x += dpp<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp<0x141>(x);  // row_half_mirror
  x += dpp<0x140>(x);  // row_mirror
  return x;
}
// all-reduce over groups of LPR consecutive lanes (LPR = 16: one DPP row, LPR = 4: one quad); values that are NOT dot4
// results (the packed cores' popcounts), so no wait states are needed
template <int LPR>
__device__ __forceinline__ uint32_t group_sum(uint32_t x) {
  x += dpp<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp<0x4E>(x);   // quad_perm [2,3,0,1]
  if constexpr (LPR >= 8) x += dpp<0x141>(x);   // row_half_mirror
  if constexpr (LPR == 16) x += dpp<0x140>(x);  // row_mirror
  return x;
}
template <int LPR>
__device__ __forceinline__ uint32_t group_or(uint32_t x) {
  x |= dpp<0xB1>(x);
  x |= dpp<0x4E>(x);
  if constexpr (LPR >= 8) x |= dpp<0x141>(x);
  if constexpr (LPR == 16) x |= dpp<0x140>(x);
  return x;
}
__device__ __forceinline__ uint32_t row16_or(uint32_t x) {
  x |= dpp<0xB1>(x);
  x |= dpp<0x4E>(x);
  x |= dpp<0x141>(x);
  x |= dpp<0x140>(x);
  return x;
}

// 4 called-bits -> four 0/1 bytes
__device__ __forceinline__ uint32_t nib_to_bytes(uint32_t nib) {
  return ((nib & 0xFu) * 0x00204081u) & 0x01010101u;
}
// bytes equal to `a` -> 0x01, others 0x00 (exact, no cross-byte carries)
__device__ __forceinline__ uint32_t eq_bytes(uint32_t x, uint32_t a4) {
  uint32_t y = x ^ a4;
  uint32_t t = ((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y;
  return (~t & 0x80808080u) >> 7;
}
__device__ __forceinline__ uint32_t or_bytes(uint32_t x) {
  x |= x >> 16;
  x |= x >> 8;
  return x & 0xFFu;
}

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// per-wave LDS scratch of the eight-group W&C kernels' regional sums (LaneTotals::kWcLdsTotals, wc_xpose_put): sixteen rows of 64 lanes
constexpr int kWcXRows = 16;
constexpr int kWcXStride = 65;
constexpr int kWcXAcc = 6 * kWcXRows;                     // running sums: 87 rows in six batches of sixteen
constexpr int kWcXWave = kWcXRows * kWcXStride + kWcXAcc;  // doubles per wave
constexpr size_t kWcXposeLdsBytes = (size_t)4 /* kWavesPerBlock */ * kWcXWave * 8;  // the host subtracts it from the masks' LDS budget
__device__ __forceinline__ double* wc_xpose_scratch() {
  __shared__ double scratch[4 * kWcXWave];
  return scratch;
}

__device__ __forceinline__ double f64_nan() { return __longlong_as_double(0x7FF8000000000000LL); }

// ------------------------------------------------------------------------------------------------
// reference formulas (each mirrors one Rust function expression by expression)
// ------------------------------------------------------------------------------------------------

// pi_from_components, stats.rs:2723-2733 (caller guarantees n >= 2)
__device__ __forceinline__ double pi_sparse(uint32_t total_called, double sum_counts_sq) {
  double n = (double)total_called;
  double inv_n = 1.0 / n;
  double sum_p2 = sum_counts_sq * inv_n * inv_n;
  return n / (n - 1.0) * (1.0 - sum_p2);
}
// dense_pi_from_counts, stats.rs:1700-1709 == general dense arms 3090-3093 / 4581-4584 (n >= 2)
__device__ __forceinline__ double pi_dense(uint32_t total_called, double sum_sq) {
  double n = (double)total_called;
  return n / (n - 1.0) * (1.0 - sum_sq / (n * n));
}
// no-missing biallelic arm, stats.rs:3239-3247 / 4500-4507 (n >= 2)
__device__ __forceinline__ double pi_dense_nomissing(uint32_t total, uint32_t alt, double sum_sq) {
  if (alt == 0 || alt == total) return 0.0;
  double n = (double)total;
  double scale = n / (n - 1.0);
  double inv_n_sq = 1.0 / (n * n);
  return scale * (1.0 - sum_sq * inv_n_sq);
}
// dense_dxy_from_biallelic_counts, stats.rs:1712-1733 (n1, n2 > 0)
__device__ __forceinline__ double dxy_dense_biallelic(uint32_t n1, uint32_t alt1, uint32_t n2, uint32_t alt2) {
  double n1_f = (double)n1, n2_f = (double)n2;
  double alt1_f = (double)alt1 / n1_f;
  double alt2_f = (double)alt2 / n2_f;
  double ref1 = 1.0 - alt1_f;
  double ref2 = 1.0 - alt2_f;
  double dot = ref1 * ref2 + alt1_f * alt2_f;
  if (dot < 0.0) dot = 0.0;
  double dxy = 1.0 - dot;
  if (dxy < 0.0) dxy = 0.0; else if (dxy > 1.0) dxy = 1.0;
  return dxy;
}
__device__ __forceinline__ double clamp01(double x) {  // f64::max(0.0).min(1.0)
  x = x > 0.0 ? x : 0.0;
  return x < 1.0 ? x : 1.0;
}

// calculate_variance_components, stats.rs:2034-2127, split in two so that nothing is evaluated twice:
// wc_shape() is the part that depends on the called counts only, wc_apply() the per-allele remainder.
// Groups are visited in group order skipping those not taking part, which is the order the reference's
// compacted `pop_stats` vector has, so every sum sees the same operands in the same sequence.
template <int R>
__host__ __device__ __forceinline__ WcShape wc_shape(const uint32_t (&n)[R], const bool (&use)[R]) {
  WcShape s;
  s.s2_den = 0.0; s.rm1_over_r = 0.0; s.nbar_m1 = 0.0; s.a_den = 1.0; s.b_fac = 0.0; s.live = 0; s.s2_ok = 0;
  int r_i = 0;
  unsigned long long total_h = 0;
#pragma unroll
  for (int i = 0; i < R; ++i) if (use[i]) { ++r_i; total_h += n[i]; }
  double r = (double)r_i;
  if (r < 2.0) return s;
  double n_bar = (double)total_h / r;
  if ((n_bar - 1.0) < 1e-9) return s;
  double sum_sq_diff_n = 0.0;
#pragma unroll
  for (int i = 0; i < R; ++i) if (use[i]) { double diff = (double)n[i] - n_bar; sum_sq_diff_n += diff * diff; }
  double c_squared = (r > 0.0 && n_bar > 0.0) ? sum_sq_diff_n / (r * n_bar * n_bar) : 0.0;
  s.s2_ok = ((r - 1.0) > 1e-9 && n_bar > 1e-9) ? 1 : 0;
  s.s2_den = (r - 1.0) * n_bar;
  s.rm1_over_r = (r - 1.0) / r;
  s.nbar_m1 = n_bar - 1.0;
  s.a_den = 1.0 - (c_squared / (r - 1.0));
  s.b_fac = n_bar / (n_bar - 1.0);
  s.live = 1;
  return s;
}

// ---- divisions that share a denominator ---------------------------------------------------------------------------------------
// Without missing data every W&C division of a site divides by a LAUNCH constant (group sizes, their sums, the WcShape terms):
// 64 divisions per biallelic four-group site, and the C3 sweep is VALU-bound on them (profiles/r02: VALU busy 0.82 per SIMD).
// An IEEE f64 division on gfx950 is hipcc's sequence  r = refine(v_rcp_f64(d));  q0 = n * r;  e = fma(-d, q0, n);
// q = fma(e, r, q0)  wrapped in v_div_scale / v_div_fmas / v_div_fixup, which only act on operands near the exponent limits,
// zero or non-finite denominators.  The reciprocal part depends on d alone, so it is computed ONCE per workgroup for every
// denominator of the launch (wc_rcp_init, into LDS) and a division is the last three operations (div_shared): the same
// instructions on the same operands as the full sequence, hence the same correctly rounded quotient bit for bit - for the
// operands this path sees (counts, frequencies, their variances: |exponent| < 200, denominators >= 1e-9).  The parity suite
// holds every per-site a, b against the oracle bit for bit (35 M values at C3's full size).
__device__ __forceinline__ double refined_rcp(double d) {
  const double r0 = __builtin_amdgcn_rcp(d);
  const double e0 = __builtin_fma(-d, r0, 1.0);
  const double r1 = __builtin_fma(r0, e0, r0);
  const double e1 = __builtin_fma(-d, r1, 1.0);
  return __builtin_fma(r1, e1, r1);
}
__device__ __forceinline__ double div_shared(double n, double d, double r) {
  const double q0 = n * r;
  const double e = __builtin_fma(-d, q0, n);
  return __builtin_fma(e, r, q0);
}
// the workgroup's table of refined reciprocals: [0, P) group sizes | [P] their sum | [P + 1 + 3 k + {0, 1, 2}] s2_den, nbar_m1, a_den
// of slot k | then the pair totals n_i + n_j in pair order
__device__ __forceinline__ double* wc_rcp_table() {
  __shared__ double table[8 + 1 + 3 * kMaxWcSlots + kMaxWcSlots];
  return table;
}
template <int P, int WAVES = 4>
__device__ __forceinline__ void wc_rcp_init(const SweepArgs& A) {  // every thread of the block, before a __syncthreads()
  constexpr int NW = 1 + (P * (P - 1)) / 2;
  constexpr int WM = WAVES - 1;  // WAVES is a power of two
  static_assert((WAVES & (WAVES - 1)) == 0, "waves per block");
  double* table = wc_rcp_table();
  // Wave w fills the entries I = w (mod WAVES): all its lanes write the same value to the same word.  The wave index is made provably
  // uniform and every index into the kernel arguments is a compile-time constant of a fully unrolled loop, so each entry is a scalar
  // branch around scalar loads (a lane-dependent or runtime index made hipcc copy the 1.7 KB argument struct to scratch memory).
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  unsigned long long total = 0;
#pragma unroll
  for (int i = 0; i < P; ++i) {
    total += A.group_size[i];
    if (wave == (i & WM)) table[i] = refined_rcp((double)A.group_size[i]);
  }
  if (wave == (P & WM)) table[P] = refined_rcp((double)total);
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    if (wave == (k & WM)) {
      table[P + 1 + 3 * k + 0] = refined_rcp(A.wc_shape[k].s2_den);
      table[P + 1 + 3 * k + 1] = refined_rcp(A.wc_shape[k].nbar_m1);
      table[P + 1 + 3 * k + 2] = refined_rcp(A.wc_shape[k].a_den);
    }
  }
  int q = 0;
#pragma unroll
  for (int x = 0; x < P; ++x) {
#pragma unroll
    for (int y = x + 1; y < P; ++y) {
      if (wave == (q & WM)) table[P + 1 + 3 * NW + q] = refined_rcp((double)((unsigned long long)A.group_size[x] + A.group_size[y]));
      ++q;
    }
  }
}

// The slots' WcShape doubles in LDS (PRE only).  As kernel arguments they are 7 x 12 SGPRs for four groups, far beyond what the epilogue can
// keep: hipcc spilled them into VGPR lanes and re-read them with 267 v_readlane per tile (14 % of the kernel's VALU instructions).  A
// ds_read_b64 of a wave-uniform address costs no VALU slot.
__device__ __forceinline__ double* wc_shape_table() {
  __shared__ double table[5 * kMaxWcSlots];
  return table;
}
template <int P, int WAVES = 4>
__device__ __forceinline__ void wc_shape_init(const SweepArgs& A) {  // every thread of the block, before a __syncthreads()
  constexpr int NW = 1 + (P * (P - 1)) / 2;
  double* table = wc_shape_table();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#pragma unroll
  for (int k = 0; k < NW; ++k) {
    if (wave == (k & (WAVES - 1))) {
      table[5 * k + 0] = A.wc_shape[k].s2_den;
      table[5 * k + 1] = A.wc_shape[k].rm1_over_r;
      table[5 * k + 2] = A.wc_shape[k].nbar_m1;
      table[5 * k + 3] = A.wc_shape[k].a_den;
      table[5 * k + 4] = A.wc_shape[k].b_fac;
    }
  }
}
// slot k's shape for a PRE kernel
__device__ __forceinline__ WcShape wc_shape_of(const SweepArgs& A, int k) {
  WcShape sh;
  const double* t = wc_shape_table() + 5 * k;
  sh.s2_den = t[0]; sh.rm1_over_r = t[1]; sh.nbar_m1 = t[2]; sh.a_den = t[3]; sh.b_fac = t[4];
  sh.live = (int)((A.wc_live_mask >> k) & 1u);
  sh.s2_ok = (int)((A.wc_s2ok_mask >> k) & 1u);
  return sh;
}
// a count or a sum of counts as f64: every operand here fits 32 bits unless the groups overlap on rows of 2^29 columns and more; the
// u64 -> f64 conversion is four VALU instructions, the u32 one is one, the value is the same
__device__ __forceinline__ double count_to_f64(unsigned long long x) {
  return x <= 0xFFFFFFFFull ? (double)(uint32_t)x : (double)x;
}

// numerator_s_squared = sum n_i (p_i - p)^2 is accumulated by the caller; this finishes a and b.  rcp3 = the slot's three shared
// reciprocals (PRE) or null.
template <bool PRE>
__device__ __forceinline__ void wc_apply(const WcShape& s, double numerator_s_squared, double global_p, double& a, double& b, const double* rcp3) {
  double s_squared = 0.0;
  if (s.s2_ok) s_squared = PRE ? div_shared(numerator_s_squared, s.s2_den, rcp3[0]) : numerator_s_squared / s.s2_den;
  double x_wc = global_p * (1.0 - global_p) - s.rm1_over_r * s_squared;
  double a_numerator_term = s_squared - (PRE ? div_shared(x_wc, s.nbar_m1, rcp3[1]) : x_wc / s.nbar_m1);
  a = PRE ? div_shared(a_numerator_term, s.a_den, rcp3[2]) : a_numerator_term / s.a_den;
  b = s.b_fac * x_wc;
}
__device__ __forceinline__ uint8_t wc_classify(double a, double b) {
  double denominator = a + b;
  if (denominator > kFstEps) return 0;   // Calculable
  if (denominator < -kFstEps) return 1;  // ComponentsYieldIndeterminateRatio
  if (fabs(a) > kFstEps) return 0;       // Calculable(+-inf)
  return 2;                              // NoInterPopulationVariance
}

// ------------------------------------------------------------------------------------------------
// per-site state: integer tallies of one site for P populations, fed allele by allele
// ------------------------------------------------------------------------------------------------
template <int P>
struct SiteTally {
  uint32_t n[P];         // called haplotypes per population
  uint32_t alt[P];       // calls of allele 1
  uint32_t distinct[P];  // alleles with count > 0
  unsigned long long ssq[P];  // sum of count^2 (exact)
  uint32_t n_all;        // called entries over ALL columns of the row
};

template <int P>
constexpr int npairs() { return P * (P - 1) / 2; }

// W&C accumulators of one site (stats.rs:1839-1985)
template <int P>
struct WcSite {
  double a[1 + (P * (P - 1)) / 2];
  double b[1 + (P * (P - 1)) / 2];
};

// calculate_fst_wc_at_site_with_membership, stats.rs:1893-1985, for NA alleles of one site at once.
// c[al][i] = calls of allele `al` in group i, n[i] = called haplotypes of group i.  Slot-major: the shape of a
// slot is obtained once (kernel argument when PRE, else computed here) and applied to every allele; the slot's a and b
// are the sums of the per-allele terms in allele order.  emit(k, a, b, has_data) is called for EVERY slot: k = 0 overall
// (has_data always; (0, 0) when fewer than two groups have data), k >= 1 pairs (has_data = both groups have data).
// Between two slots: nothing of slot k + 1 is scheduled above the end of slot k.  The slots are independent, and left alone the scheduler
// interleaves all seven of a four-group site for instruction-level parallelism the f64 pipe does not need (a dependent v_fma_f64 issues
// back to back), at the price of every slot's operands being live at once.
#ifndef FMH_SLOT_FENCE
#define FMH_SLOT_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
template <int P, int NA, bool PRE, class Emit>
__device__ __forceinline__ void wc_for_each_slot(const SweepArgs& A, const uint32_t (&n)[P], const uint32_t (&c)[NA][P], Emit&& emit) {
  constexpr int NWS = 1 + (P * (P - 1)) / 2;
  const double* rcp = PRE ? wc_rcp_table() : nullptr;  // PRE: every denominator below is a launch constant (wc_rcp_init)
  bool use[P];
  double nd[P], freq[NA][P];
  int valid = 0;
  unsigned long long total_called = 0;
#pragma unroll
  for (int i = 0; i < P; ++i) {
    use[i] = n[i] != 0;
    nd[i] = (double)n[i];
    if (use[i]) { ++valid; total_called += n[i]; }
#pragma unroll
    for (int al = 0; al < NA; ++al) {
      double f = 0.0;
      if (use[i]) f = PRE ? div_shared((double)c[al][i], nd[i], rcp[i]) : (double)c[al][i] / nd[i];
      freq[al][i] = f;
    }
  }
  {
    double wa = 0.0, wb = 0.0;
    if (valid >= 2) {  // stats.rs:1925-1930
      WcShape sh;
      if constexpr (PRE) sh = wc_shape_of(A, 0); else sh = wc_shape<P>(n, use);
      if (sh.live) {
#pragma unroll
        for (int al = 0; al < NA; ++al) {
          unsigned long long total_target = 0;
#pragma unroll
          for (int i = 0; i < P; ++i) if (use[i]) total_target += c[al][i];
          double global_freq = 0.0;
          if (total_called > 0) global_freq = PRE ? div_shared(count_to_f64(total_target), count_to_f64(total_called), rcp[P]) : (double)total_target / (double)total_called;
          double num = 0.0;
#pragma unroll
          for (int i = 0; i < P; ++i) if (use[i]) { double diff_p = freq[al][i] - global_freq; num += nd[i] * diff_p * diff_p; }
          double ca, cb;
          wc_apply<PRE>(sh, num, global_freq, ca, cb, PRE ? rcp + P + 1 : nullptr);
          wa += ca;
          wb += cb;
        }
      }
    }
    emit(0, wa, wb, true);
    FMH_SLOT_FENCE();
  }
  int k = 1;
#pragma unroll
  for (int i = 0; i < P; ++i) {
#pragma unroll
    for (int j = i + 1; j < P; ++j) {
      double wa = 0.0, wb = 0.0;
      const bool both = use[i] && use[j];  // stats.rs:1950-1952
      if (both && valid >= 2) {
        WcShape sh;
        if constexpr (PRE) {
          sh = wc_shape_of(A, k);
        } else {
          const uint32_t pn[2] = {n[i], n[j]};
          const bool pu[2] = {true, true};
          sh = wc_shape<2>(pn, pu);
        }
        if (sh.live) {
          unsigned long long pair_total = (unsigned long long)n[i] + n[j];
#pragma unroll
          for (int al = 0; al < NA; ++al) {
            double pair_global = 0.0;
            if (pair_total > 0) {
              const double pair_target = PRE ? count_to_f64((unsigned long long)c[al][i] + c[al][j]) : (double)((unsigned long long)c[al][i] + c[al][j]);
              pair_global = PRE ? div_shared(pair_target, count_to_f64(pair_total), rcp[P + 1 + 3 * NWS + (k - 1)]) : pair_target / (double)pair_total;
            }
            double num = 0.0;
            { double diff_p = freq[al][i] - pair_global; num += nd[i] * diff_p * diff_p; }
            { double diff_p = freq[al][j] - pair_global; num += nd[j] * diff_p * diff_p; }
            double pa, pb;
            wc_apply<PRE>(sh, num, pair_global, pa, pb, PRE ? rcp + P + 1 + 3 * k : nullptr);
            wa += pa;
            wb += pb;
          }
        }
      }
      emit(k, wa, wb, both);
      FMH_SLOT_FENCE();  // (eight groups: a fence after every second or fourth slot only changed nothing, 0.36 / 0.52 ms either way)
      ++k;
    }
  }
}

// allele-at-a-time accumulation (multi-allelic path): the per-allele terms are added to the site's slot sums
template <int P, int NA, bool PRE>
__device__ __forceinline__ void wc_add_alleles(const SweepArgs& A, const uint32_t (&n)[P], const uint32_t (&c)[NA][P], WcSite<P>& w) {
  wc_for_each_slot<P, NA, PRE>(A, n, c, [&](int k, double a, double b, bool has) {
    if (has) { w.a[k] += a; w.b[k] += b; }
  });
}

// ------------------------------------------------------------------------------------------------
// row counting cores.  Each returns values already all-reduced over the 16-lane group.
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ uint4 load_vec(const uint8_t* p) { return *reinterpret_cast<const uint4*>(p); }
// streaming (read-once) genotype loads.  Default cache policy on purpose: nontemporal loads measured
// 4-6 % SLOWER on this sweep (profiles/r01/ab_variants.txt).
__device__ __forceinline__ uint4 load_stream(const uint8_t* p) { return *reinterpret_cast<const uint4*>(p); }

__device__ __forceinline__ uint4 called_bytes(uint32_t bits16) {
  uint4 v;
  v.x = nib_to_bytes(bits16);
  v.y = nib_to_bytes(bits16 >> 4);
  v.z = nib_to_bytes(bits16 >> 8);
  v.w = nib_to_bytes(bits16 >> 12);
  return v;
}

// Where a sweep keeps its membership masks.  Bytes in LDS are the fast form (one ds_read_b128 per vector and group);
// when P masks of the row width exceed the LDS budget the masks are kept as BITS in LDS (one 16-bit word per vector,
// expanded to 0/1 bytes in registers: 8x the width for ~16 VALU ops per vector and group); beyond that, bytes in
// global memory (L2), for one or two groups.
constexpr int kMaskLdsBytes = 0, kMaskGlobalBytes = 1, kMaskLdsBits = 2;
// kMaskPacked: the MATRIX is bit-packed (one bit per column per plane, 128 columns per 16-byte vector) and so are the
// masks in LDS (the same bit image as kMaskLdsBits, read as 16-byte vectors); counting is AND + v_bcnt.
constexpr int kMaskPacked = 3;
template <int MM>
__device__ __forceinline__ uint4 mask_vec(const void* base, uint32_t idx) {
  if constexpr (MM == kMaskLdsBits) return called_bytes(reinterpret_cast<const uint16_t*>(base)[idx]);
  else return reinterpret_cast<const uint4*>(base)[idx];
}

// Biallelic row: alt[p] = sum of allele bytes over called members, n[p] = called members.
// The row is consumed in batches of U vectors per lane: the U global loads are issued back to back
// (U KiB in flight per wave) before the first dot4, so memory-level parallelism does not depend on
// occupancy alone.  LDS masks are zero-padded to nvec_pad (a multiple of 16*U) and the load address
// is clamped to the last vector of the row, so the loop is uniform and branch-free.
template <int P, bool MISSING, bool NEED_ALL, int U, int MM>
__device__ __forceinline__ void count_row_biallelic(const MatrixView& mv, const void* __restrict__ lds_mask,
                                                    uint32_t nvec_pad, const uint8_t* __restrict__ row_ptr,
                                                    const uint8_t* __restrict__ bits_ptr, int gl,
                                                    uint32_t (&alt)[P], uint32_t (&n)[P], uint32_t& n_all) {
#pragma unroll
  for (int p = 0; p < P; ++p) { alt[p] = 0; n[p] = 0; }
  n_all = 0;
  const uint32_t last = mv.nvec - 1;
  for (uint32_t v0 = gl; v0 < nvec_pad; v0 += 16 * U) {
    uint4 g[U];
    uint32_t bits16[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t v = v0 + 16 * u;
      const uint32_t vc = v < last ? v : last;
      g[u] = load_stream(row_ptr + (size_t)vc * 16);
      if (MISSING) bits16[u] = *reinterpret_cast<const uint16_t*>(bits_ptr + (size_t)vc * 2);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t v = v0 + 16 * u;
      uint4 cb;
      if (MISSING) {
        cb = called_bytes(bits16[u]);
        if (NEED_ALL) n_all += v <= last ? __builtin_popcount(bits16[u]) : 0;
      }
#pragma unroll
      for (int p = 0; p < P; ++p) {
        uint4 m = mask_vec<MM>(lds_mask, (uint32_t)p * nvec_pad + v);  // zero beyond the row
        if (MISSING) {
          m.x &= cb.x; m.y &= cb.y; m.z &= cb.z; m.w &= cb.w;
          n[p] = dot4(m.x, 0x01010101u, n[p]);
          n[p] = dot4(m.y, 0x01010101u, n[p]);
          n[p] = dot4(m.z, 0x01010101u, n[p]);
          n[p] = dot4(m.w, 0x01010101u, n[p]);
        }
        alt[p] = dot4(g[u].x, m.x, alt[p]);
        alt[p] = dot4(g[u].y, m.y, alt[p]);
        alt[p] = dot4(g[u].z, m.z, alt[p]);
        alt[p] = dot4(g[u].w, m.w, alt[p]);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    alt[p] = row16_sum(alt[p]);
    if (MISSING) n[p] = row16_sum(n[p]);
  }
  if (MISSING && NEED_ALL) n_all = row16_sum(n_all);
}

// Bit-packed row (fmh_matrix_pack): plane k holds bit k of the allele value (NPL planes: alleles up to 2^NPL - 1), the called
// plane one bit per column; masks are bit vectors too.  For every non-empty subset T of the planes (index T - 1, T a bit set
// over the planes) s[p][T - 1] = members whose allele has ALL bits of T set: s0 (the alt count when NPL == 1), s1, s01, s2,
// s02, s12, s012.  The exact count of every allele value follows by inclusion-exclusion (allele_count_from_planes) without
// touching the row again.  n[p] = called members, allele_or = OR of the called allele values.  Same batching as
// the byte cores: U 16-byte vectors (128 columns each) per lane in flight, clamped addresses, zero-padded masks.
// LPR lanes share a row: 16 (one DPP row, 256 B contiguous per load instruction) or 4 (one quad, 64 B per instruction and
// a two-step reduction: packed rows are short - C4 is 40 vectors - and four lanes cover them with no idle slots).
// acc + popcount(x) as ONE instruction: v_bcnt_u32_b32 adds its second operand.  Written as `acc + __builtin_popcount(x)` LLVM reassociates
// the sums of a row into a tree for instruction-level parallelism - v_bcnt x, 0; v_bcnt y, 0; v_add3 acc, a, b: three instructions for two
// popcounts (148 v_add3 per tile of the four-group kernels, 8 % of their VALU instructions) - which a VALU-bound kernel with four independent
// chains per row (one per group) does not need.
// The chain also keeps no partial sums alive: the two-group Hudson kernel went from 153 to 119 VGPRs, the four-group W&C kernel from 206 to
// 166, the fused region kernel from 203 to a third wave per SIMD (1.51 -> 1.28 ms at C4's shape), and the eight-group kernels stopped
// spilling into AGPRs.  Time is unchanged where registers were not the limit (profiles/r03/kernel_variants.jsonl).
__device__ __forceinline__ uint32_t bcnt_add(uint32_t x, uint32_t acc) {
  uint32_t r;
  asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
  return r;
}
__device__ __forceinline__ uint32_t popc128(const uint4& v, uint32_t acc) {
  acc = bcnt_add(v.x, acc); acc = bcnt_add(v.y, acc); acc = bcnt_add(v.z, acc); acc = bcnt_add(v.w, acc);
  return acc;
}
__device__ __forceinline__ uint4 and128(const uint4& a, const uint4& b) { return make_uint4(a.x & b.x, a.y & b.y, a.z & b.z, a.w & b.w); }
__device__ __forceinline__ uint32_t any128(const uint4& a) { return a.x | a.y | a.z | a.w; }

template <int P, bool MISSING, bool NEED_ALL, int NPL, int U, int LPR>
__device__ __forceinline__ void count_row_packed(const MatrixView& mv, const uint4* __restrict__ lds_mask, uint32_t nvec_pad,
                                                 const uint8_t* __restrict__ row0, const uint8_t* __restrict__ row1,
                                                 const uint8_t* __restrict__ row2, const uint8_t* __restrict__ called_ptr, int gl,
                                                 uint32_t (&n)[P], uint32_t& n_all, uint32_t& allele_or, uint32_t (&s)[P][(1 << NPL) - 1],
                                                 int live_groups = P) {
  constexpr int NS = (1 << NPL) - 1;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    n[p] = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) s[p][k] = 0;
  }
  n_all = 0;
  allele_or = 0;
  const uint32_t last = mv.nvec - 1;
  auto load_trip = [&](uint32_t v0, uint4 (&x)[NPL][U], uint4 (&cb)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t v = v0 + LPR * u;
      const uint32_t vc = v < last ? v : last;
      x[0][u] = load_stream(row0 + (size_t)vc * 16);
      // (a null upper plane: the row carries no allele above 1 - MatrixView::row_hi - and its upper planes are not read)
      if constexpr (NPL >= 2) x[1][u] = row1 ? load_stream(row1 + (size_t)vc * 16) : make_uint4(0, 0, 0, 0);
      if constexpr (NPL >= 3) x[2][u] = row2 ? load_stream(row2 + (size_t)vc * 16) : make_uint4(0, 0, 0, 0);
      // (a null called plane on a MISSING core: every column of the row is called - MatrixView::row_gap - all ones; the masks are zero beyond the row)
      if (MISSING) cb[u] = called_ptr ? load_stream(called_ptr + (size_t)vc * 16) : make_uint4(~0u, ~0u, ~0u, ~0u);
    }
  };
  auto count_trip = [&](uint32_t v0, const uint4 (&x)[NPL][U], const uint4 (&cb)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t v = v0 + LPR * u;
      const bool inside = v <= last;
      // sub[T - 1] = AND of the planes in T, restricted to called entries
      uint4 sub[NS];
      sub[0] = x[0][u];
      if constexpr (NPL >= 2) sub[1] = x[1][u];
      if constexpr (NPL >= 3) sub[3] = x[2][u];
      if (MISSING) {
        sub[0] = and128(sub[0], cb[u]);
        if constexpr (NPL >= 2) sub[1] = and128(sub[1], cb[u]);
        if constexpr (NPL >= 3) sub[3] = and128(sub[3], cb[u]);
        if (NEED_ALL) n_all = inside ? popc128(cb[u], n_all) : n_all;
      }
      if constexpr (NPL >= 2) {
        uint32_t seen = (any128(sub[0]) ? 1u : 0u) | (any128(sub[1]) ? 2u : 0u);
        if constexpr (NPL >= 3) seen |= any128(sub[3]) ? 4u : 0u;
        allele_or |= inside ? seen : 0u;
        sub[2] = and128(sub[0], sub[1]);
      }
      if constexpr (NPL >= 3) {
        sub[4] = and128(sub[0], sub[3]);
        sub[5] = and128(sub[1], sub[3]);
        sub[6] = and128(sub[2], sub[3]);
      }
#pragma unroll
      for (int p = 0; p < P; ++p) {
        if (P == 8 && p >= live_groups) continue;  // five to seven groups run the eight-group kernel: the padding's masks are zero, nothing to count
        uint4 m = lds_mask[(uint32_t)p * nvec_pad + v];  // zero beyond the row
        if (MISSING) { m = and128(m, cb[u]); n[p] = popc128(m, n[p]); }
#pragma unroll
        for (int k = 0; k < NS; ++k) s[p][k] = popc128(and128(sub[k], m), s[p][k]);
      }
    }
  };
  // Rows of many trips (200 000 columns: 25 trips of four vectors per lane) are bound by the latency of one batch of loads per trip.  With the
  // popcounts written in C hipcc unrolled this loop by two by itself (eight loads in flight per lane); the inline-asm popcount chain stopped
  // that (WIDE: 0.44 -> 0.63 ms) and `#pragma unroll 2` is refused for a loop with inline asm, so the biallelic no-missing cores pair the trips by
  // hand.  (The others keep one trip in flight: up to four planes x U x 4 registers per trip.)
  // nvec_pad is a multiple of LPR x U and every lane starts at its own gl < LPR: the trip count is the same for all lanes
  const uint32_t trips = nvec_pad / (LPR * U);
  uint32_t t = 0;
  if constexpr (NPL == 1 && !MISSING) {
    for (; t + 2 <= trips; t += 2) {  // two trips' loads in flight before the first popcount
      uint4 xa[NPL][U], xb[NPL][U], ca[U], cc[U];
      const uint32_t va = (uint32_t)gl + t * (LPR * U), vb = va + LPR * U;
      load_trip(va, xa, ca);
      load_trip(vb, xb, cc);
      count_trip(va, xa, ca);
      count_trip(vb, xb, cc);
    }
  }
  for (; t < trips; ++t) {
    uint4 x[NPL][U], cb[U];
    const uint32_t v0 = (uint32_t)gl + t * (LPR * U);
    load_trip(v0, x, cb);
    count_trip(v0, x, cb);
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (MISSING) n[p] = group_sum<LPR>(n[p]);
#pragma unroll
    for (int k = 0; k < NS; ++k) s[p][k] = group_sum<LPR>(s[p][k]);
  }
  if (MISSING && NEED_ALL) n_all = called_ptr ? group_sum<LPR>(n_all) : mv.columns;  // (all ones would count the padding of the last vector)
  if (NPL >= 2) allele_or = group_or<LPR>(allele_or);
}

// Calls of allele value `a` among n called members, from the subset sums of count_row_packed / count_row_planes
// (inclusion-exclusion over the supersets of a's bit set; a == 0: n minus the members with any bit set).  Exact integers.
template <int NPL>
__device__ __forceinline__ uint32_t allele_count_from_planes(uint32_t a, uint32_t n, const uint32_t (&s)[(1 << NPL) - 1]) {
  uint32_t c = a == 0 ? n : 0u;
#pragma unroll
  for (uint32_t T = 1; T < (1u << NPL); ++T) {
    if ((T & a) == a) c += (__builtin_popcount(T ^ a) & 1) ? 0u - s[T - 1] : s[T - 1];
  }
  return c;
}

// The LPR rows a group owns in one tile, for the commonest packed shape: biallelic, nothing missing, and a row that one
// batch of U loads per lane covers (C4: 40 vectors, 16 lanes x 3).  Then (1) a lane's mask vectors are the same for every
// row, so they are read from LDS once per tile into registers, and (2) the loads of row s + 1 are issued before the
// popcounts and the reduction of row s, so a wave always has two rows of loads in flight instead of one.
// (With many groups or a deep batch the mask vectors would take P x U x 4 registers - 80 for four groups on C3's 20-vector rows, which
// held that kernel at two waves per SIMD - so beyond 12 vectors they stay in LDS and are re-read per row: MREG = false.)
template <int P, int U, int LPR>
__device__ __forceinline__ void tile_rows_packed_prefetch(const SweepArgs& A, const MatrixView& mv, const uint4* __restrict__ lds_mask,
                                                          uint32_t nvec_pad, size_t tile_row0, int grp, int gl, uint32_t (&alt_mine)[P]) {
  constexpr bool MREG = P * U <= 12;
  const uint32_t last = mv.nvec - 1;
  const int live_groups = P == 8 ? A.n_groups : P;
  uint4 m[MREG ? P : 1][MREG ? U : 1];
  if constexpr (MREG) {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int u = 0; u < U; ++u) m[p][u] = lds_mask[(uint32_t)p * nvec_pad + (uint32_t)gl + LPR * u];  // zero beyond the row
  }
  auto load_row = [&](uint4 (&dst)[U], int s) {
    const size_t rel = tile_row0 + (size_t)grp * LPR + s;
    const size_t row = A.row_begin + (rel < A.row_count ? rel : A.row_count - 1);  // rows past the end re-read the last one
    const uint8_t* rp = mv.data + row * mv.pitch;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t v = (uint32_t)gl + LPR * u;
      dst[u] = load_stream(rp + (size_t)(v < last ? v : last) * 16);
    }
  };
  auto count_row = [&](const uint4 (&x)[U], int s) {
    // masks in LDS: keep their reads inside the row (hoisted out of the row loop they would be the P x U register image again)
    if constexpr (!MREG) asm volatile("" ::: "memory");
    uint32_t alt[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      alt[p] = 0;
      if (P == 8 && p >= live_groups) continue;  // five to seven groups run the eight-group kernel: nothing to count for the padding (its masks are zero)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (MREG) alt[p] = popc128(and128(x[u], m[p][u]), alt[p]);
        else alt[p] = popc128(and128(x[u], lds_mask[(uint32_t)p * nvec_pad + (uint32_t)gl + LPR * u]), alt[p]);
      }
      alt[p] = group_sum<LPR>(alt[p]);
    }
    if (gl == s) {
#pragma unroll
      for (int p = 0; p < P; ++p) alt_mine[p] = alt[p];
    }
  };
  uint4 a[U], b[U];
  load_row(a, 0);
  for (int s = 0; s < LPR; s += 2) {  // LPR is even
    load_row(b, s + 1);
    count_row(a, s);
    if (s + 2 < LPR) load_row(a, s + 2);
    count_row(b, s + 1);
  }
}

// General row, single pass for alleles 0..3 by bit planes: with s0 = #(bit0 set), s1 = #(bit1 set),
// s01 = #(both) over the called members, and every called allele < 4 (checked through allele_or),
//   c3 = s01, c1 = s0 - s01, c2 = s1 - s01, c0 = n - c1 - c2 - c3.
// Also returns n[p], n_all and the OR of the called allele values.  Loads are batched like the
// biallelic core (U vectors in flight per lane).
template <int P, bool MISSING, int U, int MM>
__device__ __forceinline__ void count_row_planes(const MatrixView& mv, const void* __restrict__ lds_mask,
                                                 uint32_t nvec_pad, const uint8_t* __restrict__ row_ptr,
                                                 const uint8_t* __restrict__ bits_ptr, int gl, uint32_t (&n)[P],
                                                 uint32_t& n_all, uint32_t& allele_or, uint32_t (&s)[P][3]) {
  uint32_t s0[P], s1[P], s01[P];
#pragma unroll
  for (int p = 0; p < P; ++p) { n[p] = 0; s0[p] = 0; s1[p] = 0; s01[p] = 0; }
  n_all = 0;
  allele_or = 0;
  const uint32_t last = mv.nvec - 1;
  for (uint32_t v0 = gl; v0 < nvec_pad; v0 += 16 * U) {
    uint4 g[U];
    uint32_t bits16[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t v = v0 + 16 * u;
      const uint32_t vc = v < last ? v : last;
      g[u] = load_stream(row_ptr + (size_t)vc * 16);
      if (MISSING) bits16[u] = *reinterpret_cast<const uint16_t*>(bits_ptr + (size_t)vc * 2);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t v = v0 + 16 * u;
      const bool inside = v <= last;
      uint4 x = g[u];
      uint4 cb = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
      if (MISSING) {
        cb = called_bytes(bits16[u]);
        n_all += inside ? __builtin_popcount(bits16[u]) : 0;
        x.x &= cb.x * 0xFFu; x.y &= cb.y * 0xFFu; x.z &= cb.z * 0xFFu; x.w &= cb.w * 0xFFu;
      }
      allele_or |= inside ? or_bytes(x.x | x.y | x.z | x.w) : 0u;
      const uint4 b0 = make_uint4(x.x & 0x01010101u, x.y & 0x01010101u, x.z & 0x01010101u, x.w & 0x01010101u);
      const uint4 b1 = make_uint4((x.x >> 1) & 0x01010101u, (x.y >> 1) & 0x01010101u, (x.z >> 1) & 0x01010101u, (x.w >> 1) & 0x01010101u);
      const uint4 bb = make_uint4(b0.x & b1.x, b0.y & b1.y, b0.z & b1.z, b0.w & b1.w);
#pragma unroll
      for (int p = 0; p < P; ++p) {
        uint4 m = mask_vec<MM>(lds_mask, (uint32_t)p * nvec_pad + v);  // zero beyond the row
        if (MISSING) {
          m.x &= cb.x; m.y &= cb.y; m.z &= cb.z; m.w &= cb.w;
          n[p] = dot4(m.x, 0x01010101u, n[p]); n[p] = dot4(m.y, 0x01010101u, n[p]);
          n[p] = dot4(m.z, 0x01010101u, n[p]); n[p] = dot4(m.w, 0x01010101u, n[p]);
        }
        s0[p] = dot4(b0.x, m.x, s0[p]); s0[p] = dot4(b0.y, m.y, s0[p]); s0[p] = dot4(b0.z, m.z, s0[p]); s0[p] = dot4(b0.w, m.w, s0[p]);
        s1[p] = dot4(b1.x, m.x, s1[p]); s1[p] = dot4(b1.y, m.y, s1[p]); s1[p] = dot4(b1.z, m.z, s1[p]); s1[p] = dot4(b1.w, m.w, s1[p]);
        s01[p] = dot4(bb.x, m.x, s01[p]); s01[p] = dot4(bb.y, m.y, s01[p]); s01[p] = dot4(bb.z, m.z, s01[p]); s01[p] = dot4(bb.w, m.w, s01[p]);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (MISSING) n[p] = row16_sum(n[p]);
    s[p][0] = row16_sum(s0[p]); s[p][1] = row16_sum(s1[p]); s[p][2] = row16_sum(s01[p]);  // subset order of count_row_packed
  }
  if (MISSING) n_all = row16_sum(n_all);
  allele_or = row16_or(allele_or);
}

// General row, pass per allele value a: c[p] = called members carrying allele a.
template <int P, bool MISSING, int MM>
__device__ __forceinline__ void count_row_allele(const MatrixView& mv, const void* __restrict__ lds_mask,
                                                 uint32_t nvec_pad, const uint8_t* __restrict__ row_ptr,
                                                 const uint8_t* __restrict__ bits_ptr, bool row_ok, int gl,
                                                 uint32_t a, uint32_t (&c)[P]) {
#pragma unroll
  for (int p = 0; p < P; ++p) c[p] = 0;
  const uint32_t nvec = mv.nvec;
  const uint32_t a4 = a * 0x01010101u;
  if (row_ok) {
    // The same trip count for every lane of the group (lanes past the row re-read its last vector; their mask vectors
    // are zero): the 16-lane DPP reduction below must see a converged row.  Two vectors per lane per trip.
    auto one = [&](uint32_t v) {
      const uint32_t vc = v < nvec ? v : nvec - 1;
      uint4 g = load_vec(row_ptr + (size_t)vc * 16);
      uint4 e;
      e.x = eq_bytes(g.x, a4); e.y = eq_bytes(g.y, a4); e.z = eq_bytes(g.z, a4); e.w = eq_bytes(g.w, a4);
      if (MISSING) {
        uint32_t bits16 = *reinterpret_cast<const uint16_t*>(bits_ptr + (size_t)vc * 2);
        uint4 cb = called_bytes(bits16);
        e.x &= cb.x; e.y &= cb.y; e.z &= cb.z; e.w &= cb.w;
      }
#pragma unroll
      for (int p = 0; p < P; ++p) {
        uint4 m = mask_vec<MM>(lds_mask, (uint32_t)p * nvec_pad + v);  // zero beyond the row (v < nvec_pad always)
        c[p] = dot4(e.x, m.x, c[p]);
        c[p] = dot4(e.y, m.y, c[p]);
        c[p] = dot4(e.z, m.z, c[p]);
        c[p] = dot4(e.w, m.w, c[p]);
      }
    };
    for (uint32_t v0 = 0; v0 < nvec; v0 += 32) {
      one(v0 + (uint32_t)gl);
      one(v0 + 16 + (uint32_t)gl);
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p) c[p] = row16_sum(c[p]);
}


// ------------------------------------------------------------------------------------------------
// the dense multi-allelic D_xy adds its products in FIRST-OCCURRENCE order (stats.rs:3106-3139, 2557-2590)
// ------------------------------------------------------------------------------------------------
// dense_hudson_sites_general / calculate_dxy_dense walk the `used` list of the population with fewer distinct alleles (ties: population 1)
// - the alleles in the order dense_collect_counts (stats.rs:2823-2880) first met them along the population's ascending column offsets - and
// add (c1 * inv1) * (c2 * inv2) for every allele both populations carry.  f64 addition commutes, so the order only shows in the bits when
// THREE or more alleles are shared; the sweep's general path adds in ascending allele order and repairs exactly those sites afterwards:
// the whole wave scans the site's row once per shared allele for its first member column (below), and the products are added again in that
// order.  Rare on real data (a site with three alleles shared by both populations), so the cost is a uniform branch per tile elsewhere.
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)x, off, 64); x = y < x ? y : x; }
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x += (uint32_t)__shfl_xor((int)x, off, 64);
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
}
__device__ __forceinline__ uint32_t first_bit128(const uint4& v) {  // index of the lowest set bit of a 128-bit vector (x = bits 0..31), 0xFFFFFFFF if none
  if (v.x) return (uint32_t)__builtin_ctz(v.x);
  if (v.y) return 32u + (uint32_t)__builtin_ctz(v.y);
  if (v.z) return 64u + (uint32_t)__builtin_ctz(v.z);
  if (v.w) return 96u + (uint32_t)__builtin_ctz(v.w);
  return 0xFFFFFFFFu;
}
// Column of the first CALLED member of group p that carries allele a in matrix row `row` (p, a, row wave-uniform; every lane of the wave
// takes part, 64 vectors per trip); 0xFFFFFFFF if there is none.  Packed matrices: NPL planes; byte rows: any allele value.
template <int MM, bool MISSING, int NPL>
__device__ __forceinline__ uint32_t first_member_column(const MatrixView& mv, const void* __restrict__ lds_mask, uint32_t nvec_pad, size_t row, int p, uint32_t a) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t v0 = 0; v0 < mv.nvec; v0 += 64) {  // uniform trip count
    const uint32_t v = v0 + lane;
    uint32_t pos = 0xFFFFFFFFu;
    if (v < mv.nvec) {
      if constexpr (MM == kMaskPacked) {
        uint4 ind = reinterpret_cast<const uint4*>(lds_mask)[(uint32_t)p * nvec_pad + v];
        if (MISSING) ind = and128(ind, load_vec(mv.bits + row * mv.bits_pitch + (size_t)v * 16));
        const uint4 x0 = load_vec(mv.data + row * mv.pitch + (size_t)v * 16);
        ind = (a & 1u) ? and128(ind, x0) : make_uint4(ind.x & ~x0.x, ind.y & ~x0.y, ind.z & ~x0.z, ind.w & ~x0.w);
        if constexpr (NPL >= 2) {
          const uint4 x1 = load_vec(mv.data1 + row * mv.pitch + (size_t)v * 16);
          ind = (a & 2u) ? and128(ind, x1) : make_uint4(ind.x & ~x1.x, ind.y & ~x1.y, ind.z & ~x1.z, ind.w & ~x1.w);
        }
        if constexpr (NPL >= 3) {
          const uint4 x2 = load_vec(mv.data2 + row * mv.pitch + (size_t)v * 16);
          ind = (a & 4u) ? and128(ind, x2) : make_uint4(ind.x & ~x2.x, ind.y & ~x2.y, ind.z & ~x2.z, ind.w & ~x2.w);
        }
        const uint32_t b = first_bit128(ind);
        if (b != 0xFFFFFFFFu) pos = v * 128u + b;
      } else {
        const uint4 g = load_vec(mv.data + row * mv.pitch + (size_t)v * 16);
        uint4 m = mask_vec<MM>(lds_mask, (uint32_t)p * nvec_pad + v);  // 0/1 bytes, zero beyond the row
        if (MISSING) {
          const uint4 cb = called_bytes(*reinterpret_cast<const uint16_t*>(mv.bits + row * mv.bits_pitch + (size_t)v * 2));
          m = and128(m, cb);
        }
        const uint32_t a4 = a * 0x01010101u;
        const uint4 e = make_uint4(eq_bytes(g.x, a4) & m.x, eq_bytes(g.y, a4) & m.y, eq_bytes(g.z, a4) & m.z, eq_bytes(g.w, a4) & m.w);
        const uint32_t b = first_bit128(e);  // bit 8 k of the vector = byte k
        if (b != 0xFFFFFFFFu) pos = v * 16u + (b >> 3);
      }
    }
    const uint32_t best = wave_min_u32(pos);
    if (best != 0xFFFFFFFFu) return best;  // later trips hold larger columns
  }
  return 0xFFFFFFFFu;
}
// byte rows: called members of group p carrying allele a in one row, counted by the whole wave (the rescue path of rows with alleles beyond the planes)
template <int MM, bool MISSING>
__device__ __forceinline__ uint32_t wave_member_count(const MatrixView& mv, const void* __restrict__ lds_mask, uint32_t nvec_pad, size_t row, int p, uint32_t a) {
  const uint32_t lane = threadIdx.x & 63;
  uint32_t c = 0;
  const uint32_t a4 = a * 0x01010101u;
  for (uint32_t v = lane; v < mv.nvec; v += 64) {
    const uint4 g = load_vec(mv.data + row * mv.pitch + (size_t)v * 16);
    uint4 m = mask_vec<MM>(lds_mask, (uint32_t)p * nvec_pad + v);
    if (MISSING) m = and128(m, called_bytes(*reinterpret_cast<const uint16_t*>(mv.bits + row * mv.bits_pitch + (size_t)v * 2)));
    c = dot4(eq_bytes(g.x, a4), m.x, c); c = dot4(eq_bytes(g.y, a4), m.y, c); c = dot4(eq_bytes(g.z, a4), m.z, c); c = dot4(eq_bytes(g.w, a4), m.w, c);
  }
  asm volatile("s_nop 3" : "+v"(c));  // dot4 result -> cross-lane read (see row16_sum)
  return wave_sum_u32(c);
}

// ------------------------------------------------------------------------------------------------
// regional accumulators kept per lane
// ------------------------------------------------------------------------------------------------
template <int P, int MODE>
struct LaneTotals {
  double pop_pi[P];
  unsigned long long pop_seg[P];
  unsigned long long pop_unc[P];
  double hud[kHudF64];
  unsigned long long hud_u[kHudU64];
  // W&C regional sums per slot.  Up to four groups: three registers pairs per slot and lane (kWcLaneTotals).  Eight groups (29 slots = 174
  // registers of accumulators) cannot: there the 87 values of a site (a, b, informative as 0.0 / 1.0 per slot) go through a per-wave LDS
  // transposition sixteen rows at a time - every lane writes its value of a row, lane (r, q) adds up a quarter of row r, the quad adds the
  // quarters - and one double per row behind the wave's scratch accumulates over the tiles (kWcLdsTotals, wc_xpose_put).  (Until round 3 the
  // host summed the per-site tracks in a second pass, wc_slot_reduce_kernel: half as long again as the sweep, and it needed the tracks.)
  static constexpr bool kWcLaneTotals = (MODE & kModeWc) != 0 && P <= 4;
  static constexpr bool kWcLdsTotals = (MODE & kModeWc) != 0 && P >= 5;
  static constexpr int kWcRegSlots = kWcLaneTotals ? 1 + (P * (P - 1)) / 2 : 1;
  static constexpr int kWcXposeRows = 3 * (1 + (P * (P - 1)) / 2);
  static constexpr int kWcBatches = kWcLdsTotals ? (kWcXposeRows + kWcXRows - 1) / kWcXRows : 1;
  double wc_a[kWcRegSlots];
  double wc_b[kWcRegSlots];
  unsigned long long wc_inf[kWcRegSlots];

  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int p = 0; p < P; ++p) { pop_pi[p] = 0.0; pop_seg[p] = 0; pop_unc[p] = 0; }
#pragma unroll
    for (int i = 0; i < kHudF64; ++i) hud[i] = 0.0;
#pragma unroll
    for (int i = 0; i < kHudU64; ++i) hud_u[i] = 0;
#pragma unroll
    for (int i = 0; i < kWcRegSlots; ++i) { wc_a[i] = 0.0; wc_b[i] = 0.0; wc_inf[i] = 0; }
    if constexpr (kWcLdsTotals) {  // the wave's running sums in LDS (only this wave touches them)
      static_assert(kWcBatches * kWcXRows <= kWcXAcc, "running sums of the transposition");
      double* acc = wc_xpose_scratch() + (threadIdx.x >> 6) * kWcXWave + kWcXRows * kWcXStride;
      for (int i = threadIdx.x & 63; i < kWcXAcc; i += 64) acc[i] = 0.0;
    }
  }
};

// One value per lane of transposition row `row` (a compile-time constant once the slot loops are unrolled; rows 3k, 3k + 1, 3k + 2 = a, b,
// informative of kernel slot k).  EVERY lane of the wave calls this for every row, in row order.  After the last row of a batch of sixteen
// (or the very last row) the batch is summed: lane (r, q) = (lane / 4, lane % 4) adds entries 16 q ... 16 q + 15 of row r in ascending order,
// then the quad adds its four quarters (xor 1, xor 2: the same bits in all four lanes).  LDS operations of one wave execute in order, so the
// wave needs no barrier - only the compiler has to be kept from moving the reads above the writes (wavefront-scope fences).  Rows of 65
// doubles: the sixteen rows a read instruction touches then start in sixteen different bank pairs.
template <int P, int MODE>
__device__ __forceinline__ void wc_xpose_put(LaneTotals<P, MODE>& T, int row, double v) {
  constexpr int kRows = LaneTotals<P, MODE>::kWcXposeRows;
  const int lane = threadIdx.x & 63;
  double* s = wc_xpose_scratch() + (threadIdx.x >> 6) * kWcXWave;
  s[(row % kWcXRows) * kWcXStride + lane] = v;
  if ((row % kWcXRows) == kWcXRows - 1 || row == kRows - 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double* mine = s + (lane >> 2) * kWcXStride + (lane & 3) * 16;  // (rows past the last one of a short batch hold older values: summed, never read)
    double x = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i % 4 == 0) __builtin_amdgcn_sched_barrier(0);  // four reads in flight, not sixteen: this sits at the epilogue's register peak
      x += mine[i];
    }
    x += dpp_f64<0xB1>(x);  // quad_perm [1,0,3,2]
    x += dpp_f64<0x4E>(x);  // quad_perm [2,3,0,1]
    // the running sums of the wave live behind its rows (one double per transposition row; as registers they were twelve more VGPRs over the
    // whole kernel, which cost several instantiations a wave per SIMD)
    if ((lane & 3) == 0) s[kWcXRows * kWcXStride + (row / kWcXRows) * kWcXRows + (lane >> 2)] += x;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the next batch's writes stay below these reads
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// layout of the per-block partial vectors (also used by the host to unpack)
//   f64: [p*1 + 0] pop pi_sum (p < P) | [8 + i] Hudson f64 i | [16 + k] wc_a[k] | [16 + 29 + k] ... too big
// -> W&C uses its own compact layout: f64 [k] = a, [29 + k] = b ; u64 [k] = informative
constexpr int kOffPopF64 = 0;    // P <= 8 entries
constexpr int kOffHudF64 = 8;    // 8 entries
constexpr int kOffPopSeg = 0;    // u64, P entries
constexpr int kOffPopUnc = 8;    // u64, P entries
constexpr int kOffHudU64 = 16;   // 3 entries
constexpr int kOffWcA = 0;       // f64 (W&C mode only, overlays)
constexpr int kOffWcB = 29;
constexpr int kOffWcInf = 24;    // u64

// ------------------------------------------------------------------------------------------------
// epilogue: statistics of one site from its tallies (lane-parallel, one site per lane)
// ------------------------------------------------------------------------------------------------
// Per-site tracks are written once and never read back by the sweep: streaming (non-temporal) stores.
template <class T, class V>
__device__ __forceinline__ void site_store(T* p, V v) {
#ifdef FMH_PLAIN_STORES
  *p = (T)v;
#else
  __builtin_nontemporal_store((T)v, p);
#endif
}
template <int P, int MODE, bool MISSING, bool GENERAL>
__device__ __forceinline__ void site_epilogue(const SweepArgs& A, size_t out_idx, bool row_ok,
                                              const SiteTally<P>& t, double hud_dot, const WcSite<P>& wc,
                                              LaneTotals<P, MODE>& T, const uint32_t (*c4)[P] = nullptr) {
  // the reference takes the no-missing biallelic arms only on a dense matrix without a mask whose
  // max_allele <= 1 (stats.rs:3191/3218, 4454/4485); build_dense_population_summary (1392, 1409)
  // always uses dense_pi_from_counts
  auto pi_by = [&](int formula, int p) {
    const double ssq = (double)t.ssq[p];
    if (formula == kFormulaSparse) return pi_sparse(t.n[p], ssq);
    if (formula == kFormulaDense && !MISSING && !GENERAL) return pi_dense_nomissing(t.n[p], t.alt[p], ssq);
    return pi_dense(t.n[p], ssq);
  };
  // the fused region sweep gives the population totals and the Hudson part their own formula sets (run_vcf: calculate_pi_dense for the
  // regional pi, the sparse per-site path for Hudson); every other sweep has one formula for both (hf == A.formula)
  const int hf = A.hudson_formula_p1 ? A.hudson_formula_p1 - 1 : A.formula;
  const bool dense = hf != kFormulaSparse;

  double pi[P];   // by the HUDSON formula (what the Hudson records use)
  bool pi_ok[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const uint32_t n = t.n[p];
    pi_ok[p] = n >= 2;
    double v = 0.0, v_pop = 0.0;
    if (pi_ok[p]) {
      v_pop = pi_by(A.formula, p);
      v = hf == A.formula ? v_pop : pi_by(hf, p);
    }
    pi[p] = v;
    if (row_ok) {
      if (pi_ok[p]) T.pop_pi[p] += v_pop; else T.pop_unc[p] += 1;
      if (t.distinct[p] >= 2) T.pop_seg[p] += 1;
      if (p < A.n_groups) {
        if (A.alt) site_store(A.alt + ((size_t)p * A.row_count + out_idx), t.alt[p]);
        if (A.called) site_store(A.called + ((size_t)p * A.row_count + out_idx), n);
      }
    }
  }

  if constexpr ((MODE & kModeDiversity) != 0) {
    // calculate_per_site_diversity, stats.rs:4710-4725, for every group of the sweep (one for fmh_diversity_sites, both of the pair for
    // the fused region sweep): tracks [P][row_count]
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const uint32_t n = t.n[p];
      double pv, tv;
      if (n < 2) { pv = f64_nan(); tv = f64_nan(); }
      else {
        if (t.distinct[p] > 1) { double denom = A.harmonic[n - 1]; tv = denom > 0.0 ? 1.0 / denom : 0.0; }
        else tv = 0.0;
        pv = pi_sparse(n, (double)t.ssq[p]);
      }
      if (row_ok && p < A.n_groups) {
        const size_t o = (size_t)p * A.row_count + out_idx;
        if (A.site_pi) site_store(A.site_pi + o, pv);
        if (A.site_theta) site_store(A.site_theta + o, tv);
        if (A.site_distinct) site_store(A.site_distinct + o, t.distinct[p]);
      }
    }
  }

  if constexpr ((MODE & kModeHudson) != 0 && P >= 2) {
    const uint32_t n1 = t.n[0], n2 = t.n[1];
    // ---- per-site record: hudson_site_from_variant 2969-3014 / dense twins 3072-3278 ----
    bool dxy_ok = (n1 != 0) && (n2 != 0);
    double dxy = 0.0;
    if (dxy_ok) {
      if (dense && !GENERAL) dxy = dxy_dense_biallelic(n1, t.alt[0], n2, t.alt[1]);
      else dxy = clamp01(1.0 - hud_dot);  // dxy_from_counts 2933-2934 / general 3140
    }
    double fst = f64_nan(), numc = f64_nan(), denc = f64_nan();
    bool comp_ok = false;
    if (dxy_ok && pi_ok[0] && pi_ok[1]) {  // stats.rs:2984-3001 == 1741-1756 == 3143-3158
      if (dxy > kFstEps) {
        double num = dxy - 0.5 * (pi[0] + pi[1]);
        fst = num / dxy; numc = num; denc = dxy; comp_ok = true;
      } else {
        double pi_avg = 0.5 * (pi[0] + pi[1]);
        if (fabs(pi_avg) <= kFstEps) { numc = 0.0; denc = 0.0; comp_ok = true; }
      }
    }
    if (row_ok) {
      if (A.fst) site_store(A.fst + (out_idx), fst);
      if (A.dxy) site_store(A.dxy + (out_idx), dxy_ok ? dxy : f64_nan());
      if (A.pi1) site_store(A.pi1 + (out_idx), pi_ok[0] ? pi[0] : f64_nan());
      if (A.pi2) site_store(A.pi2 + (out_idx), pi_ok[1] ? pi[1] : f64_nan());
      if (A.num) site_store(A.num + (out_idx), numc);
      if (A.den) site_store(A.den + (out_idx), denc);
      if (comp_ok) { T.hud[5] += numc; T.hud[6] += denc; T.hud_u[1] += 1; }  // hudson_component_sums 1625-1635
      // calculate_dxy_dense 2546-2596 / sparse fold 2476-2496 always use the frequency-dot form
      if (dxy_ok) T.hud[7] += clamp01(1.0 - hud_dot); else T.hud_u[2] += 1;
      // ---- aggregate_hudson_components_from_summaries, stats.rs:1565-1620 (biallelic counts) ----
      if (n1 == 0 || n2 == 0) {
        T.hud_u[0] += 1;
      } else {
        unsigned long long a1 = t.alt[0], a2 = t.alt[1];
        unsigned long long r1 = n1 - a1, r2 = n2 - a2;
        double denom_pairs = (double)((unsigned long long)n1 * n2);
        if (denom_pairs != 0.0) {
          double d = (double)(a1 * r2 + r1 * a2) / denom_pairs;
          if (d < 0.0) d = 0.0; else if (d > 1.0) d = 1.0;
          T.hud[4] += d;
          if (n1 >= 2 && n2 >= 2) {
            double denom1 = (double)((unsigned long long)n1 * (n1 - 1));
            double denom2 = (double)((unsigned long long)n2 * (n2 - 1));
            double p1 = denom1 > 0.0 ? 2.0 * (double)a1 * (double)r1 / denom1 : 0.0;
            double p2 = denom2 > 0.0 ? 2.0 * (double)a2 * (double)r2 / denom2 : 0.0;
            T.hud[2] += p1;
            T.hud[3] += p2;
            if (d > kFstEps) { T.hud[0] += d - 0.5 * (p1 + p2); T.hud[1] += d; }
          }
        }
      }
    }
  }

  if constexpr ((MODE & kModeWc) != 0) {
    constexpr int NW = 1 + (P * (P - 1)) / 2;
    // stats.rs:1987-2031.  pop_sizes_populated == at least one allele present among ALL samples.
    const bool any_allele = t.n_all != 0;
    // one slot: state, stores, regional sums.  A pair has an entry iff both totals > 0 (the overall pass was then not
    // skipped either: valid_groups >= 2 is implied)
    auto finish = [&](int k, double a, double b, bool has) {
      uint8_t st;
      double oa, ob;
      if (any_allele && has) { st = wc_classify(a, b); oa = a; ob = b; }
      else { st = 3; oa = 0.0; ob = 0.0; }
      if (row_ok) {
        const int slot = A.wc_slot[k];
        if (slot >= 0) {
          if (A.wc_a) site_store(A.wc_a + ((size_t)slot * A.row_count + out_idx), oa);
          if (A.wc_b) site_store(A.wc_b + ((size_t)slot * A.row_count + out_idx), ob);
          if (A.wc_state) site_store(A.wc_state + ((size_t)slot * A.row_count + out_idx), st);
        }
        if constexpr (LaneTotals<P, MODE>::kWcLaneTotals) {
          if (st != 3) { T.wc_a[k] += oa; T.wc_b[k] += ob; T.wc_inf[k] += 1; }  // 2172-2203
        }
      }
      if constexpr (LaneTotals<P, MODE>::kWcLdsTotals) {  // every lane, whatever its row: the wave sums these across its lanes
        const bool counts = row_ok && st != 3;
        wc_xpose_put<P, MODE>(T, 3 * k + 0, counts ? oa : 0.0);
        wc_xpose_put<P, MODE>(T, 3 * k + 1, counts ? ob : 0.0);
        wc_xpose_put<P, MODE>(T, 3 * k + 2, counts ? 1.0 : 0.0);
      }
    };
    if constexpr (!GENERAL) {
      // biallelic: allele 0 count = n - alt, allele 1 count = alt; every slot is computed and finished in turn
      uint32_t cc[2][P];
#pragma unroll
      for (int p = 0; p < P; ++p) { cc[1][p] = t.alt[p]; cc[0][p] = t.n[p] - t.alt[p]; }
      wc_for_each_slot<P, 2, !MISSING>(A, t.n, cc, finish);
    } else if constexpr (P >= 5) {
      // eight groups, multi-allelic: the counts of alleles 0..3 (c4), every slot computed and finished in turn like the biallelic
      // case - the same per-allele terms in the same allele order as the accumulating form below
      wc_for_each_slot<P, 4, !MISSING>(A, t.n, *reinterpret_cast<const uint32_t (*)[4][P]>(c4), finish);
    } else {
      finish(0, wc.a[0], wc.b[0], true);
      int k = 1;
#pragma unroll
      for (int i = 0; i < P; ++i) {
#pragma unroll
        for (int j = i + 1; j < P; ++j) {
          finish(k, wc.a[k], wc.b[k], t.n[i] != 0 && t.n[j] != 0);
          ++k;
        }
      }
    }
    (void)NW;
  }
}

// Biallelic site: the tallies that follow from (n, alt) - allele 0 count = n - alt, allele 1 count = alt - and the
// frequency dot product of the Hudson D_xy (dxy_from_counts, stats.rs:2921-2931, ascending allele order).
template <int P, int MODE>
__device__ __forceinline__ void finish_biallelic_site(SiteTally<P>& mine, double& hud_dot) {
  uint32_t c0[P], c1[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    c1[p] = mine.alt[p];
    c0[p] = mine.n[p] - mine.alt[p];
    mine.ssq[p] = (unsigned long long)c0[p] * c0[p] + (unsigned long long)c1[p] * c1[p];
    mine.distinct[p] = (c0[p] != 0 ? 1u : 0u) + (c1[p] != 0 ? 1u : 0u);
  }
  if constexpr ((MODE & kModeHudson) != 0 && P >= 2) {
    if (mine.n[0] != 0 && mine.n[1] != 0) {
      double inv1 = 1.0 / (double)mine.n[0], inv2 = 1.0 / (double)mine.n[1];
      if (c0[0] != 0 && c0[1] != 0) hud_dot += ((double)c0[0] * inv1) * ((double)c0[1] * inv2);
      if (c1[0] != 0 && c1[1] != 0) hud_dot += ((double)c1[0] * inv1) * ((double)c1[1] * inv2);
    }
  }
  // W&C of a biallelic site is computed slot by slot inside site_epilogue (no per-site slot arrays in registers)
}

// Block reduction of the per-lane regional accumulators (fixed order: lane tree, then waves 0..3) into this block's row of
// the partial vectors.  Called once, by every thread of the block, after the tile loop.
template <int P, int MODE, int WAVES = kWavesPerBlock>
__device__ __forceinline__ void reduce_block_totals(const SweepArgs& A, LaneTotals<P, MODE>& T) {

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  __shared__ double s_f64[WAVES][kMaxF64];
  __shared__ unsigned long long s_u64[WAVES][kMaxU64];
  auto put_f64 = [&](int slot, double v) { v = wave_sum(v); if (lane == 0) s_f64[wave][slot] = v; };
  auto put_u64 = [&](int slot, unsigned long long v) { v = wave_sum(v); if (lane == 0) s_u64[wave][slot] = v; };
  if (lane < kMaxF64) s_f64[wave][lane] = 0.0;
  if (lane < kMaxU64) s_u64[wave][lane] = 0;
  __syncthreads();
  if constexpr ((MODE & kModeWc) != 0) {
    if constexpr (LaneTotals<P, MODE>::kWcLaneTotals) {
      constexpr int NW = 1 + (P * (P - 1)) / 2;
#pragma unroll
      for (int k = 0; k < NW; ++k) { put_f64(kOffWcA + k, T.wc_a[k]); put_f64(kOffWcB + k, T.wc_b[k]); put_u64(kOffWcInf + k, T.wc_inf[k]); }
    }
    if constexpr (LaneTotals<P, MODE>::kWcLdsTotals) {  // lane (r, 0) holds this wave's sum of transposition row 16 B + r
#pragma unroll
      for (int B = 0; B < LaneTotals<P, MODE>::kWcBatches; ++B) {
        const int row = B * kWcXRows + (lane >> 2);
        if ((lane & 3) == 0 && row < LaneTotals<P, MODE>::kWcXposeRows) {
          const double v = wc_xpose_scratch()[wave * kWcXWave + kWcXRows * kWcXStride + row];
          const int k = row / 3, c = row - 3 * k;
          if (c == 0) s_f64[wave][kOffWcA + k] = v;
          else if (c == 1) s_f64[wave][kOffWcB + k] = v;
          else s_u64[wave][kOffWcInf + k] = (unsigned long long)v;  // a count of sites as a sum of 1.0s: exact
        }
      }
    }
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p) { put_f64(kOffPopF64 + p, T.pop_pi[p]); put_u64(kOffPopSeg + p, T.pop_seg[p]); put_u64(kOffPopUnc + p, T.pop_unc[p]); }
    if constexpr ((MODE & kModeHudson) != 0) {
#pragma unroll
      for (int i = 0; i < kHudF64; ++i) put_f64(kOffHudF64 + i, T.hud[i]);
#pragma unroll
      for (int i = 0; i < kHudU64; ++i) put_u64(kOffHudU64 + i, T.hud_u[i]);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kMaxF64 + kMaxU64; i += WAVES * 64) {  // (one pass for workgroups of two waves and more)
    if (i < kMaxF64) {
      double v = 0.0;
      for (int w = 0; w < WAVES; ++w) v += s_f64[w][i];
      A.part_f64[(size_t)blockIdx.x * kMaxF64 + i] = v;
    } else {
      unsigned long long v = 0;
      for (int w = 0; w < WAVES; ++w) v += s_u64[w][i - kMaxF64];
      A.part_u64[(size_t)blockIdx.x * kMaxU64 + (i - kMaxF64)] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// the sweep kernel
// ------------------------------------------------------------------------------------------------
// MM = where the membership masks live (kMaskLdsBytes / kMaskGlobalBytes / kMaskLdsBits, see mask_vec).
// NPL = bit planes of a packed multi-allelic matrix (2: alleles up to 3, 3: up to 7); ignored elsewhere.
// kernels that defer their epilogues (kDefer below) are held to three waves per SIMD: left alone, the register allocator keeps the hoisted
// invariants of the epilogue loop alive across the counting loop (233 VGPRs, two waves), and occupancy is worth more than that (2 -> 3 waves: 16 %)
template <int P, int MODE, bool MISSING, bool GENERAL, int MM, int LPR>
constexpr int sweep_min_waves() { return (MM == 3 /* kMaskPacked */ && !GENERAL && !MISSING && P <= 2 && LPR == 16 && (MODE & kModeWc) == 0) ? 3 : 1; }
// __launch_bounds__' second argument (workgroups of four waves per CU = waves per SIMD): the eight-group W&C kernels of a packed matrix
// (biallelic, and multi-allelic on two planes) end one register above 256 when left alone - one wave per SIMD instead of two; the
// three-plane ones would spill to scratch under that bound and are left alone
// Kernels whose register count sat just above an occupancy step (512 VGPRs per SIMD: 128 -> four waves, 168 -> three) are held to the step
// when hipcc reaches it without spilling (tools/kernel_resources.py; the CPU suite fails on any scratch).  Round 4 found the eight-group
// summaries kernel of sixteen-lane rows - the counting sweeps of more than eight W&C groups - at 171 VGPRs, two waves per SIMD, streaming its
// planes at 2.5 TB/s; at 133 VGPRs and three waves it streams at 3.75 TB/s (profiles/r04/occupancy_steps_*).  FMH_OCC_STEPS=0: the rules off.
#ifndef FMH_OCC_STEPS
#define FMH_OCC_STEPS 1
#endif
template <int P, int MODE, bool GENERAL, int MM, int NPL, int LPR = 16, bool MISSING = false>
constexpr int sweep_min_blocks() {
  if (FMH_OCC_STEPS && MM == 3 /* kMaskPacked */ && !GENERAL && !MISSING && NPL == 2) {
    if (P == 8 && MODE == kModeSummary && LPR == 16) return 3;  // 171 -> 133
    if (P == 4 && MODE == kModeWc && LPR == 16) return 3;       // 171 -> 147
    if (P == 6 && MODE == kModeWc && LPR == 4) return 3;        // 170
    if (P == 5 && MODE == kModeWc && LPR == 16) return 3;       // 173
    // (two groups on four-lane rows, 131-135 VGPRs: the diversity and W&C kernels spill two registers at 128, and one and two groups take the
    // pipelined kernel there anyway)
  }
  // (the complete-row fast path of the kernels with a called plane - row_gap - added ten registers: these three crossed the step at 128)
  if (FMH_OCC_STEPS && MM == 3 && !GENERAL && MISSING && NPL == 2 && P == 2) {
    if (LPR == 4 && (MODE == (kModeSummary | kModeHudson) || MODE == (kModeSummary | kModeHudson | kModeDiversity))) return 4;  // 129
    if (LPR == 16 && MODE == kModeWc) return 4;                                                                                 // 130
  }
  return (MM == 3 /* kMaskPacked */ && P >= 5 && (MODE & kModeWc) != 0 && NPL == 2) ? 2 : 1;
}

// Which kernels defer their epilogues (see sweep_kernel), how many u32 they park per site and how many tiles deep (LDS per workgroup =
// 4 waves x depth x 64 sites x values x 4 B: 16 or 32 KiB).  The host sizes the dynamic LDS with the same functions (defer_lds_bytes).
// ONE rule for the kernel template and the host (enqueue_sweep sizes the parked counts' LDS with it): which kernels have the deferring tile loop
constexpr bool defer_rule(int P, int mode, bool missing, bool general, int lpr) {
  // sixteen-lane rows only: on four-lane rows (at most 4 096 columns) deferral was level or behind at 1 000 haplotypes and 3-7 % ahead on long
  // launches at 2 500, while the restructured loop itself cost the C2 kernel (1 M x 1 000) 7 % - those kernels keep the plain tile loop
  // (round 3 repeated the four-lane experiment with the old loop structure behind this rule: depths 2, 4, 8 against 1 on W&C, summaries
  // and Hudson sweeps of 1 000 and 2 500 haplotypes all within +-2 %, profiles/r03/defer_four_lane_rows.jsonl)
  return !general && lpr == 16 && (P <= 2 ? (mode & kModeWc) == 0 || !missing : P == 4 && !missing);
}
template <int P, int MODE, bool MISSING, bool GENERAL, int MM, int LPR>
constexpr bool defer_kernel() { return defer_rule(P, MODE, MISSING, GENERAL, LPR); }
template <int P, int MODE, bool MISSING>
constexpr int defer_values() { return MISSING ? 2 * P + ((MODE & kModeWc) != 0 ? 1 : 0) : P; }
template <int P, int MODE, bool MISSING>
constexpr int defer_depth() { return defer_values<P, MODE, MISSING>() <= 2 ? kDeferTiles : kDeferTiles / 2; }
inline int defer_depth_host(int P, int mode, bool missing) {
  const int k = missing ? 2 * P + ((mode & kModeWc) != 0 ? 1 : 0) : P;
  return k <= 2 ? kDeferTiles : kDeferTiles / 2;
}
inline size_t defer_lds_bytes(int P, int mode, bool missing, int depth) {
  const int k = missing ? 2 * P + ((mode & kModeWc) != 0 ? 1 : 0) : P;
  return (size_t)kWavesPerBlock * depth * 64 * k * 4;
}

template <int P, int MODE, bool MISSING, bool GENERAL, int MM = kMaskLdsBytes, int LPR = 16, int NPL = 2>
__global__ __launch_bounds__(kBlock, (sweep_min_blocks<P, MODE, GENERAL, MM, NPL, LPR, MISSING>())) void sweep_kernel(const SweepArgs A) {
  static_assert(NPL == 2 || (NPL == 3 && GENERAL && MM == kMaskPacked), "a third plane exists on packed multi-allelic matrices only");
  static_assert(LPR == 16 || ((LPR == 4 || LPR == 8) && MM == kMaskPacked), "four / eight lanes per row exist for the packed cores only");
  extern __shared__ __align__(16) unsigned char smem[];
  const MatrixView mv = A.mv;
  const uint32_t nvec = mv.nvec;
  uint32_t nvec_pad = A.nvec_pad;
  const void* lds_mask;
  if constexpr (MM == kMaskGlobalBytes) {
    lds_mask = A.masks;
    nvec_pad = (uint32_t)(A.mask_pitch / 16);  // the stride between the masks of two groups, in vectors
  } else if constexpr (MM == kMaskPacked) {
    // the groups' bit masks as 16-byte vectors (128 columns each), zero beyond the row
    uint4* staged = reinterpret_cast<uint4*>(smem);
    for (uint32_t i = threadIdx.x; i < (uint32_t)P * nvec_pad; i += kBlock) {
      const uint32_t p = i / nvec_pad, v = i - p * nvec_pad;
      staged[i] = v < nvec ? *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(A.mask_bits) + (size_t)p * (A.mask_pitch / 8) + (size_t)v * 16)
                           : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    lds_mask = staged;
  } else if constexpr (MM == kMaskLdsBits) {
    // one 16-bit word per vector and group, zero beyond the row (the host lays them out with the same padded stride)
    uint16_t* staged = reinterpret_cast<uint16_t*>(smem);
    for (uint32_t i = threadIdx.x; i < (uint32_t)P * nvec_pad; i += kBlock) {
      const uint32_t p = i / nvec_pad, v = i - p * nvec_pad;
      staged[i] = v < nvec ? A.mask_bits[(size_t)p * (A.mask_pitch / 16) + v] : (uint16_t)0;
    }
    __syncthreads();
    lds_mask = staged;
  } else {
    // stage the P membership masks into LDS (16 B per thread per step), zero-padded to nvec_pad
    uint4* staged = reinterpret_cast<uint4*>(smem);
    for (uint32_t i = threadIdx.x; i < (uint32_t)P * nvec_pad; i += kBlock) {
      const uint32_t p = i / nvec_pad, v = i - p * nvec_pad;
      staged[i] = v < nvec ? load_vec(A.masks + (size_t)p * A.mask_pitch + (size_t)v * 16) : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    lds_mask = staged;
  }

  if constexpr ((MODE & kModeWc) != 0 && !MISSING) {
    wc_rcp_init<P>(A);
    wc_shape_init<P>(A);
    __syncthreads();
  }
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int grp = lane / LPR;   // the wave's 64 / LPR groups take LPR rows of the tile each, one row per step
  const int gl = lane % LPR;
  constexpr bool NEED_ALL = (MODE & kModeWc) != 0;

  LaneTotals<P, MODE> T;
  T.clear();

  const size_t ntiles = (A.row_count + kTileRows - 1) / kTileRows;
  const size_t tile_stride = (size_t)gridDim.x * kWavesPerBlock;

  // Deferred epilogues (packed biallelic kernels: one or two groups with or without missing calls, four groups without; defer_kernel()).
  // A wave counts `defer_tiles` of its tiles in a row, parking the per-site counts in LDS (4 bytes per site and value), and only then runs
  // their epilogues: its track stores leave in one burst of defer_tiles x 7 instructions instead of 7 after every tile.  The tiles, their
  // order per lane and every per-site operation are unchanged (same bits, same regional sums); only WHEN a wave writes changes.  Found with the
  // do-nothing kernel of tools/microbench/store_bursts.hip (modes 20 / 21: the sweep's reads with one row in flight per lane group, stores per
  // tile vs per eight tiles: 1.256 -> 1.150 ms at 16 waves per CU) after larger CONTIGUOUS bursts had turned out not to be the point (modes 2,
  // 3, 18 vs 19).  C4 sweep, same process: 1.289 ms undeferred, 1.175 ms sixteen tiles deep (DESIGN.md section 3 "Deferred epilogues").
  // The prefetching row loop is kept on four-lane rows only: its variants for sixteen-lane rows cost this loop structure 40 VGPRs (two waves
  // per SIMD instead of three) for a path that long launches had stopped taking anyway.
  constexpr bool kDefer = defer_kernel<P, MODE, MISSING, GENERAL, MM, LPR>();
  if constexpr (kDefer) {
    {
      // (the only tile loop of these kernels: defer_tiles = 1 is the undeferred order, through the same code)
      constexpr int K = defer_values<P, MODE, MISSING>();   // parked u32 per site: alt per group (+ called per group, + called in all, with missing calls)
      constexpr int D = defer_depth<P, MODE, MISSING>();    // tiles a wave can park
      const int cap = A.defer_cap < 1 ? 1 : (A.defer_cap > D ? D : A.defer_cap);
      const int ch = A.defer_tiles < 1 ? 1 : (A.defer_tiles > cap ? cap : A.defer_tiles);
      const uint4* lm = reinterpret_cast<const uint4*>(lds_mask);
      uint32_t* park = reinterpret_cast<uint32_t*>(smem + A.defer_offset) + (size_t)wave * cap * 64 * K;
      for (size_t tile0 = (size_t)blockIdx.x * kWavesPerBlock + wave; tile0 < ntiles; tile0 += tile_stride * ch) {
#pragma unroll 1
        for (int b = 0; b < ch; ++b) {  // counts
          const size_t tile = tile0 + (size_t)b * tile_stride;
          if (tile >= ntiles) break;
          const size_t tile_row0 = tile * kTileRows;
          uint32_t alt_mine[P], n_mine[P], n_all_mine = 0;
#pragma unroll
          for (int p = 0; p < P; ++p) { alt_mine[p] = 0; n_mine[p] = 0; }
          bool counted = false;
          if constexpr (!MISSING && LPR != 16 && MM == kMaskPacked) {
            if (A.single_trip) {  // four-lane rows keep the prefetching row loop (level or 1-6 % ahead at every launch size)
              if (A.unroll == 5) tile_rows_packed_prefetch<P, 5, LPR>(A, mv, lm, nvec_pad, tile_row0, grp, gl, alt_mine);
              else if (A.unroll == 3) tile_rows_packed_prefetch<P, 3, LPR>(A, mv, lm, nvec_pad, tile_row0, grp, gl, alt_mine);
              else if (A.unroll == 2) tile_rows_packed_prefetch<P, 2, LPR>(A, mv, lm, nvec_pad, tile_row0, grp, gl, alt_mine);
              else tile_rows_packed_prefetch<P, 1, LPR>(A, mv, lm, nvec_pad, tile_row0, grp, gl, alt_mine);
              counted = true;
            }
          }
          // rows every column of which is called do not read their called plane (MatrixView::row_gap); a step of such rows runs the
          // no-missing core and takes the group sizes as its called counts - the values the popcounts would give
          unsigned long long gap_mask = ~0ull;
          if constexpr (MISSING && MM == kMaskPacked) {
            if (mv.row_gap) {
              const size_t r = tile_row0 + (size_t)lane;
              const uint8_t f = r < A.row_count ? mv.row_gap[A.row_begin + r] : (uint8_t)0;
              gap_mask = __ballot(f != 0);
            }
          }
          if (!counted) {
            for (int s = 0; s < LPR; ++s) {
              const size_t rel = tile_row0 + (size_t)grp * LPR + s;
              const size_t row = A.row_begin + (rel < A.row_count ? rel : A.row_count - 1);
              const uint8_t* row_ptr = mv.data + row * mv.pitch;
              constexpr unsigned long long kStepRowsD = LPR == 16 ? 0x0001000100010001ull : (LPR == 8 ? 0x0101010101010101ull : 0x1111111111111111ull);
              const bool step_gap = (gap_mask & (kStepRowsD << s)) != 0;           // wave-uniform
              const bool my_gap = ((gap_mask >> (grp * LPR + s)) & 1ull) != 0;     // this lane group's row
              const uint8_t* bits_ptr = MISSING && my_gap ? mv.bits + row * mv.bits_pitch : nullptr;
              uint32_t n[P], n_all, aor, sp[P][1];
              bool counted_full = false;
              if constexpr (MISSING && MM == kMaskPacked) {
                if (!step_gap) {
#define FMH_COUNT_FULL(UV) count_row_packed<P, false, false, 1, UV, LPR>(mv, lm, nvec_pad, row_ptr, nullptr, nullptr, nullptr, gl, n, n_all, aor, sp)
                  if constexpr (LPR != 16) {
                    if (A.unroll == 5) FMH_COUNT_FULL(5);
                    else if (A.unroll == 3) FMH_COUNT_FULL(3);
                    else if (A.unroll == 2) FMH_COUNT_FULL(2);
                    else FMH_COUNT_FULL(1);
                  } else {
                    if (A.unroll == 4) FMH_COUNT_FULL(4);
                    else if (A.unroll == 3) FMH_COUNT_FULL(3);
                    else FMH_COUNT_FULL(2);
                  }
#undef FMH_COUNT_FULL
#pragma unroll
                  for (int p = 0; p < P; ++p) n[p] = A.group_size[p];
                  n_all = mv.columns;
                  counted_full = true;
                }
              }
              if (!counted_full) {
#define FMH_COUNT_DEFER(UV) count_row_packed<P, MISSING, NEED_ALL, 1, UV, LPR>(mv, lm, nvec_pad, row_ptr, nullptr, nullptr, bits_ptr, gl, n, n_all, aor, sp)
              if constexpr (MM != kMaskPacked) {  // u8 rows: the dot4 core
                uint32_t alt[P];
                if (A.unroll == 8) count_row_biallelic<P, MISSING, NEED_ALL, 8, MM>(mv, lds_mask, nvec_pad, row_ptr, bits_ptr, gl, alt, n, n_all);
                else count_row_biallelic<P, MISSING, NEED_ALL, 4, MM>(mv, lds_mask, nvec_pad, row_ptr, bits_ptr, gl, alt, n, n_all);
#pragma unroll
                for (int p = 0; p < P; ++p) sp[p][0] = alt[p];
              } else if constexpr (LPR != 16) {
                if (A.unroll == 5) FMH_COUNT_DEFER(5);
                else if (A.unroll == 3) FMH_COUNT_DEFER(3);
                else if (A.unroll == 2) FMH_COUNT_DEFER(2);
                else FMH_COUNT_DEFER(1);
              } else {
                if (A.unroll == 4) FMH_COUNT_DEFER(4);
                else if (A.unroll == 3) FMH_COUNT_DEFER(3);
                else FMH_COUNT_DEFER(2);
              }
              }  // !counted_full
#undef FMH_COUNT_DEFER
              if (gl == s) {
#pragma unroll
                for (int p = 0; p < P; ++p) { alt_mine[p] = sp[p][0]; if (MISSING) n_mine[p] = n[p]; }
                if (MISSING && NEED_ALL) n_all_mine = n_all;
              }
            }
          }
          uint32_t* slot = park + ((size_t)b * 64 + lane) * K;
#pragma unroll
          for (int p = 0; p < P; ++p) { slot[p] = alt_mine[p]; if (MISSING) slot[P + p] = n_mine[p]; }
          if (MISSING && NEED_ALL) slot[2 * P] = n_all_mine;
        }
#pragma unroll 1
        for (int b = 0; b < ch; ++b) {  // epilogues and stores of the same tiles, back to back
          const size_t tile = tile0 + (size_t)b * tile_stride;
          if (tile >= ntiles) break;
          SiteTally<P> mine;
          WcSite<P> wc;
          double hud_dot = 0.0;
          const uint32_t* slot = park + ((size_t)b * 64 + lane) * K;
#pragma unroll
          for (int p = 0; p < P; ++p) {
            // the group size is re-read opaquely per tile: as a visible loop invariant, everything the epilogue derives from it (f64 reciprocals,
            // products) is hoisted out of this loop AND kept alive across the counting loop above - 233 VGPRs instead of 160
            uint32_t gs = A.group_size[p];
            asm volatile("" : "+s"(gs));
            mine.n[p] = MISSING ? slot[P + p] : gs;
            mine.alt[p] = slot[p];
            mine.distinct[p] = 0;
            mine.ssq[p] = 0;
          }
          uint32_t cols = mv.columns;
          asm volatile("" : "+s"(cols));
          mine.n_all = MISSING ? (NEED_ALL ? slot[2 * P] : 0u) : cols;
          if constexpr ((MODE & kModeWc) != 0) {
            constexpr int NW = 1 + (P * (P - 1)) / 2;
#pragma unroll
            for (int k = 0; k < NW; ++k) { wc.a[k] = 0.0; wc.b[k] = 0.0; }
          }
          finish_biallelic_site<P, MODE>(mine, hud_dot);
          const size_t my_rel = tile * kTileRows + lane;
          site_epilogue<P, MODE, MISSING, GENERAL>(A, my_rel, my_rel < A.row_count, mine, hud_dot, wc, T);
        }
      }
      reduce_block_totals<P, MODE>(A, T);
      return;
    }
  }

  for (size_t tile = (size_t)blockIdx.x * kWavesPerBlock + wave; tile < ntiles; tile += tile_stride) {
    const size_t tile_row0 = tile * kTileRows;  // relative to row_begin
    SiteTally<P> mine;
    WcSite<P> wc;
    double hud_dot = 0.0;
#pragma unroll
    for (int p = 0; p < P; ++p) { mine.n[p] = 0; mine.alt[p] = 0; mine.distinct[p] = 0; mine.ssq[p] = 0; }
    mine.n_all = 0;
    if constexpr ((MODE & kModeWc) != 0) {
      constexpr int NW = 1 + (P * (P - 1)) / 2;
#pragma unroll
      for (int k = 0; k < NW; ++k) { wc.a[k] = 0.0; wc.b[k] = 0.0; }
    }

    bool rows_done = false;
    if constexpr (MM == kMaskPacked && !GENERAL && !MISSING) {
      if (A.single_trip) {
        const uint4* lm = reinterpret_cast<const uint4*>(lds_mask);
        uint32_t alt_mine[P];
#pragma unroll
        for (int p = 0; p < P; ++p) alt_mine[p] = 0;
        if constexpr (LPR != 16) {
          if (A.unroll == 5) tile_rows_packed_prefetch<P, 5, LPR>(A, mv, lm, nvec_pad, tile_row0, grp, gl, alt_mine);
          else if (A.unroll == 3) tile_rows_packed_prefetch<P, 3, LPR>(A, mv, lm, nvec_pad, tile_row0, grp, gl, alt_mine);
          else if (A.unroll == 2) tile_rows_packed_prefetch<P, 2, LPR>(A, mv, lm, nvec_pad, tile_row0, grp, gl, alt_mine);
          else tile_rows_packed_prefetch<P, 1, LPR>(A, mv, lm, nvec_pad, tile_row0, grp, gl, alt_mine);
        } else {
          if (A.unroll == 4) tile_rows_packed_prefetch<P, 4, 16>(A, mv, lm, nvec_pad, tile_row0, grp, gl, alt_mine);
          else if (A.unroll == 3) tile_rows_packed_prefetch<P, 3, 16>(A, mv, lm, nvec_pad, tile_row0, grp, gl, alt_mine);
          else tile_rows_packed_prefetch<P, 2, 16>(A, mv, lm, nvec_pad, tile_row0, grp, gl, alt_mine);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) { mine.alt[p] = alt_mine[p]; mine.n[p] = A.group_size[p]; }
        mine.n_all = mv.columns;
        rows_done = true;
      }
    }
    // planes the counting core separates on the GENERAL path: NPL on a packed matrix, bits 0 and 1 on byte rows
    constexpr int NPLK = !GENERAL ? 1 : (MM == kMaskPacked ? NPL : 2);
    constexpr int NS = (1 << NPLK) - 1;
    uint32_t my_s[P][NS];   // subset sums of the row this lane owns (GENERAL)
    uint32_t my_or = 0;     // OR of its called allele values
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int k = 0; k < NS; ++k) my_s[p][k] = 0;
    // packed cores: the batch depth U is the launch's (A.unroll); with eight groups only the shallow batches are built in this row loop -
    // deeper ones kept P x U mask vectors live and spilled to scratch (the one-batch-per-row loop above takes any depth: its masks stay in LDS)
    constexpr bool kShallow = P >= 5;
    // Multi-allelic packed matrices: real cohorts carry an allele above 1 at a few sites in a hundred, yet every row used to pay for
    // every plane (two planes: twice the bytes of a biallelic sweep; 0.2 % such sites: 2.0 x the time).  row_hi says which rows of the
    // tile have a bit in an upper plane; a step none of whose rows has one runs the one-plane core (its upper subset sums are zero), a
    // step with some reads the upper planes of those rows only.
    unsigned long long hi_mask = ~0ull;
    if constexpr (GENERAL && MM == kMaskPacked) {
      if (mv.row_hi) {
        const size_t r = tile_row0 + (size_t)lane;
        const uint8_t f = r < A.row_count ? mv.row_hi[A.row_begin + r] : (uint8_t)0;
        hi_mask = __ballot(f != 0);
      }
    }
    // the same for the called plane: rows every column of which is called (MatrixView::row_gap) do not read it
    unsigned long long gap_mask = ~0ull;
    if constexpr (MISSING && MM == kMaskPacked) {
      if (mv.row_gap) {
        const size_t r = tile_row0 + (size_t)lane;
        const uint8_t f = r < A.row_count ? mv.row_gap[A.row_begin + r] : (uint8_t)0;
        gap_mask = __ballot(f != 0);
      }
    }
    for (int s = 0; s < LPR && !rows_done; ++s) {
      const size_t rel = tile_row0 + (size_t)grp * LPR + s;
      const bool row_ok = rel < A.row_count;
      const size_t row = A.row_begin + (row_ok ? rel : A.row_count - 1);  // rows past the end are clamped to the last row (their results are discarded by row_ok)
      const uint8_t* row_ptr = mv.data + row * mv.pitch;
      const bool my_gap = ((gap_mask >> (grp * LPR + s)) & 1ull) != 0;
      const uint8_t* bits_ptr = MISSING && my_gap ? mv.bits + row * mv.bits_pitch : nullptr;
      const bool own = gl == s;
      uint32_t n[P], n_all, aor, sp[P][NS];
      if constexpr (MM == kMaskPacked) {
        const uint4* lm = reinterpret_cast<const uint4*>(lds_mask);
        constexpr unsigned long long kStepRows = LPR == 16 ? 0x0001000100010001ull : (LPR == 8 ? 0x0101010101010101ull : 0x1111111111111111ull);
        const bool step_hi = (hi_mask & (kStepRows << s)) != 0;                 // wave-uniform: some row of this step has an upper-plane bit
        const bool row_hi = ((hi_mask >> (grp * LPR + s)) & 1ull) != 0;         // this lane group's row
        const uint8_t* row_ptr1 = NPLK >= 2 && row_hi ? mv.data1 + row * mv.pitch : nullptr;
        const uint8_t* row_ptr2 = NPLK >= 3 && row_hi ? mv.data2 + row * mv.pitch : nullptr;
        constexpr bool NA_ALL = GENERAL || NEED_ALL;
        bool counted_low = false;
        if constexpr (MISSING && !GENERAL) {
          if ((gap_mask & (kStepRows << s)) == 0) {  // no row of this step has an uncalled column: the no-missing core, called counts = group sizes
#define FMH_COUNT_FULL(UV) count_row_packed<P, false, false, 1, UV, LPR>(mv, lm, nvec_pad, row_ptr, nullptr, nullptr, nullptr, gl, n, n_all, aor, sp, P == 8 ? A.n_groups : P)
            if constexpr (LPR != 16) {
              if constexpr (kShallow) { if (A.unroll == 2) FMH_COUNT_FULL(2); else FMH_COUNT_FULL(1); }
              else if (A.unroll == 5) FMH_COUNT_FULL(5);
              else if (A.unroll == 3) FMH_COUNT_FULL(3);
              else if (A.unroll == 2) FMH_COUNT_FULL(2);
              else FMH_COUNT_FULL(1);
            } else {
              if constexpr (kShallow) FMH_COUNT_FULL(2);
              else if (A.unroll == 4) FMH_COUNT_FULL(4);
              else if (A.unroll == 3) FMH_COUNT_FULL(3);
              else FMH_COUNT_FULL(2);
            }
#undef FMH_COUNT_FULL
#pragma unroll
            for (int p = 0; p < P; ++p) n[p] = A.group_size[p];
            n_all = mv.columns;
            counted_low = true;
          }
        }
        if constexpr (GENERAL && NPLK >= 2) {
          if (!step_hi) {  // the one-plane core; allele 1 may be present (aor = 1: an allele a site does not carry adds exact zeros)
            uint32_t low[P][1], aor_low;
#define FMH_COUNT_LOW(UV) count_row_packed<P, MISSING, NA_ALL, 1, UV, LPR>(mv, lm, nvec_pad, row_ptr, nullptr, nullptr, bits_ptr, gl, n, n_all, aor_low, low, P == 8 ? A.n_groups : P)
            if constexpr (LPR != 16) {
              if constexpr (kShallow) { if (A.unroll == 2) FMH_COUNT_LOW(2); else FMH_COUNT_LOW(1); }
              else if (A.unroll == 5) FMH_COUNT_LOW(5);
              else if (A.unroll == 3) FMH_COUNT_LOW(3);
              else if (A.unroll == 2) FMH_COUNT_LOW(2);
              else FMH_COUNT_LOW(1);
            } else {
              if constexpr (kShallow) FMH_COUNT_LOW(2);
              else if (A.unroll == 4) FMH_COUNT_LOW(4);
              else if (A.unroll == 3) FMH_COUNT_LOW(3);
              else FMH_COUNT_LOW(2);
            }
#undef FMH_COUNT_LOW
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
              for (int k = 0; k < NS; ++k) sp[p][k] = k == 0 ? low[p][0] : 0u;
            }
            aor = 1;
            counted_low = true;
          }
        }
        if (!counted_low) {
#define FMH_COUNT_PACKED(UV) count_row_packed<P, MISSING, NA_ALL, NPLK, UV, LPR>(mv, lm, nvec_pad, row_ptr, row_ptr1, row_ptr2, bits_ptr, gl, n, n_all, aor, sp, P == 8 ? A.n_groups : P)
        if constexpr (LPR != 16) {
          if constexpr (kShallow) { if (A.unroll == 2) FMH_COUNT_PACKED(2); else FMH_COUNT_PACKED(1); }
          else if (A.unroll == 5) FMH_COUNT_PACKED(5);
          else if (A.unroll == 3) FMH_COUNT_PACKED(3);
          else if (A.unroll == 2) FMH_COUNT_PACKED(2);
          else FMH_COUNT_PACKED(1);
        } else {
          if constexpr (kShallow) FMH_COUNT_PACKED(2);
          else if (A.unroll == 4) FMH_COUNT_PACKED(4);
          else if (A.unroll == 3) FMH_COUNT_PACKED(3);
          else FMH_COUNT_PACKED(2);
        }
        }  // !counted_low
#undef FMH_COUNT_PACKED
      } else if constexpr (!GENERAL) {
        uint32_t alt[P];
        if (A.unroll == 8) count_row_biallelic<P, MISSING, NEED_ALL, 8, MM>(mv, lds_mask, nvec_pad, row_ptr, bits_ptr, gl, alt, n, n_all);
        else count_row_biallelic<P, MISSING, NEED_ALL, 4, MM>(mv, lds_mask, nvec_pad, row_ptr, bits_ptr, gl, alt, n, n_all);
#pragma unroll
        for (int p = 0; p < P; ++p) sp[p][0] = alt[p];
        aor = 0;
      } else {
        count_row_planes<P, MISSING, 4, MM>(mv, lds_mask, nvec_pad, row_ptr, bits_ptr, gl, n, n_all, aor, sp);
      }
      if (own) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
          mine.n[p] = MISSING ? n[p] : A.group_size[p];
          if constexpr (!GENERAL) mine.alt[p] = sp[p][0];
#pragma unroll
          for (int k = 0; k < NS; ++k) my_s[p][k] = sp[p][k];
        }
        mine.n_all = MISSING ? n_all : mv.columns;
        my_or = aor;
      }
    }

    // W&C with eight groups on the general path: no per-slot accumulators across the allele loop (29 slots = 116 registers); the counts of
    // alleles 0..3 are kept instead and site_epilogue finishes slot by slot (the host sends max_allele > 3 to the counts route)
    constexpr bool kWc8 = GENERAL && P >= 5 && (MODE & kModeWc) != 0;
    uint32_t c4[kWc8 ? 4 : 1][P];
    if constexpr (GENERAL) {
      // The alleles of the tile's 64 sites are consumed here, one site per lane (all 64 lanes busy), from the subset sums each
      // lane kept of its own row.  `bound` is wave-uniform: the OR of the called allele values over the tile, capped by the matrix's
      // max_allele; an allele a site does not carry contributes exact zeros to every accumulator.
      uint32_t bound = my_or;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) bound |= __shfl_xor(bound, off, 64);
      bound = __builtin_amdgcn_readfirstlane(bound);
      if (bound > (uint32_t)A.max_allele) bound = (uint32_t)A.max_allele;
      const size_t my_row = tile_row0 + lane;
      const bool my_ok = my_row < A.row_count;
      double inv1 = 0.0, inv2 = 0.0;
      uint32_t shared_k = 0;  // byte rows with alleles beyond the planes: alleles both populations carry at this lane's site
      const int hud_formula = A.hudson_formula_p1 ? A.hudson_formula_p1 - 1 : A.formula;  // as site_epilogue
      (void)shared_k; (void)hud_formula;
      auto consume = [&](uint32_t a, const uint32_t (&c)[P], bool mine_ok, size_t out_row) {
        if (A.acounts && mine_ok) {
#pragma unroll
          for (int p = 0; p < P; ++p)
            if (p < A.n_groups) A.acounts[((size_t)a * A.acounts_groups + A.acounts_group0 + p) * A.row_count + out_row] = c[p];
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
          mine.ssq[p] += (unsigned long long)c[p] * c[p];
          mine.distinct[p] += c[p] != 0 ? 1u : 0u;
          if (a == 1) mine.alt[p] = c[p];
        }
        if constexpr ((MODE & kModeHudson) != 0 && P >= 2) {
          // dxy_from_counts 2921-2931 (ascending allele order); a zero count adds +0.0
          if (c[0] != 0 && c[1] != 0) hud_dot += ((double)c[0] * inv1) * ((double)c[1] * inv2);
        }
        if constexpr ((MODE & kModeWc) != 0 && !kWc8) {
          // the reference iterates only alleles present among all samples; an absent allele
          // contributes exact zeros, so iterating it is harmless
          uint32_t cc[1][P];
#pragma unroll
          for (int p = 0; p < P; ++p) cc[0][p] = c[p];
          wc_add_alleles<P, 1, !MISSING>(A, mine.n, cc, wc);
        }
      };
      if (bound < (1u << NPLK)) {
        if constexpr ((MODE & kModeHudson) != 0 && P >= 2) {
          if (mine.n[0] != 0) inv1 = 1.0 / (double)mine.n[0];
          if (mine.n[1] != 0) inv2 = 1.0 / (double)mine.n[1];
        }
        for (uint32_t a = 0; a <= bound; ++a) {
          uint32_t c[P];
#pragma unroll
          for (int p = 0; p < P; ++p) c[p] = allele_count_from_planes<NPLK>(a, mine.n[p], my_s[p]);
          consume(a, c, my_ok, my_row);
        }
        if constexpr (kWc8) {
#pragma unroll
          for (uint32_t a = 0; a < 4; ++a)
#pragma unroll
            for (int p = 0; p < P; ++p) c4[a][p] = a <= bound ? allele_count_from_planes<NPLK>(a, mine.n[p], my_s[p]) : 0u;
        }
        if constexpr ((MODE & kModeHudson) != 0 && P >= 2) {
          // dense formula set: the reference adds the products in first-occurrence order (see first_member_column); sites where three or more
          // alleles are shared get their dot product again in that order
          if (hud_formula == kFormulaDense && bound >= 2) {
            constexpr uint32_t NA = 1u << NPLK;
            uint32_t shared = 0;
#pragma unroll
            for (uint32_t a = 0; a < NA; ++a) {
              if (a <= bound && allele_count_from_planes<NPLK>(a, mine.n[0], my_s[0]) != 0 && allele_count_from_planes<NPLK>(a, mine.n[1], my_s[1]) != 0) shared |= 1u << a;
            }
            const uint32_t driver = mine.distinct[0] <= mine.distinct[1] ? 0u : 1u;  // used1.len() <= used2.len()
            unsigned long long todo = __ballot(my_ok && __builtin_popcount(shared) >= 3);
            while (todo) {  // wave-uniform
              const int L = __builtin_ctzll(todo);
              todo &= todo - 1;
              const uint32_t sh = (uint32_t)__builtin_amdgcn_readlane((int)shared, L);
              const int d = __builtin_amdgcn_readlane((int)driver, L);
              const size_t frow = A.row_begin + tile_row0 + (size_t)L;
              uint32_t fp[NA];
#pragma unroll
              for (uint32_t a = 0; a < NA; ++a)
                fp[a] = ((sh >> a) & 1u) ? first_member_column<MM, MISSING, NPLK>(mv, lds_mask, nvec_pad, frow, d, a) : 0xFFFFFFFFu;
              double dot = 0.0;  // every lane adds its own site's products in site L's order; only lane L keeps the sum
              uint32_t left = sh;
              while (left) {
                uint32_t bestv = 0xFFFFFFFFu, besta = 0;
#pragma unroll
                for (uint32_t a = 0; a < NA; ++a)
                  if (((left >> a) & 1u) && fp[a] <= bestv) { bestv = fp[a]; besta = a; }
                left &= ~(1u << besta);
                const uint32_t c0 = allele_count_from_planes<NPLK>(besta, mine.n[0], my_s[0]), c1 = allele_count_from_planes<NPLK>(besta, mine.n[1], my_s[1]);
                dot += ((double)c0 * inv1) * ((double)c1 * inv2);
              }
              if (lane == L) hud_dot = dot;
            }
          }
        }
      } else if constexpr (MM != kMaskPacked) {
        // byte rows carrying an allele beyond the planes (>= 4): one pass per allele value over each row, re-read from L2
        for (int s = 0; s < LPR; ++s) {
          const size_t rel = tile_row0 + (size_t)grp * LPR + s;
          const bool row_ok = rel < A.row_count;
          const size_t row = A.row_begin + (row_ok ? rel : A.row_count - 1);
          const uint8_t* row_ptr = mv.data + row * mv.pitch;
          const uint8_t* bits_ptr = MISSING ? mv.bits + row * mv.bits_pitch : nullptr;
          const bool own = gl == s;
          if constexpr ((MODE & kModeHudson) != 0 && P >= 2) {
            if (own) {
              inv1 = mine.n[0] != 0 ? 1.0 / (double)mine.n[0] : 0.0;
              inv2 = mine.n[1] != 0 ? 1.0 / (double)mine.n[1] : 0.0;
            }
          }
          for (uint32_t a = 0; a <= bound; ++a) {
            uint32_t c[P];
            count_row_allele<P, MISSING, MM>(mv, lds_mask, nvec_pad, row_ptr, bits_ptr, row_ok, gl, a, c);
            if (own) {
              consume(a, c, row_ok, rel);
              if constexpr ((MODE & kModeHudson) != 0 && P >= 2) shared_k += (c[0] != 0 && c[1] != 0) ? 1u : 0u;
            }
          }
        }
        if constexpr ((MODE & kModeHudson) != 0 && P >= 2) {
          // the same repair for byte rows with alleles beyond the planes: no per-allele table fits in registers, so the wave looks for the
          // next first-seen shared allele round by round (each round: one scan of the row per allele value) and recounts it
          if (hud_formula == kFormulaDense) {
            const uint32_t driver = mine.distinct[0] <= mine.distinct[1] ? 0u : 1u;
            unsigned long long todo = __ballot(my_ok && shared_k >= 3);
            while (todo) {
              const int L = __builtin_ctzll(todo);
              todo &= todo - 1;
              const int d = __builtin_amdgcn_readlane((int)driver, L);
              const uint32_t rounds = (uint32_t)__builtin_amdgcn_readlane((int)shared_k, L);
              const double i1 = 1.0 / (double)(uint32_t)__builtin_amdgcn_readlane((int)mine.n[0], L);
              const double i2 = 1.0 / (double)(uint32_t)__builtin_amdgcn_readlane((int)mine.n[1], L);
              const size_t frow = A.row_begin + tile_row0 + (size_t)L;
              double dot = 0.0;
              uint32_t prev = 0;
              for (uint32_t r = 0; r < rounds; ++r) {
                uint32_t bestv = 0xFFFFFFFFu, besta = 0;
                for (uint32_t a = 0; a <= bound; ++a) {
                  const uint32_t f = first_member_column<MM, MISSING, 2>(mv, lds_mask, nvec_pad, frow, d, a);
                  if (f == 0xFFFFFFFFu || (r != 0 && f <= prev) || f >= bestv) continue;
                  if (wave_member_count<MM, MISSING>(mv, lds_mask, nvec_pad, frow, 1 - d, a) == 0) continue;  // not shared
                  bestv = f;
                  besta = a;
                }
                if (bestv == 0xFFFFFFFFu) break;
                const uint32_t c0 = wave_member_count<MM, MISSING>(mv, lds_mask, nvec_pad, frow, 0, besta);
                const uint32_t c1 = wave_member_count<MM, MISSING>(mv, lds_mask, nvec_pad, frow, 1, besta);
                dot += ((double)c0 * i1) * ((double)c1 * i2);
                prev = bestv;
              }
              if (lane == L) hud_dot = dot;
            }
          }
        }
      }
    }

    if constexpr (!GENERAL) finish_biallelic_site<P, MODE>(mine, hud_dot);

    const size_t my_rel = tile_row0 + lane;
    site_epilogue<P, MODE, MISSING, GENERAL>(A, my_rel, my_rel < A.row_count, mine, hud_dot, wc, T, c4);
  }

  reduce_block_totals<P, MODE>(A, T);
}


// ------------------------------------------------------------------------------------------------
// the pipelined tile loop: packed, biallelic, nothing missing, one batch of loads per row (single_trip)
// ------------------------------------------------------------------------------------------------
// sweep_kernel's plain tile loop has no load in flight while a wave runs an epilogue: with the W&C epilogue of four groups (about 900 VALU
// instructions per tile, as many as the counting) and two waves per SIMD that is a third of every wave's time, and whenever both waves of a
// SIMD are there at once its share of the memory pipeline idles (profiles/r02: C3 at 0.58 of the peak with SQ_WAIT_ANY at only 47 %).  Here the
// row steps of ALL of a wave's tiles form one stream with two rows of loads in flight: the loads of the next tile's first two row steps are
// issued before the epilogue of the current tile and land under it.  Costs the 2 x U x 4 load registers across the epilogue - free for a kernel
// that sits at two waves per SIMD anyway (207 -> at most 256 VGPRs).  Tiles, their order per lane, every per-site operation and the order of
// the regional sums are those of the plain loop: same bits.
template <int P, int MODE, int U, int LPR>
__device__ __forceinline__ void tiles_pipelined(const SweepArgs& A, const MatrixView& mv, const uint4* __restrict__ lm, uint32_t nvec_pad,
                                                LaneTotals<P, MODE>& T) {
  constexpr bool MREG = P * U <= 12;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int grp = lane / LPR, gl = lane % LPR;
  const size_t ntiles = (A.row_count + kTileRows - 1) / kTileRows;
  const size_t tile_stride = (size_t)gridDim.x * kWavesPerBlock;
  size_t tile = (size_t)blockIdx.x * kWavesPerBlock + wave;
  if (tile >= ntiles) return;
  const uint32_t last = mv.nvec - 1;
  uint32_t voff[U];  // a lane's byte offsets inside any row (clamped to the row's last vector: the masks beyond it are zero)
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const uint32_t v = (uint32_t)gl + LPR * u;
    voff[u] = (v < last ? v : last) * 16;
  }
  uint4 m[MREG ? P : 1][MREG ? U : 1];
  if constexpr (MREG) {
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int u = 0; u < U; ++u) m[p][u] = lm[(uint32_t)p * nvec_pad + (uint32_t)gl + LPR * u];
  }
  auto load_row = [&](uint4 (&dst)[U], size_t t, int s) {
    const size_t rel = t * kTileRows + (size_t)grp * LPR + s;
    const size_t row = A.row_begin + (rel < A.row_count ? rel : A.row_count - 1);  // rows past the end re-read the last one
    const uint8_t* rp = mv.data + row * mv.pitch;
#pragma unroll
    for (int u = 0; u < U; ++u) dst[u] = load_stream(rp + voff[u]);
  };
  auto count_row = [&](const uint4 (&x)[U], int s, uint32_t (&alt_mine)[P]) {
    if constexpr (!MREG) asm volatile("" ::: "memory");  // keep the mask reads inside the row (hoisted they are the P x U register image)
    uint32_t alt[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      alt[p] = 0;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (MREG) alt[p] = popc128(and128(x[u], m[p][u]), alt[p]);
        else alt[p] = popc128(and128(x[u], lm[(uint32_t)p * nvec_pad + (uint32_t)gl + LPR * u]), alt[p]);
      }
      alt[p] = group_sum<LPR>(alt[p]);
    }
    if (gl == s) {
#pragma unroll
      for (int p = 0; p < P; ++p) alt_mine[p] = alt[p];
    }
  };
  uint4 a[U], b[U];
  load_row(a, tile, 0);
  load_row(b, tile, 1);
  for (; tile < ntiles; tile += tile_stride) {
    const size_t nxt = tile + tile_stride < ntiles ? tile + tile_stride : tile;  // after the last tile: its own rows once more (L2 hits, unused)
    uint32_t alt_mine[P];
#pragma unroll
    for (int p = 0; p < P; ++p) alt_mine[p] = 0;
#pragma unroll
    for (int s = 0; s < LPR; s += 2) {
      count_row(a, s, alt_mine);
      if (s + 2 < LPR) load_row(a, tile, s + 2); else load_row(a, nxt, s + 2 - LPR);
      count_row(b, s + 1, alt_mine);
      if (s + 3 < LPR) load_row(b, tile, s + 3); else load_row(b, nxt, s + 3 - LPR);
    }
    SiteTally<P> mine;
    WcSite<P> wc;
    double hud_dot = 0.0;
#pragma unroll
    for (int p = 0; p < P; ++p) { mine.n[p] = A.group_size[p]; mine.alt[p] = alt_mine[p]; mine.distinct[p] = 0; mine.ssq[p] = 0; }
    mine.n_all = mv.columns;
    if constexpr ((MODE & kModeWc) != 0) {
      constexpr int NW = 1 + (P * (P - 1)) / 2;
#pragma unroll
      for (int k = 0; k < NW; ++k) { wc.a[k] = 0.0; wc.b[k] = 0.0; }
    }
    finish_biallelic_site<P, MODE>(mine, hud_dot);
    const size_t my_rel = tile * kTileRows + lane;
    site_epilogue<P, MODE, false, false>(A, my_rel, my_rel < A.row_count, mine, hud_dot, wc, T);
  }
}

// (held to three waves per SIMD every pipelined kernel above 168 VGPRs spills: they keep the next tile's loads in registers by design)
template <int P, int MODE, int LPR>
__global__ __launch_bounds__(kBlock) void sweep_kernel_pipe(const SweepArgs A) {
  extern __shared__ __align__(16) unsigned char smem[];
  const MatrixView mv = A.mv;
  const uint32_t nvec = mv.nvec, nvec_pad = A.nvec_pad;
  uint4* staged = reinterpret_cast<uint4*>(smem);
  for (uint32_t i = threadIdx.x; i < (uint32_t)P * nvec_pad; i += kBlock) {
    const uint32_t p = i / nvec_pad, v = i - p * nvec_pad;
    staged[i] = v < nvec ? *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(A.mask_bits) + (size_t)p * (A.mask_pitch / 8) + (size_t)v * 16)
                         : make_uint4(0, 0, 0, 0);
  }
  if constexpr ((MODE & kModeWc) != 0) { wc_rcp_init<P>(A); wc_shape_init<P>(A); }
  __syncthreads();
  LaneTotals<P, MODE> T;
  T.clear();
  if (A.unroll == 5) tiles_pipelined<P, MODE, 5, LPR>(A, mv, staged, nvec_pad, T);
  else if (A.unroll == 3) tiles_pipelined<P, MODE, 3, LPR>(A, mv, staged, nvec_pad, T);
  else if (A.unroll == 2) tiles_pipelined<P, MODE, 2, LPR>(A, mv, staged, nvec_pad, T);
  else tiles_pipelined<P, MODE, 1, LPR>(A, mv, staged, nvec_pad, T);
  reduce_block_totals<P, MODE>(A, T);
}

}  // namespace fmh
