// sweep_flat.hip — instantiations and launcher of the LDS-staged flat-tile route (sweep_flat_kernels.hpp): packed biallelic matrices with
// nothing missing whose rows are at most 32 vectors (4 096 columns).
#include "abi_internal.hpp"
#include "sweep_flat_kernels.hpp"

using namespace fmh;

namespace fmhi {
namespace {

// LDS a workgroup's tile images take: waves x slots x nvec KiB (+ slack for the padded reads of the last row)
inline size_t flat_smem(uint32_t nvec, int slots) { return (size_t)kFlatWaves * slots * nvec * 1024u + kFlatLdsSlack; }

// the register-staged variant: one tile image per wave in LDS, the next tile in flight in registers
template <int P, int MODE, int NVMAX>
int launch_flat_rs(const SweepArgs& args_in, hipStream_t st, const LaunchCtx& ctx, int* grid_out) {
  auto kern = sweep_kernel_flat_rs<P, MODE, NVMAX>;
  const uint32_t nvec = args_in.mv.nvec;
  static thread_local int cached_occ[64];
  static thread_local uint32_t cached_nvec[64];
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  const size_t smem = flat_smem(nvec, 1);
  if (smem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  if (cached_nvec[dev] != nvec) {
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, kFlatBlock, smem));
    if (occ > 8) occ = 8;
    if (occ < 1) return fail(FMH_ERR_UNSUPPORTED, "a tile of %u vectors does not fit the LDS of the flat-tile route", nvec);
    cached_occ[dev] = occ;
    cached_nvec[dev] = nvec;
  }
  int occ = cached_occ[dev];
  if (const int env_occ = (int)options().max_occ.load(); env_occ > 0 && occ > env_occ) occ = env_occ;
  SweepArgs args = args_in;
  args.flat_slots = 0;
  const size_t ntiles = (args.row_count + kTileRows - 1) / kTileRows;
  size_t blocks = (ntiles + kFlatWaves - 1) / kFlatWaves;
  size_t cap = (size_t)ctx.cus * occ;
  if (const long long v = options().grid_per_cu.load(); v > 0) cap = (size_t)ctx.cus * (size_t)v;
  if (const long long v = options().grid_blocks.load(); v > 0) cap = (size_t)v;
  if (blocks > cap) blocks = cap;
  if (blocks > (size_t)ctx.max_grid) blocks = ctx.max_grid;
  if (blocks < 1) blocks = 1;
  int depth = (int)options().flat_defer.load();
  if (depth < 1) {  // by the tiles a wave sweeps: short launches keep the undeferred order
    const size_t rounds = (ntiles + blocks * kFlatWaves - 1) / (blocks * kFlatWaves);
    depth = rounds >= 16 ? kFlatDeferMax : rounds >= 8 ? 4 : 1;
  }
  args.flat_defer = depth > flat_defer_max<P>() ? flat_defer_max<P>() : depth;
  if (ctx.timing) HIP_TRY(hipEventRecord(ctx.ev0, st));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kFlatBlock), smem, st, args);
  HIP_TRY(hipGetLastError());
  if (ctx.timing) HIP_TRY(hipEventRecord(ctx.ev1, st));
  *grid_out = (int)blocks;
  return FMH_OK;
}

template <int P, int MODE, int NVMAX>
int launch_flat(const SweepArgs& args_in, hipStream_t st, const LaunchCtx& ctx, int* grid_out) {
  if (options().flat_slots.load() == 0) return launch_flat_rs<P, MODE, NVMAX>(args_in, st, ctx, grid_out);  // FMH_FLAT_SLOTS = 1 | 2: the LDS-DMA variants
  auto kern = sweep_kernel_flat<P, MODE, NVMAX>;
  const uint32_t nvec = args_in.mv.nvec;
  static thread_local int cached_occ[64][2];
  static thread_local uint32_t cached_nvec[64];
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (cached_nvec[dev] != nvec) {
    for (int slots = 1; slots <= 2; ++slots) {
      const size_t smem = flat_smem(nvec, slots);
      int occ = 0;
      if (smem <= 160 * 1024 - 8 * 1024) {
        if (smem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, kFlatBlock, smem));
      }
      if (occ > 8) occ = 8;  // the partial vectors hold 8 workgroups per CU
      cached_occ[dev][slots - 1] = occ;
    }
    cached_nvec[dev] = nvec;
  }
  SweepArgs args = args_in;
  // Tile images per wave.  Two slots keep a DMA in flight through the whole round of a wave, one slot leaves more waves resident.
  int slots = (int)options().flat_slots.load();
  if (slots != 1 && slots != 2) slots = cached_occ[dev][1] * kFlatWaves >= 8 ? 2 : 1;
  if (cached_occ[dev][slots - 1] < 1) slots = 1;
  int occ = cached_occ[dev][slots - 1];
  if (occ < 1) return fail(FMH_ERR_UNSUPPORTED, "a tile of %u vectors does not fit the LDS of the flat-tile route", nvec);
  if (const int env_occ = (int)options().max_occ.load(); env_occ > 0 && occ > env_occ) occ = env_occ;
  args.flat_slots = slots;
  const size_t smem = flat_smem(nvec, slots);
  if (smem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const size_t ntiles = (args.row_count + kTileRows - 1) / kTileRows;
  size_t blocks = (ntiles + kFlatWaves - 1) / kFlatWaves;
  size_t cap = (size_t)ctx.cus * occ;
  if (const long long v = options().grid_per_cu.load(); v > 0) cap = (size_t)ctx.cus * (size_t)v;
  if (const long long v = options().grid_blocks.load(); v > 0) cap = (size_t)v;
  if (blocks > cap) blocks = cap;
  if (blocks > (size_t)ctx.max_grid) blocks = ctx.max_grid;
  if (blocks < 1) blocks = 1;
  if (ctx.timing) HIP_TRY(hipEventRecord(ctx.ev0, st));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kFlatBlock), smem, st, args);
  HIP_TRY(hipGetLastError());
  if (ctx.timing) HIP_TRY(hipEventRecord(ctx.ev1, st));
  *grid_out = (int)blocks;
  return FMH_OK;
}

template <int P, int MODE>
int launch_nv(const SweepArgs& a, hipStream_t st, const LaunchCtx& ctx, int* grid) {
  if (a.mv.nvec <= 8) return launch_flat<P, MODE, 8>(a, st, ctx, grid);
  if (a.mv.nvec <= 20) return launch_flat<P, MODE, 20>(a, st, ctx, grid);
  return launch_flat<P, MODE, kFlatMaxVec>(a, st, ctx, grid);
}

}  // namespace

bool flat_route_builds(int P, int mode) {
  if (mode == kModeSummary) return P == 1 || P == 2 || P == 4;
  if (mode == (kModeSummary | kModeHudson)) return P == 2;
  if (mode == (kModeSummary | kModeDiversity)) return P == 1 || P == 2;
  if (mode == (kModeSummary | kModeHudson | kModeDiversity)) return P == 2;
  if (mode == kModeWc) return P == 2 || P == 4;
  return false;
}

// packed rows of at most kFlatMaxVec vectors, biallelic, nothing missing; a.mask_flat set
int launch_sweep_flat(int P, int mode, const SweepArgs& a, hipStream_t st, const LaunchCtx& ctx, int* grid) {
  if (a.mv.nvec < 1 || a.mv.nvec > (uint32_t)kFlatMaxVec || !a.mask_flat || a.mv.pitch != (size_t)a.mv.nvec * 16)
    return fail(FMH_ERR_INVALID, "the flat-tile route takes packed rows of 1..%d vectors with pitch = 16 x vectors", kFlatMaxVec);
#define CASE(PV, MODEV) return launch_nv<PV, MODEV>(a, st, ctx, grid)
  if (mode == kModeSummary) {
    if (P == 1) CASE(1, kModeSummary);
    if (P == 2) CASE(2, kModeSummary);
    if (P == 4) CASE(4, kModeSummary);
  } else if (mode == (kModeSummary | kModeHudson)) {
    if (P == 2) CASE(2, kModeSummary | kModeHudson);
  } else if (mode == (kModeSummary | kModeDiversity)) {
    if (P == 1) CASE(1, kModeSummary | kModeDiversity);
    if (P == 2) CASE(2, kModeSummary | kModeDiversity);
  } else if (mode == (kModeSummary | kModeHudson | kModeDiversity)) {
    if (P == 2) CASE(2, kModeSummary | kModeHudson | kModeDiversity);
  } else if (mode == kModeWc) {
    if (P == 2) CASE(2, kModeWc);
    if (P == 4) CASE(4, kModeWc);
  }
#undef CASE
  return fail(FMH_ERR_UNSUPPORTED, "no flat-tile sweep kernel for %d groups in mode %d", P, mode);
}

}  // namespace fmhi
