// sweep_mfma.hip — instantiations and launcher of the int8 matrix-core counting route (sweep_mfma_kernels.hpp).
#include "abi_internal.hpp"
#include "sweep_mfma_kernels.hpp"

using namespace fmh;

namespace fmhi {
namespace {

template <int P, int MODE, int U>
int launch_mfma(const SweepArgs& args, size_t smem, hipStream_t st, const LaunchCtx& ctx, int* grid_out) {
  auto kern = sweep_mfma_kernel<P, MODE, U>;
  static thread_local int cached_occ[64];
  static thread_local size_t cached_smem[64];
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (smem > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  if (cached_occ[dev] == 0 || cached_smem[dev] != smem) {
    int occ = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, kBlock, smem));
    if (occ < 1) occ = 1;
    if (occ > 8) occ = 8;
    cached_occ[dev] = occ;
    cached_smem[dev] = smem;
  }
  // FMH_MAX_OCC is applied per launch, outside the cache: fmh_set_option may change it at any time (tools/ab_env.py alternates it)
  int occ_now = cached_occ[dev];
  if (const int env_occ = (int)options().max_occ.load(); env_occ > 0 && occ_now > env_occ) occ_now = env_occ;
  const size_t ntiles = (args.row_count + kTileRows - 1) / kTileRows;
  size_t blocks = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
  size_t cap = (size_t)ctx.cus * occ_now;
  if (const long long v = options().grid_per_cu.load(); v > 0) cap = (size_t)ctx.cus * (size_t)v;  // the same grid options as the other routes
  if (const long long v = options().grid_blocks.load(); v > 0) cap = (size_t)v;
  if (blocks > cap) blocks = cap;  // persistent grid (equalising the tile rounds per workgroup was measured: fewer resident waves, slower)
  if (blocks > (size_t)ctx.max_grid) blocks = ctx.max_grid;
  if (blocks < 1) blocks = 1;
  if (ctx.timing) HIP_TRY(hipEventRecord(ctx.ev0, st));
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(kBlock), smem, st, args);
  HIP_TRY(hipGetLastError());
  if (ctx.timing) HIP_TRY(hipEventRecord(ctx.ev1, st));
  *grid_out = (int)blocks;
  return FMH_OK;
}

template <int P, int MODE>
int launch_u(int unroll, const SweepArgs& a, size_t smem, hipStream_t st, const LaunchCtx& ctx, int* grid) {
  if (unroll == 2) return launch_mfma<P, MODE, 2>(a, smem, st, ctx, grid);
  return launch_mfma<P, MODE, 4>(a, smem, st, ctx, grid);
}

}  // namespace

// u8 rows, biallelic, nothing missing, P (padded) <= 4; a.unroll = K steps issued back to back (2 or 4), a.nvec_pad =
// mfma_mask_stride(nvec, unroll), smem = P * nvec_pad * 16
int launch_sweep_mfma(int P, int mode, const SweepArgs& a, size_t smem, hipStream_t st, const LaunchCtx& ctx, int* grid) {
#define CASE(PV, MODEV) return launch_u<PV, MODEV>(a.unroll, a, smem, st, ctx, grid)
  if (mode == kModeSummary) {
    if (P == 1) CASE(1, kModeSummary);
    if (P == 2) CASE(2, kModeSummary);
    if (P == 4) CASE(4, kModeSummary);
  } else if (mode == (kModeSummary | kModeHudson)) {
    if (P == 2) CASE(2, kModeSummary | kModeHudson);
  } else if (mode == (kModeSummary | kModeDiversity)) {
    if (P == 1) CASE(1, kModeSummary | kModeDiversity);
  } else if (mode == kModeWc) {
    if (P == 2) CASE(2, kModeWc);
    if (P == 4) CASE(4, kModeWc);
  }
#undef CASE
  return fail(FMH_ERR_UNSUPPORTED, "no matrix-core sweep kernel for %d groups in mode %d", P, mode);
}

}  // namespace fmhi
