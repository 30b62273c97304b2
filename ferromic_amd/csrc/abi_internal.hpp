// abi_internal.hpp — host-side internals shared by the translation units of libferromic_hip.so (abi.hip, pairwise.hip,
// comm.hip and the sweep_*.hip files that hold the kernel instantiations).  Nothing here is part of the C-ABI: the library
// is built with -fvisibility=hidden and only include/ferromic_hip.h's functions are exported.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#pragma GCC visibility push(default)
#include "../../include/ferromic_hip.h"
#pragma GCC visibility pop

#include "sweep_kernels.hpp"

namespace fmhi {

// ---- errors: status code + thread-local message (fmh_last_error) ----------------------------------------------------
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define HIP_TRY(expr)                                                                               \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess)                                                                           \
      return ::fmhi::fail(_e == hipErrorNoDevice ? FMH_ERR_NO_DEVICE : FMH_ERR_HIP, "%s: %s (%s:%d)", \
                          #expr, hipGetErrorString(_e), __FILE__, __LINE__);                        \
  } while (0)

#define FMH_TRY(expr)            \
  do {                           \
    int _s = (expr);             \
    if (_s != FMH_OK) return _s; \
  } while (0)

int use_device(int device);  // range check + hipSetDevice

// ---- device memory pool (abi.hip) -------------------------------------------------------------------------------------
hipError_t pool_malloc(int device, void** out, size_t bytes);
void pool_free(int device, void* p);
void pool_trim(int device);

inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- per-process options (fmh_set_option / fmh_get_option, abi.hip) ------------------------------------------------------
// Every switch that routes a kernel or sizes a launch lives here as an atomic integer.  The FMH_* environment is read ONCE, when the
// library first needs an option (so `FMH_LAYOUT=bytes python ...` and a child process started with such an environment still work);
// after that only fmh_set_option changes a value.  Nothing on a launch path reads the environment: round 2 read ten variables per enqueue, which
// cost 1-2 us per sweep and raced with any host thread calling setenv.
struct Options {
  std::atomic<long long> layout_bytes{0};        // FMH_LAYOUT = bytes | packed: keep the u8 kernels on matrices that still hold their byte rows
  std::atomic<long long> mask_mode{-1};          // FMH_MASK_MODE: force a slower mask route (tests / measurements)
  std::atomic<long long> defer_tiles{0};         // FMH_DEFER_TILES: 1..16 forces the deferral depth (1 = undeferred); 0 = by the launch size
  std::atomic<long long> packed_lpr{0};          // FMH_PACKED_LPR: 4 | 16 lanes per packed row; 0 = by the row width
  std::atomic<long long> packed_unroll{0};       // FMH_PACKED_UNROLL: batch depth of the packed cores; 0 = fewest padded slots
  std::atomic<long long> packed_no_prefetch{0};  // FMH_PACKED_NO_PREFETCH: plain row loop instead of tile_rows_packed_prefetch
  std::atomic<long long> counts_mfma{0};         // FMH_COUNTS_MFMA: 1 | 2 = the int8 matrix-core counting route (BASELINE config C5)
  std::atomic<long long> grid_per_cu{0};         // FMH_GRID_PER_CU: workgroups per CU of the persistent grid; 0 = the occupancy
  std::atomic<long long> grid_blocks{0};         // FMH_GRID_BLOCKS: total workgroups (tests: many tile rounds on small inputs)
  std::atomic<long long> max_occ{0};             // FMH_MAX_OCC: cap on the occupancy used for the grid
  std::atomic<long long> unroll{0};              // FMH_UNROLL: 8 = deeper batches of the u8 cores
  std::atomic<long long> pitch_align{16};        // FMH_PITCH_ALIGN: row pitch alignment of u8 matrices
  std::atomic<long long> comm_host{0};           // FMH_COMM_TRANSPORT = host: in-process rendezvous instead of RCCL (fmh_comm_init_all)
  std::atomic<long long> upload_threads{0};      // FMH_UPLOAD_THREADS: host packer threads; 0 = the CPU share
  std::atomic<long long> pd_two_planes{0}, pd_int8{0}, pd_planes_bytes{(long long)8 << 30}, pd_kchunk{0}, pd_sb{0}, pd_occ{0}, pd_phased{1}, pd_slabs{1}, pd_slab_bytes{(long long)4 << 30};  // FMH_PD_*: pairwise path
  std::atomic<long long> pipe{-1};               // FMH_PIPE: the pipelined tile loop on four-lane rows: 1 = wherever it is built, 0 = never, -1 = where it measured ahead (one and two groups)
  std::atomic<long long> flat{-1};               // FMH_FLAT: the LDS-staged flat-tile route on short packed rows: 1 = wherever it is built, 0 = never, -1 = where it measured ahead
  std::atomic<long long> flat_slots{0};          // FMH_FLAT_SLOTS: 0 = the register-staged variant (default); 1 | 2 = the LDS-DMA variants with that many tile images per wave
  std::atomic<long long> flat_defer{0};          // FMH_FLAT_DEFER: tiles a wave counts before it runs their epilogues on the register-staged variant (1..8); 0 = by the launch size
  std::atomic<long long> wc_exact{1};            // FMH_WC_EXACT: 0 = five to seven W&C groups run the padded eight-group kernel again (A/B of the exact kernels)
  std::atomic<long long> wc_bi_totals{1};        // FMH_WC_BI_TOTALS: 0 = more than eight groups' regional sums through the general pair kernel again (A/B of the biallelic one)
  std::atomic<long long> wc_bi_replicas{0};      // FMH_WC_BI_REPLICAS: 1 | 2 | 4 | 8 threads per pair in the biallelic pair kernel; 0 = by the lanes the last wave would waste
  std::atomic<long long> wc_bi_chunks{0};        // FMH_WC_BI_CHUNKS: row chunks of the biallelic pair kernel (measurement); 0 = about 8 192 workgroups
  std::atomic<long long> row_hi{1};              // FMH_ROW_HI: 0 = packed matrices get no tables of the rows with alleles above 1 / with uncalled columns (every plane of every row is read); 2 = tables at any size
  std::atomic<long long> graph{0};               // FMH_GRAPH: 1 = replay a repeated pipelined sweep on a local communicator from a captured hipGraph
  std::atomic<unsigned long long> generation{0}; // bumped by every fmh_set_option: a captured launch is never replayed across an option change
};
Options& options();

// FMH_LAYOUT=bytes keeps the u8 kernels on matrices that still hold their byte rows
inline bool layout_bytes_forced() { return options().layout_bytes.load(std::memory_order_relaxed) != 0; }

// RAII for the scratch of one call
struct DeviceScratch {
  int device = 0;
  hipStream_t stream = nullptr;
  bool settled = false;  // set once the call has synchronised the stream its kernels ran on
  std::vector<void*> ptrs;
  // An early (error) return may leave kernels that write these blocks enqueued: wait for the stream before the blocks go
  // back to the pool, where another thread's pool_malloc could be handed them at once.
  ~DeviceScratch() {
    if (!settled && !ptrs.empty()) { (void)hipStreamSynchronize(stream); (void)hipGetLastError(); }
    for (void* p : ptrs) pool_free(device, p);
  }
  template <class T> int get(T** out, size_t count) {
    void* p = nullptr;
    HIP_TRY(pool_malloc(device, &p, (count > 0 ? count : 1) * sizeof(T)));
    ptrs.push_back(p);
    *out = (T*)p;
    return FMH_OK;
  }
};

// ---- per-device workspace: sweep leases, harmonic table, pairwise scratch -----------------------------------------------
// What ONE sweep in flight needs: block partials, the finalised totals, their pinned host copy, timing events, and a stream of its own
// (used when the caller passes the NULL stream, so that sweeps issued by different host threads overlap on the device instead of
// queueing behind each other on the legacy default stream).  Leases are pooled per device: run_vcf's region workers each take one.
struct SweepLease {
  double* part_f64 = nullptr;
  unsigned long long* part_u64 = nullptr;
  double* out_f64 = nullptr;
  unsigned long long* out_u64 = nullptr;
  double* h_f64 = nullptr;  // pinned
  unsigned long long* h_u64 = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t stream = nullptr;
};
struct Workspace {
  bool ready = false;
  int cus = 0;
  int max_grid = 0;
  std::mutex lease_mu;
  std::vector<SweepLease*> idle_leases;
  double* harmonic = nullptr;  // grows only; superseded tables stay alive (a sweep in flight may still read them)
  size_t harmonic_len = 0;
  std::vector<double*> retired_harmonic;
  uint8_t* pd_planes = nullptr;  // pairwise-differences planes, kept between calls (fmh_device_release_scratch frees them)
  size_t pd_planes_bytes = 0;
  int* pd_slabs = nullptr;       // the phased Gram kernel's per-item partial tiles
  size_t pd_slab_bytes = 0;
  std::mutex in_use;  // the harmonic table's growth and the pairwise scratch: one holder at a time per device
};
int workspace(int device, Workspace** out);

// measurement (fmh_timing_*): accumulated HIP-event time of the sweep kernels
void timing_add(double ms);
bool timing_enabled();

// ---- kernel launches that live in their own translation units ------------------------------------------------------
// What a launcher needs from the workspace.
struct LaunchCtx {
  int cus;
  int max_grid;
  hipEvent_t ev0, ev1;  // recorded around the sweep kernel when `timing`
  bool timing;
};
// Device buffers a sweep reduces into: [grid][64] block partials and the 64 + 64 regional accumulators.
struct SweepBuffers {
  double* part_f64;
  unsigned long long* part_u64;
  double* out_f64;
  unsigned long long* out_u64;
  // optional: run finalize_kernel on another stream, ordered behind the sweep by this event - the pipelined sharded sweeps put it on the
  // communicator's stream, so that the next window's sweep follows this one's without the finalize launch and its two kernel boundaries between them
  hipStream_t finalize_stream = nullptr;
  hipEvent_t swept = nullptr;
};
}  // namespace fmhi
struct fmh_matrix;
struct fmh_groups;
namespace fmhi {
// host buffers -> the (already allocated) planes of a packed matrix (upload.hip): packed on the host into pinned staging, copied
// asynchronously; *overflow = a called entry carries an allele above max_allele
int upload_planes_from_bytes(fmh_matrix* m, const uint8_t* h_data, const uint64_t* h_missing, bool* overflow);
int upload_planes_from_planes(fmh_matrix* m, const uint8_t* const h_planes[4], size_t h_pitch);
void upload_release(int device);
// planes rows [row0, row0 + rows) of a packed matrix -> byte rows of `pitch` bytes (abi.hip)
hipError_t unpack_rows(const fmh_matrix* m, size_t row0, size_t rows, uint8_t* data, size_t pitch, hipStream_t st);
// sweep + finalize enqueued on `st`, no synchronisation (abi.hip)
int enqueue_sweep(const fmh_matrix* m, const fmh_groups* g, int mode, fmh::SweepArgs& a, hipStream_t st, const LaunchCtx& ctx,
                  const SweepBuffers& b, const double* harmonic, bool* launched);

// the fused region sweep's kernel arguments and mode (abi.hip); the device's harmonic table
int pair_region_args(const fmh_groups* g, size_t row_begin, size_t row_count, int summary_formula, int hudson_formula, const fmh_pair_diversity_sites* div,
                     const fmh_hudson_sites* sites, fmh::SweepArgs& a, int* mode);
int harmonic_table(int device, size_t max_k, hipStream_t st, const double** out);
// W&C slot order of the padded kernel -> the caller's pair order; which W&C / summaries calls are one fused sweep (abi.hip)
void wc_slot_map(const fmh_matrix* m, const fmh_groups* g, fmh::SweepArgs& a, int (&slot_of)[32]);
int wc_kernel_groups(const fmh_matrix* m, const fmh_groups* g);
bool wc_fused_lane_totals(const fmh_matrix* m, const fmh_groups* g);
bool summaries_single_sweep(const fmh_matrix* m, const fmh_groups* g);

// One function per (mask route, lanes per row): dispatches on (P, mode, missing, general) to the instantiation, sizes the
// persistent grid and launches.  FMH_ERR_UNSUPPORTED for a combination the route does not build.
int launch_sweep_packed4(int P, int mode, bool missing, bool general, const fmh::SweepArgs& a, size_t smem, hipStream_t st, const LaunchCtx& ctx, int* grid);
int launch_sweep_packed16(int P, int mode, bool missing, bool general, const fmh::SweepArgs& a, size_t smem, hipStream_t st, const LaunchCtx& ctx, int* grid);
// three-plane packed matrices (alleles 4..7): the multi-allelic kernels only
int launch_sweep_packed4_3p(int P, int mode, bool missing, bool general, const fmh::SweepArgs& a, size_t smem, hipStream_t st, const LaunchCtx& ctx, int* grid);
int launch_sweep_packed16_3p(int P, int mode, bool missing, bool general, const fmh::SweepArgs& a, size_t smem, hipStream_t st, const LaunchCtx& ctx, int* grid);
int launch_sweep_bytes(int P, int mode, bool missing, bool general, const fmh::SweepArgs& a, size_t smem, hipStream_t st, const LaunchCtx& ctx, int* grid);
int launch_sweep_bits(int P, int mode, bool missing, bool general, const fmh::SweepArgs& a, size_t smem, hipStream_t st, const LaunchCtx& ctx, int* grid);
int launch_sweep_global(int P, int mode, bool missing, bool general, const fmh::SweepArgs& a, size_t smem, hipStream_t st, const LaunchCtx& ctx, int* grid);
// the LDS-staged flat-tile route (sweep_flat.hip): packed rows of at most kFlatMaskMaxVec vectors, biallelic, nothing missing
constexpr int kFlatMaskMaxVec = 32;
bool flat_route_builds(int P, int mode);
inline bool flat_route_default(int P, int mode, uint32_t pvec) { (void)P; (void)mode; (void)pvec; return false; }  // until measured
int launch_sweep_flat(int P, int mode, const fmh::SweepArgs& a, hipStream_t st, const LaunchCtx& ctx, int* grid);
// counts on the int8 matrix cores (sweep_mfma.hip): u8 rows, biallelic, nothing missing, at most 4 (padded) groups
int launch_sweep_mfma(int P, int mode, const fmh::SweepArgs& a, size_t smem, hipStream_t st, const LaunchCtx& ctx, int* grid);

}  // namespace fmhi

// ---- handles (the opaque types of the C-ABI) ----------------------------------------------------------------------------
struct fmh_matrix {
  int device = 0;
  uint8_t* data = nullptr;
  uint8_t* bits = nullptr;  // called bit-rows or null
  size_t pitch = 0, bits_pitch = 0;
  size_t variants = 0, samples = 0, ploidy = 0;
  uint32_t columns = 0, nvec = 0;
  uint8_t max_allele = 0;
  bool owns = true;
  bool has_missing = false;  // a called mask exists (bits and / or pc)
  // bit-packed image (fmh_matrix_pack): plane k = bit k of the allele value (p1: max_allele >= 2, p2: max_allele >= 4),
  // pc = called bits; plane_pitch bytes per row, pvec = ceil(columns / 128) 16-byte vectors.  data / bits may have been
  // released (nullptr).
  uint8_t *p0 = nullptr, *p1 = nullptr, *p2 = nullptr, *pc = nullptr;
  uint8_t* row_gap = nullptr; // with pc: one byte per row, non-zero when some column of the row is not called (written wherever the planes are)
  uint8_t* row_hi = nullptr;  // with p1: one byte per row, non-zero when the row has a bit in plane 1 or 2 (written wherever the planes are)
  size_t plane_pitch = 0;
  uint32_t pvec = 0;
};

struct fmh_groups {
  int device = 0;
  int n_groups = 0;   // caller's P
  int padded = 0;     // kernel P (1, 2, 4 or 8)
  uint8_t* masks = nullptr;  // [padded][mask_pitch], zero beyond the row
  size_t pitch = 0;
  size_t mask_pitch = 0;     // pitch rounded up to 2048: covers the kernels' zero-padded mask stride
  uint16_t* mask_bits = nullptr;  // [padded][mask_pitch / 16]: the same masks as one 16-bit word per 16-byte vector
  uint32_t* mask_flat = nullptr;  // [round_up(ceil(columns / 128), 4)][padded][4]: the bit masks interleaved by vector (rows of at most 4 096 columns; same block)
  uint32_t columns = 0;
  uint64_t sizes[FMH_MAX_GROUPS] = {0};
  std::vector<uint8_t> host_mask;  // [n_groups][columns] as handed in (the wide-matrix W&C route re-batches the groups)
};
