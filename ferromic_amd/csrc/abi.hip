// abi.hip — implementation of include/ferromic_hip.h: device memory, layout conversion, kernel
// dispatch and totals collection.  No CPU compute path exists here: without a GPU every entry
// point fails with FMH_ERR_NO_DEVICE.  The sweep kernels are instantiated in sweep_*.hip (one object per mask route),
// the pairwise-differences path lives in pairwise.hip, the RCCL communicator in comm.hip.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <unordered_map>
#include <string>
#include <vector>

#include "abi_internal.hpp"
#include "util_kernels.hpp"
#include "sweep_mfma_kernels.hpp"
#include "wc_counts_kernels.hpp"

using namespace fmh;
using namespace fmhi;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

int fmhi::fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

extern "C" const char* fmh_last_error(void) { return g_last_error.c_str(); }
extern "C" int fmh_abi_version(void) { return FMH_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------
// options: one table, the FMH_* environment read once, fmh_set_option afterwards
// ------------------------------------------------------------------------------------------------
namespace {
struct OptionRow {
  const char* key;
  std::atomic<long long> Options::*field;
  long long dflt;
  const char* words;  // "word=value,..." spellings accepted besides integers
};
const OptionRow kOptionRows[] = {
    {"FMH_LAYOUT", &Options::layout_bytes, 0, "bytes=1,packed=0"},
    {"FMH_MASK_MODE", &Options::mask_mode, -1, nullptr},
    {"FMH_DEFER_TILES", &Options::defer_tiles, 0, nullptr},
    {"FMH_PACKED_LPR", &Options::packed_lpr, 0, nullptr},
    {"FMH_PACKED_UNROLL", &Options::packed_unroll, 0, nullptr},
    {"FMH_PACKED_NO_PREFETCH", &Options::packed_no_prefetch, 0, nullptr},
    {"FMH_COUNTS_MFMA", &Options::counts_mfma, 0, nullptr},
    {"FMH_GRID_PER_CU", &Options::grid_per_cu, 0, nullptr},
    {"FMH_GRID_BLOCKS", &Options::grid_blocks, 0, nullptr},
    {"FMH_MAX_OCC", &Options::max_occ, 0, nullptr},
    {"FMH_UNROLL", &Options::unroll, 0, nullptr},
    {"FMH_PITCH_ALIGN", &Options::pitch_align, 16, nullptr},
    {"FMH_COMM_TRANSPORT", &Options::comm_host, 0, "host=1,rccl=0"},
    {"FMH_UPLOAD_THREADS", &Options::upload_threads, 0, nullptr},
    {"FMH_PD_TWO_PLANES", &Options::pd_two_planes, 0, nullptr},
    {"FMH_PD_INT8", &Options::pd_int8, 0, nullptr},
    {"FMH_PD_PLANES_BYTES", &Options::pd_planes_bytes, (long long)8 << 30, nullptr},
    {"FMH_PD_KCHUNK", &Options::pd_kchunk, 0, nullptr},
    {"FMH_PD_SB", &Options::pd_sb, 0, nullptr},
    {"FMH_PD_OCC", &Options::pd_occ, 0, nullptr},
    {"FMH_PD_PHASED", &Options::pd_phased, 1, nullptr},
    {"FMH_PD_SLABS", &Options::pd_slabs, 1, nullptr},
    {"FMH_PD_SLAB_BYTES", &Options::pd_slab_bytes, (long long)4 << 30, nullptr},
    {"FMH_PIPE", &Options::pipe, -1, nullptr},
    {"FMH_GRAPH", &Options::graph, 0, nullptr},
    {"FMH_FLAT", &Options::flat, -1, nullptr},
    {"FMH_FLAT_SLOTS", &Options::flat_slots, 0, nullptr},
    {"FMH_FLAT_DEFER", &Options::flat_defer, 0, nullptr},
    {"FMH_WC_EXACT", &Options::wc_exact, 1, nullptr},
    {"FMH_WC_BI_TOTALS", &Options::wc_bi_totals, 1, nullptr},
    {"FMH_WC_BI_REPLICAS", &Options::wc_bi_replicas, 0, nullptr},
    {"FMH_WC_BI_CHUNKS", &Options::wc_bi_chunks, 0, nullptr},
    {"FMH_ROW_HI", &Options::row_hi, 1, nullptr},
};
bool parse_option(const OptionRow& row, const char* text, long long* out) {
  if (row.words) {
    const size_t n = strlen(text);
    for (const char* w = row.words; *w;) {
      const char* eq = strchr(w, '=');
      const char* end = strchr(w, ',');
      if (!end) end = w + strlen(w);
      if ((size_t)(eq - w) == n && strncmp(w, text, n) == 0) { *out = atoll(eq + 1); return true; }
      w = *end ? end + 1 : end;
    }
  }
  char* stop = nullptr;
  const long long v = strtoll(text, &stop, 10);
  if (stop == text || *stop) return false;
  *out = v;
  return true;
}
}  // namespace

constexpr size_t kOptionCount = sizeof kOptionRows / sizeof kOptionRows[0];
static long long g_option_initial[kOptionCount];  // what the process started with: the environment's value, else the default

Options& fmhi::options() {
  static Options o;
  static std::once_flag once;
  std::call_once(once, [] {
    for (size_t i = 0; i < kOptionCount; ++i) {
      const OptionRow& row = kOptionRows[i];
      long long x = row.dflt;
      // a variable that is set but empty or unparsable counts as "1" for the integer switches (round 2's presence switches:
      // FMH_PACKED_NO_PREFETCH=, FMH_PD_INT8=yes); a word-valued one (FMH_LAYOUT, FMH_COMM_TRANSPORT) keeps its default and says so, as
      // fmh_set_option refuses the same text
      if (const char* v = getenv(row.key)) {
        if (!parse_option(row, v, &x)) {
          if (row.words) { x = row.dflt; fprintf(stderr, "libferromic_hip: %s='%s' is not one of %s: ignored\n", row.key, v, row.words); }
          else x = 1;
        }
      }
      g_option_initial[i] = x;
      (o.*row.field).store(x);
    }
  });
  return o;
}

extern "C" int fmh_set_option(const char* key, const char* value_or_null) {
  if (!key) return fail(FMH_ERR_INVALID, "option key is NULL");
  Options& o = options();
  for (size_t i = 0; i < kOptionCount; ++i) {
    const OptionRow& row = kOptionRows[i];
    if (strcmp(row.key, key) != 0) continue;
    long long x = g_option_initial[i];  // NULL: back to what the process started with
    if (value_or_null && !parse_option(row, value_or_null, &x))
      return fail(FMH_ERR_INVALID, "option %s: cannot parse '%s'%s%s", key, value_or_null, row.words ? " (integers or " : "", row.words ? row.words : "");
    (o.*row.field).store(x);
    o.generation.fetch_add(1);
    return FMH_OK;
  }
  return fail(FMH_ERR_INVALID, "unknown option '%s'", key);
}

extern "C" int fmh_get_option(const char* key, long long* h_value) {
  if (!key || !h_value) return fail(FMH_ERR_INVALID, "NULL argument");
  Options& o = options();
  for (const OptionRow& row : kOptionRows) {
    if (strcmp(row.key, key) == 0) { *h_value = (o.*row.field).load(); return FMH_OK; }
  }
  return fail(FMH_ERR_INVALID, "unknown option '%s'", key);
}

static int device_count_checked(int* n) {
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess || c <= 0) {
    *n = 0;
    return fail(FMH_ERR_NO_DEVICE, "no HIP device available (%s); libferromic_hip has no CPU fallback",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  }
  *n = c;
  return FMH_OK;
}

extern "C" int fmh_device_count(int* h_count) {
  if (!h_count) return fail(FMH_ERR_INVALID, "h_count is NULL");
  return device_count_checked(h_count);
}

int fmhi::use_device(int device) {
  int n = 0;
  FMH_TRY(device_count_checked(&n));
  if (device < 0 || device >= n) return fail(FMH_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
  HIP_TRY(hipSetDevice(device));
  return FMH_OK;
}

// ------------------------------------------------------------------------------------------------
// device memory pool.  A region of run_vcf or one call of the Python module allocates a dozen small device
// buffers (masks, per-site tracks, temporaries); hipMalloc/hipFree cost tens to hundreds of microseconds each
// and hipFree synchronises the device.  Blocks up to 64 MiB are recycled through size-class free lists (at most
// 2 GiB cached per device; fmh_device_release_scratch empties them).  Every entry point of this library
// synchronises before it returns, so a block handed back by the caller is idle.
// ------------------------------------------------------------------------------------------------
struct DevicePool {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;           // pooled blocks in use -> class size
  std::map<size_t, std::vector<void*>> free_lists;  // class size -> idle blocks
  size_t cached_bytes = 0;
};
static DevicePool g_pool[64];
static const size_t kPoolMaxBlock = (size_t)64 << 20, kPoolMaxCached = (size_t)2 << 30;

static size_t pool_class(size_t bytes) {  // powers of two in quarter steps, >= 256 B
  if (bytes <= 256) return 256;
  size_t p2 = 256;
  while (p2 < bytes) p2 <<= 1;
  const size_t q = p2 / 8;  // p2/2 + k*q, k = 1..4
  for (size_t c = p2 / 2 + q; c <= p2; c += q) if (c >= bytes) return c;
  return p2;
}

hipError_t fmhi::pool_malloc(int device, void** out, size_t bytes) {
  *out = nullptr;
  if (device < 0 || device >= 64 || bytes > kPoolMaxBlock) return hipMalloc(out, bytes ? bytes : 16);
  DevicePool& pool = g_pool[device];
  const size_t c = pool_class(bytes);
  {
    std::lock_guard<std::mutex> lock(pool.mu);
    auto it = pool.free_lists.find(c);
    if (it != pool.free_lists.end() && !it->second.empty()) {
      *out = it->second.back();
      it->second.pop_back();
      pool.cached_bytes -= c;
      pool.live[*out] = c;
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, c);
  if (e != hipSuccess) {  // out of memory: give the cache back and retry once
    std::vector<void*> drop;
    {
      std::lock_guard<std::mutex> lock(pool.mu);
      for (auto& kv : pool.free_lists) { drop.insert(drop.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
      pool.cached_bytes = 0;
    }
    for (void* p : drop) (void)hipFree(p);
    (void)hipGetLastError();
    e = hipMalloc(out, c);
  }
  if (e == hipSuccess) { std::lock_guard<std::mutex> lock(pool.mu); pool.live[*out] = c; }
  return e;
}

void fmhi::pool_free(int device, void* p) {
  if (!p) return;
  if (device >= 0 && device < 64) {
    DevicePool& pool = g_pool[device];
    std::lock_guard<std::mutex> lock(pool.mu);
    auto it = pool.live.find(p);
    if (it != pool.live.end()) {
      const size_t c = it->second;
      pool.live.erase(it);
      if (pool.cached_bytes + c <= kPoolMaxCached) { pool.free_lists[c].push_back(p); pool.cached_bytes += c; return; }
    }
  }
  (void)hipFree(p);
}

void fmhi::pool_trim(int device) {
  if (device < 0 || device >= 64) return;
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> lock(g_pool[device].mu);
    for (auto& kv : g_pool[device].free_lists) { drop.insert(drop.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
    g_pool[device].cached_bytes = 0;
  }
  for (void* p : drop) (void)hipFree(p);
}

extern "C" int fmh_device_info(int device, char* h_name, size_t name_cap, int* h_cus, uint64_t* h_mem) {
  FMH_TRY(use_device(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (h_name && name_cap) snprintf(h_name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  if (h_cus) *h_cus = prop.multiProcessorCount;
  if (h_mem) *h_mem = (uint64_t)prop.totalGlobalMem;
  return FMH_OK;
}

extern "C" int fmh_device_alloc(int device, size_t bytes, void** d_out) {
  if (!d_out) return fail(FMH_ERR_INVALID, "d_out is NULL");
  FMH_TRY(use_device(device));
  HIP_TRY(pool_malloc(device, d_out, bytes));
  return FMH_OK;
}
extern "C" int fmh_device_free(int device, void* d_ptr) {
  FMH_TRY(use_device(device));
  pool_free(device, d_ptr);
  return FMH_OK;
}
// FMH_STREAM_PER_THREAD (include/ferromic_hip.h) is HIP's per-thread stream handle
[[maybe_unused]] static const bool kStreamPerThreadChecked = [] {
  if ((void*)hipStreamPerThread != FMH_STREAM_PER_THREAD) { fprintf(stderr, "libferromic_hip: FMH_STREAM_PER_THREAD is not hipStreamPerThread\n"); abort(); }
  return true;
}();
extern "C" int fmh_copy_to_host(int device, void* h_dst, const void* d_src, size_t bytes, void* stream) {
  FMH_TRY(use_device(device));
  HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return FMH_OK;
}
extern "C" int fmh_copy_to_device(int device, void* d_dst, const void* h_src, size_t bytes, void* stream) {
  FMH_TRY(use_device(device));
  HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return FMH_OK;
}
extern "C" int fmh_device_zero(int device, void* d_ptr, size_t bytes, void* stream) {
  FMH_TRY(use_device(device));
  if (bytes) HIP_TRY(hipMemsetAsync(d_ptr, 0, bytes, (hipStream_t)stream));
  return FMH_OK;
}
extern "C" int fmh_stream_synchronize(int device, void* stream) {
  FMH_TRY(use_device(device));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return FMH_OK;
}

// ------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------
// LDS the sweep needs for P (padded) groups of an `nvec`-vector row, masks as bytes (fast) or as bits (8x the width);
// the kernels take at most 150 KiB.  sweep_lds_bytes() > limit means "does not fit in LDS in either form".
// The limit comes from the device: the LDS a workgroup may opt into, less the kernels' static arrays (block reduction, W&C reciprocal
// table: about 6 KiB) - 150 KiB on gfx950's 160 KiB.  g_lds[device] is filled by workspace(); 150 KiB until then.
static const size_t kSweepLdsLimit = 150 * 1024;
static std::atomic<size_t> g_lds_optin[64], g_lds_per_cu[64];
static size_t device_lds_limit(int device) {
  size_t optin = device >= 0 && device < 64 ? g_lds_optin[device].load(std::memory_order_relaxed) : 0;
  const size_t per_cu = device >= 0 && device < 64 ? g_lds_per_cu[device].load(std::memory_order_relaxed) : 0;
  if (optin < per_cu) optin = per_cu;  // runtimes that report the 64-KiB default as the opt-in limit: a workgroup may take the CU's LDS
  return optin >= (size_t)128 * 1024 ? optin - 10 * 1024 : kSweepLdsLimit;
}
static size_t device_lds_per_cu(int device) {
  const size_t v = device >= 0 && device < 64 ? g_lds_per_cu[device].load(std::memory_order_relaxed) : 0;
  return v ? v : (size_t)160 * 1024;
}
static size_t sweep_lds_limit(int device) { return device_lds_limit(device); }
// the eight-group W&C kernels keep their regional sums through a static LDS scratch (sweep_kernels.hpp, wc_xpose_scratch): that much less for masks
static size_t wc_lds_limit(int device, int padded) { return sweep_lds_limit(device) - (padded >= 5 ? fmh::kWcXposeLdsBytes + 1024 : 0); }
static size_t sweep_lds_bytes(int padded, size_t nvec) { return (size_t)padded * round_up(nvec, 64) * 2; }

static int check_dims(size_t variants, size_t samples, size_t ploidy) {
  if (ploidy == 0 || samples == 0) return fail(FMH_ERR_INVALID, "samples and ploidy must be positive");
  const unsigned long long cols = (unsigned long long)samples * ploidy;
  if (cols > 0x7FFFFFF0ull) return fail(FMH_ERR_INVALID, "samples*ploidy = %llu too large", cols);
  (void)variants;
  return FMH_OK;
}

extern "C" int fmh_matrix_alloc(size_t variants, size_t samples, size_t ploidy, int with_missing,
                                uint8_t max_allele, int device, fmh_matrix** out) {
  if (!out) return fail(FMH_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMH_TRY(check_dims(variants, samples, ploidy));
  FMH_TRY(use_device(device));
  fmh_matrix* m = new fmh_matrix();
  m->device = device;
  m->variants = variants;
  m->samples = samples;
  m->ploidy = ploidy;
  m->columns = (uint32_t)(samples * ploidy);
  const int env_align = (int)options().pitch_align.load();
  m->pitch = round_up(m->columns, env_align >= 16 && env_align % 16 == 0 ? env_align : 16);
  m->nvec = (uint32_t)(round_up(m->columns, 16) / 16);
  m->bits_pitch = with_missing ? round_up(m->pitch / 8, 4) : 0;
  m->max_allele = max_allele;
  m->has_missing = with_missing != 0;
  const size_t bytes = variants * m->pitch;
  hipError_t e = pool_malloc(device, (void**)&m->data, bytes);
  if (e == hipSuccess && with_missing) {
    const size_t bb = variants * m->bits_pitch;
    e = pool_malloc(device, (void**)&m->bits, bb);
  }
  if (e != hipSuccess) {
    pool_free(device, m->data);
    delete m;
    return fail(FMH_ERR_HIP, "hipMalloc of %zu-byte matrix failed: %s", bytes, hipGetErrorString(e));
  }
  *out = m;
  return FMH_OK;
}

// ---- bit-packed image -------------------------------------------------------------------------------------------
// Rows wider than this keep the byte layout: their masks would not fit LDS even as bits for two groups.
static const uint32_t kPackMaxColumns = 600000;
static bool packable(uint8_t max_allele, uint32_t columns) { return max_allele <= 7 && columns <= kPackMaxColumns; }

static int alloc_planes(fmh_matrix* m) {
  m->plane_pitch = round_up(((size_t)m->columns + 7) / 8, 16);
  m->pvec = (uint32_t)(m->plane_pitch / 16);
  const size_t bytes = std::max<size_t>(m->variants, 1) * m->plane_pitch;
  hipError_t e = pool_malloc(m->device, (void**)&m->p0, bytes);
  if (e == hipSuccess && m->max_allele >= 2) e = pool_malloc(m->device, (void**)&m->p1, bytes);
  if (e == hipSuccess && m->max_allele >= 4) e = pool_malloc(m->device, (void**)&m->p2, bytes);
  if (e == hipSuccess && m->has_missing) e = pool_malloc(m->device, (void**)&m->pc, bytes);
  if (e != hipSuccess) {
    pool_free(m->device, m->p0); pool_free(m->device, m->p1); pool_free(m->device, m->p2); pool_free(m->device, m->pc);
    m->p0 = m->p1 = m->p2 = m->pc = nullptr;
    return fail(FMH_ERR_HIP, "hipMalloc of the %zu-byte packed planes failed: %s", bytes, hipGetErrorString(e));
  }
  return FMH_OK;
}
static void free_planes(fmh_matrix* m) {
  pool_free(m->device, m->p0); pool_free(m->device, m->p1); pool_free(m->device, m->p2); pool_free(m->device, m->pc);
  pool_free(m->device, m->row_hi);
  pool_free(m->device, m->row_gap);
  m->p0 = m->p1 = m->p2 = m->pc = nullptr;
  m->row_hi = m->row_gap = nullptr;
}
// after the planes of a matrix have been written: which rows have a bit above plane 0 (row_hi_kernel) and which have an uncalled column
// (row_gap_kernel) - the sweeps read the upper / called planes of those rows only.  FMH_ROW_HI=0: no tables, every plane of every row is read
// as before round 4.  (A matrix of a few thousand rows is swept in one launch-bound round either way: no tables, no extra launches and
// synchronisation per small region of run_vcf; FMH_ROW_HI=2 builds them for any size - tests.)
static int mark_upper_plane_rows(fmh_matrix* m) {
  const long long mode = options().row_hi.load();
  if ((!m->p1 && !m->pc) || m->variants == 0 || mode == 0 || (mode != 2 && m->variants < 4096)) return FMH_OK;
  const int blocks = (int)std::min<size_t>((m->variants * 16 + 255) / 256, 1 << 16);
  hipError_t e = hipSuccess;
  if (m->p1) {
    if (!m->row_hi && pool_malloc(m->device, (void**)&m->row_hi, m->variants) != hipSuccess) m->row_hi = nullptr;  // no table: every row is read in full
    if (m->row_hi) hipLaunchKernelGGL(row_hi_kernel, dim3(blocks), dim3(256), 0, 0, (const uint8_t*)m->p1, (const uint8_t*)m->p2, m->plane_pitch, m->variants, m->row_hi);
    e = hipGetLastError();
  }
  if (e == hipSuccess && m->pc) {
    if (!m->row_gap && pool_malloc(m->device, (void**)&m->row_gap, m->variants) != hipSuccess) m->row_gap = nullptr;
    if (m->row_gap) hipLaunchKernelGGL(row_gap_kernel, dim3(blocks), dim3(256), 0, 0, (const uint8_t*)m->pc, m->plane_pitch, m->variants, m->columns, m->row_gap);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(0);
  if (e != hipSuccess) return fail(FMH_ERR_HIP, "marking the rows with alleles above 1 / uncalled columns failed: %s", hipGetErrorString(e));
  return FMH_OK;
}
// byte rows (and, when `bits` is given, their called rows) -> planes rows [row0, row0 + rows)
// `d_overflow` (one zeroed device word, may be null) is set when a called entry carries an allele bit the planes do not store
static hipError_t pack_rows(fmh_matrix* m, const uint8_t* data, size_t pitch, const uint8_t* bits, size_t bits_pitch, size_t row0, size_t rows,
                            hipStream_t st, unsigned int* d_overflow = nullptr) {
  if (rows == 0) return hipSuccess;
  const size_t total = rows * (m->plane_pitch / 4);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 1 << 20);
  const size_t off = row0 * m->plane_pitch;
  hipLaunchKernelGGL(pack_rows_kernel, dim3(blocks), dim3(256), 0, st, data, pitch, bits, bits_pitch, rows, m->columns, m->p0 + off,
                     m->p1 ? m->p1 + off : nullptr, m->p2 ? m->p2 + off : nullptr, m->pc ? m->pc + off : nullptr, m->plane_pitch, d_overflow);
  return hipGetLastError();
}
// planes rows [row0, row0 + rows) -> byte rows of `pitch` bytes
hipError_t fmhi::unpack_rows(const fmh_matrix* m, size_t row0, size_t rows, uint8_t* data, size_t pitch, hipStream_t st) {
  if (rows == 0) return hipSuccess;
  const size_t total = rows * (pitch / 16);
  const int blocks = (int)std::min<size_t>((total + 255) / 256, 1 << 20);
  const size_t off = row0 * m->plane_pitch;
  hipLaunchKernelGGL(unpack_rows_kernel, dim3(blocks), dim3(256), 0, st, m->p0 + off, m->p1 ? m->p1 + off : nullptr, m->p2 ? m->p2 + off : nullptr,
                     m->pc ? m->pc + off : nullptr, m->plane_pitch, rows, data, pitch);
  return hipGetLastError();
}

extern "C" int fmh_matrix_pack(fmh_matrix* m, int release_bytes) {
  if (!m) return fail(FMH_ERR_INVALID, "matrix is NULL");
  if (!m->data) return m->p0 ? FMH_OK : fail(FMH_ERR_INVALID, "matrix has no byte image to pack");
  if (!packable(m->max_allele, m->columns))
    return fail(FMH_ERR_UNSUPPORTED, "the packed layout holds alleles 0..7 on rows of at most %u columns (max_allele %u, %u columns)", kPackMaxColumns,
                (unsigned)m->max_allele, m->columns);
  FMH_TRY(use_device(m->device));
  if (m->p0 && (((m->max_allele >= 2) != (m->p1 != nullptr)) || ((m->max_allele >= 4) != (m->p2 != nullptr)))) free_planes(m);  // max_allele changed since the last pack
  if (!m->p0) FMH_TRY(alloc_planes(m));
  unsigned int* d_overflow = nullptr;
  unsigned int overflow = 0;
  hipError_t e = pool_malloc(m->device, (void**)&d_overflow, 4);
  if (e == hipSuccess) e = hipMemset(d_overflow, 0, 4);
  if (e == hipSuccess) e = pack_rows(m, m->data, m->pitch, m->bits, m->bits_pitch, 0, m->variants, 0, d_overflow);
  if (e == hipSuccess) e = hipMemcpy(&overflow, d_overflow, 4, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  pool_free(m->device, d_overflow);
  if (e != hipSuccess) return fail(FMH_ERR_HIP, "packing failed: %s", hipGetErrorString(e));
  if (overflow) {  // the planes would silently count allele 2 as 0 or allele 5 as 1: refuse, the byte rows stay in use
    free_planes(m);
    return fail(FMH_ERR_INVALID, "a called entry holds an allele above max_allele = %u: the packed layout cannot represent it", (unsigned)m->max_allele);
  }
  FMH_TRY(mark_upper_plane_rows(m));
  if (release_bytes && m->owns) {
    pool_free(m->device, m->data);
    pool_free(m->device, m->bits);
    m->data = nullptr;
    m->bits = nullptr;
  }
  return FMH_OK;
}

extern "C" int fmh_matrix_create(const uint8_t* h_data, const uint64_t* h_missing, size_t variants, size_t samples,
                                 size_t ploidy, uint8_t max_allele, int device, fmh_matrix** out) {
  if (!out) return fail(FMH_ERR_INVALID, "out is NULL");
  if (!h_data && variants) return fail(FMH_ERR_INVALID, "h_data is NULL");
  FMH_TRY(check_dims(variants, samples, ploidy));
  const bool packed = packable(max_allele, (uint32_t)(samples * ploidy)) && !layout_bytes_forced();
  if (packed) {
    // the resident image is the packed one: the bytes pass through a staging slab and are never kept
    *out = nullptr;
    FMH_TRY(use_device(device));
    fmh_matrix* m = new fmh_matrix();
    m->device = device;
    m->variants = variants; m->samples = samples; m->ploidy = ploidy;
    m->columns = (uint32_t)(samples * ploidy);
    m->pitch = round_up(m->columns, 16);
    m->nvec = (uint32_t)(m->pitch / 16);
    m->max_allele = max_allele;
    m->has_missing = h_missing != nullptr;
    auto bail = [&](int code) { fmh_matrix_destroy(m); return code; };
    if (alloc_planes(m) != FMH_OK) return bail(FMH_ERR_HIP);
    if (variants == 0) { *out = m; return FMH_OK; }
    // packed on the host (threads + SSE2) into pinned staging, ceil(H / 8) bytes per site and plane over PCIe (upload.hip)
    bool overflow = false;
    const int rc = upload_planes_from_bytes(m, h_data, h_missing, &overflow);
    if (rc != FMH_OK) return bail(rc);
    if (overflow) return bail(fail(FMH_ERR_INVALID, "a called entry holds an allele above max_allele = %u: the packed layout cannot represent it", (unsigned)max_allele));
    if (const int rh = mark_upper_plane_rows(m); rh != FMH_OK) return bail(rh);
    *out = m;
    return FMH_OK;
  }
  FMH_TRY(fmh_matrix_alloc(variants, samples, ploidy, h_missing != nullptr, max_allele, device, out));
  fmh_matrix* m = *out;
  auto bail = [&](int code) { fmh_matrix_destroy(m); *out = nullptr; return code; };
  if (variants == 0) return FMH_OK;
  // zero the padding columns, then a pitched copy of the packed rows
  hipError_t e = hipMemset(m->data, 0, variants * m->pitch);
  if (e == hipSuccess)
    e = hipMemcpy2D(m->data, m->pitch, h_data, m->columns, m->columns, variants, hipMemcpyHostToDevice);
  if (e != hipSuccess) return bail(fail(FMH_ERR_HIP, "matrix upload failed: %s", hipGetErrorString(e)));
  if (h_missing) {
    const size_t words = (variants * (size_t)m->columns + 63) / 64;
    unsigned long long* d_words = nullptr;
    e = pool_malloc(device, (void**)&d_words, words * 8);
    if (e == hipSuccess) e = hipMemcpy(d_words, h_missing, words * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      const size_t total = variants * m->bits_pitch;
      const int blocks = (int)std::min<size_t>((total + 255) / 256, 65535);
      hipLaunchKernelGGL(missing_to_called_rows, dim3(blocks), dim3(256), 0, 0, d_words, variants, m->columns,
                         m->bits, m->bits_pitch);
      e = hipGetLastError();
      if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    pool_free(device, d_words);
    if (e != hipSuccess) return bail(fail(FMH_ERR_HIP, "missing-mask upload failed: %s", hipGetErrorString(e)));
  }
  return FMH_OK;
}

extern "C" int fmh_matrix_create_packed(const uint8_t* h_plane0, const uint8_t* h_plane1, const uint8_t* h_plane2, const uint8_t* h_called,
                                        size_t h_pitch, size_t variants, size_t samples, size_t ploidy, uint8_t max_allele, int device, fmh_matrix** out) {
  if (!out) return fail(FMH_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMH_TRY(check_dims(variants, samples, ploidy));
  const uint32_t columns = (uint32_t)(samples * ploidy);
  if (!packable(max_allele, columns)) return fail(FMH_ERR_UNSUPPORTED, "the packed layout holds alleles 0..7 on rows of at most %u columns", kPackMaxColumns);
  if (!h_plane0 && variants) return fail(FMH_ERR_INVALID, "h_plane0 is NULL");
  if ((max_allele >= 2) != (h_plane1 != nullptr) || (max_allele >= 4) != (h_plane2 != nullptr))
    return fail(FMH_ERR_INVALID, "max_allele %u needs %d allele plane(s)", (unsigned)max_allele, max_allele >= 4 ? 3 : (max_allele >= 2 ? 2 : 1));
  if (variants && h_pitch < ((size_t)columns + 7) / 8) return fail(FMH_ERR_INVALID, "h_pitch %zu is smaller than a row of %u columns", h_pitch, columns);
  FMH_TRY(use_device(device));
  fmh_matrix* m = new fmh_matrix();
  m->device = device;
  m->variants = variants; m->samples = samples; m->ploidy = ploidy;
  m->columns = columns;
  m->pitch = round_up(columns, 16);
  m->nvec = (uint32_t)(m->pitch / 16);
  m->max_allele = max_allele;
  m->has_missing = h_called != nullptr;
  if (alloc_planes(m) != FMH_OK) { fmh_matrix_destroy(m); return FMH_ERR_HIP; }
  const uint8_t* planes[4] = {h_plane0, h_plane1, h_plane2, h_called};
  const int rc = upload_planes_from_planes(m, planes, h_pitch);
  if (rc != FMH_OK) { fmh_matrix_destroy(m); return rc; }
  if (const int rh = mark_upper_plane_rows(m); rh != FMH_OK) { fmh_matrix_destroy(m); return rh; }
  *out = m;
  return FMH_OK;
}

extern "C" int fmh_matrix_wrap(void* d_data, size_t pitch, void* d_bits, size_t bits_pitch, size_t variants,
                               size_t samples, size_t ploidy, uint8_t max_allele, int device, fmh_matrix** out) {
  if (!out) return fail(FMH_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMH_TRY(check_dims(variants, samples, ploidy));
  FMH_TRY(use_device(device));
  const size_t cols = samples * ploidy;
  if (!d_data) return fail(FMH_ERR_INVALID, "d_data is NULL");
  if (pitch % 16 != 0 || pitch < cols) return fail(FMH_ERR_INVALID, "pitch %zu must be a multiple of 16 and >= %zu", pitch, cols);
  if (((uintptr_t)d_data) % 16 != 0) return fail(FMH_ERR_INVALID, "d_data must be 16-byte aligned");
  // the u8 kernels read a called row as 16-bit words, pack_rows_kernel as 4-byte groups: what the header states is enforced
  if (d_bits && (bits_pitch % 4 != 0 || bits_pitch * 8 < round_up(cols, 16)))
    return fail(FMH_ERR_INVALID, "bits_pitch %zu must be a multiple of 4 and hold %zu bits", bits_pitch, round_up(cols, 16));
  if (d_bits && ((uintptr_t)d_bits) % 4 != 0) return fail(FMH_ERR_INVALID, "d_called_bits must be 4-byte aligned");
  fmh_matrix* m = new fmh_matrix();
  m->device = device;
  m->data = (uint8_t*)d_data;
  m->bits = (uint8_t*)d_bits;
  m->pitch = pitch;
  m->bits_pitch = d_bits ? bits_pitch : 0;
  m->variants = variants;
  m->samples = samples;
  m->ploidy = ploidy;
  m->columns = (uint32_t)cols;
  m->nvec = (uint32_t)(round_up(cols, 16) / 16);
  m->max_allele = max_allele;
  m->owns = false;
  m->has_missing = d_bits != nullptr;
  *out = m;
  return FMH_OK;
}

extern "C" int fmh_matrix_destroy(fmh_matrix* m) {
  if (!m) return FMH_OK;
  (void)hipSetDevice(m->device);
  if (m->owns) {
    pool_free(m->device, m->data);
    pool_free(m->device, m->bits);
  }
  free_planes(m);
  delete m;
  return FMH_OK;
}

extern "C" int fmh_matrix_info(const fmh_matrix* m, size_t* variants, size_t* samples, size_t* ploidy, size_t* pitch,
                               size_t* bits_pitch, int* has_missing, uint8_t* max_allele, int* device) {
  if (!m) return fail(FMH_ERR_INVALID, "matrix is NULL");
  if (variants) *variants = m->variants;
  if (samples) *samples = m->samples;
  if (ploidy) *ploidy = m->ploidy;
  if (pitch) *pitch = m->pitch;
  if (bits_pitch) *bits_pitch = m->bits_pitch;
  if (has_missing) *has_missing = m->has_missing;
  if (max_allele) *max_allele = m->max_allele;
  if (device) *device = m->device;
  return FMH_OK;
}

extern "C" int fmh_matrix_device_ptrs(const fmh_matrix* m, void** d_data, void** d_bits) {
  if (!m) return fail(FMH_ERR_INVALID, "matrix is NULL");
  if (d_data) *d_data = m->data;
  if (d_bits) *d_bits = m->bits;
  return FMH_OK;
}

extern "C" int fmh_matrix_download(const fmh_matrix* m, uint8_t* h_data, uint64_t* h_missing) {
  if (!m) return fail(FMH_ERR_INVALID, "matrix is NULL");
  if (!h_data) return fail(FMH_ERR_INVALID, "h_data is NULL");
  FMH_TRY(use_device(m->device));
  if (m->variants == 0) return FMH_OK;
  if (m->data) {
    HIP_TRY(hipMemcpy2D(h_data, m->columns, m->data, m->pitch, m->columns, m->variants, hipMemcpyDeviceToHost));
  } else {  // packed image only: unpack through a staging slab
    const size_t slab_rows = std::max<size_t>(1, std::min<size_t>(m->variants, ((size_t)256 << 20) / m->pitch));
    uint8_t* d_slab = nullptr;
    HIP_TRY(pool_malloc(m->device, (void**)&d_slab, slab_rows * m->pitch));
    hipError_t e = hipSuccess;
    for (size_t r0 = 0; r0 < m->variants && e == hipSuccess; r0 += slab_rows) {
      const size_t rows = std::min(slab_rows, m->variants - r0);
      e = unpack_rows(m, r0, rows, d_slab, m->pitch, 0);
      if (e == hipSuccess) e = hipMemcpy2D(h_data + r0 * m->columns, m->columns, d_slab, m->pitch, m->columns, rows, hipMemcpyDeviceToHost);
    }
    pool_free(m->device, d_slab);
    if (e != hipSuccess) return fail(FMH_ERR_HIP, "matrix download failed: %s", hipGetErrorString(e));
  }
  if (m->has_missing) {
    if (!h_missing) return fail(FMH_ERR_INVALID, "matrix has a missing mask but h_missing is NULL");
    const size_t words = (m->variants * (size_t)m->columns + 63) / 64;
    unsigned long long* d_words = nullptr;
    HIP_TRY(hipMalloc((void**)&d_words, words * 8));
    const int blocks = (int)std::min<size_t>((words + 255) / 256, 65535);
    hipLaunchKernelGGL(called_rows_to_missing, dim3(blocks), dim3(256), 0, 0, m->bits ? m->bits : m->pc, m->bits ? m->bits_pitch : m->plane_pitch,
                       m->variants, m->columns, d_words, words);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(h_missing, d_words, words * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_words);
    if (e != hipSuccess) return fail(FMH_ERR_HIP, "missing-mask download failed: %s", hipGetErrorString(e));
  }
  return FMH_OK;
}

extern "C" int fmh_matrix_scan_max_allele(const fmh_matrix* m, uint8_t* h_max, void* stream) {
  if (!m || !h_max) return fail(FMH_ERR_INVALID, "NULL argument");
  FMH_TRY(use_device(m->device));
  unsigned int* d_out = nullptr;
  HIP_TRY(hipMalloc((void**)&d_out, 4));
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(d_out, 0, 4, st);
  unsigned int host = 0;
  if (e == hipSuccess && m->variants && m->data) {
    const size_t total = m->variants * (size_t)m->columns;
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(max_allele_kernel, dim3(blocks), dim3(256), 0, st, m->data, m->pitch, m->bits, m->bits_pitch,
                       m->variants, m->columns, d_out);
    e = hipGetLastError();
  } else if (e == hipSuccess && m->variants) {
    const size_t total = m->variants * (m->plane_pitch / 4);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(packed_max_allele_kernel, dim3(blocks), dim3(256), 0, st, m->p0, m->p1, m->p2, m->pc, m->plane_pitch, m->variants, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(&host, d_out, 4, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_out);
  if (e != hipSuccess) return fail(FMH_ERR_HIP, "max-allele scan failed: %s", hipGetErrorString(e));
  *h_max = (uint8_t)host;
  return FMH_OK;
}

extern "C" int fmh_matrix_generate(fmh_matrix* m, uint64_t seed, uint64_t first_site, const uint32_t* h_thr,
                                   const uint8_t* h_pop_of_column, int n_pops, uint32_t missing_thr, void* stream) {
  if (!m || !h_thr || !h_pop_of_column) return fail(FMH_ERR_INVALID, "NULL argument");
  if (n_pops < 1 || n_pops > 255) return fail(FMH_ERR_INVALID, "n_pops out of range");
  for (uint32_t h = 0; h < m->columns; ++h)
    if (h_pop_of_column[h] >= n_pops) return fail(FMH_ERR_INVALID, "pop_of_column[%u] = %u >= n_pops", h, h_pop_of_column[h]);
  if (!m->data) return fail(FMH_ERR_INVALID, "the generator writes the byte image, which this matrix has released");
  if (missing_thr && !m->bits) return fail(FMH_ERR_INVALID, "missing_threshold set but the matrix has no mask");
  FMH_TRY(use_device(m->device));
  if (m->variants == 0) return FMH_OK;
  hipStream_t st = (hipStream_t)stream;
  uint32_t* d_thr = nullptr;
  uint8_t* d_pop = nullptr;
  const size_t thr_bytes = (size_t)n_pops * m->variants * 4;
  HIP_TRY(hipMalloc((void**)&d_thr, thr_bytes));
  hipError_t e = hipMalloc((void**)&d_pop, m->columns);
  if (e == hipSuccess) e = hipMemcpyAsync(d_thr, h_thr, thr_bytes, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_pop, h_pop_of_column, m->columns, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    const size_t total = m->variants * (m->pitch / 16);
    const int blocks = (int)std::min<size_t>((total + 255) / 256, 1 << 20);
    hipLaunchKernelGGL(generate_kernel, dim3(blocks), dim3(256), 0, st, m->data, m->pitch, m->bits, m->bits_pitch,
                       m->variants, m->columns, m->nvec, seed, first_site, d_thr, d_pop, missing_thr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d_thr);
  if (d_pop) (void)hipFree(d_pop);
  if (e != hipSuccess) return fail(FMH_ERR_HIP, "generate failed: %s", hipGetErrorString(e));
  if (m->max_allele < 1) m->max_allele = 1;
  if (m->p0) FMH_TRY(fmh_matrix_pack(m, 0));  // keep an existing packed image in step
  return FMH_OK;
}

// ------------------------------------------------------------------------------------------------
// groups
// ------------------------------------------------------------------------------------------------
static int padded_groups(int n) { return n <= 1 ? 1 : n <= 2 ? 2 : n <= 4 ? 4 : 8; }
// The group count the W&C kernel of a sweep is instantiated with: the padded one (1, 2, 4, 8), or EXACTLY five, six or seven on a packed
// biallelic matrix with nothing missing - the slots of the padding are then not even compiled in (sweep_launch.inc).
int fmhi::wc_kernel_groups(const fmh_matrix* m, const fmh_groups* g) {
  if (!m || !g) return 0;
  const bool packed = m->p0 && !(m->data && layout_bytes_forced());
  if (g->n_groups >= 5 && g->n_groups <= 7 && packed && !m->has_missing && m->max_allele <= 1 && options().wc_exact.load() != 0) return g->n_groups;
  return g->padded;
}

extern "C" int fmh_groups_create(const fmh_matrix* m, const uint8_t* h_mask, int n_groups, fmh_groups** out) {
  if (!out) return fail(FMH_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!m || !h_mask) return fail(FMH_ERR_INVALID, "NULL argument");
  if (n_groups < 1 || n_groups > FMH_MAX_GROUPS) return fail(FMH_ERR_INVALID, "n_groups %d out of range 1..%d", n_groups, FMH_MAX_GROUPS);
  FMH_TRY(use_device(m->device));
  fmh_groups* g = new fmh_groups();
  g->device = m->device;
  g->n_groups = n_groups;
  g->padded = padded_groups(n_groups);
  g->pitch = m->pitch;
  g->mask_pitch = round_up(m->pitch, 2048);
  g->columns = m->columns;
  g->host_mask.assign(h_mask, h_mask + (size_t)n_groups * m->columns);
  // byte masks and, behind them, the same masks as one bit per column: ONE device block and ONE copy (every statistic of the Python module
  // makes its groups per call; two blocking copies were 24 us of a 50-us call, tools/measure_call_overheads.py)
  const size_t bytes_len = (size_t)g->padded * g->mask_pitch, bits_len = (size_t)g->padded * (g->mask_pitch / 16) * 2;
  // third form, for the flat-tile route of short packed rows (sweep_flat_kernels.hpp): the bit masks interleaved [vector][group][4 dwords], the
  // vector count padded to a multiple of 4 with zeros - scalar loads fetch every group's mask of one vector at once
  const size_t pvec = ((size_t)m->columns + 127) / 128;
  const size_t flat_vecs = pvec <= (size_t)kFlatMaskMaxVec ? round_up(pvec, 4) : 0;
  const size_t flat_len = flat_vecs * g->padded * 16;
  std::vector<uint8_t> staged(bytes_len + bits_len + flat_len, 0);
  uint16_t* bits = reinterpret_cast<uint16_t*>(staged.data() + bytes_len);  // bytes_len is a multiple of 2048
  for (int p = 0; p < n_groups; ++p) {
    uint64_t cnt = 0;
    for (uint32_t h = 0; h < m->columns; ++h) {
      const uint8_t v = h_mask[(size_t)p * m->columns + h] ? 1 : 0;
      staged[(size_t)p * g->mask_pitch + h] = v;
      if (v) bits[(size_t)p * (g->mask_pitch / 16) + (h >> 4)] |= (uint16_t)(1u << (h & 15));
      cnt += v;
    }
    g->sizes[p] = cnt;
  }
  if (flat_len) {  // bits_len is a multiple of 256: the image is 16-byte aligned
    uint8_t* flat = staged.data() + bytes_len + bits_len;
    for (int p = 0; p < n_groups; ++p)
      for (size_t v = 0; v < pvec; ++v)
        memcpy(flat + (v * g->padded + p) * 16, reinterpret_cast<const uint8_t*>(bits) + (size_t)p * (g->mask_pitch / 8) + v * 16, 16);
  }
  hipError_t e = pool_malloc(g->device, (void**)&g->masks, staged.size());
  // (on the calling thread's own stream: a plain hipMemcpy runs on the legacy default stream, which waits for - and holds up - every blocking
  // stream of the device, i.e. the sweeps of all other host threads; the block is a fresh allocation)
  if (e == hipSuccess) e = hipMemcpyAsync(g->masks, staged.data(), staged.size(), hipMemcpyHostToDevice, hipStreamPerThread);
  if (e == hipSuccess) e = hipStreamSynchronize(hipStreamPerThread);
  if (e != hipSuccess) {
    pool_free(g->device, g->masks);
    delete g;
    return fail(FMH_ERR_HIP, "group mask upload failed: %s", hipGetErrorString(e));
  }
  g->mask_bits = reinterpret_cast<uint16_t*>(g->masks + bytes_len);
  g->mask_flat = flat_len ? reinterpret_cast<uint32_t*>(g->masks + bytes_len + bits_len) : nullptr;
  *out = g;
  return FMH_OK;
}

extern "C" int fmh_groups_destroy(fmh_groups* g) {
  if (!g) return FMH_OK;
  (void)hipSetDevice(g->device);
  pool_free(g->device, g->masks);  // (mask_bits lives in the same block)
  delete g;
  return FMH_OK;
}

extern "C" int fmh_groups_sizes(const fmh_groups* g, int* n_groups, uint64_t* h_sizes) {
  if (!g) return fail(FMH_ERR_INVALID, "groups is NULL");
  if (n_groups) *n_groups = g->n_groups;
  if (h_sizes) for (int p = 0; p < g->n_groups; ++p) h_sizes[p] = g->sizes[p];
  return FMH_OK;
}

// ------------------------------------------------------------------------------------------------
// per-device workspace: block partials, totals, harmonic table, timing events
// ------------------------------------------------------------------------------------------------
static Workspace g_ws[64];
static std::mutex g_ws_mutex;
static std::atomic<int> g_timing{0};             // 0 = off, n = time every n-th sweep (fmh_timing_enable)
static std::atomic<uint64_t> g_timing_seq{0};
static double g_timing_ms = 0.0, g_timing_min = 0.0, g_timing_max = 0.0;
static uint64_t g_timing_launches = 0;

int fmhi::workspace(int device, Workspace** out) {
  if (device < 0 || device >= 64) return fail(FMH_ERR_INVALID, "device index %d unsupported", device);
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  Workspace& w = g_ws[device];
  if (!w.ready) {
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    w.cus = prop.multiProcessorCount;
    w.max_grid = w.cus * 8;
    g_lds_optin[device].store(prop.sharedMemPerBlockOptin ? prop.sharedMemPerBlockOptin : prop.sharedMemPerBlock);
    g_lds_per_cu[device].store(prop.maxSharedMemoryPerMultiProcessor);
    w.ready = true;
  }
  *out = &w;
  return FMH_OK;
}

// H_k = sum_{i=1..k} 1/i summed ascending (harmonic(), stats.rs:4234-4240); table index k.
static int ensure_harmonic(Workspace* w, size_t max_k, hipStream_t st) {
  if (w->harmonic && w->harmonic_len > max_k) return FMH_OK;
  std::vector<double> table(max_k + 2);
  double sum = 0.0;
  table[0] = 0.0;
  for (size_t k = 1; k < table.size(); ++k) {
    sum += 1.0 / (double)k;
    table[k] = sum;
  }
  double* fresh = nullptr;
  HIP_TRY(hipMalloc((void**)&fresh, table.size() * 8));
  HIP_TRY(hipMemcpyAsync(fresh, table.data(), table.size() * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (w->harmonic) w->retired_harmonic.push_back(w->harmonic);  // another thread's sweep may still be reading it
  w->harmonic = fresh;
  w->harmonic_len = table.size();
  return FMH_OK;
}

// one sweep's buffers, taken from / returned to the device's pool
static int lease_acquire(Workspace* w, SweepLease** out) {
  {
    std::lock_guard<std::mutex> lock(w->lease_mu);
    if (!w->idle_leases.empty()) { *out = w->idle_leases.back(); w->idle_leases.pop_back(); return FMH_OK; }
  }
  SweepLease* l = new SweepLease();
  HIP_TRY(hipMalloc((void**)&l->part_f64, (size_t)w->max_grid * kMaxF64 * 8));
  HIP_TRY(hipMalloc((void**)&l->part_u64, (size_t)w->max_grid * kMaxU64 * 8));
  HIP_TRY(hipMalloc((void**)&l->out_f64, kMaxF64 * 8));
  HIP_TRY(hipMalloc((void**)&l->out_u64, kMaxU64 * 8));
  HIP_TRY(hipHostMalloc((void**)&l->h_f64, kMaxF64 * 8, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&l->h_u64, kMaxU64 * 8, hipHostMallocDefault));
  HIP_TRY(hipEventCreate(&l->ev0));
  HIP_TRY(hipEventCreate(&l->ev1));
  // a BLOCKING stream: it still orders itself against the legacy default stream (work a caller queued there on the sweep's inputs or
  // outputs - a torch kernel on a wrapped tensor, fmh_device_zero - is seen, as it was when sweeps ran on the NULL stream itself), while
  // the leases' streams do not order against each other
  HIP_TRY(hipStreamCreate(&l->stream));
  *out = l;
  return FMH_OK;
}
struct LeaseHolder {
  Workspace* w = nullptr;
  SweepLease* l = nullptr;
  ~LeaseHolder() {
    if (w && l) { std::lock_guard<std::mutex> lock(w->lease_mu); w->idle_leases.push_back(l); }
  }
};

extern "C" int fmh_timing_enable(int on) { g_timing = on < 0 ? 0 : on; return FMH_OK; }
extern "C" int fmh_timing_reset(void) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  g_timing_ms = 0.0;
  g_timing_min = g_timing_max = 0.0;
  g_timing_launches = 0;
  return FMH_OK;
}
extern "C" int fmh_timing_read_minmax(double* h_min_ms, double* h_max_ms) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  if (h_min_ms) *h_min_ms = g_timing_min;
  if (h_max_ms) *h_max_ms = g_timing_max;
  return FMH_OK;
}
extern "C" int fmh_timing_read(double* ms, uint64_t* launches) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  if (ms) *ms = g_timing_ms;
  if (launches) *launches = g_timing_launches;
  return FMH_OK;
}

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
struct SweepResult {
  double f64[kMaxF64];
  unsigned long long u64[kMaxU64];
};

// Validates, fills the kernel arguments, picks the mask route and enqueues sweep + finalize on `st`: the 128 regional
// accumulators land in b.out_f64 / b.out_u64 (device).  No synchronisation and no use of the shared workspace buffers, so
// callers with private buffers (the pipelined sharded sweeps of comm.hip) need no device lock.  `*launched` = false when
// the row range is empty (nothing was enqueued; the totals are all zero).
int fmhi::enqueue_sweep(const fmh_matrix* m, const fmh_groups* g, int mode, SweepArgs& a, hipStream_t st, const LaunchCtx& ctx,
                        const SweepBuffers& b, const double* harmonic, bool* launched) {
  *launched = false;
  if (!m || !g) return fail(FMH_ERR_INVALID, "matrix or groups is NULL");
  if (g->device != m->device || g->pitch != m->pitch || g->columns != m->columns)
    return fail(FMH_ERR_INVALID, "groups were built for a different matrix geometry");
  if (a.row_begin > m->variants || a.row_count > m->variants - a.row_begin)
    return fail(FMH_ERR_INVALID, "row range [%zu, +%zu) exceeds %zu variants", a.row_begin, a.row_count, m->variants);
  FMH_TRY(use_device(m->device));
  // the packed image when there is one (FMH_LAYOUT=bytes keeps the byte kernels on matrices that still hold their bytes)
  const bool packed = m->p0 && !(m->data && layout_bytes_forced());
  if (packed) {
    a.mv.data = m->p0;
    a.mv.data1 = m->p1;
    a.mv.data2 = m->p2;
    a.mv.bits = m->pc;
    a.mv.row_hi = m->row_hi;
    a.mv.row_gap = m->row_gap;
    a.mv.pitch = m->plane_pitch;
    a.mv.bits_pitch = m->plane_pitch;
    a.mv.nvec = m->pvec;
  } else {
    if (!m->data) return fail(FMH_ERR_INVALID, "matrix holds neither a byte nor a packed image");
    a.mv.data = m->data;
    a.mv.data1 = nullptr;
    a.mv.data2 = nullptr;
    a.mv.bits = m->bits;
    a.mv.row_hi = nullptr;
    a.mv.row_gap = nullptr;
    a.mv.pitch = m->pitch;
    a.mv.bits_pitch = m->bits_pitch;
    a.mv.nvec = m->nvec;
  }
  a.mv.columns = m->columns;
  a.masks = g->masks;
  a.mask_pitch = g->mask_pitch;
  a.mask_bits = g->mask_bits;
  a.mask_flat = g->mask_flat;
  a.flat_slots = 0;
  a.flat_defer = 1;
  for (int p = 0; p < 8; ++p) a.group_size[p] = p < g->n_groups ? (uint32_t)g->sizes[p] : 0;
  a.n_groups = g->n_groups;
  a.max_allele = m->max_allele;
  if (mode & kModeWc) {
    // without missing data every site has n_i = group size: the allele-independent W&C terms are per launch
    const int P = mode == kModeWc ? wc_kernel_groups(m, g) : g->padded;
    uint32_t n8[8];
    bool use8[8];
    for (int i = 0; i < 8; ++i) { n8[i] = i < P ? a.group_size[i] : 0; use8[i] = n8[i] != 0; }
    a.wc_shape[0] = wc_shape<8>(n8, use8);
    int k = 1;
    for (int i = 0; i < P; ++i)
      for (int j = i + 1; j < P; ++j, ++k) {
        const uint32_t pn[2] = {n8[i], n8[j]};
        const bool pu[2] = {true, true};
        a.wc_shape[k] = wc_shape<2>(pn, pu);
      }
    a.wc_live_mask = a.wc_s2ok_mask = 0;
    for (int q = 0; q < k; ++q) {
      if (a.wc_shape[q].live) a.wc_live_mask |= 1u << q;
      if (a.wc_shape[q].s2_ok) a.wc_s2ok_mask |= 1u << q;
    }
  }
  a.part_f64 = b.part_f64;
  a.part_u64 = b.part_u64;
  if (mode & kModeDiversity) {
    if (!harmonic) return fail(FMH_ERR_INVALID, "diversity sweep without a harmonic table");
    a.harmonic = harmonic;
  }
  if (a.row_count == 0) return FMH_OK;
  const bool missing = m->has_missing;
  const bool general = m->max_allele > 1;
  const int P = mode == kModeWc ? wc_kernel_groups(m, g) : g->padded;
  const Options& opt = options();  // one snapshot of the switches per enqueue (atomics; the environment is never read here)
  a.unroll = opt.unroll.load() == 8 ? 8 : 4;
  a.nvec_pad = (uint32_t)round_up(m->nvec, 16 * a.unroll);
  size_t smem = (size_t)P * a.nvec_pad * 16;
  int mask_mode = kMaskLdsBytes;
  int lpr = 16;
  const size_t lds_limit = mode == kModeWc ? wc_lds_limit(m->device, P) : sweep_lds_limit(m->device);
  if (packed) {
    // 128 columns per vector.  Rows of up to 32 vectors (4 096 columns) are shared by FOUR lanes (no idle vector slots
    // on short rows, a two-step reduction: C2 0.081 -> 0.042 ms, C3 0.94 -> 0.63 ms), wider ones by the sixteen lanes of a
    // DPP row (C4's 40 vectors: 1.32 vs 1.34 ms, 200 000 columns: 0.46 vs 0.66 ms); the batch depth U (vectors per lane in
    // flight per trip) is the one with the fewest padded slots, ties to the deeper batch.  (Eight lanes per row measured between
    // the two everywhere - C4 1.37 ms - and is not built.  Round 3 built it once more for the narrow rows, where eight lanes read whole
    // 128-byte lines - 1 000 haplotypes: one contiguous KB per load instruction -: Hudson +4...+10 % at 1 000 haplotypes, +-1 % at 2 500; four-group
    // W&C +3...+18 %, summaries -1...+8 % (profiles/r03/ab_eight_lanes_per_row.jsonl).  The 64-byte segments of the four-lane rows are not what holds them back.)
    const int env_punroll = (int)opt.packed_unroll.load();
    const int env_lpr = (int)opt.packed_lpr.load();
    lpr = env_lpr == 4 || env_lpr == 16 ? env_lpr : (m->pvec <= 32 ? 4 : 16);
    // eight groups: the row loop of many batches is built with the shallow batches only (deeper ones kept P x U mask vectors and subset sums live and
    // spilled).  A biallelic row with nothing missing that ONE batch of loads per lane covers takes any depth: that loop (tile_rows_packed_prefetch,
    // MREG = false) re-reads its masks from LDS per row, and a 2 500-haplotype row is then one batch of five loads per lane instead of five trips
    // of one with a single vector in flight (five groups, which run the eight-group kernel: DESIGN.md section 3).
    const int us4[4] = {1, 2, 3, 5}, us16[3] = {2, 3, 4};
    const int* us = lpr != 16 ? us4 : us16;
    auto pick = [&](bool shallow) {
      const int nus = shallow ? (lpr != 16 ? 2 : 1) : (lpr != 16 ? 4 : 3);
      int best_u = us[0];
      size_t best = SIZE_MAX;
      for (int k = 0; k < nus; ++k) {
        const size_t slots = round_up(m->pvec, (size_t)lpr * us[k]);
        if (slots <= best) { best = slots; best_u = us[k]; }
      }
      a.unroll = best_u;
      for (int k = 0; k < nus; ++k) if (env_punroll == us[k]) a.unroll = env_punroll;
      a.nvec_pad = (uint32_t)round_up(m->pvec, (size_t)lpr * a.unroll);
    };
    const bool no_prefetch = opt.packed_no_prefetch.load() != 0;
    pick(P >= 5 && (general || missing || no_prefetch));
    if (P >= 5 && a.nvec_pad != (uint32_t)(lpr * a.unroll)) pick(true);  // not one batch per row: the shallow set
    // The prefetching row loop (tile_rows_packed_prefetch) is taken when one batch of loads covers a row.  Same-process A/Bs (tools/ab_env.py
    // FMH_PACKED_NO_PREFETCH=1): on four-lane rows (1 000 and 2 500 haplotypes) it is level or 1-6 % ahead at every launch size; on sixteen-lane
    // rows with two groups (5 000 haplotypes) it was 2.4-2.9 % ahead at 625 k sites, level at 1 M and 0.6-2.4 % behind from 1.25 M to 10 M sites -
    // those kernels (one and two groups, sixteen lanes) have since dropped it altogether for the deferred-epilogue loop (sweep_kernel, defer_kernel()).
    a.single_trip = a.nvec_pad == (uint32_t)(lpr * a.unroll) && !no_prefetch;
    smem = (size_t)P * a.nvec_pad * 16;
    mask_mode = kMaskPacked;
    if (smem > lds_limit)
      return fail(FMH_ERR_UNSUPPORTED, "%d group masks of %u columns exceed the LDS budget: sweep fewer groups at a time on rows this wide", P, m->columns);
  } else if (smem > lds_limit) {  // byte masks do not fit LDS: bits in LDS if those fit, else bytes in global memory (L2)
    smem = (size_t)P * a.nvec_pad * 2;
    mask_mode = kMaskLdsBits;
    if (smem > lds_limit) { smem = 0; mask_mode = kMaskGlobalBytes; }
  }
  // BASELINE config C5: the counts as an int8 matrix-core contraction (sweep_mfma_kernels.hpp), for u8 rows that are biallelic with
  // nothing missing and at most four (padded) groups.  An alternative route: the contraction has <= 4 output rows and stays HBM-bound,
  // so it is measured beside the dot4 route, not chosen by default.  Read per call: tests flip it.
  const int env_mfma = (int)opt.counts_mfma.load();
  const int mfma_unroll = env_mfma == 2 ? 2 : 4;
  // (rows whose byte masks do not fit LDS stay on the dot4 routes)
  const bool fused_region = (mode & kModeDiversity) != 0 && P == 2;  // not built on the matrix-core route
  const bool mfma = !packed && !fused_region && env_mfma != 0 && !missing && !general && P <= 4 &&
                    (size_t)P * mfma_mask_stride(m->nvec, mfma_unroll) * 16 <= lds_limit;
  if (mfma) {
    a.unroll = mfma_unroll;
    a.nvec_pad = mfma_mask_stride(m->nvec, a.unroll);
    smem = (size_t)P * a.nvec_pad * 16;
  }
  const int want = packed || mfma ? -1 : (int)opt.mask_mode.load();
  if (want >= 0) {  // tests and measurements: take a slower mask route than needed
    const bool global_ok = P <= 2 && mode != kModeWc;
    if (want == kMaskLdsBits && mask_mode == kMaskLdsBytes) { smem = (size_t)P * a.nvec_pad * 2; mask_mode = kMaskLdsBits; }
    if (want == kMaskGlobalBytes && global_ok) { smem = 0; mask_mode = kMaskGlobalBytes; }
  }
  // Deferred epilogues (sweep_kernel, defer_kernel()): room behind the mask image for the counts a wave parks, as deep as leaves three workgroups
  // per CU their LDS (wide rows: the mask image takes it, and a tile of megabytes has nothing to gain from deferral anyway; measured: 200 000
  // columns fell from 3 to 2 workgroups per CU, 0.45 -> 0.69 ms, before the cap).  The deferring kernels have no other tile loop, so one tile's
  // room is always added.
  if (!mfma && defer_rule(P, mode, missing, general, lpr)) {  // the rule the kernel template is instantiated with
    const int e = (int)opt.defer_tiles.load();  // measurements (1 = the undeferred order); default -1 = by the launch size (launch_one)
    a.defer_tiles = e >= 1 && e <= kDeferTiles ? e : -1;
    smem = round_up(smem, 16);
    int depth = defer_depth_host(P, mode, missing);
    while (depth > 1 && smem + defer_lds_bytes(P, mode, missing, depth) > device_lds_per_cu(m->device) / 3 - 1024) depth /= 2;
    a.defer_cap = depth;
    a.defer_offset = (uint32_t)smem;
    smem += defer_lds_bytes(P, mode, missing, depth);
  }
  int grid = 0;
  // argument checks shared by every route, then the route's own launcher (sweep_*.hip)
  if (mode == (kModeSummary | kModeHudson) && P != 2) return fail(FMH_ERR_INVALID, "Hudson sweep needs exactly 2 groups");
  if (mode == (kModeSummary | kModeDiversity) && P > 2) return fail(FMH_ERR_INVALID, "diversity sweep needs 1 group (or the 2 of a fused region sweep)");
  if (mode == (kModeSummary | kModeHudson | kModeDiversity) && P != 2) return fail(FMH_ERR_INVALID, "the fused region sweep needs exactly 2 groups");
  if (mode == kModeWc && P == 1) return fail(FMH_ERR_INVALID, "W&C sweep needs at least 2 groups");
  if (mode != kModeSummary && mode != (kModeSummary | kModeHudson) && mode != (kModeSummary | kModeDiversity) && mode != (kModeSummary | kModeHudson | kModeDiversity) &&
      mode != kModeWc)
    return fail(FMH_ERR_UNSUPPORTED, "unsupported sweep mode %d", mode);
  if (mask_mode == kMaskGlobalBytes && (P > 2 || mode == kModeWc))
    return fail(FMH_ERR_UNSUPPORTED, "%d group masks of %u columns exceed the LDS budget: sweep at most two groups at a time on rows this wide", P, m->columns);
  // Short packed rows, biallelic, nothing missing: the LDS-staged flat-tile route (sweep_flat_kernels.hpp: one row per lane, scalar masks).
  // FMH_FLAT: 1 = wherever it is built, 0 = never, -1 = where it measured ahead of the four-lane route.
  const long long want_flat = opt.flat.load();
  const bool flat = mask_mode == kMaskPacked && !mfma && !missing && !general && m->pvec <= (uint32_t)kFlatMaskMaxVec && g->mask_flat &&
                    flat_route_builds(P, mode) && want_flat != 0 && (want_flat > 0 || flat_route_default(P, mode, m->pvec));
  int rc;
  if (flat) rc = launch_sweep_flat(P, mode, a, st, ctx, &grid);
  else if (mfma) rc = launch_sweep_mfma(P, mode, a, smem, st, ctx, &grid);
  else if (mask_mode == kMaskPacked && general && m->p2)  // alleles 4..7: three planes
    rc = lpr == 4 ? launch_sweep_packed4_3p(P, mode, missing, general, a, smem, st, ctx, &grid) : launch_sweep_packed16_3p(P, mode, missing, general, a, smem, st, ctx, &grid);
  else if (mask_mode == kMaskPacked) rc = lpr == 4 ? launch_sweep_packed4(P, mode, missing, general, a, smem, st, ctx, &grid) : launch_sweep_packed16(P, mode, missing, general, a, smem, st, ctx, &grid);
  else if (mask_mode == kMaskGlobalBytes) rc = launch_sweep_global(P, mode, missing, general, a, smem, st, ctx, &grid);
  else if (mask_mode == kMaskLdsBits) rc = launch_sweep_bits(P, mode, missing, general, a, smem, st, ctx, &grid);
  else rc = launch_sweep_bytes(P, mode, missing, general, a, smem, st, ctx, &grid);
  FMH_TRY(rc);
  hipStream_t fin = st;
  if (b.finalize_stream && b.swept) {
    HIP_TRY(hipEventRecord(b.swept, st));
    HIP_TRY(hipStreamWaitEvent(b.finalize_stream, b.swept, 0));
    fin = b.finalize_stream;
  }
  hipLaunchKernelGGL(finalize_kernel, dim3(kMaxF64 + kMaxU64), dim3(256), 0, fin, b.part_f64, b.part_u64, grid, b.out_f64, b.out_u64);
  HIP_TRY(hipGetLastError());
  *launched = true;
  return FMH_OK;
}

void fmhi::timing_add(double ms) {
  std::lock_guard<std::mutex> lock(g_ws_mutex);
  if (g_timing_launches == 0 || ms < g_timing_min) g_timing_min = ms;
  if (g_timing_launches == 0 || ms > g_timing_max) g_timing_max = ms;
  g_timing_ms += ms;
  g_timing_launches += 1;
}
// one decision per sweep: with fmh_timing_enable(n), n > 1, every n-th sweep is bracketed by events (the two event records cost a few
// microseconds of stream time each, which matters next to a 0.16 ms kernel)
bool fmhi::timing_enabled() {
  const int n = g_timing.load();
  if (n <= 0) return false;
  return n == 1 || g_timing_seq.fetch_add(1) % (uint64_t)n == 0;
}

static int run_sweep(const fmh_matrix* m, const fmh_groups* g, int mode, SweepArgs& a, void* stream, SweepResult* res) {
  if (!m || !g) return fail(FMH_ERR_INVALID, "matrix or groups is NULL");
  FMH_TRY(use_device(m->device));
  Workspace* w = nullptr;
  FMH_TRY(workspace(m->device, &w));
  LeaseHolder hold;
  hold.w = w;
  FMH_TRY(lease_acquire(w, &hold.l));
  SweepLease* l = hold.l;
  // The caller's stream, or - for the NULL stream - the lease's own (blocking: ordered against the legacy default stream like the NULL
  // stream itself, but not against the other leases), so that sweeps of different host threads (run_vcf's region workers) overlap on the device.
  hipStream_t st = stream ? (hipStream_t)stream : l->stream;
  const bool timing = timing_enabled();  // one snapshot per sweep: another thread may flip the switch while this one runs
  const double* harmonic = nullptr;
  if (mode & kModeDiversity) {
    std::lock_guard<std::mutex> grow(w->in_use);
    FMH_TRY(ensure_harmonic(w, m->columns + 1, st));
    harmonic = w->harmonic;
  }
  memset(res, 0, sizeof *res);
  const LaunchCtx ctx{w->cus, w->max_grid, l->ev0, l->ev1, timing};
  // finalize_kernel writes the 64 + 64 totals straight into the lease's PINNED host vectors (device-visible): a blocking sweep is then two
  // launches and one stream synchronisation.  (Two hipMemcpyAsync of 512 B behind the kernels cost more than the kernels on a small cohort:
  // a lone 512-byte copy_to_host is 22 us on the GPU box, tools/measure_call_overheads.py.)
  const SweepBuffers bufs{l->part_f64, l->part_u64, l->h_f64, l->h_u64};
  bool launched = false;
  FMH_TRY(enqueue_sweep(m, g, mode, a, st, ctx, bufs, harmonic, &launched));
  if (!launched) return FMH_OK;
  HIP_TRY(hipStreamSynchronize(st));
  memcpy(res->f64, l->h_f64, sizeof res->f64);
  memcpy(res->u64, l->h_u64, sizeof res->u64);
  if (timing) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, l->ev0, l->ev1) == hipSuccess) timing_add(ms);  // a timing failure never fails a sweep that has its results
    else (void)hipGetLastError();
  }
  return FMH_OK;
}

static void fill_pop_totals(const uint64_t* capacity, const SweepResult& r, int p, fmh_pop_totals* t) {
  t->haplotype_capacity = capacity[p];
  t->segregating_sites = r.u64[kOffPopSeg + p];
  t->uncallable_sites = r.u64[kOffPopUnc + p];
  t->pi_sum = r.f64[kOffPopF64 + p];
}

static void fill_hudson_totals(const uint64_t* capacity, const SweepResult& r, fmh_hudson_totals* t) {
  memset(t, 0, sizeof *t);
  t->numerator_sum = r.f64[kOffHudF64 + 0];
  t->denominator_sum = r.f64[kOffHudF64 + 1];
  t->pi1_sum = r.f64[kOffHudF64 + 2];
  t->pi2_sum = r.f64[kOffHudF64 + 3];
  t->dxy_sum_all = r.f64[kOffHudF64 + 4];
  t->site_num_sum = r.f64[kOffHudF64 + 5];
  t->site_den_sum = r.f64[kOffHudF64 + 6];
  t->site_dxy_sum = r.f64[kOffHudF64 + 7];
  t->dxy_uncallable_sites = r.u64[kOffHudU64 + 0];
  t->sites_with_components = r.u64[kOffHudU64 + 1];
  t->site_dxy_skipped = r.u64[kOffHudU64 + 2];
  fill_pop_totals(capacity, r, 0, &t->pop[0]);
  fill_pop_totals(capacity, r, 1, &t->pop[1]);
}

static int check_formula(int formula) {
  if (formula != FMH_FORMULA_SPARSE && formula != FMH_FORMULA_DENSE && formula != FMH_FORMULA_SUMMARY)
    return fail(FMH_ERR_INVALID, "unknown formula %d", formula);
  return FMH_OK;
}

extern "C" int fmh_population_summaries(const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count,
                                        int formula, uint32_t* d_alt, uint32_t* d_called, fmh_pop_totals* h_totals,
                                        void* stream) {
  FMH_TRY(check_formula(formula));
  SweepArgs a{};
  a.row_begin = row_begin;
  a.row_count = row_count;
  a.formula = formula;
  a.alt = d_alt;
  a.called = d_called;
  if (m && g && sweep_lds_bytes(g->padded, m->nvec) > sweep_lds_limit(m->device) && g->n_groups > 2) {
    // rows too wide for all masks at once: the populations are independent, sweep them in smaller batches
    int batch = g->n_groups;
    while (batch > 2 && sweep_lds_bytes(padded_groups(batch), m->nvec) > sweep_lds_limit(m->device)) batch = (batch + 1) / 2;
    for (int p0 = 0; p0 < g->n_groups; p0 += batch) {
      const int cnt = std::min(batch, g->n_groups - p0);
      fmh_groups* sub = nullptr;
      FMH_TRY(fmh_groups_create(m, g->host_mask.data() + (size_t)p0 * m->columns, cnt, &sub));
      const int rc = fmh_population_summaries(m, sub, row_begin, row_count, formula, d_alt ? d_alt + (size_t)p0 * row_count : nullptr,
                                              d_called ? d_called + (size_t)p0 * row_count : nullptr, h_totals ? h_totals + p0 : nullptr, stream);
      fmh_groups_destroy(sub);
      FMH_TRY(rc);
    }
    return FMH_OK;
  }
  SweepResult r;
  FMH_TRY(run_sweep(m, g, kModeSummary, a, stream, &r));
  if (h_totals) for (int p = 0; p < g->n_groups; ++p) fill_pop_totals(g->sizes, r, p, &h_totals[p]);
  return FMH_OK;
}

extern "C" int fmh_hudson_sweep(const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count,
                                int formula, const fmh_hudson_sites* sites, fmh_hudson_totals* h_totals, void* stream) {
  FMH_TRY(check_formula(formula));
  if (g && g->n_groups != 2) return fail(FMH_ERR_INVALID, "Hudson sweep needs exactly 2 groups, got %d", g->n_groups);
  SweepArgs a{};
  a.row_begin = row_begin;
  a.row_count = row_count;
  a.formula = formula;
  if (sites) {
    a.fst = sites->d_fst; a.dxy = sites->d_dxy; a.pi1 = sites->d_pi1; a.pi2 = sites->d_pi2;
    a.num = sites->d_num; a.den = sites->d_den; a.alt = sites->d_alt; a.called = sites->d_called;
  }
  SweepResult r;
  FMH_TRY(run_sweep(m, g, kModeSummary | kModeHudson, a, stream, &r));
  if (h_totals) fill_hudson_totals(g->sizes, r, h_totals);
  return FMH_OK;
}

// Hudson totals (and per-site records) of two populations from their count tables - populations of different matrices included
extern "C" int fmh_hudson_from_counts(int device, const uint32_t* d_called1, const uint32_t* d_alt1, uint64_t capacity1, const uint32_t* d_called2,
                                      const uint32_t* d_alt2, uint64_t capacity2, size_t row_count, int formula, int any_missing,
                                      const fmh_hudson_sites* sites, fmh_hudson_totals* h_totals, void* stream) {
  FMH_TRY(check_formula(formula));
  if (row_count && (!d_called1 || !d_alt1 || !d_called2 || !d_alt2)) return fail(FMH_ERR_INVALID, "NULL count table");
  if (h_totals) { memset(h_totals, 0, sizeof *h_totals); h_totals->pop[0].haplotype_capacity = capacity1; h_totals->pop[1].haplotype_capacity = capacity2; }
  if (row_count == 0) return FMH_OK;
  FMH_TRY(use_device(device));
  Workspace* w = nullptr;
  FMH_TRY(workspace(device, &w));
  LeaseHolder hold;
  hold.w = w;
  FMH_TRY(lease_acquire(w, &hold.l));
  SweepLease* l = hold.l;
  hipStream_t st = stream ? (hipStream_t)stream : l->stream;
  SweepArgs a{};
  a.row_begin = 0;
  a.row_count = row_count;
  a.formula = formula;
  a.n_groups = 2;
  if (sites) {
    a.fst = sites->d_fst; a.dxy = sites->d_dxy; a.pi1 = sites->d_pi1; a.pi2 = sites->d_pi2;
    a.num = sites->d_num; a.den = sites->d_den; a.alt = sites->d_alt; a.called = sites->d_called;
  }
  a.part_f64 = l->part_f64;
  a.part_u64 = l->part_u64;
  const int grid = (int)std::min<size_t>((size_t)w->max_grid, (row_count + kBlock - 1) / kBlock);
  if (any_missing) hipLaunchKernelGGL(hudson_from_counts_kernel<true>, dim3((unsigned)grid), dim3(kBlock), 0, st, a, d_called1, d_alt1, d_called2, d_alt2);
  else hipLaunchKernelGGL(hudson_from_counts_kernel<false>, dim3((unsigned)grid), dim3(kBlock), 0, st, a, d_called1, d_alt1, d_called2, d_alt2);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(finalize_kernel, dim3(kMaxF64 + kMaxU64), dim3(256), 0, st, l->part_f64, l->part_u64, grid, l->h_f64, l->h_u64);  // pinned, as run_sweep
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(st));
  if (h_totals) {
    SweepResult r;
    memcpy(r.f64, l->h_f64, sizeof r.f64);
    memcpy(r.u64, l->h_u64, sizeof r.u64);
    const uint64_t caps[2] = {capacity1, capacity2};
    fill_hudson_totals(caps, r, h_totals);
  }
  return FMH_OK;
}

extern "C" int fmh_diversity_sites(const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count,
                                   double* d_pi, double* d_theta, uint32_t* d_called, uint32_t* d_distinct,
                                   fmh_pop_totals* h_totals, void* stream) {
  if (g && g->n_groups != 1) return fail(FMH_ERR_INVALID, "diversity sweep needs exactly 1 group, got %d", g->n_groups);
  SweepArgs a{};
  a.row_begin = row_begin;
  a.row_count = row_count;
  a.formula = FMH_FORMULA_SPARSE;
  a.site_pi = d_pi;
  a.site_theta = d_theta;
  a.called = d_called;
  a.site_distinct = d_distinct;
  SweepResult r;
  FMH_TRY(run_sweep(m, g, kModeSummary | kModeDiversity, a, stream, &r));
  if (h_totals) fill_pop_totals(g->sizes, r, 0, h_totals);
  return FMH_OK;
}

// What run_vcf's region driver needs of groups 0 and 1 of a region matrix (process.rs:2468-3653: process_variants for each group, then
// the Hudson pair) in ONE read of the matrix: the population summaries of both groups by `summary_formula`, the per-site diversity of both
// groups, and - with hudson_formula >= 0 - the Hudson per-site records and totals by THAT formula set (the region driver takes
// calculate_pi_dense for the regional pi of a diploid dense matrix and the sparse per-site path for Hudson).  Round 2 issued
// fmh_population_summaries + 2 x fmh_diversity_sites + fmh_hudson_sweep: four reads.
int fmhi::pair_region_args(const fmh_groups* g, size_t row_begin, size_t row_count, int summary_formula, int hudson_formula, const fmh_pair_diversity_sites* div,
                           const fmh_hudson_sites* sites, SweepArgs& a, int* mode) {
  FMH_TRY(check_formula(summary_formula));
  if (hudson_formula >= 0) FMH_TRY(check_formula(hudson_formula));
  if (g && g->n_groups != 2) return fail(FMH_ERR_INVALID, "the fused region sweep needs exactly 2 groups, got %d", g->n_groups);
  a.row_begin = row_begin;
  a.row_count = row_count;
  a.formula = summary_formula;
  a.hudson_formula_p1 = hudson_formula >= 0 && hudson_formula != summary_formula ? hudson_formula + 1 : 0;
  if (div) { a.site_pi = div->d_pi; a.site_theta = div->d_theta; }
  if (sites && hudson_formula >= 0) {
    a.fst = sites->d_fst; a.dxy = sites->d_dxy; a.pi1 = sites->d_pi1; a.pi2 = sites->d_pi2;
    a.num = sites->d_num; a.den = sites->d_den;
  }
  if (sites) { a.alt = sites->d_alt; a.called = sites->d_called; }
  *mode = kModeSummary | kModeDiversity | (hudson_formula >= 0 ? kModeHudson : 0);
  return FMH_OK;
}

extern "C" int fmh_pair_region_sweep(const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, int summary_formula,
                                     int hudson_formula, const fmh_pair_diversity_sites* diversity_or_null, const fmh_hudson_sites* sites_or_null,
                                     fmh_hudson_totals* h_totals, void* stream) {
  SweepArgs a{};
  int mode = 0;
  FMH_TRY(pair_region_args(g, row_begin, row_count, summary_formula, hudson_formula, diversity_or_null, sites_or_null, a, &mode));
  SweepResult r;
  FMH_TRY(run_sweep(m, g, mode, a, stream, &r));
  if (h_totals) fill_hudson_totals(g->sizes, r, h_totals);
  return FMH_OK;
}

// the device's harmonic table, grown on demand (comm.hip's sharded fused sweep needs it outside run_sweep)
int fmhi::harmonic_table(int device, size_t max_k, hipStream_t st, const double** out) {
  Workspace* w = nullptr;
  FMH_TRY(workspace(device, &w));
  std::lock_guard<std::mutex> grow(w->in_use);
  FMH_TRY(ensure_harmonic(w, max_k, st));
  *out = w->harmonic;
  return FMH_OK;
}

// regional W&C sums per slot from the per-site tracks (two small launches, deterministic)
static int wc_slot_sums(DeviceScratch& scratch, hipStream_t st, size_t nslots, size_t rows, const double* d_a, const double* d_b,
                        const uint8_t* d_state, double** sa, double** sb, unsigned long long** si) {
  const size_t chunks = std::max<size_t>(1, std::min<size_t>(64, (rows + 32767) / 32768));
  double *pa = nullptr, *pb = nullptr;
  unsigned long long* pi = nullptr;
  FMH_TRY(scratch.get(&pa, nslots * chunks));
  FMH_TRY(scratch.get(&pb, nslots * chunks));
  FMH_TRY(scratch.get(&pi, nslots * chunks));
  FMH_TRY(scratch.get(sa, nslots));
  FMH_TRY(scratch.get(sb, nslots));
  FMH_TRY(scratch.get(si, nslots));
  hipLaunchKernelGGL(wc_slot_reduce_kernel, dim3((unsigned)nslots, (unsigned)chunks), dim3(256), 0, st, rows, d_a, d_b, d_state, pa, pb, pi);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(wc_slot_finalize_kernel, dim3((unsigned)((nslots + 63) / 64)), dim3(64), 0, st, nslots, chunks, (const double*)pa,
                     (const double*)pb, (const unsigned long long*)pi, *sa, *sb, *si);
  HIP_TRY(hipGetLastError());
  return FMH_OK;
}

// W&C kernel slots follow the padded-P pair order; maps them to the caller's G-group order (a.wc_slot for the kernel's stores,
// slot_of for the host's unpacking; -1 = a padded group takes part, never reported)
void fmhi::wc_slot_map(const fmh_matrix* m, const fmh_groups* g, SweepArgs& a, int (&slot_of)[32]) {
  for (int k = 0; k < 32; ++k) { a.wc_slot[k] = -1; slot_of[k] = -1; }
  if (!g) return;
  const int P = m ? wc_kernel_groups(m, g) : g->padded, G = g->n_groups;
  a.wc_slot[0] = 0;
  slot_of[0] = 0;
  int k = 1;
  for (int i = 0; i < P; ++i)
    for (int j = i + 1; j < P; ++j, ++k) {
      if (i < G && j < G) {
        int idx = 1;
        for (int x = 0; x < G; ++x)
          for (int y = x + 1; y < G; ++y, ++idx)
            if (x == i && y == j) slot_of[k] = idx;
        a.wc_slot[k] = (int8_t)slot_of[k];
      }
    }
}
// true when fmh_wc_sweep runs ONE fused kernel that keeps the regional sums itself (2..8 groups, masks in LDS; registers up to four groups,
// the per-wave LDS transposition beyond): the route the pipelined sharded sweep can finalise and reduce on the device; everything else
// (alleles beyond 3 with five to eight groups, rows too wide for all masks) goes through the counts route
bool fmhi::wc_fused_lane_totals(const fmh_matrix* m, const fmh_groups* g) {
  if (!m || !g || (g->padded == 8 && m->max_allele > 3)) return false;
  return sweep_lds_bytes(g->padded, m->nvec) <= wc_lds_limit(m->device, g->padded);
}
bool fmhi::summaries_single_sweep(const fmh_matrix* m, const fmh_groups* g) {
  return m && g && !(sweep_lds_bytes(g->padded, m->nvec) > sweep_lds_limit(m->device) && g->n_groups > 2);
}

extern "C" int fmh_wc_sweep(const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, double* d_a,
                            double* d_b, uint8_t* d_state, uint32_t* d_group_called, fmh_wc_totals* h_totals,
                            void* stream) {
  if (g && g->n_groups < 2) return fail(FMH_ERR_INVALID, "W&C sweep needs at least 2 groups, got %d", g->n_groups);
  SweepArgs a{};
  a.row_begin = row_begin;
  a.row_count = row_count;
  a.formula = FMH_FORMULA_SPARSE;
  a.wc_a = d_a;
  a.wc_b = d_b;
  a.wc_state = d_state;
  a.called = d_group_called;
  int slot_of[32];
  wc_slot_map(m, g, a, slot_of);
  // the fused kernel for 5..8 groups keeps the counts of alleles 0..3 per site; cohorts with alleles beyond 3 take the counts route
  const bool many_alleles8 = m && g && g->padded == 8 && m->max_allele > 3;
  if (m && g && (many_alleles8 || sweep_lds_bytes(g->padded, m->nvec) > wc_lds_limit(m->device, g->padded))) {
    // (or: rows too wide for all groups' masks to sit in LDS at once) count in smaller batches, components from the count tables
    const size_t nslots = 1 + (size_t)g->n_groups * (g->n_groups - 1) / 2;
    std::vector<double> sa(nslots), sb(nslots);
    std::vector<uint64_t> si(nslots);
    FMH_TRY(fmh_wc_sweep_many(m, g->host_mask.data(), g->n_groups, row_begin, row_count, d_a, d_b, d_state, d_group_called, sa.data(),
                              sb.data(), si.data(), stream));
    if (h_totals) {
      memset(h_totals, 0, sizeof *h_totals);
      h_totals->sites_attempted = row_count;
      for (size_t k = 0; k < nslots; ++k) { h_totals->sum_a[k] = sa[k]; h_totals->sum_b[k] = sb[k]; h_totals->informative_sites[k] = si[k]; }
    }
    return FMH_OK;
  }
  SweepResult r;
  FMH_TRY(run_sweep(m, g, kModeWc, a, stream, &r));
  if (h_totals) {
    memset(h_totals, 0, sizeof *h_totals);
    h_totals->sites_attempted = row_count;
    const int P = wc_kernel_groups(m, g);
    const int nw = 1 + P * (P - 1) / 2;
    for (int k = 0; k < nw; ++k) {
      if (slot_of[k] < 0) continue;
      h_totals->sum_a[slot_of[k]] = r.f64[kOffWcA + k];
      h_totals->sum_b[slot_of[k]] = r.f64[kOffWcB + k];
      h_totals->informative_sites[slot_of[k]] = r.u64[kOffWcInf + k];
    }
  }
  return FMH_OK;
}

extern "C" int fmh_wc_sweep_many(const fmh_matrix* m, const uint8_t* h_column_mask, int n_groups, size_t row_begin,
                                 size_t row_count, double* d_a, double* d_b, uint8_t* d_state, uint32_t* d_group_called,
                                 double* h_sum_a, double* h_sum_b, uint64_t* h_informative_sites, void* stream) {
  if (!m || !h_column_mask) return fail(FMH_ERR_INVALID, "NULL argument");
  if (n_groups < 2 || n_groups > FMH_MAX_GROUPS_MANY) return fail(FMH_ERR_INVALID, "n_groups %d out of range 2..%d", n_groups, FMH_MAX_GROUPS_MANY);
  if (row_begin > m->variants || row_count > m->variants - row_begin)
    return fail(FMH_ERR_INVALID, "row range [%zu, +%zu) exceeds %zu variants", row_begin, row_count, m->variants);
  FMH_TRY(use_device(m->device));
  const size_t G = (size_t)n_groups, nslots = 1 + G * (G - 1) / 2;
  if (h_sum_a) for (size_t k = 0; k < nslots; ++k) h_sum_a[k] = 0.0;
  if (h_sum_b) for (size_t k = 0; k < nslots; ++k) h_sum_b[k] = 0.0;
  if (h_informative_sites) for (size_t k = 0; k < nslots; ++k) h_informative_sites[k] = 0;
  if (row_count == 0) return FMH_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool general = m->max_allele > 1;
  const int n_alleles = (int)m->max_allele + 1;
  DeviceScratch scratch;
  scratch.device = m->device;
  scratch.stream = st;
  uint32_t *called = d_group_called, *alt = nullptr, *acounts = nullptr, *n_all = nullptr, *all_alt = nullptr;
  if (!called && m->has_missing) FMH_TRY(scratch.get(&called, G * row_count));  // (nothing missing: the called counts are the group sizes, no table)
  FMH_TRY(scratch.get(&n_all, row_count));
  if (general) {
    FMH_TRY(scratch.get(&acounts, (size_t)n_alleles * G * row_count));
    HIP_TRY(hipMemsetAsync(acounts, 0, (size_t)n_alleles * G * row_count * sizeof(uint32_t), st));
    HIP_TRY(hipStreamSynchronize(st));  // the counting sweeps may run on another stream (run_sweep's lease): the zeros must be in place first
  } else {
    FMH_TRY(scratch.get(&alt, G * row_count));
  }
  // no per-site track asked for: the regional sums straight from the count tables (wc_pair_totals_kernel), nothing per site and slot in memory
  const size_t tot_lds = ((size_t)(1 + (general ? n_alleles : 1)) * G * kWcTotTile + kWcTotTile) * sizeof(uint32_t);
  const size_t tot_cells = (size_t)(1 + (general ? n_alleles : 1)) * G * kWcTotTile;
  const bool totals_only = !d_a && !d_b && !d_state && tot_cells <= (size_t)kWcTotCellsMax * 256;
  if (!totals_only) {
    if (!d_a) FMH_TRY(scratch.get(&d_a, nslots * row_count));
    if (!d_b) FMH_TRY(scratch.get(&d_b, nslots * row_count));
    if (!d_state) FMH_TRY(scratch.get(&d_state, nslots * row_count));
  }
  // A matrix with nothing missing: every group's called count is its size at every site, so each slot's shape and denominators are launch
  // constants (wc_many_pre_kernel) and the all-columns sweep (1) is not needed
  WcSlotPre* pre = nullptr;
  double* grcp = nullptr;
  uint32_t* gsize = nullptr;
  if (!m->has_missing) {
    std::vector<uint32_t> sizes(G, 0);
    for (size_t gi = 0; gi < G; ++gi)
      for (size_t col = 0; col < m->columns; ++col) sizes[gi] += h_column_mask[gi * m->columns + col] != 0;
    FMH_TRY(scratch.get(&pre, nslots));
    FMH_TRY(scratch.get(&grcp, G));
    FMH_TRY(scratch.get(&gsize, G));
    HIP_TRY(hipMemcpyAsync(gsize, sizes.data(), G * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // `sizes` leaves scope; the table is tiny
    hipLaunchKernelGGL(wc_many_pre_kernel, dim3((unsigned)((nslots + 255) / 256)), dim3(256), 0, st, n_groups, (const uint32_t*)gsize, pre, grcp);
    HIP_TRY(hipGetLastError());
  }
  // (1) called entries over ALL columns: pop_sizes_populated (stats.rs:1987)
  if (!pre) {
    std::vector<uint8_t> ones(m->columns, 1);
    fmh_groups* g = nullptr;
    FMH_TRY(fmh_groups_create(m, ones.data(), 1, &g));
    SweepArgs a{};
    a.row_begin = row_begin; a.row_count = row_count; a.formula = FMH_FORMULA_SPARSE; a.called = n_all;
    SweepResult r;
    const int rc = run_sweep(m, g, kModeSummary, a, stream, &r);
    fmh_groups_destroy(g);
    FMH_TRY(rc);
  }
  (void)all_alt;
  // (2) counts of every group, eight groups per sweep
  size_t batch = FMH_MAX_GROUPS;  // as many groups per sweep as the LDS holds masks for
  while (batch > 2 && sweep_lds_bytes((int)batch, m->nvec) > sweep_lds_limit(m->device)) batch /= 2;  // two groups fit any width (global-mask route)
  for (size_t g0 = 0; g0 < G; g0 += batch) {
    const int cnt = (int)std::min<size_t>(batch, G - g0);
    fmh_groups* g = nullptr;
    FMH_TRY(fmh_groups_create(m, h_column_mask + g0 * m->columns, cnt, &g));
    SweepArgs a{};
    a.row_begin = row_begin; a.row_count = row_count; a.formula = FMH_FORMULA_SPARSE;
    // nothing missing: a group's called count is its size at every site - the kernels below take the sizes, the table is written only for a caller who asked for it
    if (!pre || d_group_called) a.called = called + g0 * row_count;
    if (alt) a.alt = alt + g0 * row_count;
    a.acounts = acounts; a.acounts_groups = (uint32_t)G; a.acounts_group0 = (uint32_t)g0;
    SweepResult r;
    const int rc = run_sweep(m, g, kModeSummary, a, stream, &r);
    fmh_groups_destroy(g);
    FMH_TRY(rc);
  }
  double *sa = nullptr, *sb = nullptr;
  unsigned long long* si = nullptr;
  if (totals_only) {
    // chunks of whole tiles of sixteen sites; enough of them to fill the chip with the pairs' workgroups, at most 1 024 (the partials are
    // nslots x chunks x 24 B), at least one tile each
    // biallelic and nothing missing, up to 64 groups: wc_pair_totals_biallelic_kernel (R replicas of every pair, blocks sized to the items)
    const bool bi = pre && !general && n_groups <= kWcBiGroupsMax && options().wc_bi_totals.load() != 0;
    const size_t npairs = nslots - 1;
    size_t R = 1, bi_blocks = 1, bi_threads = 256;
    if (bi) {
      // replicas: lanes launched per item (the last wave of every workgroup is partly idle) against what every further workgroup of a chunk
      // costs - each stages the whole tile and waits out its barriers with less arithmetic per tile.  5 % per workgroup orders the measured
      // replica counts of 12, 26 and 40 groups at 2 M sites correctly (profiles/r04/wc_many_groups_pair_kernel_trace.csv): 4, 4, 1.
      double best = 1e30;
      for (size_t r = 1; r <= 8; r *= 2) {
        const size_t blocks = (npairs * r + 511) / 512;
        const size_t threads = std::max<size_t>(std::max<size_t>(256, round_up(8 * G, (size_t)64)), round_up((npairs * r + blocks - 1) / blocks, (size_t)64));
        const double cost = (double)(blocks * threads) / (double)(npairs * r) * (1.0 + 0.05 * (double)blocks);
        if (cost < best) { best = cost; R = r; }
      }
      if (const long long v = options().wc_bi_replicas.load(); v == 1 || v == 2 || v == 4 || v == 8) R = (size_t)v;
      bi_blocks = (npairs * R + 511) / 512;
      // at least 8 threads per group: a thread stages at most four cells of a tile (G x 32 cells)
      bi_threads = std::max<size_t>(std::max<size_t>(256, round_up(8 * G, (size_t)64)), round_up((npairs * R + bi_blocks - 1) / bi_blocks, (size_t)64));
    }
    const size_t pair_blocks = bi ? bi_blocks : (nslots - 1 + 255) / 256;
    // (about 8 000 workgroups: a CU holds five or so at a time and a launch of only eight per CU ran in two uneven rounds; the partials stay below 4 M entries)
    size_t chunks = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(8192 / R, ((size_t)4 << 20) / (nslots * R)), (8192 + pair_blocks - 1) / pair_blocks));
    if (const long long v = options().wc_bi_chunks.load(); bi && v > 0) chunks = std::max<size_t>(1, std::min<size_t>((size_t)v, ((size_t)4 << 20) / (nslots * R)));
    const size_t tile_rows = bi ? (size_t)kWcBiTile : (size_t)kWcTotTile;
    size_t chunk_rows = round_up((row_count + chunks - 1) / chunks, tile_rows);
    chunks = (row_count + chunk_rows - 1) / chunk_rows;
    const size_t parts = chunks * R;  // partial sums per slot; the overall kernel is launched over as many (shorter) chunks
    // (the overall slot: one thread per site; at most 1 024 of the `parts` workgroups take rows - eight or so sites per thread amortise the block
    // reduction - the others write the zeros of their partials)
    const size_t overall_parts = std::min<size_t>(parts, 1024);
    const size_t overall_rows = (row_count + overall_parts - 1) / overall_parts;
    double *pa = nullptr, *pb = nullptr;
    unsigned long long* pi = nullptr;
    FMH_TRY(scratch.get(&pa, nslots * parts));
    FMH_TRY(scratch.get(&pb, nslots * parts));
    FMH_TRY(scratch.get(&pi, nslots * parts));
    FMH_TRY(scratch.get(&sa, nslots));
    FMH_TRY(scratch.get(&sb, nslots));
    FMH_TRY(scratch.get(&si, nslots));
    hipLaunchKernelGGL(wc_overall_totals_kernel, dim3((unsigned)parts), dim3(256), 0, st, n_groups, n_alleles, row_count, overall_rows, parts,
                       (const uint32_t*)called, (const uint32_t*)alt, (const uint32_t*)acounts, (const uint32_t*)n_all, pa, pb, pi,
                       (const WcSlotPre*)pre, (const double*)grcp, (const uint32_t*)gsize);
    HIP_TRY(hipGetLastError());
    if (bi) {
      const size_t bi_lds = (size_t)kWcBiTile * G * sizeof(WcBiCell);  // at most 64 KiB (64 groups)
      const size_t cells = (G * kWcBiTile + bi_threads - 1) / bi_threads;  // <= 4
      auto* bi_kernel = cells <= 2 ? wc_pair_totals_biallelic_kernel<2> : wc_pair_totals_biallelic_kernel<4>;
      if (bi_lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)bi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bi_lds));
      hipLaunchKernelGGL(bi_kernel, dim3((unsigned)(8 * ((chunks + 7) / 8) * bi_blocks)), dim3((unsigned)bi_threads), bi_lds, st, n_groups, (int)R,
                         (int)bi_blocks, row_count, chunk_rows, chunks, (const uint32_t*)alt, pa, pb, pi, (const WcSlotPre*)pre, (const double*)grcp, (const uint32_t*)gsize);
    } else {
      auto* pair_kernel = tot_cells <= 4 * 256 ? wc_pair_totals_kernel<4> : wc_pair_totals_kernel<kWcTotCellsMax>;
      if (tot_lds > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tot_lds));
      hipLaunchKernelGGL(pair_kernel, dim3((unsigned)pair_blocks, (unsigned)chunks), dim3(256), tot_lds, st, n_groups, n_alleles, row_count,
                         chunk_rows, chunks, (const uint32_t*)called, (const uint32_t*)alt, (const uint32_t*)acounts, (const uint32_t*)n_all, pa, pb, pi,
                         (const WcSlotPre*)pre, (const double*)grcp, (const uint32_t*)gsize);
    }
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(wc_slot_finalize_wave_kernel, dim3((unsigned)nslots), dim3(64), 0, st, parts, (const double*)pa, (const double*)pb,
                       (const unsigned long long*)pi, sa, sb, si);
    HIP_TRY(hipGetLastError());
  } else {
    // (3) per-site components from the count tables, (4) regional sums per slot
    hipLaunchKernelGGL(wc_from_counts_kernel, dim3((unsigned)((row_count + 255) / 256)), dim3(256), 0, st, n_groups, n_alleles, row_count,
                       (const uint32_t*)called, (const uint32_t*)alt, (const uint32_t*)acounts, (const uint32_t*)n_all, d_a, d_b, d_state,
                       (const WcSlotPre*)pre, (const double*)grcp, (const uint32_t*)gsize);
    HIP_TRY(hipGetLastError());
    FMH_TRY(wc_slot_sums(scratch, st, nslots, row_count, d_a, d_b, d_state, &sa, &sb, &si));
  }
  if (h_sum_a) HIP_TRY(hipMemcpyAsync(h_sum_a, sa, nslots * 8, hipMemcpyDeviceToHost, st));
  if (h_sum_b) HIP_TRY(hipMemcpyAsync(h_sum_b, sb, nslots * 8, hipMemcpyDeviceToHost, st));
  if (h_informative_sites) HIP_TRY(hipMemcpyAsync(h_informative_sites, si, nslots * 8, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  scratch.settled = true;
  return FMH_OK;
}

extern "C" int fmh_device_release_scratch(int device) {
  FMH_TRY(use_device(device));
  Workspace* w = nullptr;
  FMH_TRY(workspace(device, &w));
  std::lock_guard<std::mutex> busy(w->in_use);
  if (w->pd_planes) (void)hipFree(w->pd_planes);
  w->pd_planes = nullptr;
  w->pd_planes_bytes = 0;
  if (w->pd_slabs) (void)hipFree(w->pd_slabs);
  w->pd_slabs = nullptr;
  w->pd_slab_bytes = 0;
  pool_trim(device);
  upload_release(device);
  return FMH_OK;
}
