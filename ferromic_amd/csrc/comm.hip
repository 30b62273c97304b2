// comm.hip — fmh_comm: the sum of the regional accumulators over the ranks of a region-sharded sweep.
//
// Transport 0: RCCL (ncclAllReduce over xGMI), bound at run time with dlopen so that a single-GPU host never loads the
// library and a process that already maps a copy (PyTorch's wheel bundles one) shares it.  Transport 1: an in-process
// rendezvous that sums in rank order on the host, for one process whose "ranks" alias a device (RCCL refuses two ranks on
// one GPU) - the rehearsal path of run_vcf --devices 0,0 and of the tests on a one-GPU box.  Transport 2: a local one-rank
// communicator (fmh_comm_init_local): nothing to exchange, no RCCL - the pipelined begin / end calls for a single GPU that scans
// many windows, and what bench.py runs at N = 1 so that every N has the same step structure.
//
// The reference has no counterpart file: its reduce is rayon's fold/reduce inside one address space
// (stats.rs:1365-1461) and the serial sums of 1554-1623 / 2145-2374.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <condition_variable>
#include <cstdlib>
#include <memory>
#include <set>

#include "abi_internal.hpp"

using namespace fmh;
using namespace fmhi;

// ---- the slice of the RCCL API this file uses (types as in <rccl/rccl.h>, NCCL 2.x ABI) ----------------------------
namespace {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
static_assert(sizeof(ncclUniqueId) == FMH_COMM_ID_BYTES, "FMH_COMM_ID_BYTES is NCCL_UNIQUE_ID_BYTES");
typedef int ncclResult_t;  // ncclSuccess == 0
constexpr int kNcclSum = 0;
constexpr int kNcclUint64 = 5, kNcclFloat64 = 8;  // ncclDataType_t

struct RcclApi {
  void* handle = nullptr;
  std::string origin;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;     // optional
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

int rccl_api(RcclApi** out) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!g_rccl.handle) {
    void* h = nullptr;
    std::string origin;
    // (1) a copy this process already maps (torch.distributed's), (2) an explicit path, (3) the system's
    for (const char* name : {"librccl.so", "librccl.so.1"}) {
      if (!h && (h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) origin = std::string(name) + " (already loaded)";
    }
    if (!h) {
      if (const char* env = getenv("FMH_RCCL_LIBRARY")) { if ((h = dlopen(env, RTLD_NOW | RTLD_LOCAL))) origin = env; }
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      if (!h && (h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) origin = name;
    }
    if (!h) return fail(FMH_ERR_UNSUPPORTED, "RCCL is not available (dlopen librccl.so: %s); set FMH_RCCL_LIBRARY", dlerror());
    RcclApi api;
    api.handle = h;
    api.origin = origin;
#define BIND(field, sym)                                                     \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, sym));          \
  if (!api.field) return fail(FMH_ERR_UNSUPPORTED, "RCCL (%s) lacks %s", origin.c_str(), sym)
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommInitAll, "ncclCommInitAll");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(AllReduce, "ncclAllReduce");
    BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd");
    BIND(GetErrorString, "ncclGetErrorString");
#undef BIND
    api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(dlsym(h, "ncclGetVersion"));
    api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(h, "ncclCommAbort"));
    if (origin.find('/') == std::string::npos) {  // a bare soname: report the file the dynamic linker actually mapped
      Dl_info info;
      if (dladdr(reinterpret_cast<void*>(api.AllReduce), &info) && info.dli_fname) origin += std::string(" -> ") + info.dli_fname;
    }
    g_rccl = api;
  }
  *out = &g_rccl;
  return FMH_OK;
}

#define RCCL_TRY(api, expr)                                                                                           \
  do {                                                                                                                \
    ncclResult_t _r = (expr);                                                                                         \
    if (_r != 0) return fail(FMH_ERR_HIP, "%s: %s (%s:%d)", #expr, (api)->GetErrorString(_r), __FILE__, __LINE__);   \
  } while (0)

// in-process rendezvous shared by the handles of one fmh_comm_init_all call (transport 1)
struct HostGroup {
  std::mutex mu;
  std::condition_variable cv;
  int n = 0, arrived = 0;
  uint64_t generation = 0;
  bool mismatch = false;
  bool aborted = false;  // fmh_comm_abort: every waiter and every later caller returns an error instead of blocking
  int aborted_by = -1;
  std::vector<std::vector<double>> f;
  std::vector<std::vector<uint64_t>> u;
  std::vector<double> rf;
  std::vector<uint64_t> ru;

  // sums in rank order (deterministic), every rank leaves with the same vectors
  int allreduce(int rank, double* f64, size_t nf, uint64_t* u64, size_t nu) {
    std::unique_lock<std::mutex> lock(mu);
    if (aborted) return fail(FMH_ERR_INVALID, "the communicator was aborted by rank %d: a peer failed before its collective", aborted_by);
    f[rank].assign(f64, f64 + nf);
    u[rank].assign(u64, u64 + nu);
    const uint64_t gen = generation;
    if (++arrived == n) {
      mismatch = false;
      for (int r = 1; r < n; ++r) mismatch |= f[r].size() != f[0].size() || u[r].size() != u[0].size();
      if (!mismatch) {
        rf.assign(f[0].size(), 0.0);
        ru.assign(u[0].size(), 0);
        for (int r = 0; r < n; ++r) {
          for (size_t i = 0; i < rf.size(); ++i) rf[i] += f[r][i];
          for (size_t i = 0; i < ru.size(); ++i) ru[i] += u[r][i];
        }
      }
      arrived = 0;
      ++generation;
      cv.notify_all();
    } else {
      cv.wait(lock, [&] { return generation != gen || aborted; });
      if (aborted && generation == gen) return fail(FMH_ERR_INVALID, "the communicator was aborted by rank %d: a peer failed before its collective", aborted_by);
    }
    if (mismatch) return fail(FMH_ERR_INVALID, "fmh_allreduce_totals: the ranks passed vectors of different lengths");
    // still under the lock: no rank can complete the NEXT round (and overwrite rf / ru) before this one has re-entered
    for (size_t i = 0; i < nf; ++i) f64[i] = rf[i];
    for (size_t i = 0; i < nu; ++i) u64[i] = ru[i];
    return FMH_OK;
  }
  void abort(int rank) {
    std::lock_guard<std::mutex> lock(mu);
    if (!aborted) { aborted = true; aborted_by = rank; }
    cv.notify_all();
  }
};

// one pipelined sharded sweep in flight
struct ShardSlot {
  double* part_f64 = nullptr;
  unsigned long long* part_u64 = nullptr;
  double* out_f64 = nullptr;   // device: finalised local totals, reduced in place
  unsigned long long* out_u64 = nullptr;
  double* h_f64 = nullptr;     // pinned
  unsigned long long* h_u64 = nullptr;
  hipEvent_t swept = nullptr, reduced = nullptr, ev0 = nullptr, ev1 = nullptr;
  hipEvent_t red0 = nullptr, red1 = nullptr;  // around the grouped all-reduce on the communicator's stream (fmh_timing_read_reduce)
  bool busy = false, launched = false, timed = false;
  int kind = 0;          // ShardKind
  int n_groups = 0, padded = 0;
  int slot_of[32];       // W&C: kernel slot -> caller slot
  uint64_t sizes[FMH_MAX_GROUPS] = {0};
  size_t row_count = 0;
  unsigned long long* h_aux = nullptr;  // pinned: scalars added to the finalised vector before the reduce (W&C sites_attempted)
  // FMH_GRAPH=1, local communicator: the slot's whole step (sweep, finalize, D2H) captured once and replayed while the call repeats
  hipGraphExec_t graph = nullptr;
  std::string graph_key;
  hipEvent_t graph_done = nullptr;
  bool via_graph = false;
};
enum ShardKind : int { kShardHudson = 0, kShardWc = 1, kShardPops = 2 };
const char* kind_name(int k) { return k == kShardHudson ? "Hudson" : k == kShardWc ? "W&C" : "population-summaries"; }
constexpr int kOffWcAttempted = 62;  // u64 slot of the W&C vector that carries sites_attempted (rows swept: a plain sum over slabs)
}  // namespace

struct fmh_comm {
  int world = 1, rank = 0, device = 0;
  int transport = 0;
  ncclComm_t nccl = nullptr;  // read and cleared under nccl_mu only: fmh_comm_abort may come from any thread, while a peer thread is enqueueing on it
  std::mutex nccl_mu;
  std::shared_ptr<HostGroup> host;
  hipStream_t stream = nullptr;  // the reduce runs here, beside the sweeps on the caller's stream
  int cus = 0, max_grid = 0;
  // fmh_allreduce_totals staging
  double* d_f64 = nullptr;
  unsigned long long* d_u64 = nullptr;
  double* h_f64 = nullptr;  // pinned
  unsigned long long* h_u64 = nullptr;
  hipEvent_t done = nullptr;
  bool pending = false;
  std::atomic<bool> aborted{false};
  size_t pend_nf = 0, pend_nu = 0;
  // fmh_hudson_sweep_sharded_*: FIFO of slots
  ShardSlot slot[FMH_SHARDED_IN_FLIGHT];
  unsigned long long head = 0, tail = 0;
};

namespace {

int comm_alloc(fmh_comm* c) {
  FMH_TRY(use_device(c->device));
  Workspace* w = nullptr;
  FMH_TRY(workspace(c->device, &w));
  c->cus = w->cus;
  c->max_grid = w->max_grid;
  HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  HIP_TRY(hipMalloc((void**)&c->d_f64, FMH_COMM_MAX_VALUES * 8));
  HIP_TRY(hipMalloc((void**)&c->d_u64, FMH_COMM_MAX_VALUES * 8));
  HIP_TRY(hipHostMalloc((void**)&c->h_f64, FMH_COMM_MAX_VALUES * 8, hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void**)&c->h_u64, FMH_COMM_MAX_VALUES * 8, hipHostMallocDefault));
  HIP_TRY(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
  for (auto& s : c->slot) {
    HIP_TRY(hipMalloc((void**)&s.part_f64, (size_t)c->max_grid * kMaxF64 * 8));
    HIP_TRY(hipMalloc((void**)&s.part_u64, (size_t)c->max_grid * kMaxU64 * 8));
    HIP_TRY(hipMalloc((void**)&s.out_f64, kMaxF64 * 8));
    HIP_TRY(hipMalloc((void**)&s.out_u64, kMaxU64 * 8));
    HIP_TRY(hipHostMalloc((void**)&s.h_f64, kMaxF64 * 8, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void**)&s.h_u64, kMaxU64 * 8, hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&s.swept, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&s.reduced, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&s.ev0));
    HIP_TRY(hipEventCreate(&s.ev1));
    HIP_TRY(hipEventCreate(&s.red0));
    HIP_TRY(hipEventCreate(&s.red1));
    HIP_TRY(hipHostMalloc((void**)&s.h_aux, 8 * 8, hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&s.graph_done, hipEventDisableTiming));
  }
  return FMH_OK;
}

void comm_free(fmh_comm* c) {
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  ncclComm_t last = nullptr;
  { std::lock_guard<std::mutex> hold(c->nccl_mu); last = c->nccl; c->nccl = nullptr; }
  if (last && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(last);
  (void)hipFree(c->d_f64); (void)hipFree(c->d_u64);
  (void)hipHostFree(c->h_f64); (void)hipHostFree(c->h_u64);
  if (c->done) (void)hipEventDestroy(c->done);
  for (auto& s : c->slot) {
    (void)hipFree(s.part_f64); (void)hipFree(s.part_u64); (void)hipFree(s.out_f64); (void)hipFree(s.out_u64);
    (void)hipHostFree(s.h_f64); (void)hipHostFree(s.h_u64);
    (void)hipHostFree(s.h_aux);
    if (s.graph) (void)hipGraphExecDestroy(s.graph);
    if (s.graph_done) (void)hipEventDestroy(s.graph_done);
    for (hipEvent_t e : {s.swept, s.reduced, s.ev0, s.ev1, s.red0, s.red1}) if (e) (void)hipEventDestroy(e);
  }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  (void)hipGetLastError();
  delete c;
}

// the two vectors, resident on the device, summed over the ranks on the communicator's stream (RCCL transport)
int rccl_reduce_on_stream(fmh_comm* c, double* d_f64, size_t nf, unsigned long long* d_u64, size_t nu) {
  RcclApi* api = nullptr;
  FMH_TRY(rccl_api(&api));
  // the handle stays valid while this thread enqueues on it: fmh_comm_abort takes it out of the struct under the same mutex before it calls
  // ncclCommAbort, so an abort either comes first (error below) or waits until the enqueue has returned
  std::lock_guard<std::mutex> hold(c->nccl_mu);
  if (c->aborted || !c->nccl) return fail(FMH_ERR_INVALID, "the communicator was aborted");
  RCCL_TRY(api, api->GroupStart());
  if (nf) RCCL_TRY(api, api->AllReduce(d_f64, d_f64, nf, kNcclFloat64, kNcclSum, c->nccl, c->stream));
  if (nu) RCCL_TRY(api, api->AllReduce(d_u64, d_u64, nu, kNcclUint64, kNcclSum, c->nccl, c->stream));
  RCCL_TRY(api, api->GroupEnd());
  return FMH_OK;
}

}  // namespace

extern "C" int fmh_comm_get_unique_id(void* h_id) {
  if (!h_id) return fail(FMH_ERR_INVALID, "h_id is NULL");
  int n = 0;
  FMH_TRY(fmh_device_count(&n));
  RcclApi* api = nullptr;
  FMH_TRY(rccl_api(&api));
  ncclUniqueId id;
  RCCL_TRY(api, api->GetUniqueId(&id));
  memcpy(h_id, &id, sizeof id);
  return FMH_OK;
}

extern "C" int fmh_comm_init_rank(const void* h_id, int world, int rank, int device, fmh_comm** out) {
  if (!out) return fail(FMH_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (!h_id) return fail(FMH_ERR_INVALID, "h_id is NULL");
  if (world < 1 || rank < 0 || rank >= world) return fail(FMH_ERR_INVALID, "rank %d of %d out of range", rank, world);
  FMH_TRY(use_device(device));
  RcclApi* api = nullptr;
  FMH_TRY(rccl_api(&api));
  fmh_comm* c = new fmh_comm();
  c->world = world; c->rank = rank; c->device = device; c->transport = 0;
  int rc = comm_alloc(c);
  if (rc == FMH_OK) {
    ncclUniqueId id;
    memcpy(&id, h_id, sizeof id);
    ncclResult_t r = api->CommInitRank(&c->nccl, world, id, rank);
    if (r != 0) rc = fail(FMH_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, device, api->GetErrorString(r));
  }
  if (rc != FMH_OK) { comm_free(c); return rc; }
  *out = c;
  return FMH_OK;
}

extern "C" int fmh_comm_init_all(const int* h_devices, int n, fmh_comm** h_out) {
  if (!h_devices || !h_out) return fail(FMH_ERR_INVALID, "NULL argument");
  if (n < 1 || n > 64) return fail(FMH_ERR_INVALID, "communicator size %d out of range 1..64", n);
  for (int i = 0; i < n; ++i) h_out[i] = nullptr;
  std::set<int> distinct(h_devices, h_devices + n);
  const bool host = distinct.size() != (size_t)n || options().comm_host.load() != 0;
  std::vector<fmh_comm*> made;
  auto bail = [&](int code) { for (fmh_comm* c : made) comm_free(c); for (int i = 0; i < n; ++i) h_out[i] = nullptr; return code; };
  std::shared_ptr<HostGroup> group;
  if (host) {
    group = std::make_shared<HostGroup>();
    group->n = n;
    group->f.resize(n);
    group->u.resize(n);
  }
  for (int i = 0; i < n; ++i) {
    if (use_device(h_devices[i]) != FMH_OK) return bail(FMH_ERR_INVALID);
    fmh_comm* c = new fmh_comm();
    made.push_back(c);
    c->world = n; c->rank = i; c->device = h_devices[i]; c->transport = host ? 1 : 0;
    c->host = group;
    const int rc = comm_alloc(c);
    if (rc != FMH_OK) return bail(rc);
  }
  if (!host) {
    RcclApi* api = nullptr;
    const int rc = rccl_api(&api);
    if (rc != FMH_OK) return bail(rc);
    std::vector<ncclComm_t> comms(n, nullptr);
    ncclResult_t r = api->CommInitAll(comms.data(), n, h_devices);
    if (r != 0) return bail(fail(FMH_ERR_HIP, "ncclCommInitAll over %d devices: %s", n, api->GetErrorString(r)));
    for (int i = 0; i < n; ++i) made[i]->nccl = comms[i];
  }
  for (int i = 0; i < n; ++i) h_out[i] = made[i];
  return FMH_OK;
}

// One rank, no transport at all (the sum over one rank is the identity): the pipelined begin / end calls for a single GPU that scans
// many windows - the next window's sweep is enqueued while the previous one's totals travel to the host.
extern "C" int fmh_comm_init_local(int device, fmh_comm** out) {
  if (!out) return fail(FMH_ERR_INVALID, "out is NULL");
  *out = nullptr;
  FMH_TRY(use_device(device));
  fmh_comm* c = new fmh_comm();
  c->world = 1; c->rank = 0; c->device = device; c->transport = 2;
  const int rc = comm_alloc(c);
  if (rc != FMH_OK) { comm_free(c); return rc; }
  *out = c;
  return FMH_OK;
}

extern "C" int fmh_comm_destroy(fmh_comm* c) {
  if (c) comm_free(c);
  return FMH_OK;
}

extern "C" int fmh_comm_info(const fmh_comm* c, int* world, int* rank, int* device, int* transport) {
  if (!c) return fail(FMH_ERR_INVALID, "communicator is NULL");
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  if (device) *device = c->device;
  if (transport) *transport = c->transport;
  return FMH_OK;
}

// What the communicator is, for reports (bench.py prints it for N > 1): transport, world, rank, device and - for RCCL - which
// library file the process bound and its version.
extern "C" int fmh_comm_describe(const fmh_comm* c, char* h_text, size_t cap) {
  if (!c || !h_text || cap == 0) return fail(FMH_ERR_INVALID, "NULL argument");
  const char* names[3] = {"rccl", "host", "local"};
  std::string lib = "none", ver = "n/a";
  if (c->transport == 0) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    lib = g_rccl.origin;
    int v = 0;
    if (g_rccl.GetVersion && g_rccl.GetVersion(&v) == 0) ver = std::to_string(v);
  }
  snprintf(h_text, cap, "transport=%s world=%d rank=%d device=%d rccl_library=%s rccl_version=%s in_flight=%d", names[c->transport < 0 || c->transport > 2 ? 1 : c->transport],
           c->world, c->rank, c->device, lib.c_str(), ver.c_str(), FMH_SHARDED_IN_FLIGHT);
  return FMH_OK;
}

// A rank that failed before its collective: wakes the peers of an in-process group with an error (they would otherwise wait for ever)
// and aborts the RCCL communicator.  The communicator is unusable afterwards (destroy it); every later collective on it fails.
extern "C" int fmh_comm_abort(fmh_comm* c) {
  if (!c) return fail(FMH_ERR_INVALID, "communicator is NULL");
  c->aborted = true;
  if (c->host) c->host->abort(c->rank);
  // Exactly one caller gets the handle (several slab threads of run_vcf may abort every communicator of a failed group at once): it is taken
  // out of the struct under the mutex that rccl_reduce_on_stream holds while it enqueues, and only then aborted - never twice, never under a
  // peer's ncclAllReduce call.
  ncclComm_t mine = nullptr;
  if (c->transport == 0) {
    std::lock_guard<std::mutex> hold(c->nccl_mu);
    mine = c->nccl;
    c->nccl = nullptr;
  }
  if (mine && g_rccl.CommAbort) (void)g_rccl.CommAbort(mine);  // also releases a peer thread of this process that is blocked on this communicator's stream
  return FMH_OK;
}

extern "C" int fmh_allreduce_totals_begin(fmh_comm* c, const double* h_f64, size_t n_f64, const uint64_t* h_u64, size_t n_u64) {
  if (!c) return fail(FMH_ERR_INVALID, "communicator is NULL");
  if ((n_f64 && !h_f64) || (n_u64 && !h_u64)) return fail(FMH_ERR_INVALID, "NULL vector");
  if (n_f64 > FMH_COMM_MAX_VALUES || n_u64 > FMH_COMM_MAX_VALUES) return fail(FMH_ERR_INVALID, "at most %d values per vector", FMH_COMM_MAX_VALUES);
  if (c->pending) return fail(FMH_ERR_INVALID, "a reduce is already in flight on this communicator: call fmh_allreduce_totals_end first");
  if (c->aborted) return fail(FMH_ERR_INVALID, "the communicator was aborted");
  FMH_TRY(use_device(c->device));
  if (n_f64) memcpy(c->h_f64, h_f64, n_f64 * 8);
  if (n_u64) memcpy(c->h_u64, h_u64, n_u64 * 8);
  c->pend_nf = n_f64;
  c->pend_nu = n_u64;
  if (c->transport == 0) {
    if (n_f64) HIP_TRY(hipMemcpyAsync(c->d_f64, c->h_f64, n_f64 * 8, hipMemcpyHostToDevice, c->stream));
    if (n_u64) HIP_TRY(hipMemcpyAsync(c->d_u64, c->h_u64, n_u64 * 8, hipMemcpyHostToDevice, c->stream));
    FMH_TRY(rccl_reduce_on_stream(c, c->d_f64, n_f64, c->d_u64, n_u64));
    if (n_f64) HIP_TRY(hipMemcpyAsync(c->h_f64, c->d_f64, n_f64 * 8, hipMemcpyDeviceToHost, c->stream));
    if (n_u64) HIP_TRY(hipMemcpyAsync(c->h_u64, c->d_u64, n_u64 * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(c->done, c->stream));
  }
  c->pending = true;
  return FMH_OK;
}

extern "C" int fmh_allreduce_totals_end(fmh_comm* c, double* h_f64, uint64_t* h_u64) {
  if (!c) return fail(FMH_ERR_INVALID, "communicator is NULL");
  if (!c->pending) return fail(FMH_ERR_INVALID, "no reduce in flight on this communicator");
  if ((c->pend_nf && !h_f64) || (c->pend_nu && !h_u64)) return fail(FMH_ERR_INVALID, "NULL vector");
  c->pending = false;
  if (c->transport == 0) {
    FMH_TRY(use_device(c->device));
    HIP_TRY(hipEventSynchronize(c->done));
    // ncclCommAbort ends the reduce kernels in flight, so the event completes - over unreduced values
    if (c->aborted) return fail(FMH_ERR_INVALID, "the communicator was aborted while this reduce was in flight");
  } else if (c->transport == 1) {
    FMH_TRY(c->host->allreduce(c->rank, c->h_f64, c->pend_nf, reinterpret_cast<uint64_t*>(c->h_u64), c->pend_nu));
  }
  if (c->pend_nf) memcpy(h_f64, c->h_f64, c->pend_nf * 8);
  if (c->pend_nu) memcpy(h_u64, c->h_u64, c->pend_nu * 8);
  return FMH_OK;
}

extern "C" int fmh_allreduce_totals(fmh_comm* c, double* h_f64, size_t n_f64, uint64_t* h_u64, size_t n_u64) {
  FMH_TRY(fmh_allreduce_totals_begin(c, h_f64, n_f64, h_u64, n_u64));
  return fmh_allreduce_totals_end(c, h_f64, h_u64);
}

// ---- pipelined sharded sweeps: Hudson pair, Weir & Cockerham, population summaries ---------------------------------------------
// One code path for the three: sweep on the caller's stream -> finalize_kernel on the communicator's stream (behind the sweep's
// event) -> grouped ncclAllReduce of the 64 + 64 finalised accumulators in place -> one D2H into pinned memory.  _begin returns once
// the work is enqueued; _end waits for the OLDEST sweep in flight (whatever its kind) and unpacks it.
namespace {

double g_reduce_ms = 0.0;   // accumulated event time of the all-reduces of timed sharded sweeps (fmh_timing_read_reduce)
uint64_t g_reduce_n = 0;
std::mutex g_reduce_mu;

// Everything that can be refused is refused here, BEFORE anything is enqueued: a rank that returns an error from _begin has not
// entered the collective, and the caller (run_vcf's slab threads) can abort the group instead of leaving the peers inside RCCL.
int sharded_check(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count) {
  if (!c || !m || !g) return fail(FMH_ERR_INVALID, "NULL argument");
  if (c->aborted) return fail(FMH_ERR_INVALID, "the communicator was aborted");
  if (m->device != c->device) return fail(FMH_ERR_INVALID, "matrix lives on device %d, the communicator on device %d", m->device, c->device);
  if (g->device != m->device || g->pitch != m->pitch || g->columns != m->columns) return fail(FMH_ERR_INVALID, "groups were built for a different matrix geometry");
  if (row_begin > m->variants || row_count > m->variants - row_begin) return fail(FMH_ERR_INVALID, "row range [%zu, +%zu) exceeds %zu variants", row_begin, row_count, m->variants);
  if (c->head - c->tail >= FMH_SHARDED_IN_FLIGHT) return fail(FMH_ERR_INVALID, "%d sharded sweeps already in flight: call the matching _end first", FMH_SHARDED_IN_FLIGHT);
  return FMH_OK;
}

// After a failure inside _begin (a HIP error once work was enqueued): nothing of this slot may still be running when it is reused
void settle_after_failure(fmh_comm* c, hipStream_t st) {
  (void)hipStreamSynchronize(st);
  (void)hipStreamSynchronize(c->stream);
  (void)hipGetLastError();
}

// sweep (mode, args) + device-side reduce + D2H, enqueued into the next free slot.  `host_vec`: the slot's totals were computed by a
// blocking call and sit in s.h_f64 / s.h_u64 already (the routes that are not one fused sweep): they are copied up and reduced the same way.
int sharded_enqueue(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, int mode, SweepArgs& a, hipStream_t st, ShardSlot& s, bool host_vec,
                    const double* harmonic = nullptr) {
  s.timed = timing_enabled();
  s.launched = false;
  auto body = [&]() -> int {
    if (!host_vec) {
      const LaunchCtx ctx{c->cus, c->max_grid, s.ev0, s.ev1, s.timed};
      // the finalize kernel goes to the communicator's stream (behind the sweep's event): on the caller's stream the next window's sweep
      // follows this one directly
      const SweepBuffers bufs{s.part_f64, s.part_u64, s.out_f64, s.out_u64, c->stream, s.swept};
      FMH_TRY(enqueue_sweep(m, g, mode, a, st, ctx, bufs, harmonic, &s.launched));
      if (!s.launched) {  // an empty slab still takes part in the collective, with zeros
        HIP_TRY(hipMemsetAsync(s.out_f64, 0, kMaxF64 * 8, st));
        HIP_TRY(hipMemsetAsync(s.out_u64, 0, kMaxU64 * 8, st));
        HIP_TRY(hipEventRecord(s.swept, st));
        HIP_TRY(hipStreamWaitEvent(c->stream, s.swept, 0));
      }
      if (s.kind == kShardWc) {  // rows swept: not a kernel accumulator
        s.h_aux[0] = (unsigned long long)a.row_count;
        HIP_TRY(hipMemcpyAsync(s.out_u64 + kOffWcAttempted, s.h_aux, 8, hipMemcpyHostToDevice, c->stream));
      }
    } else {
      HIP_TRY(hipMemcpyAsync(s.out_f64, s.h_f64, kMaxF64 * 8, hipMemcpyHostToDevice, c->stream));
      HIP_TRY(hipMemcpyAsync(s.out_u64, s.h_u64, kMaxU64 * 8, hipMemcpyHostToDevice, c->stream));
    }
    if (c->transport == 0) {
      if (s.timed) HIP_TRY(hipEventRecord(s.red0, c->stream));
      FMH_TRY(rccl_reduce_on_stream(c, s.out_f64, kMaxF64, s.out_u64, kMaxU64));
      if (s.timed) HIP_TRY(hipEventRecord(s.red1, c->stream));
    }
    HIP_TRY(hipMemcpyAsync(s.h_f64, s.out_f64, kMaxF64 * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(s.h_u64, s.out_u64, kMaxU64 * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipEventRecord(s.reduced, c->stream));
    return FMH_OK;
  };
  // FMH_GRAPH=1 (opt-in, measurement of the fixed cost per step): on a local communicator with an explicit stream, a call that repeats the
  // previous one of this slot - same matrix and groups (pointers AND geometry), same rows, outputs, mode, stream and option generation - is
  // replayed from the hipGraph captured the first time: one hipGraphLaunch instead of sweep + event + finalize + two copies + event.
  s.via_graph = false;
  if (options().graph.load() != 0 && c->transport == 2 && st != nullptr && !host_vec && !s.timed) {
    std::string key;
    auto put = [&](const void* p, size_t n) { key.append(reinterpret_cast<const char*>(p), n); };
    const unsigned long long gen = options().generation.load();
    const void* ptrs[8] = {m, g, st, harmonic, m->p0, m->data, g->masks, g->mask_bits};
    put(ptrs, sizeof ptrs); put(&mode, sizeof mode); put(&gen, sizeof gen); put(&m->variants, sizeof m->variants); put(&m->columns, sizeof m->columns);
    put(&m->plane_pitch, sizeof m->plane_pitch); put(&m->pitch, sizeof m->pitch); put(g->sizes, sizeof g->sizes); put(&s.kind, sizeof s.kind);
    const void* outs[16] = {a.alt, a.called, a.fst, a.dxy, a.pi1, a.pi2, a.num, a.den, a.site_pi, a.site_theta, a.site_distinct, a.wc_a, a.wc_b, a.wc_state, a.acounts, nullptr};
    put(outs, sizeof outs); put(&a.row_begin, sizeof a.row_begin); put(&a.row_count, sizeof a.row_count); put(&a.formula, sizeof a.formula);
    put(&a.hudson_formula_p1, sizeof a.hudson_formula_p1);
    if (!s.graph || s.graph_key != key) {
      if (s.graph) { (void)hipGraphExecDestroy(s.graph); s.graph = nullptr; }
      hipGraph_t graph = nullptr;
      hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
      int rc = e == hipSuccess ? body() : fail(FMH_ERR_HIP, "hipStreamBeginCapture: %s", hipGetErrorString(e));
      if (e == hipSuccess) {
        if (rc == FMH_OK && hipStreamWaitEvent(st, s.reduced, 0) != hipSuccess) rc = fail(FMH_ERR_HIP, "joining the communicator's stream into the capture failed");
        const hipError_t e2 = hipStreamEndCapture(st, &graph);  // always ended, also after a failure inside
        if (rc == FMH_OK && e2 != hipSuccess) rc = fail(FMH_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e2));
      }
      if (rc == FMH_OK && hipGraphInstantiate(&s.graph, graph, nullptr, nullptr, 0) != hipSuccess) rc = fail(FMH_ERR_HIP, "hipGraphInstantiate failed");
      if (graph) (void)hipGraphDestroy(graph);
      if (rc != FMH_OK) { s.graph = nullptr; (void)hipGetLastError(); return rc; }  // nothing ran: a capture executes nothing
      s.graph_key = key;
    }
    HIP_TRY(hipGraphLaunch(s.graph, st));
    HIP_TRY(hipEventRecord(s.graph_done, st));
    s.launched = a.row_count != 0;
    s.via_graph = true;
    s.busy = true;
    ++c->head;
    return FMH_OK;
  }
  const int rc = body();
  if (rc != FMH_OK) { settle_after_failure(c, st); return rc; }  // the slot was never marked busy: nothing of it may still be in flight
  s.busy = true;
  ++c->head;
  return FMH_OK;
}

// waits for the oldest sweep in flight; the slot is released only once its work is known to have finished (or, after an error, has
// been waited for), so the next _begin never reuses buffers under running kernels
int sharded_collect(fmh_comm* c, int kind, ShardSlot** out) {
  if (!c) return fail(FMH_ERR_INVALID, "communicator is NULL");
  if (c->head == c->tail) return fail(FMH_ERR_INVALID, "no sharded sweep in flight on this communicator");
  FMH_TRY(use_device(c->device));
  ShardSlot& s = c->slot[c->tail % FMH_SHARDED_IN_FLIGHT];
  if (s.kind != kind) return fail(FMH_ERR_INVALID, "the oldest sharded sweep in flight is a %s sweep: collect it with its own _end", kind_name(s.kind));
  int rc = FMH_OK;
  hipError_t e = hipEventSynchronize(s.via_graph ? s.graph_done : s.reduced);
  if (e != hipSuccess) {
    rc = fail(FMH_ERR_HIP, "hipEventSynchronize(reduced): %s", hipGetErrorString(e));
    (void)hipStreamSynchronize(c->stream);
    (void)hipGetLastError();
  }
  if (rc == FMH_OK && c->transport == 1) rc = c->host->allreduce(c->rank, s.h_f64, kMaxF64, reinterpret_cast<uint64_t*>(s.h_u64), kMaxU64);
  // RCCL transport: ncclCommAbort ends the reduce kernels in flight and the event above completes - over unreduced totals.  The header promises
  // that every collective on an aborted communicator fails, so it does (the slot is still released below).
  if (rc == FMH_OK && c->transport == 0 && c->world > 1 && c->aborted) rc = fail(FMH_ERR_INVALID, "the communicator was aborted while this sweep's reduce was in flight");
  // (transport 2, a local one-rank communicator: nothing to add)
  ++c->tail;
  s.busy = false;
  if (rc != FMH_OK) return rc;
  if (s.timed) {
    float ms = 0.f;
    if (s.launched) { if (hipEventElapsedTime(&ms, s.ev0, s.ev1) == hipSuccess) timing_add(ms); else (void)hipGetLastError(); }
    if (c->transport == 0) {
      if (hipEventElapsedTime(&ms, s.red0, s.red1) == hipSuccess) { std::lock_guard<std::mutex> lock(g_reduce_mu); g_reduce_ms += ms; g_reduce_n += 1; }
      else (void)hipGetLastError();
    }
  }
  *out = &s;
  return FMH_OK;
}

}  // namespace

extern "C" int fmh_timing_read_reduce(double* h_total_ms, uint64_t* h_reduces) {
  std::lock_guard<std::mutex> lock(g_reduce_mu);
  if (h_total_ms) *h_total_ms = g_reduce_ms;
  if (h_reduces) *h_reduces = g_reduce_n;
  return FMH_OK;
}
extern "C" int fmh_timing_reset_reduce(void) {
  std::lock_guard<std::mutex> lock(g_reduce_mu);
  g_reduce_ms = 0.0;
  g_reduce_n = 0;
  return FMH_OK;
}

extern "C" int fmh_hudson_sweep_sharded_begin(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count,
                                              int formula, const fmh_hudson_sites* sites, void* stream) {
  FMH_TRY(sharded_check(c, m, g, row_begin, row_count));
  if (formula != FMH_FORMULA_SPARSE && formula != FMH_FORMULA_DENSE && formula != FMH_FORMULA_SUMMARY) return fail(FMH_ERR_INVALID, "unknown formula %d", formula);
  if (g->n_groups != 2) return fail(FMH_ERR_INVALID, "Hudson sweep needs exactly 2 groups, got %d", g->n_groups);
  FMH_TRY(use_device(c->device));
  ShardSlot& s = c->slot[c->head % FMH_SHARDED_IN_FLIGHT];
  SweepArgs a{};
  a.row_begin = row_begin;
  a.row_count = row_count;
  a.formula = formula;
  if (sites) {
    a.fst = sites->d_fst; a.dxy = sites->d_dxy; a.pi1 = sites->d_pi1; a.pi2 = sites->d_pi2;
    a.num = sites->d_num; a.den = sites->d_den; a.alt = sites->d_alt; a.called = sites->d_called;
  }
  s.kind = kShardHudson;
  s.n_groups = 2;
  s.sizes[0] = g->sizes[0];
  s.sizes[1] = g->sizes[1];
  s.row_count = row_count;
  return sharded_enqueue(c, m, g, kModeSummary | kModeHudson, a, (hipStream_t)stream, s, false);
}

extern "C" int fmh_hudson_sweep_sharded_end(fmh_comm* c, fmh_hudson_totals* t) {
  ShardSlot* sp = nullptr;
  FMH_TRY(sharded_collect(c, kShardHudson, &sp));
  const ShardSlot& s = *sp;
  if (t) {
    memset(t, 0, sizeof *t);
    const double* f = s.h_f64;
    const unsigned long long* u = s.h_u64;
    t->numerator_sum = f[kOffHudF64 + 0]; t->denominator_sum = f[kOffHudF64 + 1];
    t->pi1_sum = f[kOffHudF64 + 2]; t->pi2_sum = f[kOffHudF64 + 3]; t->dxy_sum_all = f[kOffHudF64 + 4];
    t->site_num_sum = f[kOffHudF64 + 5]; t->site_den_sum = f[kOffHudF64 + 6]; t->site_dxy_sum = f[kOffHudF64 + 7];
    t->dxy_uncallable_sites = u[kOffHudU64 + 0]; t->sites_with_components = u[kOffHudU64 + 1]; t->site_dxy_skipped = u[kOffHudU64 + 2];
    for (int p = 0; p < 2; ++p) {
      t->pop[p].haplotype_capacity = s.sizes[p];
      t->pop[p].segregating_sites = u[kOffPopSeg + p];
      t->pop[p].uncallable_sites = u[kOffPopUnc + p];
      t->pop[p].pi_sum = f[kOffPopF64 + p];
    }
  }
  return FMH_OK;
}

extern "C" int fmh_hudson_sweep_sharded(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, int formula,
                                        const fmh_hudson_sites* sites, fmh_hudson_totals* t, void* stream) {
  if (c && c->head != c->tail) return fail(FMH_ERR_INVALID, "pipelined sharded sweeps are in flight: collect them with their _end first");
  FMH_TRY(fmh_hudson_sweep_sharded_begin(c, m, g, row_begin, row_count, formula, sites, stream));
  return fmh_hudson_sweep_sharded_end(c, t);
}

// ---- the fused region sweep (fmh_pair_region_sweep) over a slab: population summaries + Hudson totals in one vector --------------------
extern "C" int fmh_pair_region_sweep_sharded_begin(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, int summary_formula,
                                                   int hudson_formula, const fmh_pair_diversity_sites* div, const fmh_hudson_sites* sites, void* stream) {
  FMH_TRY(sharded_check(c, m, g, row_begin, row_count));
  SweepArgs a{};
  int mode = 0;
  FMH_TRY(pair_region_args(g, row_begin, row_count, summary_formula, hudson_formula, div, sites, a, &mode));
  FMH_TRY(use_device(c->device));
  const double* harmonic = nullptr;
  FMH_TRY(harmonic_table(c->device, (size_t)m->columns + 1, (hipStream_t)stream, &harmonic));
  ShardSlot& s = c->slot[c->head % FMH_SHARDED_IN_FLIGHT];
  s.kind = kShardHudson;
  s.n_groups = 2;
  s.sizes[0] = g->sizes[0];
  s.sizes[1] = g->sizes[1];
  s.row_count = row_count;
  return sharded_enqueue(c, m, g, mode, a, (hipStream_t)stream, s, false, harmonic);
}

extern "C" int fmh_pair_region_sweep_sharded(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, int summary_formula,
                                             int hudson_formula, const fmh_pair_diversity_sites* div, const fmh_hudson_sites* sites, fmh_hudson_totals* t, void* stream) {
  if (c && c->head != c->tail) return fail(FMH_ERR_INVALID, "pipelined sharded sweeps are in flight: collect them with their _end first");
  FMH_TRY(fmh_pair_region_sweep_sharded_begin(c, m, g, row_begin, row_count, summary_formula, hudson_formula, div, sites, stream));
  return fmh_hudson_sweep_sharded_end(c, t);
}

// ---- W&C (calculate_overall_fst_wc's sums, stats.rs:2145-2374, over the ranks) -------------------------------------------------
extern "C" int fmh_wc_sweep_sharded_begin(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, double* d_a,
                                          double* d_b, uint8_t* d_state, uint32_t* d_group_called, void* stream) {
  FMH_TRY(sharded_check(c, m, g, row_begin, row_count));
  if (g->n_groups < 2) return fail(FMH_ERR_INVALID, "W&C sweep needs at least 2 groups, got %d", g->n_groups);
  FMH_TRY(use_device(c->device));
  ShardSlot& s = c->slot[c->head % FMH_SHARDED_IN_FLIGHT];
  s.kind = kShardWc;
  s.n_groups = g->n_groups;
  s.padded = g->padded;
  s.row_count = row_count;
  SweepArgs a{};
  a.row_begin = row_begin;
  a.row_count = row_count;
  a.formula = FMH_FORMULA_SPARSE;
  a.wc_a = d_a; a.wc_b = d_b; a.wc_state = d_state; a.called = d_group_called;
  wc_slot_map(m, g, a, s.slot_of);
  if (wc_fused_lane_totals(m, g)) return sharded_enqueue(c, m, g, kModeWc, a, (hipStream_t)stream, s, false);
  // alleles beyond 3 with five to eight groups, or rows too wide for all masks at once: the blocking call (the counts route), then the same
  // device-side reduce of its totals, laid out like the fused kernel's vector in CALLER slot order
  fmh_wc_totals local;
  FMH_TRY(fmh_wc_sweep(m, g, row_begin, row_count, d_a, d_b, d_state, d_group_called, &local, stream));
  memset(s.h_f64, 0, kMaxF64 * 8);
  memset(s.h_u64, 0, kMaxU64 * 8);
  const int slots = 1 + g->n_groups * (g->n_groups - 1) / 2;
  for (int k = 0; k < slots; ++k) { s.h_f64[kOffWcA + k] = local.sum_a[k]; s.h_f64[kOffWcB + k] = local.sum_b[k]; s.h_u64[kOffWcInf + k] = local.informative_sites[k]; }
  s.h_u64[kOffWcAttempted] = local.sites_attempted;
  for (int k = 0; k < 32; ++k) s.slot_of[k] = k < slots ? k : -1;
  return sharded_enqueue(c, m, g, kModeWc, a, (hipStream_t)stream, s, true);
}

extern "C" int fmh_wc_sweep_sharded_end(fmh_comm* c, fmh_wc_totals* t) {
  ShardSlot* sp = nullptr;
  FMH_TRY(sharded_collect(c, kShardWc, &sp));
  const ShardSlot& s = *sp;
  if (t) {
    memset(t, 0, sizeof *t);
    t->sites_attempted = s.h_u64[kOffWcAttempted];
    for (int k = 0; k < 29; ++k) {
      const int slot = s.slot_of[k];
      if (slot < 0) continue;
      t->sum_a[slot] = s.h_f64[kOffWcA + k];
      t->sum_b[slot] = s.h_f64[kOffWcB + k];
      t->informative_sites[slot] = s.h_u64[kOffWcInf + k];
    }
  }
  return FMH_OK;
}

extern "C" int fmh_wc_sweep_sharded(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, double* d_a, double* d_b,
                                    uint8_t* d_state, uint32_t* d_group_called, fmh_wc_totals* t, void* stream) {
  if (c && c->head != c->tail) return fail(FMH_ERR_INVALID, "pipelined sharded sweeps are in flight: collect them with their _end first");
  FMH_TRY(fmh_wc_sweep_sharded_begin(c, m, g, row_begin, row_count, d_a, d_b, d_state, d_group_called, stream));
  return fmh_wc_sweep_sharded_end(c, t);
}

// ---- population summaries (build_dense_population_summary's scalars, stats.rs:1367-1470, over the ranks) -----------------------
extern "C" int fmh_population_summaries_sharded_begin(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count,
                                                      int formula, uint32_t* d_alt, uint32_t* d_called, void* stream) {
  FMH_TRY(sharded_check(c, m, g, row_begin, row_count));
  if (formula != FMH_FORMULA_SPARSE && formula != FMH_FORMULA_DENSE && formula != FMH_FORMULA_SUMMARY) return fail(FMH_ERR_INVALID, "unknown formula %d", formula);
  FMH_TRY(use_device(c->device));
  ShardSlot& s = c->slot[c->head % FMH_SHARDED_IN_FLIGHT];
  s.kind = kShardPops;
  s.n_groups = g->n_groups;
  s.padded = g->padded;
  s.row_count = row_count;
  for (int p = 0; p < g->n_groups; ++p) s.sizes[p] = g->sizes[p];
  SweepArgs a{};
  a.row_begin = row_begin;
  a.row_count = row_count;
  a.formula = formula;
  a.alt = d_alt;
  a.called = d_called;
  if (summaries_single_sweep(m, g)) return sharded_enqueue(c, m, g, kModeSummary, a, (hipStream_t)stream, s, false);
  // rows too wide for all masks at once: the blocking call re-batches the groups; its totals take the same device-side reduce
  fmh_pop_totals local[FMH_MAX_GROUPS];
  FMH_TRY(fmh_population_summaries(m, g, row_begin, row_count, formula, d_alt, d_called, local, stream));
  memset(s.h_f64, 0, kMaxF64 * 8);
  memset(s.h_u64, 0, kMaxU64 * 8);
  for (int p = 0; p < g->n_groups; ++p) {
    s.h_f64[kOffPopF64 + p] = local[p].pi_sum;
    s.h_u64[kOffPopSeg + p] = local[p].segregating_sites;
    s.h_u64[kOffPopUnc + p] = local[p].uncallable_sites;
  }
  return sharded_enqueue(c, m, g, kModeSummary, a, (hipStream_t)stream, s, true);
}

extern "C" int fmh_population_summaries_sharded_end(fmh_comm* c, fmh_pop_totals* t) {
  ShardSlot* sp = nullptr;
  FMH_TRY(sharded_collect(c, kShardPops, &sp));
  const ShardSlot& s = *sp;
  if (t) {
    for (int p = 0; p < s.n_groups; ++p) {
      t[p].haplotype_capacity = s.sizes[p];
      t[p].segregating_sites = s.h_u64[kOffPopSeg + p];
      t[p].uncallable_sites = s.h_u64[kOffPopUnc + p];
      t[p].pi_sum = s.h_f64[kOffPopF64 + p];
    }
  }
  return FMH_OK;
}

extern "C" int fmh_population_summaries_sharded(fmh_comm* c, const fmh_matrix* m, const fmh_groups* g, size_t row_begin, size_t row_count, int formula,
                                                uint32_t* d_alt, uint32_t* d_called, fmh_pop_totals* t, void* stream) {
  if (c && c->head != c->tail) return fail(FMH_ERR_INVALID, "pipelined sharded sweeps are in flight: collect them with their _end first");
  FMH_TRY(fmh_population_summaries_sharded_begin(c, m, g, row_begin, row_count, formula, d_alt, d_called, stream));
  return fmh_population_summaries_sharded_end(c, t);
}

// ---- packing of the totals structs ----------------------------------------------------------------------------------------
extern "C" int fmh_hudson_totals_pack(const fmh_hudson_totals* t, double* f, uint64_t* u) {
  if (!t || !f || !u) return fail(FMH_ERR_INVALID, "NULL argument");
  f[0] = t->numerator_sum; f[1] = t->denominator_sum; f[2] = t->pi1_sum; f[3] = t->pi2_sum;
  f[4] = t->dxy_sum_all; f[5] = t->site_num_sum; f[6] = t->site_den_sum; f[7] = t->site_dxy_sum;
  f[8] = t->pop[0].pi_sum; f[9] = t->pop[1].pi_sum;
  u[0] = t->dxy_uncallable_sites; u[1] = t->sites_with_components; u[2] = t->site_dxy_skipped;
  u[3] = t->pop[0].segregating_sites; u[4] = t->pop[0].uncallable_sites;
  u[5] = t->pop[1].segregating_sites; u[6] = t->pop[1].uncallable_sites;
  u[7] = t->pop[0].haplotype_capacity; u[8] = t->pop[1].haplotype_capacity;  // identical on every rank; divide after a sum
  u[9] = 1;  // ranks summed
  return FMH_OK;
}

extern "C" int fmh_hudson_totals_unpack(fmh_hudson_totals* t, const double* f, const uint64_t* u) {
  if (!t || !f || !u) return fail(FMH_ERR_INVALID, "NULL argument");
  memset(t, 0, sizeof *t);
  t->numerator_sum = f[0]; t->denominator_sum = f[1]; t->pi1_sum = f[2]; t->pi2_sum = f[3];
  t->dxy_sum_all = f[4]; t->site_num_sum = f[5]; t->site_den_sum = f[6]; t->site_dxy_sum = f[7];
  t->pop[0].pi_sum = f[8]; t->pop[1].pi_sum = f[9];
  t->dxy_uncallable_sites = u[0]; t->sites_with_components = u[1]; t->site_dxy_skipped = u[2];
  t->pop[0].segregating_sites = u[3]; t->pop[0].uncallable_sites = u[4];
  t->pop[1].segregating_sites = u[5]; t->pop[1].uncallable_sites = u[6];
  const uint64_t ranks = u[9] ? u[9] : 1;
  t->pop[0].haplotype_capacity = u[7] / ranks;
  t->pop[1].haplotype_capacity = u[8] / ranks;
  return FMH_OK;
}

extern "C" int fmh_pop_totals_pack(const fmh_pop_totals* t, int n, double* f, uint64_t* u) {
  if (!t || !f || !u) return fail(FMH_ERR_INVALID, "NULL argument");
  if (n < 1 || n > FMH_MAX_GROUPS_MANY) return fail(FMH_ERR_INVALID, "population count %d out of range", n);
  for (int p = 0; p < n; ++p) {
    f[p] = t[p].pi_sum;
    u[3 * p + 0] = t[p].segregating_sites;
    u[3 * p + 1] = t[p].uncallable_sites;
    u[3 * p + 2] = t[p].haplotype_capacity;  // identical on every rank; divide after a sum
  }
  u[3 * n] = 1;  // ranks summed
  return FMH_OK;
}

extern "C" int fmh_pop_totals_unpack(fmh_pop_totals* t, int n, const double* f, const uint64_t* u) {
  if (!t || !f || !u) return fail(FMH_ERR_INVALID, "NULL argument");
  if (n < 1 || n > FMH_MAX_GROUPS_MANY) return fail(FMH_ERR_INVALID, "population count %d out of range", n);
  const uint64_t ranks = u[3 * n] ? u[3 * n] : 1;
  for (int p = 0; p < n; ++p) {
    t[p].pi_sum = f[p];
    t[p].segregating_sites = u[3 * p + 0];
    t[p].uncallable_sites = u[3 * p + 1];
    t[p].haplotype_capacity = u[3 * p + 2] / ranks;
  }
  return FMH_OK;
}

extern "C" int fmh_wc_totals_pack(const fmh_wc_totals* t, int n_groups, double* f, uint64_t* u) {
  if (!t || !f || !u) return fail(FMH_ERR_INVALID, "NULL argument");
  if (n_groups < 2 || n_groups > FMH_MAX_GROUPS) return fail(FMH_ERR_INVALID, "n_groups %d out of range 2..%d", n_groups, FMH_MAX_GROUPS);
  const int slots = 1 + n_groups * (n_groups - 1) / 2;
  for (int k = 0; k < slots; ++k) { f[k] = t->sum_a[k]; f[slots + k] = t->sum_b[k]; u[k] = t->informative_sites[k]; }
  u[slots] = t->sites_attempted;  // rows swept: a plain sum over slabs
  return FMH_OK;
}

extern "C" int fmh_wc_totals_unpack(fmh_wc_totals* t, int n_groups, const double* f, const uint64_t* u) {
  if (!t || !f || !u) return fail(FMH_ERR_INVALID, "NULL argument");
  if (n_groups < 2 || n_groups > FMH_MAX_GROUPS) return fail(FMH_ERR_INVALID, "n_groups %d out of range 2..%d", n_groups, FMH_MAX_GROUPS);
  const int slots = 1 + n_groups * (n_groups - 1) / 2;
  memset(t, 0, sizeof *t);
  for (int k = 0; k < slots; ++k) { t->sum_a[k] = f[k]; t->sum_b[k] = f[slots + k]; t->informative_sites[k] = u[k]; }
  t->sites_attempted = u[slots];
  return FMH_OK;
}
