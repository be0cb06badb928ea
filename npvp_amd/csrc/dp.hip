// The data-parallel exchange under the C ABI (npvp_dp_init / npvp_dp_allreduce_async / npvp_dp_wait).
//
// What the reference gets from Lightning's DDP strategy (ref/train_Predictor_lightning.py:40-42) is ONE exchange
// per step: the all-reduce(mean) of the parameter gradients.  Here it is RCCL over xGMI, one communicator per process (= per
// GPU), driven by a host that has NO torch.distributed: buckets of the flat gradient buffer are reduced IN PLACE on a side
// stream the caller owns, the compute stream is ordered after them by an event.  npvp_amd/dp.py takes this path with
// NPVP_DP_COMM=c (default: torch.distributed's ProcessGroupNCCL, which is the same RCCL).
//
// RCCL is NOT a link-time dependency of libnpvp_hip.so: it is looked up when npvp_dp_unique_id / npvp_dp_init is first called -
// the copy already in the process if there is one (a PyTorch process has its own librccl.so loaded; two copies of a collective
// library in one process must not be mixed on one communicator, and nothing here shares one), else $NPVP_RCCL_LIB, else
// librccl.so.1 from the loader's path.  A process that never calls these entry points never loads it.
//
// State: unlike the kernels' entry points this layer HAS process state - the communicator, its rank / world and the event of the
// last reduction - guarded by a mutex; one communicator per process, npvp_dp_finalize releases it.
#include "common.h"
#include <rccl/rccl.h>      // types and enumerators only: every function is reached through dlsym
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {

struct Rccl {
  void* so = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};

struct DpState {
  Rccl api;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 0, device = -1;
  hipEvent_t done = nullptr;         // recorded on the side stream after the latest all-reduce
  bool pending = false;
};

DpState g_dp;
std::mutex g_mu;

int fail(const char* what, const char* detail) {
  char msg[512];
  snprintf(msg, sizeof(msg), "%s: %s", what, detail ? detail : "?");
  npvp_set_error(msg);
  return NPVP_ERR_LAUNCH;
}

// the library, once: already-loaded copy first (RTLD_NOLOAD), then $NPVP_RCCL_LIB, then the loader's path
int load_rccl(Rccl& a) {
  if (a.so) return NPVP_OK;
  // order (the header and INTEGRATION.md say the same): (1) the copy ALREADY in the process - a PyTorch host has loaded its own
  // RCCL, and a second copy beside it would be a second set of communicator state and IPC handles; (2) $NPVP_RCCL_LIB; (3) the
  // loader path
  const char* env = getenv("NPVP_RCCL_LIB");
  void* h = nullptr;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (int i = 0; i < 2 && !h; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
  if (!h && env && *env) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
  for (int i = 1; i >= 0 && !h; --i) h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!h) return fail("npvp_dp: RCCL not found (set NPVP_RCCL_LIB to librccl.so)", dlerror());
  a.get_unique_id = (decltype(a.get_unique_id))dlsym(h, "ncclGetUniqueId");
  a.comm_init_rank = (decltype(a.comm_init_rank))dlsym(h, "ncclCommInitRank");
  a.comm_destroy = (decltype(a.comm_destroy))dlsym(h, "ncclCommDestroy");
  a.all_reduce = (decltype(a.all_reduce))dlsym(h, "ncclAllReduce");
  a.error_string = (decltype(a.error_string))dlsym(h, "ncclGetErrorString");
  if (!a.get_unique_id || !a.comm_init_rank || !a.comm_destroy || !a.all_reduce || !a.error_string) {
    dlclose(h);
    return fail("npvp_dp: the RCCL library lacks an entry point", "ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce / ncclGetErrorString");
  }
  a.so = h;
  return NPVP_OK;
}

}  // namespace

static_assert(NCCL_UNIQUE_ID_BYTES == 128, "include/npvp_hip.h documents a 128-byte id");

// rank 0: a fresh communicator id (128 bytes at `id_out`); the host carries it to the other ranks by whatever it has (a file, a
// socket, MPI, torch.distributed's store) before every rank calls npvp_dp_init with the same bytes
extern "C" int npvp_dp_unique_id(void* id_out) {
  NPVP_CHECK_ARG(id_out, "dp_unique_id: null buffer");
  std::lock_guard<std::mutex> lock(g_mu);
  if (int rc = load_rccl(g_dp.api)) return rc;
  ncclUniqueId id;
  const ncclResult_t r = g_dp.api.get_unique_id(&id);
  if (r != ncclSuccess) return fail("dp_unique_id: ncclGetUniqueId", g_dp.api.error_string(r));
  memcpy(id_out, &id, sizeof(id));
  return NPVP_OK;
}

// every rank, on the device it trains on (hipSetDevice before the call); collective: returns when all `world` ranks have joined
extern "C" int npvp_dp_init(int rank, int world, const void* unique_id) {
  NPVP_CHECK_ARG(unique_id && world >= 1 && rank >= 0 && rank < world, "dp_init: needs 0 <= rank < world and the 128-byte id of npvp_dp_unique_id");
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_dp.comm) { npvp_set_error("dp_init: already initialised (one communicator per process; npvp_dp_finalize first)"); return NPVP_ERR_ARG; }
  if (int rc = load_rccl(g_dp.api)) return rc;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return fail("dp_init", "no HIP device");
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = g_dp.api.comm_init_rank(&comm, world, id, rank);
  if (r != ncclSuccess) return fail("dp_init: ncclCommInitRank", g_dp.api.error_string(r));
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
    g_dp.api.comm_destroy(comm);
    return fail("dp_init", "hipEventCreate failed");
  }
  g_dp.comm = comm; g_dp.rank = rank; g_dp.world = world; g_dp.device = dev; g_dp.done = ev; g_dp.pending = false;
  return NPVP_OK;
}

extern "C" int npvp_dp_world(void) { std::lock_guard<std::mutex> lock(g_mu); return g_dp.comm ? g_dp.world : 0; }
extern "C" int npvp_dp_rank(void) { std::lock_guard<std::mutex> lock(g_mu); return g_dp.comm ? g_dp.rank : -1; }

// bucket[0..n) <- mean over the ranks of bucket[0..n), in place, enqueued on `side` (the caller has ordered `side` after the
// producers of the bucket).  Every rank must call it for the same buckets in the same order.  Nothing blocks the host.
extern "C" int npvp_dp_allreduce_async(float* bucket, size_t n, hipStream_t side) {
  NPVP_CHECK_ARG(bucket && n > 0, "dp_allreduce_async: empty bucket");
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_dp.comm) { npvp_set_error("dp_allreduce_async: npvp_dp_init has not run"); return NPVP_ERR_ARG; }
  const ncclResult_t r = g_dp.api.all_reduce(bucket, bucket, n, ncclFloat32, ncclAvg, g_dp.comm, side);
  if (r != ncclSuccess) return fail("dp_allreduce_async: ncclAllReduce", g_dp.api.error_string(r));
  if (hipEventRecord(g_dp.done, side) != hipSuccess) return fail("dp_allreduce_async", "hipEventRecord failed");
  g_dp.pending = true;
  return NPVP_OK;
}

// `compute` waits (on the device, not the host) for every all-reduce enqueued so far
extern "C" int npvp_dp_wait(hipStream_t compute) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_dp.comm) { npvp_set_error("dp_wait: npvp_dp_init has not run"); return NPVP_ERR_ARG; }
  if (!g_dp.pending) return NPVP_OK;
  if (hipStreamWaitEvent(compute, g_dp.done, 0) != hipSuccess) return fail("dp_wait", "hipStreamWaitEvent failed");
  return NPVP_OK;
}

// the host waits for the reductions in flight, then the communicator goes (call on every rank)
extern "C" int npvp_dp_finalize(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (!g_dp.comm) return NPVP_OK;
  if (g_dp.pending) hipEventSynchronize(g_dp.done);
  g_dp.api.comm_destroy(g_dp.comm);
  hipEventDestroy(g_dp.done);
  g_dp.comm = nullptr; g_dp.done = nullptr; g_dp.pending = false; g_dp.world = 0; g_dp.rank = 0; g_dp.device = -1;
  return NPVP_OK;
}
