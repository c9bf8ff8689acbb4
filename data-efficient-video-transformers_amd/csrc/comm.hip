// comm.hip -- the one collective of the path behind the C ABI: RCCL gradient all-reduce over xGMI
// (SURVEY section 8b "dvt_comm_{init,allreduce,destroy}", 8e: pure data parallelism, one SUM per step).
//
// The reference is single-GPU (pl.Trainer(gpus=1), src/main.py:87); this is the MI355X-side functionality north_star
// asks for.  One process per GPU; the communicator belongs to the host thread that created it.  Every call only
// ENQUEUES on the caller's stream (no host synchronisation), so the exchange can sit inside a captured hipGraph next
// to the backward kernels it overlaps with.
//
// RCCL is bound at run time (dlopen / dlsym), not at link time: a process that has already loaded an RCCL (PyTorch
// ships its own librccl.so and loads it with torch.distributed) must keep exactly one instance of the library -- two
// copies would each bring their own bootstrap state and kernels.  RTLD_NOLOAD finds the loaded one; otherwise the
// ROCm installation's librccl.so.1 is loaded.
#include "common.h"

#include <dlfcn.h>
#include <string.h>
#include <mutex>

namespace {

// the slice of rccl.h this file needs (ABI-stable since NCCL 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSum = 0 };
enum { ncclFloat16 = 6, ncclFloat32 = 7, ncclBfloat16 = 9 };

struct Rccl {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*);
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t);
  const char* (*GetErrorString)(ncclResult_t);
  bool ok = false;
  char why[256] = "";
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
  void* h = nullptr;
  const char* names[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);     // an instance this process already holds
  for (const char* n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    snprintf(g_rccl.why, sizeof(g_rccl.why), "librccl.so not found: %s", dlerror());
    return;
  }
  bool all = true;
  auto sym = [&](const char* n) {
    void* p = dlsym(h, n);
    if (!p) { all = false; snprintf(g_rccl.why, sizeof(g_rccl.why), "RCCL symbol %s missing", n); }
    return p;
  };
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
  g_rccl.Broadcast = (decltype(g_rccl.Broadcast))sym("ncclBroadcast");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
  g_rccl.ok = all;
}

int need_rccl() {
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.ok) return dvt_fail(DVT_ERR_UNSUPPORTED, "dvt_comm: %s", g_rccl.why);
  return DVT_OK;
}

int rccl_fail(ncclResult_t r, const char* where) {
  return dvt_fail(DVT_ERR_HIP, "%s: RCCL error %d (%s)", where, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
}

int nccl_dtype(int dtype) { return dtype == DVT_F32 ? ncclFloat32 : dtype == DVT_BF16 ? ncclBfloat16 : dtype == DVT_F16 ? ncclFloat16 : -1; }

}  // namespace

extern "C" {

int dvt_comm_unique_id(void* id_out) {
  DVT_REQUIRE(id_out, "dvt_comm_unique_id: null output");
  int rc = need_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r) return rccl_fail(r, "ncclGetUniqueId");
  __builtin_memcpy(id_out, &id, DVT_COMM_ID_BYTES);
  return DVT_OK;
}

int dvt_comm_init(dvt_comm_t* comm_out, const void* unique_id, int world, int rank) {
  DVT_REQUIRE(comm_out && unique_id && world >= 1 && rank >= 0 && rank < world, "dvt_comm_init: bad arguments");
  int rc = need_rccl();
  if (rc) return rc;
  ncclUniqueId id;
  __builtin_memcpy(&id, unique_id, DVT_COMM_ID_BYTES);
  ncclComm_t c = nullptr;
  const ncclResult_t r = g_rccl.CommInitRank(&c, world, id, rank);      // collective over all ranks; binds the current device
  if (r) return rccl_fail(r, "ncclCommInitRank");
  *comm_out = (dvt_comm_t)c;
  return DVT_OK;
}

int dvt_comm_allreduce(dvt_comm_t comm, void* buf, int64_t count, int dtype, dvt_stream_t stream) {
  DVT_REQUIRE(comm && (buf || count == 0) && count >= 0, "dvt_comm_allreduce: bad arguments");
  const int dt = nccl_dtype(dtype);
  DVT_REQUIRE(dt >= 0, "dvt_comm_allreduce: dtype %d unsupported", dtype);
  if (count == 0) return DVT_OK;
  int rc = need_rccl();
  if (rc) return rc;
  const ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)count, dt, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
  if (r) return rccl_fail(r, "ncclAllReduce");
  return DVT_OK;
}

int dvt_comm_broadcast(dvt_comm_t comm, void* buf, int64_t count, int dtype, int root, dvt_stream_t stream) {
  DVT_REQUIRE(comm && (buf || count == 0) && count >= 0 && root >= 0, "dvt_comm_broadcast: bad arguments");
  const int dt = nccl_dtype(dtype);
  DVT_REQUIRE(dt >= 0, "dvt_comm_broadcast: dtype %d unsupported", dtype);
  if (count == 0) return DVT_OK;
  int rc = need_rccl();
  if (rc) return rc;
  const ncclResult_t r = g_rccl.Broadcast(buf, buf, (size_t)count, dt, root, (ncclComm_t)comm, (hipStream_t)stream);
  if (r) return rccl_fail(r, "ncclBroadcast");
  return DVT_OK;
}

int dvt_comm_destroy(dvt_comm_t comm) {
  if (!comm) return DVT_OK;
  int rc = need_rccl();
  if (rc) return rc;
  const ncclResult_t r = g_rccl.CommDestroy((ncclComm_t)comm);
  if (r) return rccl_fail(r, "ncclCommDestroy");
  return DVT_OK;
}

}  // extern "C"
