// Gradient exchange of the data-parallel step over RCCL (mi355_comm_init / mi355_allreduce_bucket).
//
// One process per GPU, ONE communicator per process.  RCCL is resolved at run time (dlopen): a process that has PyTorch loaded
// already holds a librccl.so (torch.distributed's "nccl" backend IS that library on ROCm) and gets the same copy; a plain C
// host gets /opt/rocm/lib/librccl.so.1.  Nothing here is on the single-GPU path, and libmi355conv.so carries no link-time
// dependency on RCCL.  The caller distributes the 128-byte id of rank 0 (any side channel: a file, MPI, a torch.distributed
// store) and passes the stream the bucket's last writer was issued on, or a stream that waits for it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/mi355conv.h"

void mi355_set_error(const char* fmt, ...);

namespace {

struct UniqueId { char internal[128]; };                 // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* comm_t;                                    // ncclComm_t
typedef int (*get_unique_id_t)(UniqueId*);
typedef int (*comm_init_rank_t)(comm_t*, int, UniqueId, int);
typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, comm_t, hipStream_t);
typedef int (*comm_destroy_t)(comm_t);
typedef const char* (*get_error_string_t)(int);

struct Rccl {
  void* handle = nullptr;
  get_unique_id_t get_unique_id = nullptr;
  comm_init_rank_t comm_init_rank = nullptr;
  all_reduce_t all_reduce = nullptr;
  comm_destroy_t comm_destroy = nullptr;
  get_error_string_t error_string = nullptr;
};

std::mutex g_mu;
Rccl g_rccl;
comm_t g_comm = nullptr;
int g_world = 0;

bool load_rccl() {
  if (g_rccl.handle) return true;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names)
    if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;          // the copy already in the process (PyTorch's), if any
  for (int i = 0; !h && i < 3; ++i) h = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    mi355_set_error("comm: cannot load librccl.so: %s", dlerror());
    return false;
  }
  Rccl r;
  r.handle = h;
  r.get_unique_id = (get_unique_id_t)dlsym(h, "ncclGetUniqueId");
  r.comm_init_rank = (comm_init_rank_t)dlsym(h, "ncclCommInitRank");
  r.all_reduce = (all_reduce_t)dlsym(h, "ncclAllReduce");
  r.comm_destroy = (comm_destroy_t)dlsym(h, "ncclCommDestroy");
  r.error_string = (get_error_string_t)dlsym(h, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) {
    mi355_set_error("comm: librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclAllReduce / ncclCommDestroy");
    return false;
  }
  g_rccl = r;
  return true;
}

int fail(const char* what, int rc) {
  mi355_set_error("comm: %s failed: %s (ncclResult %d)", what, g_rccl.error_string ? g_rccl.error_string(rc) : "?", rc);
  return MI355_ERR_RUNTIME;
}

}  // namespace

extern "C" int mi355_comm_unique_id(void* id128) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!id128) { mi355_set_error("comm_unique_id: null pointer"); return MI355_ERR_ARG; }
  if (!load_rccl()) return MI355_ERR_RUNTIME;
  const int rc = g_rccl.get_unique_id(static_cast<UniqueId*>(id128));
  return rc ? fail("ncclGetUniqueId", rc) : MI355_OK;
}

extern "C" int mi355_comm_init(int rank, int world, const void* id128) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!id128 || world < 1 || rank < 0 || rank >= world) { mi355_set_error("comm_init: bad rank %d / world %d / id", rank, world); return MI355_ERR_ARG; }
  if (g_comm) { mi355_set_error("comm_init: this process already holds a communicator (mi355_comm_destroy first)"); return MI355_ERR_ARG; }
  if (!load_rccl()) return MI355_ERR_RUNTIME;
  UniqueId id = *static_cast<const UniqueId*>(id128);
  comm_t c = nullptr;
  const int rc = g_rccl.comm_init_rank(&c, world, id, rank);      // collective: every rank calls it with the same id
  if (rc) return fail("ncclCommInitRank", rc);
  g_comm = c;
  g_world = world;
  return MI355_OK;
}

extern "C" int mi355_comm_world(void) { return g_world; }

extern "C" int mi355_allreduce_bucket(void* ptr, long long count, int dtype, mi355_stream_t s) {
  if (!g_comm) { mi355_set_error("allreduce_bucket: no communicator (mi355_comm_init)"); return MI355_ERR_ARG; }
  if (!ptr || count <= 0) { mi355_set_error("allreduce_bucket: null pointer or count %lld", count); return MI355_ERR_ARG; }
  int nccl_type;
  switch (dtype) {
    case MI355_F32: nccl_type = 7; break;                          // ncclFloat32
    case MI355_F16: nccl_type = 6; break;                          // ncclFloat16
    case MI355_BF16: nccl_type = 9; break;                         // ncclBfloat16
    default: mi355_set_error("allreduce_bucket: unknown dtype %d", dtype); return MI355_ERR_UNSUPPORTED;
  }
  const int rc = g_rccl.all_reduce(ptr, ptr, (size_t)count, nccl_type, /*ncclSum*/ 0, g_comm, (hipStream_t)s);   // in place, on `s`
  return rc ? fail("ncclAllReduce", rc) : MI355_OK;
}

extern "C" int mi355_comm_destroy(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_comm) return MI355_OK;
  const int rc = g_rccl.comm_destroy(g_comm);
  g_comm = nullptr;
  g_world = 0;
  return rc ? fail("ncclCommDestroy", rc) : MI355_OK;
}
