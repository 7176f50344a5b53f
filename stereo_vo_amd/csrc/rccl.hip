// RCCL binding of the sharded bundle adjustment (one all-reduce of the reduced camera system per LM
// iteration over xGMI).  librccl is bound at run time — the librccl already mapped into the process if there is one
// (a PyTorch process carries its own copy, and a communicator must be used with the library that created it), else
// ROCm's — so that libsvo_hip.so loads on machines without RCCL and single-GPU users never touch it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>

#include "common.h"

namespace {
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  const char* load_error = "librccl not found";
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* loaded[] = {"librccl.so.1", "librccl.so"};
    for (const char* nm : loaded)
      if (!r.lib) r.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);  // reuse the copy the process already holds
    const char* fresh[] = {"/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    for (const char* nm : fresh)
      if (!r.lib) r.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) return;
    r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(dlsym(r.lib, "ncclAllReduce"));
    r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.lib, "ncclGetErrorString"));
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.error_string) {
      r.load_error = "librccl lacks a required symbol";
      r.lib = nullptr;
    }
  });
  return r.lib ? &r : nullptr;
}
}  // namespace

// sum `count` doubles in place over the communicator, asynchronously on `stream`
int svo_rccl_allreduce_f64(void* buf, size_t count, void* comm, hipStream_t stream, const char** err) {
  Rccl* r = rccl();
  if (!r) { if (err) *err = "librccl not available"; return SVO_ERR_HIP; }
  const ncclResult_t rc = r->all_reduce(buf, buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(comm), stream);
  if (rc != ncclSuccess) { if (err) *err = r->error_string(rc); return SVO_ERR_HIP; }
  return SVO_OK;
}

extern "C" int svo_rccl_unique_id(void* id128) {
  static_assert(sizeof(ncclUniqueId) == 128, "svo.h promises a 128-byte id");
  Rccl* r = rccl();
  if (!r || !id128) return SVO_ERR_INVALID;
  return r->get_unique_id(static_cast<ncclUniqueId*>(id128)) == ncclSuccess ? SVO_OK : SVO_ERR_HIP;
}

extern "C" int svo_rccl_comm_create(void** nccl_comm, int n_ranks, int rank, const void* id128, int device) {
  Rccl* r = rccl();
  if (!r || !nccl_comm || !id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return SVO_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return SVO_ERR_NO_DEVICE;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t rc = r->comm_init_rank(&c, n_ranks, id, rank);
  if (rc != ncclSuccess) { fprintf(stderr, "svo_rccl_comm_create: %s\n", r->error_string(rc)); return SVO_ERR_HIP; }
  *nccl_comm = c;
  return SVO_OK;
}

extern "C" int svo_rccl_comm_destroy(void* nccl_comm) {
  Rccl* r = rccl();
  if (!r || !nccl_comm) return SVO_ERR_INVALID;
  return r->comm_destroy(static_cast<ncclComm_t>(nccl_comm)) == ncclSuccess ? SVO_OK : SVO_ERR_HIP;
}
