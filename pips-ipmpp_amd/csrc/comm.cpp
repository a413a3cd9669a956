// RCCL all-reduce of the Schur complement / b0 across the GPUs of a node.
// Replaces MPI_Allreduce in DistributedRootLinearSystem::reduceKKTdense (:860-881) and sLinsysRootAug::Lsolve (:340-341).
// librccl is resolved at run time (dlopen) so that the library loads on hosts without it; in a PyTorch process this
// binds to the librccl.so.1 torch already loaded.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>

#include "common.h"
#include "pips_hip.h"

namespace {
typedef struct { char internal[128]; } nccl_id_t;
typedef void* nccl_comm_t;
typedef int (*fn_get_id)(nccl_id_t*);
typedef int (*fn_init_rank)(nccl_comm_t*, int, nccl_id_t, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_destroy)(nccl_comm_t);
typedef const char* (*fn_errstr)(int);

struct Rccl {
   void* lib = nullptr;
   fn_get_id get_id = nullptr;
   fn_init_rank init_rank = nullptr;
   fn_allreduce allreduce = nullptr;
   fn_destroy destroy = nullptr;
   fn_errstr errstr = nullptr;
   bool load() {
      if (lib) return true;
      const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (const char* n : names) {
         lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
         if (lib) break;
      }
      if (!lib) return false;
      get_id = (fn_get_id)dlsym(lib, "ncclGetUniqueId");
      init_rank = (fn_init_rank)dlsym(lib, "ncclCommInitRank");
      allreduce = (fn_allreduce)dlsym(lib, "ncclAllReduce");
      destroy = (fn_destroy)dlsym(lib, "ncclCommDestroy");
      errstr = (fn_errstr)dlsym(lib, "ncclGetErrorString");
      return get_id && init_rank && allreduce && destroy;
   }
};
Rccl g_rccl;
struct Comm {
   nccl_comm_t comm;
   int device;
   pips_hip_allreduce_cb external = nullptr;   // host-supplied reduction (GPU-aware MPI, torch.distributed, ...)
   void* user = nullptr;
};
constexpr int kNcclDouble = 8;  // ncclFloat64
constexpr int kNcclSum = 0;
}  // namespace

extern "C" {

int pips_hip_comm_unique_id(void* id128) {
   if (!id128) PIPS_FAIL(pips::PIPS_ERR_ARG, "null id buffer");
   if (!g_rccl.load()) PIPS_FAIL(pips::PIPS_ERR_RCCL, "librccl not found");
   nccl_id_t id;
   const int rc = g_rccl.get_id(&id);
   if (rc) PIPS_FAIL(pips::PIPS_ERR_RCCL, "ncclGetUniqueId failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
   std::memcpy(id128, &id, 128);
   return 0;
}

int pips_hip_comm_create(void** comm, const void* id128, int n_ranks, int rank, int device) {
   if (!comm || !id128 || n_ranks <= 0 || rank < 0 || rank >= n_ranks) PIPS_FAIL(pips::PIPS_ERR_ARG, "pips_hip_comm_create: bad arguments");
   if (!g_rccl.load()) PIPS_FAIL(pips::PIPS_ERR_RCCL, "librccl not found");
   if (hipSetDevice(device) != hipSuccess) PIPS_FAIL(pips::PIPS_ERR_HIP, "hipSetDevice(%d) failed", device);
   nccl_id_t id;
   std::memcpy(&id, id128, 128);
   Comm* c = new Comm{nullptr, device};
   const int rc = g_rccl.init_rank(&c->comm, n_ranks, id, rank);
   if (rc) {
      delete c;
      PIPS_FAIL(pips::PIPS_ERR_RCCL, "ncclCommInitRank failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
   }
   *comm = c;
   return 0;
}

int pips_hip_comm_create_external(void** comm, pips_hip_allreduce_cb allreduce, void* user) {
   if (!comm || !allreduce) PIPS_FAIL(pips::PIPS_ERR_ARG, "pips_hip_comm_create_external: bad arguments");
   Comm* c = new Comm{nullptr, -1};
   c->external = allreduce;
   c->user = user;
   *comm = c;
   return 0;
}

int pips_hip_allreduce_sum(void* comm, double* buf_dev, size_t n, void* stream) {
   Comm* c = (Comm*)comm;
   if (!c || !buf_dev) PIPS_FAIL(pips::PIPS_ERR_ARG, "pips_hip_allreduce_sum: bad arguments");
   if (c->external) {
      // the callback works on its own stream / on the host: hand over a quiescent buffer, take back a finished one
      if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) PIPS_FAIL(pips::PIPS_ERR_HIP, "stream sync before the external all-reduce failed");
      const int rc = c->external(c->user, buf_dev, n);
      if (rc) PIPS_FAIL(pips::PIPS_ERR_RCCL, "external all-reduce callback returned %d", rc);
      return 0;
   }
   const int rc = g_rccl.allreduce(buf_dev, buf_dev, n, kNcclDouble, kNcclSum, c->comm, (hipStream_t)stream);
   if (rc) PIPS_FAIL(pips::PIPS_ERR_RCCL, "ncclAllReduce failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
   return 0;
}

void pips_hip_comm_destroy(void* comm) {
   Comm* c = (Comm*)comm;
   if (!c) return;
   if (c->comm) g_rccl.destroy(c->comm);
   delete c;
}

}  // extern "C"
