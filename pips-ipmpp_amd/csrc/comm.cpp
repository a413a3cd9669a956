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
typedef int (*fn_reduce_scatter)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t);
typedef int (*fn_broadcast)(const void*, void*, size_t, int, int, nccl_comm_t, hipStream_t);
typedef int (*fn_destroy)(nccl_comm_t);
typedef const char* (*fn_errstr)(int);

struct Rccl {
   void* lib = nullptr;
   fn_get_id get_id = nullptr;
   fn_init_rank init_rank = nullptr;
   fn_allreduce allreduce = nullptr;
   fn_reduce_scatter reduce_scatter = nullptr;
   fn_all_gather all_gather = nullptr;
   fn_broadcast broadcast = nullptr;
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
      reduce_scatter = (fn_reduce_scatter)dlsym(lib, "ncclReduceScatter");
      all_gather = (fn_all_gather)dlsym(lib, "ncclAllGather");
      broadcast = (fn_broadcast)dlsym(lib, "ncclBroadcast");
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
   int n_ranks = 1, rank = 0;
   pips_hip_reduce_scatter_cb ext_reduce_scatter = nullptr;   // optional: MPI_Reduce_scatter_block / MPI_Allgather of the host
   pips_hip_all_gather_cb ext_all_gather = nullptr;
   pips_hip_broadcast_cb ext_broadcast = nullptr;             // optional: MPI_Bcast of the host
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
   c->n_ranks = n_ranks; c->rank = rank;
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

int pips_hip_comm_set_external_rsag(void* comm, int n_ranks, int rank, pips_hip_reduce_scatter_cb reduce_scatter, pips_hip_all_gather_cb all_gather) {
   Comm* c = (Comm*)comm;
   if (!c || !c->external || n_ranks < 1 || rank < 0 || rank >= n_ranks || !reduce_scatter || !all_gather)
      PIPS_FAIL(pips::PIPS_ERR_ARG, "pips_hip_comm_set_external_rsag: bad arguments");
   c->n_ranks = n_ranks; c->rank = rank;
   c->ext_reduce_scatter = reduce_scatter;
   c->ext_all_gather = all_gather;
   return 0;
}

/* Sum over the ranks as reduce-scatter + all-gather: rank r ends up owning the reduced slice [r * chunk, (r + 1) * chunk) after the
 * first phase (what a distributed root factorisation would keep), the second phase replicates it.  On xGMI (point-to-point
 * links, 7 per GPU) both phases are one direct exchange with every peer; the ring all-reduce RCCL may pick instead is bound by
 * a single link.  buf_dev must hold n_padded = chunk * n_ranks doubles, chunk = ceil(n / n_ranks); entries n .. n_padded are
 * scratch. */
int pips_hip_allreduce_sum_rsag(void* comm, double* buf_dev, size_t n, void* stream) {
   Comm* c = (Comm*)comm;
   if (!c || !buf_dev) PIPS_FAIL(pips::PIPS_ERR_ARG, "pips_hip_allreduce_sum_rsag: bad arguments");
   const size_t P = (size_t)c->n_ranks, chunk = (n + P - 1) / P;
   if (c->external) {
      if (!c->ext_reduce_scatter || !c->ext_all_gather) return pips_hip_allreduce_sum(comm, buf_dev, n, stream);   // host has no such pair
      if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) PIPS_FAIL(pips::PIPS_ERR_HIP, "stream sync before the external reduce-scatter failed");
      int rc = c->ext_reduce_scatter(c->user, buf_dev, chunk);
      if (rc) PIPS_FAIL(pips::PIPS_ERR_RCCL, "external reduce-scatter callback returned %d", rc);
      rc = c->ext_all_gather(c->user, buf_dev, chunk);
      if (rc) PIPS_FAIL(pips::PIPS_ERR_RCCL, "external all-gather callback returned %d", rc);
      return 0;
   }
   if (!g_rccl.reduce_scatter || !g_rccl.all_gather) return pips_hip_allreduce_sum(comm, buf_dev, n, stream);
   double* mine = buf_dev + (size_t)c->rank * chunk;
   int rc = g_rccl.reduce_scatter(buf_dev, mine, chunk, kNcclDouble, kNcclSum, c->comm, (hipStream_t)stream);
   if (rc) PIPS_FAIL(pips::PIPS_ERR_RCCL, "ncclReduceScatter failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
   rc = g_rccl.all_gather(mine, buf_dev, chunk, kNcclDouble, c->comm, (hipStream_t)stream);
   if (rc) PIPS_FAIL(pips::PIPS_ERR_RCCL, "ncclAllGather failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
   return 0;
}

/* In-place all-gather of n_parts (= number of ranks) equal parts: rank r's part lies at buf_dev + r * chunk on entry, all n_parts * chunk
 * doubles are every rank's on return.  RCCL: ncclAllGather; a host-supplied communicator: its all-gather callback (pips_hip_comm_set_external_rsag), or - it has none -
 * the all-reduce, for which the OTHER ranks' parts must be zero on entry (exact either way; the all-reduce moves n_ranks times the bytes).
 * Used by deterministic mode over several ranks: every rank's group buffers to every rank (Engine::det_global). */
int pips_hip_all_gather(void* comm, double* buf_dev, size_t chunk, int n_parts, void* stream) {
   Comm* c = (Comm*)comm;
   if (!c || !buf_dev || n_parts < 1) PIPS_FAIL(pips::PIPS_ERR_ARG, "pips_hip_all_gather: bad arguments");
   const size_t P = (size_t)n_parts;   // (= the number of ranks; a host-supplied communicator that was given no rank count still knows how to all-reduce)
   if (c->external) {
      if (!c->ext_all_gather || c->n_ranks != n_parts) return pips_hip_allreduce_sum(comm, buf_dev, P * chunk, stream);
      if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) PIPS_FAIL(pips::PIPS_ERR_HIP, "stream sync before the external all-gather failed");
      const int rc = c->ext_all_gather(c->user, buf_dev, chunk);
      if (rc) PIPS_FAIL(pips::PIPS_ERR_RCCL, "external all-gather callback returned %d", rc);
      return 0;
   }
   if (!g_rccl.all_gather || c->n_ranks != n_parts) return pips_hip_allreduce_sum(comm, buf_dev, P * chunk, stream);
   const int rc = g_rccl.all_gather(buf_dev + (size_t)c->rank * chunk, buf_dev, chunk, kNcclDouble, c->comm, (hipStream_t)stream);
   if (rc) PIPS_FAIL(pips::PIPS_ERR_RCCL, "ncclAllGather failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
   return 0;
}

int pips_hip_comm_set_external_broadcast(void* comm, int n_ranks, int rank, pips_hip_broadcast_cb broadcast) {
   Comm* c = (Comm*)comm;
   if (!c || !c->external || n_ranks < 1 || rank < 0 || rank >= n_ranks || !broadcast) PIPS_FAIL(pips::PIPS_ERR_ARG, "pips_hip_comm_set_external_broadcast: bad arguments");
   c->n_ranks = n_ranks; c->rank = rank;
   c->ext_broadcast = broadcast;
   return 0;
}

/* buf_dev of rank `root` goes to every rank (the panel of the distributed root factorisation).  RCCL: ncclBroadcast; a host-supplied
 * communicator: its broadcast callback, or - it has none - an all-reduce in which the other ranks contribute zeros (exact, twice the
 * bytes on the wire: the caller's buffer on the other ranks is cleared for that). */
int pips_hip_broadcast(void* comm, double* buf_dev, size_t n, int root, void* stream) {
   Comm* c = (Comm*)comm;
   if (!c || !buf_dev || root < 0) PIPS_FAIL(pips::PIPS_ERR_ARG, "pips_hip_broadcast: bad arguments");
   if (c->external) {
      if (c->ext_broadcast) {
         if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) PIPS_FAIL(pips::PIPS_ERR_HIP, "stream sync before the external broadcast failed");
         const int rc = c->ext_broadcast(c->user, buf_dev, n, root);
         if (rc) PIPS_FAIL(pips::PIPS_ERR_RCCL, "external broadcast callback returned %d", rc);
         return 0;
      }
      return pips_hip_allreduce_sum(comm, buf_dev, n, stream);   // (the caller cleared the buffer on the other ranks: pips_hip_comm_has_broadcast == 0)
   }
   if (!g_rccl.broadcast) return pips_hip_allreduce_sum(comm, buf_dev, n, stream);
   const int rc = g_rccl.broadcast(buf_dev, buf_dev, n, kNcclDouble, root, c->comm, (hipStream_t)stream);
   if (rc) PIPS_FAIL(pips::PIPS_ERR_RCCL, "ncclBroadcast failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
   return 0;
}

/* 1: pips_hip_broadcast moves the root's bytes only (RCCL, or an external broadcast callback); 0: it falls back to the all-reduce of zeros,
 * and the non-root ranks must clear their buffer before the call */
int pips_hip_comm_has_broadcast(void* comm) {
   Comm* c = (Comm*)comm;
   if (!c) return 0;
   return c->external ? (c->ext_broadcast ? 1 : 0) : (g_rccl.broadcast ? 1 : 0);
}

int pips_hip_comm_size(void* comm) { return comm ? ((Comm*)comm)->n_ranks : 1; }

void pips_hip_comm_destroy(void* comm) {
   Comm* c = (Comm*)comm;
   if (!c) return;
   if (c->comm) g_rccl.destroy(c->comm);
   delete c;
}

}  // extern "C"
