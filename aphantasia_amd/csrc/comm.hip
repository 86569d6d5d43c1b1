// The one collective of the multi-GPU path: an all-reduce (sum, f32) of the parameter gradient
// over the ranks of one node, RCCL called directly -- no torch.distributed on the data path.
//
//   rank 0:     aph_comm_unique_id(uid)            (ncclGetUniqueId; the 128 bytes travel to the other ranks out of band)
//   every rank: aph_comm_init(rank, nranks, uid, &comm)      one process per GPU, the device current at the call
//   per step:   aph_allreduce_f32(comm, d_buf, n, stream)    in place, asynchronous on the caller's compute stream
//               (capturable into a hipGraph together with the kernels around it)
//
// RCCL is bound at run time (dlopen): the library has no link-time dependency on it, single-GPU users never load it, and
// inside a PyTorch process the copy PyTorch already mapped is reused, so the process holds ONE RCCL (the same reason
// `import torch` must precede loading this library: one HIP runtime per process, aphantasia_amd/_ffi.py).
#include "aph_device.h"
#include "aph_host.h"

#ifndef APH_EMU
#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace {

// the slice of rccl.h this file needs (ABI-stable NCCL 2.x API)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
constexpr int kNcclSuccess = 0, kNcclFloat32 = 7, kNcclSum = 0;

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;      // optional
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // a copy that is already mapped (PyTorch's) first, then the ROCm installation's
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.h) break; }
    if (!r.h) for (const char* n : names) { r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.h) break; }
    if (!r.h) { r.err = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "?"); return; }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))dlsym(r.h, "ncclCommCount");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.CommDestroy || !r.GetErrorString) r.err = "librccl.so lacks the NCCL 2.x entry points";
  });
  return r;
}

int rccl_fail(const char* where, ncclResult_t e) { return aph_fail(APH_ERR_COMM, "%s: RCCL error %d (%s)", where, e, rccl().GetErrorString(e)); }

}  // namespace

struct aph_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
};

extern "C" {

int aph_comm_unique_id(void* uid128) {
  APH_TRY
  if (!uid128) return aph_fail(APH_ERR_ARG, "aph_comm_unique_id: null argument");
  Rccl& r = rccl();
  if (!r.err.empty()) return aph_fail(APH_ERR_COMM, "aph_comm_unique_id: %s", r.err.c_str());
  ncclUniqueId id;
  if (ncclResult_t e = r.GetUniqueId(&id)) return rccl_fail("aph_comm_unique_id", e);
  memcpy(uid128, id.internal, 128);
  return APH_OK;
  APH_CATCH
}

int aph_comm_init(int rank, int nranks, const void* uid128, aph_comm** out) {
  APH_TRY
  if (!uid128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return aph_fail(APH_ERR_ARG, "aph_comm_init: bad argument (rank %d of %d)", rank, nranks);
  Rccl& r = rccl();
  if (!r.err.empty()) return aph_fail(APH_ERR_COMM, "aph_comm_init: %s", r.err.c_str());
  ncclUniqueId id;
  memcpy(id.internal, uid128, 128);
  auto* c = new aph_comm();
  c->rank = rank; c->nranks = nranks;
  if (ncclResult_t e = r.CommInitRank(&c->comm, nranks, id, rank)) { delete c; return rccl_fail("aph_comm_init", e); }
  *out = c;
  return APH_OK;
  APH_CATCH
}

int aph_allreduce_f32(aph_comm* c, float* d_buf, size_t n, void* stream_) {
  APH_TRY
  if (!c || !c->comm || !d_buf) return aph_fail(APH_ERR_ARG, "aph_allreduce_f32: null argument");
  if (ncclResult_t e = rccl().AllReduce(d_buf, d_buf, n, kNcclFloat32, kNcclSum, c->comm, (hipStream_t)stream_)) return rccl_fail("aph_allreduce_f32", e);
  return APH_OK;
  APH_CATCH
}

// number of ranks the COMMUNICATOR itself reports (ncclCommCount): what bench.py prints as `rccl_ranks_seen`
int aph_comm_ranks(aph_comm* c, int* nranks) {
  APH_TRY
  if (!c || !c->comm || !nranks) return aph_fail(APH_ERR_ARG, "aph_comm_ranks: null argument");
  if (!rccl().CommCount) return aph_fail(APH_ERR_UNSUPPORTED, "aph_comm_ranks: this librccl.so has no ncclCommCount");
  if (ncclResult_t e = rccl().CommCount(c->comm, nranks)) return rccl_fail("aph_comm_ranks", e);
  return APH_OK;
  APH_CATCH
}

int aph_comm_destroy(aph_comm* c) {
  if (!c) return APH_OK;
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  delete c;
  return APH_OK;
}

}  // extern "C"

#else   // the host interpreter has no collective library: single-rank only

struct aph_comm { int unused; };
extern "C" {
int aph_comm_unique_id(void*) { return aph_fail(APH_ERR_UNSUPPORTED, "aph_comm_unique_id: not available in the host interpreter"); }
int aph_comm_init(int, int, const void*, aph_comm**) { return aph_fail(APH_ERR_UNSUPPORTED, "aph_comm_init: not available in the host interpreter"); }
int aph_allreduce_f32(aph_comm*, float*, size_t, void*) { return aph_fail(APH_ERR_UNSUPPORTED, "aph_allreduce_f32: not available in the host interpreter"); }
int aph_comm_ranks(aph_comm*, int*) { return aph_fail(APH_ERR_UNSUPPORTED, "aph_comm_ranks: not available in the host interpreter"); }
int aph_comm_destroy(aph_comm*) { return APH_OK; }
}
#endif
