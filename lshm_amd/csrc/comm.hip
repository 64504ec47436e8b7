// Collectives of the data-parallel step behind the C ABI: RCCL over xGMI, one rank per GPU,
// enqueued on the caller's HIP stream so they order with the kernels without a host round trip.  The
// reference has no distributed code at all; the call sites replaced are the two places where upstream's
// single process consumes a full-batch quantity: loss.backward() + the logged terms
// (src/kharmonic_lofar.py:175-181) and the optimizer step that follows (:185).
//
// RCCL is bound at run time (dlopen / dlsym): the library builds, loads and passes its CPU-side tests on a
// machine without RCCL or without a GPU, and a process that already carries a copy of RCCL (PyTorch does)
// shares that copy instead of loading a second one.
#include "../../include/lshm.h"
#include "kernels.h"

#include <dlfcn.h>
#include <stdio.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r = [] {
    Rccl q;
    // LSHM_RCCL_LIB: bind the eight entry points from this library instead (the rehearsal tests point it at a
    // stand-in that sums through host shared memory, so two ranks on ONE GPU can drive the engine-attached path)
    const char* override_path = getenv("LSHM_RCCL_LIB");
    if (override_path && *override_path) {
      // (said aloud: the collectives of this process then come from a library the environment named, not from RCCL)
      fprintf(stderr, "lshm: RCCL entry points bound from LSHM_RCCL_LIB=%s\n", override_path);
      q.handle = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
    } else {
      const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (const char* n : names)
        if ((q.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;  // a copy the process already has
      for (const char* n : names)
        if (!q.handle) q.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    }
    if (!q.handle) return q;
#define LSHM_SYM(field, name) q.field = reinterpret_cast<decltype(q.field)>(dlsym(q.handle, name))
    LSHM_SYM(GetUniqueId, "ncclGetUniqueId");
    LSHM_SYM(CommInitRank, "ncclCommInitRank");
    LSHM_SYM(CommDestroy, "ncclCommDestroy");
    LSHM_SYM(AllReduce, "ncclAllReduce");
    LSHM_SYM(GroupStart, "ncclGroupStart");
    LSHM_SYM(GroupEnd, "ncclGroupEnd");
    LSHM_SYM(GetErrorString, "ncclGetErrorString");
#undef LSHM_SYM
    q.ok = q.GetUniqueId && q.CommInitRank && q.CommDestroy && q.AllReduce && q.GroupStart && q.GroupEnd;
    return q;
  }();
  return r;
}

int fail(const char* what, ncclResult_t rc) {
  char msg[200];
  const char* s = rccl().GetErrorString ? rccl().GetErrorString(rc) : "?";
  snprintf(msg, sizeof msg, "%s: RCCL error %d (%s)", what, (int)rc, s);
  lshm::set_last_error(msg);
  return LSHM_ERR_COMM;
}

}  // namespace

struct lshm_comm {
  ncclComm_t comm;
  int rank, world, device;
};

extern "C" {

int lshm_comm_available(void) { return rccl().ok ? 1 : 0; }

int lshm_comm_unique_id(char* id128) {
  if (!id128) { lshm::set_last_error("comm_unique_id: null buffer"); return LSHM_ERR_ARG; }
  if (!rccl().ok) { lshm::set_last_error("comm: RCCL is not available in this process"); return LSHM_ERR_UNSUPPORTED; }
  static_assert(sizeof(ncclUniqueId) == LSHM_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id;
  const ncclResult_t rc = rccl().GetUniqueId(&id);
  if (rc != ncclSuccess) return fail("comm_unique_id", rc);
  memcpy(id128, &id, sizeof id);
  return LSHM_OK;
}

int lshm_comm_init(const char* id128, int rank, int world, lshm_comm** out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) {
    lshm::set_last_error("comm_init: bad argument");
    return LSHM_ERR_ARG;
  }
  if (!rccl().ok) { lshm::set_last_error("comm: RCCL is not available in this process"); return LSHM_ERR_UNSUPPORTED; }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  lshm_comm* c = new lshm_comm();
  c->rank = rank;
  c->world = world;
  c->device = -1;
  (void)hipGetDevice(&c->device);  // the communicator belongs to the device that is current now
  const ncclResult_t rc = rccl().CommInitRank(&c->comm, world, id, rank);
  if (rc != ncclSuccess) { delete c; return fail("comm_init", rc); }
  *out = c;
  return LSHM_OK;
}

void lshm_comm_destroy(lshm_comm* c) {
  if (!c) return;
  if (rccl().ok) (void)rccl().CommDestroy(c->comm);
  delete c;
}

int lshm_comm_rank(const lshm_comm* c) { return c ? c->rank : -1; }
int lshm_comm_world(const lshm_comm* c) { return c ? c->world : 0; }

int lshm_comm_allreduce_flat(lshm_comm* c, float* buf, size_t n, double* tail, size_t ntail, lshm_stream_t s) {
  if (!c || (!buf && n) || (!tail && ntail)) { lshm::set_last_error("comm_allreduce_flat: bad argument"); return LSHM_ERR_ARG; }
  float* bufs[1] = {buf};
  size_t ns[1] = {n};
  return lshm::comm_allreduce_segments(c, bufs, ns, n ? 1 : 0, tail, ntail, reinterpret_cast<hipStream_t>(s));
}

}  // extern "C"

namespace lshm {
// SUM all-reduce, in place, of up to a few float ranges and one double range as ONE group (one fused launch)
int comm_allreduce_segments(lshm_comm* c, float* const* bufs, const size_t* ns, int nseg, double* tail, size_t ntail,
                            hipStream_t st) {
  Rccl& r = rccl();
  if (!r.ok) { set_last_error("comm: RCCL is not available in this process"); return LSHM_ERR_UNSUPPORTED; }
  ncclResult_t rc = r.GroupStart();
  if (rc != ncclSuccess) return fail("comm_allreduce", rc);
  for (int i = 0; i < nseg && rc == ncclSuccess; ++i)
    if (ns[i]) rc = r.AllReduce(bufs[i], bufs[i], ns[i], ncclFloat32, ncclSum, c->comm, st);
  if (rc == ncclSuccess && ntail) rc = r.AllReduce(tail, tail, ntail, ncclFloat64, ncclSum, c->comm, st);
  const ncclResult_t rc2 = r.GroupEnd();
  if (rc != ncclSuccess) return fail("comm_allreduce", rc);
  if (rc2 != ncclSuccess) return fail("comm_allreduce", rc2);
  return LSHM_OK;
}
int comm_world(const lshm_comm* c) { return c ? c->world : 1; }
}  // namespace lshm
