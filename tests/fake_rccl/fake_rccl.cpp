// TEST INFRASTRUCTURE, not a product path: a stand-in for the eight RCCL entry points that
// lshm_amd/csrc/comm.hip binds at run time (LSHM_RCCL_LIB points the library at this file's .so).
// RCCL refuses two ranks on one device, and the test box has one GPU; with this stand-in two processes
// sharing that GPU drive the ENGINE-ATTACHED collective path (lshm_engine_set_comm: the early netT / netF
// bucket on its own stream, the closing group, the loss terms of the gradient-free closures) end to end.
//
// Semantics kept: in-place SUM all-reduce of float32 / float64 ranges, grouped calls, the result visible to
// work enqueued on any stream after the call returns.  Not kept: asynchrony -- every collective synchronises
// its stream and exchanges through POSIX shared memory on the host (ranks add the slots in rank order, so
// every rank ends with bitwise the same sums, like a ring would not necessarily give).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <vector>

namespace {

constexpr size_t kSlotBytes = 64u << 20;  // per rank; the gradient arena is 6.9 MB
struct Header {
  std::atomic<int> arrive;
  std::atomic<int> gen;
  char pad[56];
};
struct Op { const void* send; void* recv; size_t count; ncclDataType_t type; hipStream_t stream; };
thread_local int g_depth = 0;
thread_local std::vector<std::pair<ncclComm_t, Op>> g_ops;

}  // namespace

struct ncclComm {
  int rank, world;
  char name[64];
  Header* hdr;
  char* slots;
  size_t map_bytes;
};

namespace {

bool barrier(ncclComm* c) {
  Header* h = c->hdr;
  const int g = h->gen.load();
  if (h->arrive.fetch_add(1) + 1 == c->world) {
    h->arrive.store(0);
    h->gen.fetch_add(1);
    return true;
  }
  const time_t t0 = time(nullptr);
  while (h->gen.load() == g) {
    usleep(50);
    if (time(nullptr) - t0 > 300) return false;  // a rank never arrived: fail the test instead of hanging it
  }
  return true;
}

template <class T>
void sum_slots(ncclComm* c, size_t count, std::vector<char>& out) {
  out.resize(count * sizeof(T));
  T* o = reinterpret_cast<T*>(out.data());
  for (size_t i = 0; i < count; ++i) o[i] = 0;
  for (int r = 0; r < c->world; ++r) {
    const T* s = reinterpret_cast<const T*>(c->slots + (size_t)r * kSlotBytes);
    for (size_t i = 0; i < count; ++i) o[i] += s[i];
  }
}

ncclResult_t run(ncclComm* c, const Op& op) {
  const size_t es = op.type == ncclFloat64 ? 8 : 4;
  if ((op.type != ncclFloat32 && op.type != ncclFloat64) || op.count * es > kSlotBytes) return ncclInvalidArgument;
  if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(c->slots + (size_t)c->rank * kSlotBytes, op.send, op.count * es, hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;
  std::vector<char> out;
  if (op.type == ncclFloat64) sum_slots<double>(c, op.count, out);
  else sum_slots<float>(c, op.count, out);
  if (!barrier(c)) return ncclSystemError;  // every rank has read the slots: they may be overwritten
  if (hipMemcpy(op.recv, out.data(), op.count * es, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  static std::atomic<int> counter{0};
  memset(id, 0, sizeof *id);
  snprintf(id->internal, sizeof id->internal, "/lshm_fake_rccl_%d_%d", (int)getpid(), counter.fetch_add(1));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
  if (!out || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->rank = rank;
  c->world = world;
  strncpy(c->name, id.internal, sizeof c->name - 1);
  c->map_bytes = sizeof(Header) + (size_t)world * kSlotBytes;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return ncclSystemError; }
  void* m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { delete c; return ncclSystemError; }
  c->hdr = reinterpret_cast<Header*>(m);  // a fresh segment is zero-filled: arrive = gen = 0
  c->slots = reinterpret_cast<char*>(m) + sizeof(Header);
  *out = c;
  return barrier(c) ? ncclSuccess : ncclSystemError;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  munmap(c->hdr, c->map_bytes);
  if (c->rank == 0) shm_unlink(c->name);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth > 0) return ncclSuccess;
  ncclResult_t rc = ncclSuccess;
  for (auto& co : g_ops)
    if (rc == ncclSuccess) rc = run(co.first, co.second);
  g_ops.clear();
  return rc;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream) {
  if (!comm || !send || !recv || op != ncclSum) return ncclInvalidArgument;
  const Op o{send, recv, count, type, stream};
  if (g_depth > 0) { g_ops.emplace_back(comm, o); return ncclSuccess; }
  return run(comm, o);
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake RCCL stand-in error"; }

}  // extern "C"
