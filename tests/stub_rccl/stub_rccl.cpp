// stub_rccl.cpp — TEST INFRASTRUCTURE: a loop-back stand-in for librccl that exports the eight entry points
// velesdb_amd/csrc/shard_group.hip binds with dlsym (ncclGetUniqueId, ncclCommInitRank, ncclCommInitAll,
// ncclCommDestroy, ncclAllGather, ncclGroupStart, ncclGroupEnd, ncclGetErrorString).
//
// Why: the in-process multi-device branch of the product (ncclCommInitAll + one grouped ncclAllGather per shard,
// shard_group.hip ensure_group_comms / group_exchange_merge) only executes when a handle's shards sit on DISTINCT
// devices, and the test boxes have one GPU.  With VELESDB_RCCL_LIB=<this library> and VELESDB_SHARD_FORCE_COLLECTIVE=1
// the product takes exactly that branch over co-located shards; this transport then moves the bytes the way an
// all-gather would (every communicator's send chunk into every communicator's receive buffer, stream-ordered).
// Never shipped, never linked: loaded by tests/test_gpu_sharded.py only.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

namespace {

struct Group;
struct Comm {
  int rank = 0, nranks = 1, device = 0;
  Group* grp = nullptr;
};
struct Op {
  const void* send = nullptr;
  void* recv = nullptr;
  size_t bytes = 0;
  hipStream_t st = nullptr;
  bool set = false;
};
struct Group {
  std::vector<Comm*> members;
  std::vector<Op> pending;
  int alive = 0;
};

std::mutex g_mu;
thread_local int t_depth = 0;
thread_local std::vector<Group*> t_touched;
int g_allgathers = 0, g_init_all = 0;

size_t dtype_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

// every member has queued its op: dst d receives src s's chunk at recv_d + s * bytes, ordered behind s's stream;
// every source stream then waits for all its readers (its send buffer may be rewritten right after)
ncclResult_t run_group(Group* g) {
  const size_t n = g->members.size();
  for (size_t i = 0; i < n; i++)
    if (!g->pending[i].set) return ncclInvalidUsage;
  std::vector<hipEvent_t> ready(n), done(n);
  for (size_t s = 0; s < n; s++) {
    if (hipSetDevice(g->members[s]->device) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&ready[s], hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&done[s], hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(ready[s], g->pending[s].st) != hipSuccess) return ncclUnhandledCudaError;
  }
  for (size_t d = 0; d < n; d++) {
    if (hipSetDevice(g->members[d]->device) != hipSuccess) return ncclUnhandledCudaError;
    const Op& od = g->pending[d];
    for (size_t s = 0; s < n; s++) {
      const Op& os = g->pending[s];
      if (os.bytes != od.bytes) return ncclInvalidArgument;
      char* dst = static_cast<char*>(od.recv) + s * od.bytes;
      if (s != d && hipStreamWaitEvent(od.st, ready[s], 0) != hipSuccess) return ncclUnhandledCudaError;
      if (dst != os.send && od.bytes &&
          hipMemcpyAsync(dst, os.send, od.bytes, hipMemcpyDefault, od.st) != hipSuccess)
        return ncclUnhandledCudaError;
    }
    if (hipEventRecord(done[d], od.st) != hipSuccess) return ncclUnhandledCudaError;
  }
  for (size_t s = 0; s < n; s++) {
    if (hipSetDevice(g->members[s]->device) != hipSuccess) return ncclUnhandledCudaError;
    for (size_t d = 0; d < n; d++)
      if (d != s && hipStreamWaitEvent(g->pending[s].st, done[d], 0) != hipSuccess) return ncclUnhandledCudaError;
  }
  for (size_t s = 0; s < n; s++) {
    (void)hipEventDestroy(ready[s]);  // (destruction is deferred until the recorded work has completed)
    (void)hipEventDestroy(done[s]);
    g->pending[s] = Op{};
  }
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof(*id));
  std::memcpy(id->internal, "velesdb-stub", 12);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId, int rank) {
  if (nranks != 1 || rank != 0) return ncclInvalidUsage;  // a loop-back transport has nobody else to talk to
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
  std::lock_guard<std::mutex> lk(g_mu);
  Group* g = new Group();
  Comm* c = new Comm();
  c->device = dev;
  c->grp = g;
  g->members.push_back(c);
  g->pending.resize(1);
  g->alive = 1;
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  if (ndev < 1) return ncclInvalidArgument;
  std::lock_guard<std::mutex> lk(g_mu);
  g_init_all++;
  Group* g = new Group();
  g->pending.resize((size_t)ndev);
  g->alive = ndev;
  for (int i = 0; i < ndev; i++) {
    Comm* c = new Comm();
    c->rank = i;
    c->nranks = ndev;
    c->device = devlist ? devlist[i] : i;
    c->grp = g;
    g->members.push_back(c);
    comms[i] = reinterpret_cast<ncclComm_t>(c);
  }
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  std::lock_guard<std::mutex> lk(g_mu);
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclInvalidArgument;
  Group* g = c->grp;
  if (--g->alive == 0) {
    for (Comm* m : g->members) delete m;
    delete g;
  }
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  t_depth++;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (t_depth <= 0) return ncclInvalidUsage;
  if (--t_depth > 0) return ncclSuccess;
  std::lock_guard<std::mutex> lk(g_mu);
  ncclResult_t r = ncclSuccess;
  for (Group* g : t_touched) {
    ncclResult_t e = run_group(g);
    if (e != ncclSuccess) r = e;
  }
  t_touched.clear();
  return r;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || !recvbuff || (!sendbuff && sendcount)) return ncclInvalidArgument;
  std::lock_guard<std::mutex> lk(g_mu);
  g_allgathers++;
  Group* g = c->grp;
  Op& op = g->pending[(size_t)c->rank];
  if (op.set) return ncclInvalidUsage;
  op = Op{sendbuff, recvbuff, sendcount * dtype_bytes(datatype), stream, true};
  if (t_depth > 0) {
    bool seen = false;
    for (Group* t : t_touched) seen |= t == g;
    if (!seen) t_touched.push_back(g);
    return ncclSuccess;
  }
  if (g->members.size() != 1) return ncclInvalidUsage;  // an ungrouped call can only complete on a 1-rank communicator
  return run_group(g);
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "stub transport: HIP call failed";
    case ncclInvalidArgument: return "stub transport: invalid argument";
    case ncclInvalidUsage: return "stub transport: invalid usage";
    default: return "stub transport: error";
  }
}

// test probes (not part of RCCL's API)
int velesdb_stub_allgathers() { return g_allgathers; }
int velesdb_stub_init_all_calls() { return g_init_all; }

}  // extern "C"
