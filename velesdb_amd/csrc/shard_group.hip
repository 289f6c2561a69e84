// shard_group.hip — one VectorIndex object over several GPUs (north-star configs[4]).
//
// The reference's seam is one `VectorIndex` whatever is behind it (crates/velesdb-core/src/index/mod.rs:30-83) and a
// backend chosen at construction (core/gpu.rs:45-58 ComputeBackend::best_available).  Here:
//   * VDB_SHARD_RANGE — the exact sweep shards by contiguous ranges of the internal rows.  Every shard is a complete
//     single-device index (index.hip); a search = per-shard top-k (the sweep kernels) -> pack_shard_records (12-byte
//     (u64 id, f32 score) records, the count folded into sentinel records) -> ONE all-gather of nq * k records per
//     shard -> merge_shards_topk (S * k -> k per query, IEEE total-order key, ties by global row order = (shard,
//     position in the shard's list)).  Transport: RCCL (ncclAllGather inside one ncclGroup over the communicators of
//     ncclCommInitAll) when the shards sit on distinct devices; device-to-device copies when shards are co-located on
//     one device (a test configuration: RCCL refuses duplicate devices).
//   * VDB_SHARD_REPLICA — the graph path: every device holds everything, a query batch is split (query_slice), no
//     collective.
//   * one process per GPU (bench.py under torchrun): vdb_hip_index_join_group attaches an RCCL communicator
//     (ncclCommInitRank) to a plain single-device index; its searches then end with the same pack / all-gather /
//     merge on the caller's stream and every rank holds the global top-k.
// RCCL is bound at first use with dlopen (librccl.so.1: the copy already in the process — PyTorch's — or ROCm's), so
// single-GPU users never load it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <functional>
#include <thread>

#include "vdb_probe_env.hpp"
#include "vdb_device.hpp"
#include "vdb_index.hpp"
#include "vdb_kernels.hpp"
#include "vdb_shard_wire.hpp"

namespace vdb {

// ---- RCCL, bound at first use -----------------------------------------------------------------
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// why rccl() returned nullptr — kept OUTSIDE the nullable object, so that the error paths can quote it
static std::string& rccl_load_error() {
  static std::string e;
  return e;
}

// VELESDB_RCCL_LIB names the library to bind instead of the default search (a site-specific RCCL build; the loop-back
// transport of tests/stub_rccl that drives the collective branch on one GPU; a name that does not exist = "no RCCL").
static Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    std::string& err = rccl_load_error();
    const char* forced = probe_env("VELESDB_RCCL_LIB");
    if (forced && forced[0]) {
      r.h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
      if (!r.h) {
        const char* de = dlerror();
        err = std::string("cannot load ") + forced + " (VELESDB_RCCL_LIB): " + (de ? de : "unknown error");
        return;
      }
      fprintf(stderr, "velesdb-hip: collective transport loaded from VELESDB_RCCL_LIB=%s\n", forced);
    } else {
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.h) break;
      }
      if (!r.h) {
        const char* de = dlerror();
        err = std::string("cannot load librccl.so.1: ") + (de ? de : "unknown error");
        return;
      }
    }
    bool ok = true;
    auto sym = [&](const char* n) {
      void* p = dlsym(r.h, n);
      if (!p) {
        ok = false;
        err = std::string("the RCCL library lacks ") + n;
      }
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) {
      dlclose(r.h);
      r.h = nullptr;
    }
  });
  return r.h ? &r : nullptr;
}
static int32_t rccl_fail(const char* what, ncclResult_t e) {
  Rccl* r = rccl();
  return fail(VDB_ERR_HIP, std::string(what) + ": " + (r ? r->GetErrorString(e) : "rccl not loaded"));
}
#define VDB_NCCL(call)                                    \
  do {                                                    \
    ncclResult_t _r = (call);                             \
    if (_r != ncclSuccess) return rccl_fail(#call, _r);   \
  } while (0)

// ---- records and the merge kernel ---------------------------------------------------------------
// (the record layout, the sentinels, the selection key and the rank rule: vdb_shard_wire.hpp — shared with the host model of the CPU tests)

__global__ __launch_bounds__(256) void pack_shard_records(const uint64_t* ids, const float* scores, const uint32_t* n,
                                                          uint32_t* rec, uint32_t nq, uint32_t k) {
  const uint64_t total = (uint64_t)nq * k;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t)gridDim.x * 256) {
    const uint32_t q = (uint32_t)(i / k), p = (uint32_t)(i % k);
    const uint32_t c = n[q];
    const bool live = c != 0xFFFFFFFFu && p < c;  // (slots past the count are never read: the arrays hold k entries per query either way)
    wire::pack(live ? ids[i] : 0ull, live ? __float_as_uint(scores[i]) : 0u, c, p, rec + i * 3);
  }
}

// One block per query.  rec = [S][nq][k] records, every shard's list best first (ascending selection key).  The rank of
// record (s, p) in the merged order is p + sum over the other shards t of the number of their records that precede it:
// a binary search per (record, shard) — wire::merged_rank.  DistanceMetric::sort_results order (core/distance.rs:95-103).
template <bool HIB>
__global__ __launch_bounds__(256) void merge_shards_topk(const uint32_t* rec, uint64_t* out_ids, float* out_scores,
                                                         uint32_t* out_n, uint32_t S, uint32_t nq, uint32_t k) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* keys = reinterpret_cast<uint32_t*>(smem);  // [S][k] selection keys (smaller = better)
  uint32_t* ns = keys + (size_t)S * k;                  // [S] records per shard
  uint32_t* ovf = ns + S;
  const uint32_t q = blockIdx.x, tid = threadIdx.x, T = S * k;
  for (uint32_t s = tid; s < S; s += 256) ns[s] = 0;
  if (tid == 0) *ovf = 0;
  __syncthreads();
  for (uint32_t i = tid; i < T; i += 256) {
    const uint32_t s = i / k, p = i - s * k;
    const uint32_t* r = rec + (((size_t)s * nq + q) * k + p) * 3;
    if (wire::is_overflow(r)) *ovf = 1;
    keys[i] = wire::select_key(r, HIB);
    if (!wire::is_empty(r)) atomicAdd(&ns[s], 1u);
  }
  __syncthreads();
  uint32_t total = 0;
  for (uint32_t s = 0; s < S; s++) total += ns[s];
  for (uint32_t i = tid; i < T; i += 256) {
    const uint32_t s = i / k, p = i - s * k;
    if (p >= ns[s]) continue;
    const uint32_t rank = wire::merged_rank(keys, ns, S, k, s, p);
    if (rank < k) {
      const uint32_t* r = rec + (((size_t)s * nq + q) * k + p) * 3;
      out_ids[(size_t)q * k + rank] = wire::id_of(r);
      out_scores[(size_t)q * k + rank] = __uint_as_float(r[2]);
    }
  }
  const uint32_t cnt = min(total, k);
  for (uint32_t e = cnt + tid; e < k; e += 256) {  // same filler as merge_topk (sweep.hip)
    out_ids[(size_t)q * k + e] = ~0ull;
    out_scores[(size_t)q * k + e] = __uint_as_float(0x7FC00000u);
  }
  if (tid == 0) out_n[q] = *ovf ? 0xFFFFFFFFu : cnt;
}

static int32_t launch_pack(const uint64_t* ids, const float* scores, const uint32_t* n, uint32_t* rec, uint32_t nq,
                           uint32_t k, hipStream_t st) {
  const uint64_t total = (uint64_t)nq * k;
  if (total == 0) return VDB_OK;
  hipLaunchKernelGGL(pack_shard_records, dim3((unsigned)std::min<uint64_t>((total + 255) / 256, 2048)), dim3(256), 0, st,
                     ids, scores, n, rec, nq, k);
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}
static int32_t launch_shard_merge(bool hib, const uint32_t* rec, uint64_t* out_ids, float* out_scores, uint32_t* out_n,
                                  uint32_t S, uint32_t nq, uint32_t k, hipStream_t st) {
  if (nq == 0) return VDB_OK;
  const size_t lds = ((size_t)S * k * 4 + (size_t)S * 4 + 16 + 15) & ~(size_t)15;
  if (lds > 64 * 1024) return fail(VDB_ERR_UNSUPPORTED, "shards x k too large for the merge kernel (> 16 K records per query)");
  if (hib)
    hipLaunchKernelGGL((merge_shards_topk<true>), dim3(nq), dim3(256), lds, st, rec, out_ids, out_scores, out_n, S, nq, k);
  else
    hipLaunchKernelGGL((merge_shards_topk<false>), dim3(nq), dim3(256), lds, st, rec, out_ids, out_scores, out_n, S, nq, k);
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}

// ---- the group ------------------------------------------------------------------------------------
struct ShardGroup {
  int mode = VDB_SHARD_REPLICA;
  std::vector<vdb_hip_index*> shards;
  uint64_t rows_per_shard = 0;   // C of the range rule
  bool distinct = true;          // all shards on different devices -> RCCL; otherwise device-to-device copies
  bool broken = false;           // a replica insert failed on some devices only: the replicas may differ, every later call fails
  std::vector<ncclComm_t> comms; // in-process communicators, created at the first sharded search
  std::vector<DevBuf> gath;      // per shard: [S][nq][k] records (every device receives everything)
  std::vector<hipEvent_t> ev;    // per shard: "records packed"
  DevBuf m_ids, m_scores, m_n;   // merged result on shard 0's device
};

void shard_group_free(ShardGroup* g) {
  if (!g) return;
  Rccl* r = g->comms.empty() ? nullptr : rccl();
  for (size_t s = 0; s < g->shards.size(); s++) {
    vdb_hip_index* c = g->shards[s];
    if (!c) continue;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (r && s < g->comms.size() && g->comms[s]) (void)r->CommDestroy(g->comms[s]);
    if (s < g->gath.size()) g->gath[s].release();
    if (s < g->ev.size() && g->ev[s]) (void)hipEventDestroy(g->ev[s]);
    if (s == 0) {
      g->m_ids.release();
      g->m_scores.release();
      g->m_n.release();
    }
    destroy_single(c);
  }
  delete g;
}

int32_t group_create(vdb_hip_index* parent, const int32_t* devices, int32_t n_devices, int32_t shard_mode,
                     uint64_t max_elements) {
  std::unique_ptr<ShardGroup, void (*)(ShardGroup*)> g(new ShardGroup(), shard_group_free);
  g->mode = shard_mode;
  const uint64_t S = (uint64_t)n_devices;
  g->rows_per_shard = std::max<uint64_t>(1, (std::max<uint64_t>(max_elements, 1) + S - 1) / S);
  for (int32_t a = 0; a < n_devices; a++)
    for (int32_t b = a + 1; b < n_devices; b++)
      if (devices[a] == devices[b]) g->distinct = false;
  // test hook: run the collective branch (ncclCommInitAll + grouped ncclAllGather) over co-located shards.  Real RCCL
  // refuses duplicate devices; the loop-back transport of tests/stub_rccl (VELESDB_RCCL_LIB) does not.
  // (honoured only together with VELESDB_RCCL_LIB, and announced: neither belongs in a deployment's environment)
  if (const char* fc = probe_env("VELESDB_SHARD_FORCE_COLLECTIVE"))
    if (fc[0] == '1' && probe_env("VELESDB_RCCL_LIB")) {
      g->distinct = true;
      fprintf(stderr, "velesdb-hip: VELESDB_SHARD_FORCE_COLLECTIVE=1 with VELESDB_RCCL_LIB=%s: co-located shards use the collective transport (test hook)\n",
              probe_env("VELESDB_RCCL_LIB"));
    }
  g->gath.resize(S);
  g->ev.assign(S, nullptr);
  for (int32_t s = 0; s < n_devices; s++) {
    vdb_hip_index* c = nullptr;
    const uint64_t cap = shard_mode == VDB_SHARD_RANGE ? g->rows_per_shard : max_elements;
    int32_t rc = create_single(parent->dim, parent->metric, parent->M, parent->efc, cap, devices[s], &c);
    if (rc != VDB_OK) return rc;
    g->shards.push_back(c);
    VDB_HIP(hipSetDevice(c->device));
    VDB_HIP(hipEventCreateWithFlags(&g->ev[s], hipEventDisableTiming));
  }
  parent->device = devices[0];
  parent->group = g.release();
  return VDB_OK;
}

vdb_hip_index* group_shard(const vdb_hip_index* ix, size_t s) { return ix->group->shards[s]; }
size_t group_size(const vdb_hip_index* ix) { return ix->group->shards.size(); }
int group_mode(const vdb_hip_index* ix) { return ix->group->mode; }

// runs fn(shard index) for every shard, one host thread per shard; the first failing status wins and its message is
// carried over to the calling thread (last_error is thread-local)
static int32_t for_each_shard(ShardGroup* g, const std::function<int32_t(size_t)>& fn, bool parallel = true) {
  const size_t S = g->shards.size();
  std::vector<int32_t> rc(S, VDB_OK);
  std::vector<std::string> msg(S);
  auto run = [&](size_t s) {
    rc[s] = guarded([&]() -> int32_t { return fn(s); });
    if (rc[s] < 0) msg[s] = vdb_hip_last_error();
  };
  if (!parallel || S == 1) {
    for (size_t s = 0; s < S; s++) run(s);
  } else {
    std::vector<std::thread> th;
    th.reserve(S);
    size_t started = 0;
    try {
      for (; started < S; started++) th.emplace_back(run, started);
    } catch (...) {  // thread creation failed (EAGAIN): the shards without a thread run here; a joinable thread must never
                     // be destroyed (std::terminate)
      for (size_t s = started; s < S; s++) run(s);
    }
    for (auto& t : th) t.join();
  }
  for (size_t s = 0; s < S; s++)
    if (rc[s] < 0) return fail(rc[s], "shard " + std::to_string(s) + ": " + msg[s]);
  return VDB_OK;
}

static size_t shard_of_row(const ShardGroup* g, uint64_t row) {
  return (size_t)std::min<uint64_t>(row / g->rows_per_shard, g->shards.size() - 1);
}

static const char* const kBrokenMsg =
    "replica group is inconsistent: an insert succeeded on some devices only; destroy the handle and rebuild it";

// kind 0 insert_batch, 1 insert_batch_parallel, 2 upload
static int32_t child_insert(vdb_hip_index* c, const uint64_t* ids, const float* vecs, uint64_t n, int kind,
                            uint32_t max_batch) {
  uint64_t ins = 0;
  int32_t rc;
  if (kind == 0)
    rc = vdb_hip_index_insert_batch(c, ids, vecs, n, &ins);
  else if (kind == 1)
    rc = vdb_hip_index_insert_batch_parallel(c, ids, vecs, n, max_batch, &ins);
  else
    rc = vdb_hip_index_upload(c, ids, vecs, n, &ins);
  if (rc < 0) return rc;
  if (ins != n) return fail(VDB_ERR_STATE, "shard rejected ids the group accepted");
  return VDB_OK;
}

int32_t group_insert(vdb_hip_index* ix, const uint64_t* ids, const float* vecs, uint64_t n, int kind, uint32_t max_batch,
                     uint64_t* inserted) {
  ShardGroup* g = ix->group;
  std::lock_guard<vdb::IndexMutex> lk(ix->mu);
  if (inserted) *inserted = 0;
  if (g->broken) return fail(VDB_ERR_STATE, kBrokenMsg);
  // duplicates (against the index and inside the batch) are skipped once, here (trait_impl.rs:23-25)
  std::vector<uint64_t> src;
  src.reserve(n);
  {
    std::unordered_map<uint64_t, char> seen;
    for (uint64_t i = 0; i < n; i++) {
      if (ix->id_to_idx.count(ids[i]) || !seen.emplace(ids[i], 1).second) continue;
      src.push_back(i);
    }
  }
  const uint64_t m = src.size();
  if (m == 0) return VDB_OK;
  const bool contiguous = m == n;
  std::vector<uint64_t> cids;
  std::vector<float> cvecs;
  if (!contiguous) {
    cids.resize(m);
    cvecs.resize((size_t)m * ix->dim);
    for (uint64_t j = 0; j < m; j++) {
      cids[j] = ids[src[j]];
      std::memcpy(cvecs.data() + (size_t)j * ix->dim, vecs + (size_t)src[j] * ix->dim, (size_t)ix->dim * 4);
    }
    ids = cids.data();
    vecs = cvecs.data();
  }
  auto commit = [&](uint64_t upto) {  // rows [0, upto) of the accepted batch are in their shards
    for (uint64_t j = 0; j < upto; j++) {
      ix->id_to_idx[ids[j]] = ix->n_rows + j;
      ix->idx_to_id.push_back(ids[j]);
      ix->idx_live.push_back(1);
    }
    ix->n_rows += upto;
    ix->live += upto;
    if (inserted) *inserted = upto;
  };
  if (g->mode == VDB_SHARD_REPLICA) {
    // all replicas or none: rows a device took while another failed (out of memory, a lost device) cannot be taken back
    // (node ids are insertion order), so a partial failure leaves replicas that differ — the handle refuses further work
    // instead of answering from diverged copies
    std::vector<uint64_t> before(g->shards.size());
    for (size_t s = 0; s < g->shards.size(); s++) before[s] = g->shards[s]->n_rows;
    int32_t rc = for_each_shard(g, [&](size_t s) { return child_insert(g->shards[s], ids, vecs, m, kind, max_batch); });
    if (rc != VDB_OK) {
      const std::string why = vdb_hip_last_error();
      for (size_t s = 0; s < g->shards.size(); s++) g->broken |= g->shards[s]->n_rows != before[s];
      return g->broken ? fail(rc, why + " — " + kBrokenMsg) : rc;
    }
    commit(m);
    return VDB_OK;
  }
  // range shards: global row = insertion order; consecutive rows of one shard travel together
  uint64_t j = 0;
  while (j < m) {
    const size_t s = shard_of_row(g, ix->n_rows + j);
    uint64_t e = j + 1;
    while (e < m && shard_of_row(g, ix->n_rows + e) == s) e++;
    int32_t rc = child_insert(g->shards[s], ids + j, vecs + (size_t)j * ix->dim, e - j, kind, max_batch);
    if (rc != VDB_OK) {
      commit(j);
      return rc;
    }
    j = e;
  }
  commit(m);
  return VDB_OK;
}

int32_t group_remove(vdb_hip_index* ix, uint64_t id, int32_t* removed) {
  ShardGroup* g = ix->group;
  std::lock_guard<vdb::IndexMutex> lk(ix->mu);
  if (removed) *removed = 0;
  if (g->broken) return fail(VDB_ERR_STATE, kBrokenMsg);
  auto it = ix->id_to_idx.find(id);
  if (it == ix->id_to_idx.end()) return VDB_OK;
  const uint64_t row = it->second;
  int32_t rc;
  if (g->mode == VDB_SHARD_REPLICA) {
    rc = for_each_shard(g, [&](size_t s) { return vdb_hip_index_remove(g->shards[s], id, nullptr); }, false);
  } else {
    rc = vdb_hip_index_remove(g->shards[shard_of_row(g, row)], id, nullptr);
  }
  if (rc < 0) return rc;
  ix->idx_live[row] = 0;
  ix->id_to_idx.erase(it);
  ix->live--;
  ix->any_dead = true;
  if (removed) *removed = 1;
  return VDB_OK;
}

int32_t group_for_all(vdb_hip_index* ix, int op, uint32_t arg) {
  ShardGroup* g = ix->group;
  std::lock_guard<vdb::IndexMutex> lk(ix->mu);
  if (g->broken) return fail(VDB_ERR_STATE, kBrokenMsg);
  if (op == 3 && g->mode == VDB_SHARD_RANGE)
    return fail(VDB_ERR_UNSUPPORTED, "the int8 quantiser is trained on the first rows of ONE index: replicas only");
  const int32_t rc = for_each_shard(g, [&](size_t s) -> int32_t {
    vdb_hip_index* c = g->shards[s];
    switch (op) {
      case 0: return vdb_hip_index_build_graph(c, arg);
      case 1: return vdb_hip_index_enable_bf16(c);
      case 2: return vdb_hip_index_set_storage_mode(c, (int32_t)arg);
      default: return vdb_hip_index_train_quantizer(c, arg);
    }
  });
  // the handle the caller holds answers is_quantizer_trained / search_with_config (search_front.hip): every replica is trained
  // (exclusive lock held; a failure leaves the flag as it was — the replicas that did train simply keep their codes)
  if (rc == VDB_OK && op == 3) ix->quantizer_trained = true;
  return rc;
}

int32_t group_set_option(vdb_hip_index* ix, int32_t option, int64_t value) {
  ShardGroup* g = ix->group;
  std::lock_guard<vdb::IndexMutex> lk(ix->mu);
  return for_each_shard(g, [&](size_t s) -> int32_t { return vdb_hip_index_set_option(g->shards[s], option, value); });
}
vdb_hip_index* group_first_shard(vdb_hip_index* ix) { return ix->group->shards[0]; }

// ---- sharded search ---------------------------------------------------------------------------------
static int32_t ensure_group_comms(ShardGroup* g) {
  if (!g->distinct || !g->comms.empty()) return VDB_OK;
  Rccl* r = rccl();
  if (!r) return fail(VDB_ERR_UNSUPPORTED, "range-sharded search needs RCCL: " + rccl_load_error());
  std::vector<int> devs;
  for (auto* c : g->shards) devs.push_back(c->device);
  g->comms.assign(devs.size(), nullptr);
  ncclResult_t e = r->CommInitAll(g->comms.data(), (int)devs.size(), devs.data());
  if (e != ncclSuccess) {
    g->comms.clear();
    return rccl_fail("ncclCommInitAll", e);
  }
  return VDB_OK;
}

// the per-shard results sit in every child's s_out_ids / s_out_scores / s_out_n and its stream is idle or ordered:
// pack -> exchange -> merge on shard 0's device into (m_ids, m_scores, m_n), enqueued on shard 0's stream
static int32_t group_exchange_merge(ShardGroup* g, uint32_t nq, uint32_t k, bool hib) {
  const size_t S = g->shards.size();
  const size_t chunk = (size_t)nq * k * 12;
  int32_t rc = ensure_group_comms(g);
  if (rc != VDB_OK) return rc;
  for (size_t s = 0; s < S; s++) {
    vdb_hip_index* c = g->shards[s];
    VDB_HIP(hipSetDevice(c->device));
    // co-located shards only ever gather on shard 0; RCCL delivers to everybody
    if (g->distinct || s == 0) {
      if (g->gath[s].reserve(S * chunk, false, c->stream) != hipSuccess) return fail(VDB_ERR_OOM, "gather buffer");
    } else if (g->gath[s].reserve(chunk, false, c->stream) != hipSuccess) {
      return fail(VDB_ERR_OOM, "gather buffer");
    }
    unsigned char* mine = g->gath[s].as<unsigned char>() + ((g->distinct || s == 0) ? s * chunk : 0);
    rc = launch_pack(c->s_out_ids.as<uint64_t>(), c->s_out_scores.as<float>(), c->s_out_n.as<uint32_t>(),
                     reinterpret_cast<uint32_t*>(mine), nq, k, c->stream);
    if (rc != VDB_OK) return rc;
    if (!g->distinct) VDB_HIP(hipEventRecord(g->ev[s], c->stream));
  }
  vdb_hip_index* c0 = g->shards[0];
  if (g->distinct) {
    Rccl* r = rccl();
    VDB_NCCL(r->GroupStart());
    for (size_t s = 0; s < S; s++) {
      unsigned char* buf = g->gath[s].as<unsigned char>();
      ncclResult_t e = r->AllGather(buf + s * chunk, buf, chunk, ncclChar, g->comms[s], g->shards[s]->stream);
      if (e != ncclSuccess) {
        (void)r->GroupEnd();
        return rccl_fail("ncclAllGather", e);
      }
    }
    VDB_NCCL(r->GroupEnd());
  } else {
    VDB_HIP(hipSetDevice(c0->device));
    for (size_t s = 1; s < S; s++) {
      VDB_HIP(hipStreamWaitEvent(c0->stream, g->ev[s], 0));
      VDB_HIP(hipMemcpyAsync(g->gath[0].as<unsigned char>() + s * chunk, g->gath[s].p, chunk, hipMemcpyDefault, c0->stream));
    }
  }
  VDB_HIP(hipSetDevice(c0->device));
  const size_t kk = std::max<uint32_t>(k, 1);
  if (g->m_ids.reserve((size_t)nq * kk * 8, false, c0->stream) != hipSuccess ||
      g->m_scores.reserve((size_t)nq * kk * 4, false, c0->stream) != hipSuccess ||
      g->m_n.reserve((size_t)nq * 4, false, c0->stream) != hipSuccess)
    return fail(VDB_ERR_OOM, "merge buffers");
  return launch_shard_merge(hib, g->gath[0].as<uint32_t>(), g->m_ids.as<uint64_t>(), g->m_scores.as<float>(),
                            g->m_n.as<uint32_t>(), (uint32_t)S, nq, k, c0->stream);
}

static void query_slice(uint32_t nq, size_t r, size_t world, uint32_t* lo, uint32_t* hi) {
  const uint32_t base = nq / (uint32_t)world, rem = nq % (uint32_t)world;
  *lo = (uint32_t)r * base + std::min<uint32_t>((uint32_t)r, rem);
  *hi = *lo + base + (r < rem ? 1u : 0u);
}

// search_with_quality's size switch (search.rs:75-77) is decided on the whole index, not per shard
static int32_t resolve_mode(const vdb_hip_index* ix, int32_t mode) {
  if (mode == VDB_SEARCH_AUTO) return ix->live <= 100 ? VDB_SEARCH_BRUTE : VDB_SEARCH_HNSW;
  return mode;
}

int32_t group_search_host(vdb_hip_index* ix, const float* queries, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode,
                          uint32_t rerank_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
  ShardGroup* g = ix->group;
  if (nq == 0) return VDB_OK;
  std::lock_guard<vdb::IndexMutex> lk(ix->mu);
  if (g->broken) return fail(VDB_ERR_STATE, kBrokenMsg);
  const size_t S = g->shards.size();
  if (g->mode == VDB_SHARD_REPLICA) {
    return for_each_shard(g, [&](size_t s) -> int32_t {
      uint32_t lo, hi;
      query_slice(nq, s, S, &lo, &hi);
      if (hi == lo) return VDB_OK;
      const float* q = queries + (size_t)lo * ix->dim;
      // (mode AND rerank_k travel as they are: with VDB_SEARCH_HNSW_INT8 rerank_k is the call's oversampling ratio —
      // search_with_config — not a rerank depth of the AUTO mode)
      return search_batch_host(g->shards[s], q, hi - lo, k, ef, mode, rerank_k, out_ids + (size_t)lo * k, out_scores + (size_t)lo * k,
                               out_n + lo);
    });
  }
  const int32_t m = resolve_mode(ix, mode);
  if (m == VDB_SEARCH_HNSW_INT8) return fail(VDB_ERR_UNSUPPORTED, "int8 traversal: replicas only");
  int32_t rc = for_each_shard(g, [&](size_t s) -> int32_t {
    vdb_hip_index* c = g->shards[s];
    std::lock_guard<vdb::IndexMutex> cl(c->mu);
    VDB_HIP(hipSetDevice(c->device));
    std::vector<uint32_t> hn(nq);
    if (c->n_rows == 0) {  // a shard no row has reached yet contributes nothing
      const int32_t ro = reserve_out(c, nq, std::max<uint32_t>(k, 1), c->stream);
      if (ro != VDB_OK) return ro;
      VDB_HIP(hipMemsetAsync(c->s_out_n.p, 0, (size_t)nq * 4, c->stream));
      return VDB_OK;
    }
    return search_to_device(c, queries, nq, k, ef, m, rerank_k, hn.data());
  });
  if (rc != VDB_OK) return rc;
  if (k == 0) {
    std::memset(out_n, 0, (size_t)nq * 4);
    return VDB_OK;
  }
  // rerank results are raw scores in the metric's order, like the exact modes
  rc = group_exchange_merge(g, nq, k, mode_higher_is_better(ix->metric, rerank_k ? VDB_SEARCH_BRUTE : m));
  if (rc != VDB_OK) return rc;
  vdb_hip_index* c0 = g->shards[0];
  VDB_HIP(hipMemcpyAsync(out_ids, g->m_ids.p, (size_t)nq * k * 8, hipMemcpyDeviceToHost, c0->stream));
  VDB_HIP(hipMemcpyAsync(out_scores, g->m_scores.p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, c0->stream));
  VDB_HIP(hipMemcpyAsync(out_n, g->m_n.p, (size_t)nq * 4, hipMemcpyDeviceToHost, c0->stream));
  VDB_HIP(hipStreamSynchronize(c0->stream));
  if (g->distinct)
    for (size_t s = 1; s < S; s++) {  // the collective is complete everywhere before the scratch is reused
      VDB_HIP(hipSetDevice(g->shards[s]->device));
      VDB_HIP(hipStreamSynchronize(g->shards[s]->stream));
    }
  return VDB_OK;
}

// device-resident variant on a multi-device handle: pointers live on devices[0]; the call returns when the work is
// complete (the exchange spans devices, so "enqueue only" cannot be kept — documented in velesdb_hip.h)
int32_t group_search_dev(vdb_hip_index* ix, const float* d_q, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode,
                         uint64_t* d_ids, float* d_scores, uint32_t* d_n, hipStream_t st) {
  ShardGroup* g = ix->group;
  if (nq == 0) return VDB_OK;
  std::lock_guard<vdb::IndexMutex> lk(ix->mu);
  if (g->broken) return fail(VDB_ERR_STATE, kBrokenMsg);
  const size_t S = g->shards.size();
  VDB_HIP(hipSetDevice(g->shards[0]->device));
  VDB_HIP(hipStreamSynchronize(st));  // the queries are complete
  const bool replica = g->mode == VDB_SHARD_REPLICA;
  const int32_t m = replica ? mode : resolve_mode(ix, mode);
  if (!replica && m == VDB_SEARCH_HNSW_INT8) return fail(VDB_ERR_UNSUPPORTED, "int8 traversal: replicas only");
  const size_t kk = std::max<uint32_t>(k, 1);
  int32_t rc = for_each_shard(g, [&](size_t s) -> int32_t {
    vdb_hip_index* c = g->shards[s];
    std::lock_guard<vdb::IndexMutex> cl(c->mu);
    VDB_HIP(hipSetDevice(c->device));
    uint32_t lo = 0, hi = nq;
    if (replica) query_slice(nq, s, S, &lo, &hi);
    const uint32_t n = hi - lo;
    if (n == 0) return VDB_OK;
    if (c->s_queries.reserve((size_t)n * c->row_stride * 4, false, c->stream) != hipSuccess) return fail(VDB_ERR_OOM, "search scratch");
    const int32_t ro = reserve_out(c, n, kk, c->stream);
    if (ro != VDB_OK) return ro;
    if (c->row_stride != c->dim) VDB_HIP(hipMemsetAsync(c->s_queries.p, 0, (size_t)n * c->row_stride * 4, c->stream));
    VDB_HIP(hipMemcpy2DAsync(c->s_queries.p, c->row_stride * 4, d_q + (size_t)lo * ix->dim, (size_t)ix->dim * 4,
                             (size_t)ix->dim * 4, n, hipMemcpyDefault, c->stream));
    if (c->n_rows == 0) {
      VDB_HIP(hipMemsetAsync(c->s_out_n.p, 0, (size_t)n * 4, c->stream));
    } else {
      int32_t r1 = search_dev(c, c->s_queries.as<float>(), c->row_stride, n, k, ef, m, c->s_out_ids.as<uint64_t>(),
                              c->s_out_scores.as<float>(), c->s_out_n.as<uint32_t>(), c->stream);
      if (r1 != VDB_OK) return r1;
    }
    if (replica) {  // straight into the caller's buffers
      if (k) {
        VDB_HIP(hipMemcpyAsync(d_ids + (size_t)lo * k, c->s_out_ids.p, (size_t)n * k * 8, hipMemcpyDefault, c->stream));
        VDB_HIP(hipMemcpyAsync(d_scores + (size_t)lo * k, c->s_out_scores.p, (size_t)n * k * 4, hipMemcpyDefault, c->stream));
      }
      VDB_HIP(hipMemcpyAsync(d_n + lo, c->s_out_n.p, (size_t)n * 4, hipMemcpyDefault, c->stream));
    }
    VDB_HIP(hipStreamSynchronize(c->stream));
    return VDB_OK;
  });
  if (rc != VDB_OK || replica) return rc;
  vdb_hip_index* c0 = g->shards[0];
  VDB_HIP(hipSetDevice(c0->device));
  if (k == 0) {
    VDB_HIP(hipMemsetAsync(d_n, 0, (size_t)nq * 4, st));
    return VDB_OK;
  }
  rc = group_exchange_merge(g, nq, k, mode_higher_is_better(ix->metric, m));
  if (rc != VDB_OK) return rc;
  VDB_HIP(hipMemcpyAsync(d_ids, g->m_ids.p, (size_t)nq * k * 8, hipMemcpyDeviceToDevice, c0->stream));
  VDB_HIP(hipMemcpyAsync(d_scores, g->m_scores.p, (size_t)nq * k * 4, hipMemcpyDeviceToDevice, c0->stream));
  VDB_HIP(hipMemcpyAsync(d_n, g->m_n.p, (size_t)nq * 4, hipMemcpyDeviceToDevice, c0->stream));
  for (size_t s = 0; s < S; s++) {
    VDB_HIP(hipSetDevice(g->shards[s]->device));
    VDB_HIP(hipStreamSynchronize(g->shards[s]->stream));
  }
  VDB_HIP(hipSetDevice(c0->device));
  return VDB_OK;
}

// ---- one process per GPU -------------------------------------------------------------------------------
struct ProcComm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  DevBuf gath;  // [world][nq][k] records
};

void proc_comm_free(ProcComm* p) {
  if (!p) return;
  Rccl* r = rccl();
  if (r && p->comm) (void)r->CommDestroy(p->comm);
  p->gath.release();
  delete p;
}

int32_t pcomm_exchange_merge(vdb_hip_index* ix, uint32_t nq, uint32_t k, bool hib, uint64_t* d_ids, float* d_scores,
                             uint32_t* d_n, hipStream_t st) {
  ProcComm* p = ix->pcomm;
  if (nq == 0 || k == 0) return VDB_OK;
  Rccl* r = rccl();
  if (!r) return fail(VDB_ERR_UNSUPPORTED, "RCCL is not available: " + rccl_load_error());
  const size_t chunk = (size_t)nq * k * 12;
  if (p->gath.cap < (size_t)p->world * chunk) {
    // growing frees the old buffer: an earlier collective enqueued on another stream must have left it
    VDB_HIP(hipDeviceSynchronize());
    if (p->gath.reserve((size_t)p->world * chunk, false, st) != hipSuccess) return fail(VDB_ERR_OOM, "gather buffer");
  }
  unsigned char* buf = p->gath.as<unsigned char>();
  int32_t rc = launch_pack(d_ids, d_scores, d_n, reinterpret_cast<uint32_t*>(buf + (size_t)p->rank * chunk), nq, k, st);
  if (rc != VDB_OK) return rc;
  VDB_NCCL(r->AllGather(buf + (size_t)p->rank * chunk, buf, chunk, ncclChar, p->comm, st));
  return launch_shard_merge(hib, reinterpret_cast<const uint32_t*>(buf), d_ids, d_scores, d_n, (uint32_t)p->world, nq, k, st);
}

}  // namespace vdb

using namespace vdb;

extern "C" {

int32_t vdb_hip_comm_unique_id(uint8_t* id) {
  return guarded([&]() -> int32_t {
    if (!id) return fail(VDB_ERR_INVALID_ARG, "null argument");
    static_assert(VDB_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    Rccl* r = rccl();
    if (!r) return fail(VDB_ERR_UNSUPPORTED, "RCCL is not available: " + rccl_load_error());
    ncclUniqueId u;
    VDB_NCCL(r->GetUniqueId(&u));
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return VDB_OK;
  });
}

int32_t vdb_hip_index_join_group(vdb_hip_index* ix, const uint8_t* id, int32_t rank, int32_t world) {
  return guarded([&]() -> int32_t {
    if (!ix || !id) return fail(VDB_ERR_INVALID_ARG, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(VDB_ERR_INVALID_ARG, "bad rank / world");
    if (ix->group) return fail(VDB_ERR_STATE, "a multi-device handle cannot join a process group");
    std::lock_guard<vdb::IndexMutex> lk(ix->mu);
    if (ix->pcomm) return fail(VDB_ERR_STATE, "already member of a process group");
    Rccl* r = rccl();
    if (!r) return fail(VDB_ERR_UNSUPPORTED, "RCCL is not available: " + rccl_load_error());
    VDB_HIP(hipSetDevice(ix->device));
    std::unique_ptr<ProcComm, void (*)(ProcComm*)> p(new ProcComm(), proc_comm_free);
    p->rank = rank;
    p->world = world;
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    VDB_NCCL(r->CommInitRank(&p->comm, world, u, rank));
    ix->pcomm = p.release();
    return VDB_OK;
  });
}

int32_t vdb_hip_index_shard_info(vdb_hip_index* ix, int32_t* n_shards, int32_t* shard_mode, int32_t* rank, int32_t* world,
                                 int32_t* transport) {
  return guarded([&]() -> int32_t {
    if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
    std::shared_lock<vdb::IndexMutex> lk(ix->mu);
    if (n_shards) *n_shards = ix->group ? (int32_t)ix->group->shards.size() : 1;
    if (shard_mode) *shard_mode = ix->group ? ix->group->mode : (ix->pcomm ? VDB_SHARD_RANGE : VDB_SHARD_REPLICA);
    if (rank) *rank = ix->pcomm ? ix->pcomm->rank : 0;
    if (world) *world = ix->pcomm ? ix->pcomm->world : 1;
    if (transport)
      *transport = ix->pcomm ? 1 : (ix->group && ix->group->mode == VDB_SHARD_RANGE ? (ix->group->distinct ? 1 : 2) : 0);
    return VDB_OK;
  });
}

}  // extern "C"
