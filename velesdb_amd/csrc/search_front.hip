// search_front.hip — the host-pointer search entry points of the C ABI (VectorIndex::search, HnswIndex::search_with_quality /
// search_brute_force / search_batch_parallel / search_with_rerank) and the COMBINING FRONT in front of them.
//
// The reference's calling pattern: many threads, each searching ONE query under a read lock (index/hnsw/index/search.rs:80; its
// stress tests index/hnsw/native/tests.rs:264-416; the server calls collection.search once per request,
// velesdb-server/src/handlers/search.rs:34-73).  A GPU walk or sweep of one query leaves the chip empty (a graph walk occupies one
// CU of 256; a one-query sweep pays a whole corpus pass) and every call pays launch + copies + a synchronisation — one launch per
// caller tops out at a few thousand calls per second whatever the number of threads.  What the hardware needs is callers that
// arrive together SHARING a launch: 64 graph walks in one launch cost 1.4 ms, not 64 x 1.1 ms.
//
// Protocol (flat combining; per handle, state in vdb::Combiner):
//   * a call of <= kCombineMaxCall queries becomes a request in the handle's queue;
//   * a caller that finds fewer batches running than may run beside each other (leader_limit) becomes a LEADER: it takes its own
//     request and every queued request with the same (k, ef, mode, rerank_k) up to COMBINE_MAX_BATCH queries, leases a search
//     context (shared lock on the handle, like any search), stages all the queries in the context's pinned buffer, issues ONE
//     search, and hands every request its slice of the pinned result block; then it wakes the callers of its batch and hands its
//     slot to the first call still queued;
//   * everybody else sleeps on its own request until it is done — or until a leader slot is handed to it.
// A lone caller is its own leader at once (no timer in its way); under load the requests that arrive while the launches in front
// of them run form the next batch, so the batch size follows the load by itself; a leader that has evidence of company waits for
// the callers of the batch that just finished to come back first (at most COMBINE_WINDOW_US; see search_combined).
// Bits: every kernel behind search_dev answers a query independently of the batch it sits in (declared arithmetic per metric and
// mode; the tests compare single-query and batched calls bit for bit), so combining never changes a result.
#include <cstring>

#include "vdb_combiner.hpp"
#include "vdb_index.hpp"

namespace vdb {

void note_last_context(vdb_hip_index* handle, vdb_hip_index* ctx);  // index.hip
void note_last_kernels(uint32_t mask);                               // index.hip

void combiner_free(Combiner* c) { delete c; }

// one search of nq host queries on a leased context of `handle`; results land in ctx->h_out (the layout of reserve_out) and
// `deliver(ctx)` copies them out while the lease is still held
template <class Stage, class Deliver>
static int32_t run_search(vdb_hip_index* handle, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode, uint32_t rerank_k, Stage&& stage,
                          Deliver&& deliver) {
  // searches share the handle (search.rs:80 takes the read lock); a member of a process group searches collectively on one
  // gather buffer: those calls stay exclusive
  std::shared_lock<IndexMutex> rd(handle->mu, std::defer_lock);
  std::unique_lock<IndexMutex> wr(handle->mu, std::defer_lock);
  if (handle->pcomm) wr.lock(); else rd.lock();
  CtxLease lease(handle);  // this search's scratch + stream: the handle itself, or one of its search contexts when it is busy
  if (lease.rc != VDB_OK) return lease.rc;
  vdb_hip_index* ix = lease.ctx;
  VDB_ENTER_SHARED(ix);
  int32_t rc = stage(ix);
  if (rc != VDB_OK) return rc;
  if (ix->pcomm && k) {  // member of a process group: every rank ends with the global top-k
    hipStream_t st = ix->stream;
    const int32_t m = (mode == VDB_SEARCH_AUTO) ? (ix->live <= 100 ? VDB_SEARCH_BRUTE : VDB_SEARCH_HNSW) : mode;
    rc = pcomm_exchange_merge(ix, nq, k, mode_higher_is_better(ix->metric, (rerank_k && mode != VDB_SEARCH_HNSW_INT8) ? VDB_SEARCH_BRUTE : m), ix->s_out_ids.as<uint64_t>(),
                              ix->s_out_scores.as<float>(), ix->s_out_n.as<uint32_t>(), st);
    if (rc != VDB_OK) return rc;
    VDB_HIP(hipMemcpyAsync(ix->h_out.p, ix->s_out.p, ix->s_out_bytes, hipMemcpyDeviceToHost, st));
    VDB_HIP(hipStreamSynchronize(st));
  }
  deliver(ix);
  return VDB_OK;
}

// slice [at, at + nq) of a context's pinned result block (a batch of nq_total queries x k results) into a caller's arrays
static void deliver_slice(const vdb_hip_index* ix, uint32_t at, uint32_t nq, uint32_t nq_total, uint32_t k, uint64_t* out_ids,
                          float* out_scores, uint32_t* out_n) {
  const size_t kk = std::max<uint32_t>(k, 1);
  const unsigned char* h = ix->h_out.as<unsigned char>();
  const size_t o_sc = (size_t)nq_total * kk * 8, o_n = o_sc + (size_t)nq_total * kk * 4;
  if (k) {
    std::memcpy(out_ids, h + (size_t)at * k * 8, (size_t)nq * k * 8);
    std::memcpy(out_scores, h + o_sc + (size_t)at * k * 4, (size_t)nq * k * 4);
  }
  std::memcpy(out_n, h + o_n + (size_t)at * 4, (size_t)nq * 4);
}

static int32_t search_direct(vdb_hip_index* handle, const float* queries, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode,
                             uint32_t rerank_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
  return run_search(
      handle, nq, k, ef, mode, rerank_k,
      [&](vdb_hip_index* ix) { return search_to_device(ix, queries, nq, k, ef, mode, rerank_k, nullptr); },
      [&](vdb_hip_index* ix) { deliver_slice(ix, 0, nq, nq, k, out_ids, out_scores, out_n); });
}

// the leader's part: one launch for `batch` (same shape; batch[0] is the leader's own request)
static void run_batch(vdb_hip_index* handle, CombineReq* const* reqs, size_t n_reqs) {
  struct Span {
    CombineReq* const* b;
    size_t n;
    CombineReq* const* begin() const { return b; }
    CombineReq* const* end() const { return b + n; }
  } batch{reqs, n_reqs};
  uint32_t total = 0;
  for (CombineReq* r : batch) total += r->nq;
  const CombineReq& s = *reqs[0];
  vdb_hip_index* served = nullptr;
  uint32_t served_kernels = 0;
  const int32_t rc = guarded([&]() -> int32_t {
    return run_search(
        handle, total, s.k, s.ef, s.mode, s.rerank_k,
        [&](vdb_hip_index* ix) -> int32_t {
          served = ix;
          uint32_t at = 0;
          for (CombineReq* r : batch) {
            const int32_t rs = stage_queries(ix, r->queries, at, r->nq, total);
            if (rs != VDB_OK) return rs;
            at += r->nq;
          }
          return search_staged(ix, total, s.k, s.ef, s.mode, s.rerank_k);
        },
        [&](vdb_hip_index* ix) {
          served_kernels = ix->last_kernels;  // (the context is still leased: a later search on it cannot have overwritten the mask)
          uint32_t at = 0;
          for (CombineReq* r : batch) {
            deliver_slice(ix, at, r->nq, total, s.k, r->out_ids, r->out_scores, r->out_n);
            at += r->nq;
          }
        });
  });
  const std::string err = rc != VDB_OK ? std::string(vdb_hip_last_error()) : std::string();
  for (CombineReq* r : batch) {
    r->rc = rc;
    r->err = err;
    r->served_by = served;
    r->kernels = served_kernels;
  }
}

// how many batches may run beside each other when this request leads one: graph walks are latency-bound (one CU per query in a
// small call: two launches overlap for free), sweeps are bandwidth-bound (a second launch beside the first only halves both
// batches: 16 callers on the 1 M exact sweep 26 K q/s with one batch in flight, 15 K with two)
static int leader_limit(const vdb_hip_index* handle, const CombineReq& r) {
  const int64_t o = opt_value(handle, VDB_OPT_COMBINE_INFLIGHT);
  if (o > 0) return (int)o;
  const bool walk = r.mode == VDB_SEARCH_HNSW || r.mode == VDB_SEARCH_HNSW_INT8 || r.mode == VDB_SEARCH_AUTO;
  return walk ? 2 : 1;
}

// the protocol (vdb_combiner.hpp) over a handle: its options, one launch per batch
struct HandleFront {
  vdb_hip_index* handle;
  uint32_t max_batch() const { return (uint32_t)opt_value(handle, VDB_OPT_COMBINE_MAX_BATCH); }
  uint32_t window_us() const { return (uint32_t)opt_value(handle, VDB_OPT_COMBINE_WINDOW_US); }
  int leader_limit(const CombineReq& r) const { return vdb::leader_limit(handle, r); }
  void run_batch(CombineReq* const* reqs, size_t n) { vdb::run_batch(handle, reqs, n); }
  void finish(CombineReq& me) {
    if (me.served_by) {
      note_last_context(handle, me.served_by);
      if (me.rc == VDB_OK) note_last_kernels(me.kernels);  // this thread's diagnostics describe ITS search, whoever ran it
    }
    if (me.rc != VDB_OK) set_last_error(me.err);
  }
};

// HnswIndex::search_batch_parallel (batch.rs:159-197) / search_with_quality / search_brute_force
int32_t search_batch_host(vdb_hip_index* ix, const float* queries, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode,
                          uint32_t rerank_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
  if (!ix || (nq && (!queries || !out_n)) || (nq && k && (!out_ids || !out_scores)))
    return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (nq == 0) return VDB_OK;
  if (ix->group) return group_search_host(ix, queries, nq, k, ef, mode, rerank_k, out_ids, out_scores, out_n);
  // (options are read without the handle's lock: set_option on a handle that is being searched is a benign race on an int)
  if (ix->combiner && !ix->pcomm && nq <= kCombineMaxCall && opt_value(ix, VDB_OPT_COMBINE_MAX_BATCH) >= (int64_t)nq &&
      opt_value(ix, VDB_OPT_COMBINE_MAX_BATCH) > 1) {
    CombineReq me;
    me.queries = queries;
    me.nq = nq;
    me.k = k;
    me.ef = ef;
    me.mode = mode;
    me.rerank_k = rerank_k;
    me.out_ids = out_ids;
    me.out_scores = out_scores;
    me.out_n = out_n;
    HandleFront front{ix};
    return search_combined(front, ix->combiner, me);
  }
  return search_direct(ix, queries, nq, k, ef, mode, rerank_k, out_ids, out_scores, out_n);
}

Combiner* combiner_new() { return new Combiner(); }

}  // namespace vdb

using namespace vdb;

extern "C" {

int32_t vdb_hip_index_search_batch(vdb_hip_index* ix, const float* queries, uint32_t nq, uint32_t k, uint32_t ef,
                                   int32_t mode, uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
  return vdb::guarded([&]() -> int32_t {
  return search_batch_host(ix, queries, nq, k, ef, mode, 0, out_ids, out_scores, out_n);
  });
}

// HnswIndex::search_with_rerank / search_with_rerank_quality — search.rs:118-160,297-350
int32_t vdb_hip_index_search_rerank(vdb_hip_index* ix, const float* queries, uint32_t nq, uint32_t k, uint32_t rerank_k,
                                    uint32_t ef, uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
  return vdb::guarded([&]() -> int32_t {
  if (rerank_k == 0) return fail(VDB_ERR_INVALID_ARG, "rerank_k must be > 0");
  return search_batch_host(ix, queries, nq, k, ef, VDB_SEARCH_AUTO, rerank_k, out_ids, out_scores, out_n);
  });
}

// DualPrecisionHnsw::is_quantizer_trained — native/dual_precision.rs:117-120
int32_t vdb_hip_index_quantizer_trained(const vdb_hip_index* ix, int32_t* trained) {
  return vdb::guarded([&]() -> int32_t {
    if (!ix || !trained) return fail(VDB_ERR_INVALID_ARG, "null argument");
    // (written under the exclusive lock by train_quantizer — on a replica group by group_for_all once every replica is trained)
    std::shared_lock<IndexMutex> rd(const_cast<vdb_hip_index*>(ix)->mu);
    *trained = ix->quantizer_trained ? 1 : 0;
    return VDB_OK;
  });
}

// DualPrecisionHnsw::search_with_config — native/dual_precision.rs:259-278: the int8 traversal (+ exact f32 re-scoring of the
// k * oversampling_ratio best) only with a trained quantiser, use_int8_traversal and at least min_index_size vectors; otherwise
// the plain f32 graph search.  The configuration travels WITH the call, as in the reference.
int32_t vdb_hip_index_search_with_config(vdb_hip_index* ix, const float* queries, uint32_t nq, uint32_t k, uint32_t ef_search,
                                         uint32_t oversampling_ratio, int32_t use_int8_traversal, uint64_t min_index_size,
                                         uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
  return vdb::guarded([&]() -> int32_t {
    if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
    // the same bound vdb_hip_set_int8_oversampling holds the handle's option to: k * ratio sizes nbmax, ef and the LDS list of the walk
    if (oversampling_ratio == 0 || oversampling_ratio > 64)
      return fail(VDB_ERR_INVALID_ARG, "oversampling_ratio must be in 1..64 (DualPrecisionConfig's default is 4)");
    // (n_rows: NativeHnsw::len() counts inserted vectors; the state is read under the shared lock, the search takes its own)
    bool int8;
    {
      std::shared_lock<IndexMutex> rd(ix->mu);
      int8 = ix->quantizer_trained && use_int8_traversal && ix->n_rows >= min_index_size;
    }
    // The f32 branch is NativeHnsw::search with ITS ef_search (dual_precision.rs:269,274 -> graph.rs:251-270: search_layer(ef_search)
    // as given, results cut to k): VDB_SEARCH_HNSW applies HnswIndex's SearchQuality rules on top (0 = Balanced, max(ef, k)).  The
    // two agree whenever 0 < k <= ef_search; the other calls — at most ef_search results, ef_search = 0 acting as 1 — take the
    // NativeHnsw-level entry point with one entry point (no draw from the graph's stream: graph.rs:303).  A device group has no such
    // entry point: its f32 branch keeps the HnswIndex rule.
    if (!int8 && !ix->group && !ix->pcomm && nq && k && (ef_search == 0 || ef_search < k))
      return vdb_hip_index_search_multi_entry(ix, queries, nq, k, ef_search, 1, out_ids, out_scores, out_n);
    return search_batch_host(ix, queries, nq, k, ef_search, int8 ? VDB_SEARCH_HNSW_INT8 : VDB_SEARCH_HNSW, int8 ? oversampling_ratio : 0, out_ids,
                             out_scores, out_n);
  });
}

// VectorIndex::search — index/mod.rs:58; trait_impl.rs:38-42
int32_t vdb_hip_index_search(vdb_hip_index* ix, const float* query, uint32_t query_len, uint32_t k, uint32_t ef,
                             int32_t mode, uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !query) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (query_len != ix->dim)
    return fail(VDB_ERR_DIM_MISMATCH, "Query dimension mismatch: expected " + std::to_string(ix->dim) + ", got " +
                                          std::to_string(query_len));
  return vdb_hip_index_search_batch(ix, query, 1, k, ef, mode, out_ids, out_scores, out_n);
  });
}

// NativeHnsw::search_multi_entry — index/hnsw/native/graph.rs:288-348.  The extra entry points come out of the graph's own
// xorshift stream (the one that draws insertion levels): the call CHANGES the index (exclusive lock), query i of the batch takes
// draws i (min(num_probes, 4) - 1) .. of it, exactly as nq sequential calls of the reference would.
int32_t vdb_hip_index_search_multi_entry(vdb_hip_index* ix, const float* queries, uint32_t nq, uint32_t k, uint32_t ef, uint32_t num_probes,
                                         uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
  return vdb::guarded([&]() -> int32_t {
    if (!ix || (nq && (!queries || !out_n)) || (nq && k && (!out_ids || !out_scores))) return fail(VDB_ERR_INVALID_ARG, "null argument");
    VDB_NO_GROUP(ix, "search_multi_entry");
    if (nq == 0) return VDB_OK;
    if (ix->pcomm) return fail(VDB_ERR_UNSUPPORTED, "search_multi_entry: not available on a member of a process group");
    std::lock_guard<vdb::IndexMutex> g(ix->mu);
    VDB_ENTER(ix);
    if (!ix->graph_valid) return fail(VDB_ERR_STATE, "HNSW graph not built for all rows (use mode BRUTE or build it)");
    // NativeHnsw level: ef_search goes to search_layer AS GIVEN (graph.rs:343) — no SearchQuality rule, no max(ef, k): a call with
    // ef_search < k returns at most ef_search results (more entry points than ef_search: that many, graph.rs:463-468)
    // (ef_search = 0 is legal there and acts as 1: the entry point is pushed uncut and every later push pops one)
    const uint32_t draws = (num_probes > 1 && ix->graph_nodes > 10) ? std::min<uint32_t>(num_probes, 4) - 1 : 0;
    const uint32_t* d_extra = nullptr;
    if (draws) {
      std::vector<uint32_t> extra((size_t)nq * 3, 0xFFFFFFFFu);
      for (uint32_t q = 0; q < nq; q++)
        for (uint32_t p = 0; p < draws; p++) {
          uint64_t s = ix->rng_state;  // graph.rs:320-334 (no zero-state reseed here)
          s ^= s << 13;
          s ^= s >> 7;
          s ^= s << 17;
          ix->rng_state = s;
          extra[(size_t)q * 3 + p] = (uint32_t)(s % ix->graph_nodes);
        }
      if (ix->s_part_cnt.reserve(extra.size() * 4, false, ix->stream) != hipSuccess) return fail(VDB_ERR_OOM, "entry-point scratch");
      VDB_HIP(hipMemcpyAsync(ix->s_part_cnt.p, extra.data(), extra.size() * 4, hipMemcpyHostToDevice, ix->stream));
      VDB_HIP(hipStreamSynchronize(ix->stream));  // `extra` is host memory
      d_extra = ix->s_part_cnt.as<uint32_t>();
    }
    int32_t rc = stage_queries(ix, queries, 0, nq, nq);
    if (rc != VDB_OK) return rc;
    struct RawEf {  // (exclusive lock held: nobody else reads the flag; reset on every way out, an exception caught by guarded() included)
      vdb_hip_index* h;
      explicit RawEf(vdb_hip_index* x) : h(x) { h->raw_ef = true; }
      ~RawEf() { h->raw_ef = false; }
    };
    {
      RawEf raw(ix);
      rc = search_staged(ix, nq, k, ef, VDB_SEARCH_HNSW, 0, d_extra);
    }
    if (rc != VDB_OK) return rc;
    deliver_slice(ix, 0, nq, nq, k, out_ids, out_scores, out_n);
    return VDB_OK;
  });
}

int32_t vdb_hip_index_combine_stats(vdb_hip_index* ix, uint64_t* launches, uint64_t* calls, uint64_t* queries, uint64_t* max_batch) {
  return vdb::guarded([&]() -> int32_t {
    if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
    uint64_t v[4] = {0, 0, 0, 0};
    auto add = [&](vdb_hip_index* h) {
      if (!h->combiner) return;
      std::lock_guard<std::mutex> lk(h->combiner->mu);
      v[0] += h->combiner->launches;
      v[1] += h->combiner->calls;
      v[2] += h->combiner->queries;
      v[3] = std::max(v[3], h->combiner->max_batch);
    };
    if (ix->group)
      for (size_t s = 0; s < group_size(ix); s++) add(group_shard(ix, s));
    else
      add(ix);
    if (launches) *launches = v[0];
    if (calls) *calls = v[1];
    if (queries) *queries = v[2];
    if (max_batch) *max_batch = v[3];
    return VDB_OK;
  });
}

}  // extern "C"
