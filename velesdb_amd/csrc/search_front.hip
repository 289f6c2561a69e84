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
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <chrono>
#include <cstring>

#include "vdb_index.hpp"

namespace vdb {

void note_last_context(vdb_hip_index* handle, vdb_hip_index* ctx);  // index.hip

constexpr uint32_t kCombineMaxCall = 64;  // larger calls fill the chip by themselves: they launch alone

struct CombineReq {
  const float* queries;
  uint32_t nq, k, ef;
  int32_t mode;
  uint32_t rerank_k;
  uint64_t* out_ids;
  float* out_scores;
  uint32_t* out_n;
  enum { kQueued, kTaken } state = kQueued;           // (under Combiner::mu) still in the queue / in some leader's batch
  enum : uint32_t { kWait = 0, kDone = 1, kLead = 2 };
  std::atomic<uint32_t> word{kWait};                  // what its sleeping caller waits on
  int32_t rc = VDB_OK;
  std::string err;
  vdb_hip_index* served_by = nullptr;
  bool same_shape(const CombineReq& o) const { return k == o.k && ef == o.ef && mode == o.mode && rerank_k == o.rerank_k; }
};

struct Combiner {
  std::mutex mu;
  std::deque<CombineReq*> queue;
  int leaders = 0;  // batches in flight
  uint64_t arrivals = 0;          // calls ever queued (a waiting leader watches it move)
  uint32_t last_batch_calls = 1;  // calls the batch that finished last carried: > 1 = callers are arriving together
  uint64_t last_batch_done_at_arrival = 0;  // `arrivals` when that batch finished (its callers re-arrive behind this mark)
  uint64_t launches = 0, calls = 0, queries = 0, max_batch = 0;
};
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
    rc = pcomm_exchange_merge(ix, nq, k, mode_higher_is_better(ix->metric, rerank_k ? VDB_SEARCH_BRUTE : m), ix->s_out_ids.as<uint64_t>(),
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
  }
}

// Sleeping callers wait on a word of their OWN request (futex): a finished batch wakes exactly its callers, and a freed leader
// slot wakes exactly one queued caller.  (One condition variable for everybody was the first version: every completion woke
// every sleeper into a fight for one mutex — with 64 callers on the box's 16 cores the stragglers came back after the next
// launch had left and the callers split into groups that took turns.)
static void futex_wait(std::atomic<uint32_t>* w, uint32_t expect) {
  syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, expect, nullptr, nullptr, 0);
}
static void futex_wake_one(std::atomic<uint32_t>* w) {
  syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
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

static int32_t search_combined(vdb_hip_index* handle, Combiner* cb, CombineReq& me) {
  const uint32_t max_batch = (uint32_t)opt_value(handle, VDB_OPT_COMBINE_MAX_BATCH);
  const uint32_t window_us = (uint32_t)opt_value(handle, VDB_OPT_COMBINE_WINDOW_US);
  bool lead = false;
  {
    std::lock_guard<std::mutex> lk(cb->mu);
    cb->arrivals++;
    if (cb->leaders < leader_limit(handle, me)) {
      cb->leaders++;
      me.state = CombineReq::kTaken;
      lead = true;
    } else {
      cb->queue.push_back(&me);
    }
  }
  if (!lead) {
    uint32_t w;
    while ((w = me.word.load(std::memory_order_acquire)) == CombineReq::kWait) futex_wait(&me.word, CombineReq::kWait);
    if (w == CombineReq::kLead) lead = true;  // a leader slot came free while this call was queued: it was handed over, taken and counted
  }
  if (lead) {
    std::unique_lock<std::mutex> lk(cb->mu);
    // my request first (the shape of the batch is mine), then every queued request of the same shape while the batch has room
    // (room for a full batch up front: nothing below can throw once other callers' requests are in it; without the room — out of
    // host memory — the leader runs alone)
    std::vector<CombineReq*> batch;
    size_t room = 1;
    try {
      batch.reserve(std::max<uint32_t>(max_batch, 1u));
      room = batch.capacity();
    } catch (const std::bad_alloc&) {
    }
    CombineReq* alone[1] = {&me};
    if (room > 1) batch.push_back(&me);
    uint32_t total = me.nq;
    auto gather = [&] {
      if (room <= 1) return;
      for (auto it = cb->queue.begin(); it != cb->queue.end();) {
        CombineReq* r = *it;
        if (batch.size() < room && r->state == CombineReq::kQueued && r->same_shape(me) && total + r->nq <= max_batch) {
          r->state = CombineReq::kTaken;
          batch.push_back(r);
          total += r->nq;
          it = cb->queue.erase(it);
        } else {
          ++it;
        }
      }
    };
    gather();
    // The callers of a finished batch come back within tens of microseconds of each other (as fast as the host wakes their
    // threads).  A leader that launched the moment it arrived would take the one or two that beat it to the queue and leave the
    // rest to the next launch: the callers split into groups that take turns, every call waits for the other group's launch
    // before its own, and each launch carries half of what it could (64 callers on the exact sweep: 1.33 ms per call where one
    // batch of 64 takes 0.6).  So a leader with EVIDENCE of company — the batch that finished last carried several calls —
    // waits for as many arrivals as that batch had callers, at most COMBINE_WINDOW_US.  A lone caller has no such evidence
    // (the batch before it was its own) and never waits; callers that went away cost the ones that stayed one window.
    if (window_us && total < max_batch && cb->last_batch_calls > 1) {
      using clk = std::chrono::steady_clock;
      const auto t_cap = clk::now() + std::chrono::microseconds(window_us);
      const uint64_t want = cb->last_batch_done_at_arrival + cb->last_batch_calls;  // everybody of that batch is back
      uint64_t seen = cb->arrivals;
      while (total < max_batch && cb->arrivals < want) {
        lk.unlock();
        std::this_thread::yield();
        lk.lock();
        if (cb->arrivals != seen) {
          seen = cb->arrivals;
          gather();
        }
        if (clk::now() >= t_cap) break;
      }
      gather();
    }
    const size_t n_calls = room > 1 ? batch.size() : 1;
    cb->launches++;
    cb->calls += n_calls;
    cb->queries += total;
    cb->max_batch = std::max<uint64_t>(cb->max_batch, total);
    lk.unlock();
    run_batch(handle, room > 1 ? batch.data() : alone, n_calls);
    lk.lock();
    cb->last_batch_calls = (uint32_t)n_calls;
    cb->last_batch_done_at_arrival = cb->arrivals;
    cb->leaders--;
    // the freed slot goes to the first queued call that may lead (it takes the others of its shape with it)
    CombineReq* next = nullptr;
    for (auto it = cb->queue.begin(); it != cb->queue.end(); ++it)
      if ((*it)->state == CombineReq::kQueued && cb->leaders < leader_limit(handle, **it)) {
        next = *it;
        cb->queue.erase(it);
        next->state = CombineReq::kTaken;
        cb->leaders++;
        break;
      }
    lk.unlock();
    // (a request is not touched after its word is set: its caller may be gone the next instant)
    for (size_t i = 1; i < batch.size(); i++) {
      std::atomic<uint32_t>* w = &batch[i]->word;
      w->store(CombineReq::kDone, std::memory_order_release);
      futex_wake_one(w);
    }
    if (next) {
      std::atomic<uint32_t>* w = &next->word;
      w->store(CombineReq::kLead, std::memory_order_release);
      futex_wake_one(w);
    }
  }
  if (me.served_by) note_last_context(handle, me.served_by);
  if (me.rc != VDB_OK) set_last_error(me.err);
  return me.rc;
}

// HnswIndex::search_batch_parallel (batch.rs:159-197) / search_with_quality / search_brute_force
static int32_t search_batch_host(vdb_hip_index* ix, const float* queries, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode,
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
    return search_combined(ix, ix->combiner, me);
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
    ef = std::max(ef ? ef : std::max<uint32_t>(128, 4 * k), k);  // (SearchQuality rules of vdb_hip_index_search)
    const uint32_t draws = (num_probes > 1 && ix->graph_nodes > 10) ? std::min<uint32_t>(num_probes, 4) - 1 : 0;
    if (draws && ef < 4) return fail(VDB_ERR_UNSUPPORTED, "search_multi_entry: ef_search >= 4 with several entry points");
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
    rc = search_staged(ix, nq, k, ef, VDB_SEARCH_HNSW, 0, d_extra);
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
