// vdb_combiner.hpp — the protocol of the COMBINING FRONT (search_front.hip) with nothing of HIP or of the index in it: requests,
// the per-handle queue, the leader election, the gathering window and the wake-ups.  search_front.hip instantiates it over the
// handle (options, one launch per batch); tests/combiner_model.cpp instantiates the SAME text over a mock launch and runs it
// under ThreadSanitizer with many callers of mixed shapes (tests/test_host_sync_tsan_cpu.py) — the race detector cannot see into a
// process that talks to the GPU driver, so the protocol is kept where it can.
//
// Env (duck-typed):  uint32_t max_batch() const;  uint32_t window_us() const;  int leader_limit(const CombineReq&) const;
//                    void run_batch(CombineReq* const* reqs, size_t n);   // one launch; sets rc / err / served_by of every request
//                    void finish(CombineReq& me);                          // on the caller's own thread, after its request is done
#pragma once
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

struct vdb_hip_index;

namespace vdb {

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
  int32_t rc = 0;  // VDB_OK
  std::string err;
  vdb_hip_index* served_by = nullptr;
  uint32_t kernels = 0;  // what the batch that served it ran (the context's diagnostic mask, read while the leader still held the context)
  uint32_t passed = 0;  // (under Combiner::mu) leaders admitted AHEAD of this request while it headed the queue: the fairness rule's clock
  bool same_shape(const CombineReq& o) const { return k == o.k && ef == o.ef && mode == o.mode && rerank_k == o.rerank_k; }
};
// FAIRNESS.  A request may lead while fewer batches than its kind's limit are in flight (graph walks overlap two launches, sweeps run
// alone), so without a rule a queued SWEEP needs `leaders == 0` — which back-to-back walk callers never let happen: every new walk leads
// at once while one walk is in flight, and the hand-off of a finished walk skips the sweep.  The rule: once the request at the HEAD of
// the queue has been passed kCombineMaxPassed times, nobody is admitted ahead of it any more — new arrivals queue behind it and a
// freed slot is kept free — until the batches in flight have drained far enough for it to lead (at zero in flight every kind may).
// Its wait is then bounded by kCombineMaxPassed admissions plus the batches in flight at that moment (tests/combiner_model.cpp
// `starve` mode: endless walk traffic plus one sweep caller).
constexpr uint32_t kCombineMaxPassed = 4;

struct Combiner {
  std::mutex mu;
  std::deque<CombineReq*> queue;
  int leaders = 0;  // batches in flight
  uint64_t arrivals = 0;          // calls ever queued (a waiting leader watches it move)
  uint32_t last_batch_calls = 1;  // calls the batch that finished last carried: > 1 = callers are arriving together
  uint64_t last_batch_done_at_arrival = 0;  // `arrivals` when that batch finished (its callers re-arrive behind this mark)
  uint64_t launches = 0, calls = 0, queries = 0, max_batch = 0;
};

// Sleeping callers wait on a word of their OWN request (futex): a finished batch wakes exactly its callers, and a freed leader
// slot wakes exactly one queued caller.  (One condition variable for everybody was the first version: every completion woke
// every sleeper into a fight for one mutex — with 64 callers on the box's 16 cores the stragglers came back after the next
// launch had left and the callers split into groups that took turns.)
static inline void futex_wait(std::atomic<uint32_t>* w, uint32_t expect) {
  syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, expect, nullptr, nullptr, 0);
}
static inline void futex_wake_one(std::atomic<uint32_t>* w) {
  syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
}

template <class Env>
int32_t search_combined(Env& env, Combiner* cb, CombineReq& me) {
  const uint32_t max_batch = env.max_batch();
  const uint32_t window_us = env.window_us();
  bool lead = false;
  {
    std::lock_guard<std::mutex> lk(cb->mu);
    cb->arrivals++;
    const bool head_starving = !cb->queue.empty() && cb->queue.front()->passed >= kCombineMaxPassed;
    if (!head_starving && cb->leaders < env.leader_limit(me)) {
      if (!cb->queue.empty()) cb->queue.front()->passed++;  // (admitted ahead of the queue's head)
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
    env.run_batch(room > 1 ? batch.data() : alone, n_calls);
    lk.lock();
    cb->last_batch_calls = (uint32_t)n_calls;
    cb->last_batch_done_at_arrival = cb->arrivals;
    cb->leaders--;
    // the freed slot goes to the first queued call that may lead (it takes the others of its shape with it)
    // (fairness: a head that has been passed kCombineMaxPassed times is the only one that may take it — the slot stays free until
    // the head can, which at the latest is when nothing is in flight)
    CombineReq* next = nullptr;
    for (auto it = cb->queue.begin(); it != cb->queue.end(); ++it) {
      if ((*it)->state == CombineReq::kQueued && cb->leaders < env.leader_limit(**it)) {
        next = *it;
        if (it != cb->queue.begin()) cb->queue.front()->passed++;
        cb->queue.erase(it);
        next->state = CombineReq::kTaken;
        cb->leaders++;
        break;
      }
      if (it == cb->queue.begin() && (*it)->passed >= kCombineMaxPassed) break;
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
  env.finish(me);  // (the caller's own thread: thread-local diagnostics of the call)
  return me.rc;
}


}  // namespace vdb
