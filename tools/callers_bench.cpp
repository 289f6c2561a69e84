// callers_bench.cpp — the reference's calling pattern against the C ABI: T host threads, each calling vdb_hip_index_search
// (host pointers, ONE query per call) in a loop — what velesdb-server does per request (velesdb-server/src/handlers/search.rs:34-73)
// and what the reference's stress tests do (index/hnsw/native/tests.rs:264-416).  Measurement / test infrastructure (bench.py's
// `concurrent_callers` leg, tests/test_gpu_callers.py); plain C++17 over include/velesdb_hip.h, no HIP headers.  Python threads
// cannot drive this: at 50 K calls per second the interpreter lock is the bottleneck, not the library.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "velesdb_hip.h"

extern "C" {

// Runs `threads` callers for `seconds` (at least `min_calls` calls per thread).  Caller t searches queries t, t + T, t + 2T, ...
// (mod n_queries), `nq_per_call` consecutive queries per call.  When ref_ids / ref_scores are given ([n_queries][k] from a
// batched call), every result is compared bit for bit and mismatches are counted.
// out[0] = queries per second, out[1] = p50 us per call, out[2] = p99 us, out[3] = mean us, out[4] = calls, out[5] = mismatching
// calls, out[6] = failed calls (status < 0)
int callers_run(vdb_hip_index* ix, const float* queries, uint32_t n_queries, uint32_t dim, uint32_t k, uint32_t ef, int32_t mode,
                int threads, double seconds, uint32_t min_calls, uint32_t nq_per_call, const uint64_t* ref_ids,
                const float* ref_scores, const uint32_t* ref_n, double* out) {
  if (!ix || !queries || threads < 1 || nq_per_call < 1 || n_queries < nq_per_call) return -1;
  using clk = std::chrono::steady_clock;
  std::vector<std::vector<float>> lat(threads);
  std::vector<uint64_t> bad(threads, 0), failed(threads, 0), served(threads, 0);
  std::atomic<int> ready{0};
  std::atomic<bool> go{false};
  clk::time_point t_end;
  const uint32_t slots = n_queries / nq_per_call;
  auto body = [&](int t) {
    std::vector<uint64_t> ids((size_t)nq_per_call * k);
    std::vector<float> sc((size_t)nq_per_call * k);
    std::vector<uint32_t> n(nq_per_call);
    lat[t].reserve(1 << 16);
    ready.fetch_add(1);
    while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
    uint32_t slot = (uint32_t)t % slots, calls = 0;
    for (;;) {
      const auto now = clk::now();
      if (now >= t_end && calls >= min_calls) break;
      const uint32_t q0 = slot * nq_per_call;
      int32_t rc;
      if (nq_per_call == 1)
        rc = vdb_hip_index_search(ix, queries + (size_t)q0 * dim, dim, k, ef, mode, ids.data(), sc.data(), n.data());
      else
        rc = vdb_hip_index_search_batch(ix, queries + (size_t)q0 * dim, nq_per_call, k, ef, mode, ids.data(), sc.data(), n.data());
      const auto t1 = clk::now();
      lat[t].push_back((float)std::chrono::duration<double, std::micro>(t1 - now).count());
      calls++;
      if (rc < 0) {
        failed[t]++;
      } else if (ref_ids) {
        bool same = true;
        for (uint32_t i = 0; i < nq_per_call && same; i++) {
          const uint32_t q = q0 + i;
          same = n[i] == ref_n[q] && std::memcmp(&ids[(size_t)i * k], ref_ids + (size_t)q * k, (size_t)n[i] * 8) == 0 &&
                 std::memcmp(&sc[(size_t)i * k], ref_scores + (size_t)q * k, (size_t)n[i] * 4) == 0;
        }
        if (!same) bad[t]++;
      }
      served[t] += nq_per_call;
      slot = (slot + (uint32_t)threads) % slots;
    }
  };
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++) th.emplace_back(body, t);
  while (ready.load() < threads) std::this_thread::yield();
  const auto t0 = clk::now();
  t_end = t0 + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double>(seconds));
  go.store(true, std::memory_order_release);
  for (auto& x : th) x.join();
  const double wall = std::chrono::duration<double>(clk::now() - t0).count();
  std::vector<float> all;
  uint64_t q = 0, b = 0, f = 0;
  for (int t = 0; t < threads; t++) {
    all.insert(all.end(), lat[t].begin(), lat[t].end());
    q += served[t];
    b += bad[t];
    f += failed[t];
  }
  std::sort(all.begin(), all.end());
  double sum = 0;
  for (float v : all) sum += v;
  out[0] = (double)q / wall;
  out[1] = all.empty() ? 0 : all[all.size() / 2];
  out[2] = all.empty() ? 0 : all[(size_t)((double)all.size() * 0.99)];
  out[3] = all.empty() ? 0 : sum / (double)all.size();
  out[4] = (double)all.size();
  out[5] = (double)b;
  out[6] = (double)f;
  return 0;
}

}  // extern "C"
