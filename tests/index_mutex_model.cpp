// index_mutex_model.cpp — the handle's reader / writer lock (velesdb_amd/csrc/vdb_host_sync.hpp, the text the library uses) under
// ThreadSanitizer: searching threads that take the shared lock back to back, inserting threads that take it exclusively.
// Test infrastructure (tests/test_host_sync_tsan_cpu.py); no GPU, no HIP.  The reference's pattern: searches under
// `inner.read()`, inserts under `inner.write()` on a parking_lot::RwLock that does not starve writers
// (index/hnsw/index/search.rs:80, index/hnsw/native/tests.rs:264-416).
//
// usage: index_mutex_model <readers> <writers> <seconds> <max_writer_wait_ms>
// Checks: a writer is alone (no reader, no other writer inside), readers overlap each other (the lock IS shared), the plain
// (non-atomic) state the lock protects is consistent, and no writer waits longer than the bound while readers arrive back to
// back — the property std::shared_mutex alone does not give on glibc (0.7 s per insert measured under eight searching threads).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vdb_host_sync.hpp"

int main(int argc, char** argv) {
  const int readers = argc > 1 ? std::atoi(argv[1]) : 8;
  const int writers = argc > 2 ? std::atoi(argv[2]) : 2;
  const double seconds = argc > 3 ? std::atof(argv[3]) : 1.0;
  const double bound_ms = argc > 4 ? std::atof(argv[4]) : 250.0;
  using clk = std::chrono::steady_clock;
  vdb::IndexMutex mu;
  // what the lock protects: two plain words that a writer moves together (a torn pair = a reader inside a writer's section)
  uint64_t a = 0, b = 0;
  std::atomic<int> readers_in{0}, writers_in{0}, fail{0}, max_readers_in{0};
  std::atomic<uint64_t> reads{0}, writes{0};
  std::atomic<bool> stop{false};
  std::atomic<uint64_t> worst_wait_us{0};
  std::vector<std::thread> pool;
  for (int r = 0; r < readers; r++)
    pool.emplace_back([&] {
      while (!stop.load(std::memory_order_relaxed)) {
        std::shared_lock<vdb::IndexMutex> lk(mu);
        const int now = readers_in.fetch_add(1) + 1;
        int seen = max_readers_in.load();
        while (now > seen && !max_readers_in.compare_exchange_weak(seen, now)) {
        }
        if (writers_in.load() != 0) fail.store(1);
        if (a != b) fail.store(2);
        for (volatile int spin = 0; spin < 200; spin = spin + 1) {
        }  // a search holds the lock for a while
        readers_in.fetch_sub(1);
        reads.fetch_add(1, std::memory_order_relaxed);
      }
    });
  for (int w = 0; w < writers; w++)
    pool.emplace_back([&] {
      while (!stop.load(std::memory_order_relaxed)) {
        const auto t0 = clk::now();
        {
          std::lock_guard<vdb::IndexMutex> lk(mu);
          const uint64_t us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(clk::now() - t0).count();
          uint64_t seen = worst_wait_us.load();
          while (us > seen && !worst_wait_us.compare_exchange_weak(seen, us)) {
          }
          if (writers_in.fetch_add(1) != 0) fail.store(3);
          if (readers_in.load() != 0) fail.store(4);
          a++;
          for (volatile int spin = 0; spin < 100; spin = spin + 1) {
          }
          b++;
          writers_in.fetch_sub(1);
        }
        writes.fetch_add(1, std::memory_order_relaxed);
        std::this_thread::sleep_for(std::chrono::microseconds(200));  // inserts arrive now and then, searches all the time
      }
    });
  std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
  stop.store(true);
  for (auto& t : pool) t.join();
  // try_lock / try_lock_shared on a free lock, and that a held exclusive lock refuses both
  bool tries = mu.try_lock();
  if (tries) {
    std::thread([&] { tries = !mu.try_lock_shared() && !mu.try_lock(); }).join();  // (another thread: a second request by the owner is undefined)
    mu.unlock();
  }
  if (tries) {
    tries = mu.try_lock_shared();
    if (tries) mu.unlock_shared();
  }
  const double worst_ms = (double)worst_wait_us.load() / 1000.0;
  const bool ok = fail.load() == 0 && a == b && a == writes.load() && tries && worst_ms <= bound_ms && writes.load() > 0 && reads.load() > 0 &&
                  (readers < 2 || max_readers_in.load() >= 2);
  std::printf("{\"readers\": %d, \"writers\": %d, \"reads\": %llu, \"writes\": %llu, \"worst_writer_wait_ms\": %.3f, \"max_readers_inside\": %d, "
              "\"violation\": %d, \"ok\": %s}\n",
              readers, writers, (unsigned long long)reads.load(), (unsigned long long)writes.load(), worst_ms, max_readers_in.load(), fail.load(),
              ok ? "true" : "false");
  return ok ? 0 : 1;
}
