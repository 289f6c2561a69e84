// vdb_host_sync.hpp — host-side synchronisation of a handle with nothing of HIP in it, so that tests/index_mutex_model.cpp can
// run the same text under ThreadSanitizer (tests/test_host_sync_tsan_cpu.py).
#pragma once
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <thread>

namespace vdb {

// The handle's reader / writer lock.  std::shared_mutex on glibc prefers readers: with searches arriving back to back an
// insert waited for a moment without any reader — 0.7 s per insert under eight searching threads
// (tests/test_gpu_hardening.py::test_concurrent_search_and_insert).  Writers announce themselves and new readers step aside
// while one waits (the reference's parking_lot::RwLock does not starve writers either, index/hnsw/index/search.rs:80).
class IndexMutex {
 public:
  void lock() {
    writers_waiting_.fetch_add(1, std::memory_order_acq_rel);
    m_.lock();
    writers_waiting_.fetch_sub(1, std::memory_order_acq_rel);
  }
  bool try_lock() { return m_.try_lock(); }
  void unlock() { m_.unlock(); }
  // NOT recursive: a thread that holds the shared lock and asks for it again while a writer waits would wait for the writer,
  // which waits for the first hold — no entry point nests them
  void lock_shared() {
    while (writers_waiting_.load(std::memory_order_acquire) > 0) std::this_thread::yield();
    m_.lock_shared();
  }
  bool try_lock_shared() { return writers_waiting_.load(std::memory_order_acquire) == 0 && m_.try_lock_shared(); }
  void unlock_shared() { m_.unlock_shared(); }

 private:
  std::shared_mutex m_;
  std::atomic<int> writers_waiting_{0};
};

}  // namespace vdb
