// abi_driver.cpp — TEST INFRASTRUCTURE: a compiled, non-Python caller of the C ABI (include/velesdb_hip.h), built with a
// plain host compiler (g++ -std=c++17, no HIP headers) and linked against libvelesdb_hip.so.  It walks the life cycle a
// Rust `impl VectorIndex for HipHnswIndex` would drive (crates/velesdb-core/src/index/mod.rs:30-83,
// index/hnsw/index/trait_impl.rs:8-71): create -> insert (batch + single + duplicate) -> search (exact, graph) -> remove
// -> save_dir / destroy / load_dir -> search -> destroy, checks the ownership / error rules from the caller's side and
// prints every result as one JSON object; tests/test_gpu_hardening.py compares them with the ctypes path.
//   usage: abi_driver <scratch directory>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "velesdb_hip.h"

namespace {

constexpr uint32_t kDim = 64, kRows = 1000, kK = 5, kQueries = 5;

// the same rows tests/test_gpu_hardening.py builds: small dyadic rationals, exact in f32
float value(uint32_t i, uint32_t j) { return (float)((i * 131u + j * 71u + (i * j) % 13u) % 257u) / 128.0f - 1.0f; }

#define CHECK(call)                                                                                    \
  do {                                                                                                 \
    const int32_t rc_ = (call);                                                                        \
    if (rc_ < 0) {                                                                                     \
      std::fprintf(stderr, "%s -> %d: %s\n", #call, (int)rc_, vdb_hip_last_error());                   \
      return 1;                                                                                        \
    }                                                                                                  \
  } while (0)

struct Result {
  std::vector<uint64_t> ids;
  std::vector<float> scores;
  std::vector<uint32_t> n;
};

int search(vdb_hip_index* ix, const std::vector<float>& q, int32_t mode, uint32_t ef, Result* r) {
  r->ids.assign((size_t)kQueries * kK, 0);
  r->scores.assign((size_t)kQueries * kK, 0.0f);
  r->n.assign(kQueries, 0);
  CHECK(vdb_hip_index_search_batch(ix, q.data(), kQueries, kK, ef, mode, r->ids.data(), r->scores.data(), r->n.data()));
  return 0;
}

void print_result(const char* name, const Result& r, bool last) {
  std::printf("\"%s\": {\"ids\": [", name);
  for (size_t i = 0; i < r.ids.size(); i++) std::printf("%s%" PRIu64, i ? ", " : "", r.ids[i]);
  std::printf("], \"score_bits\": [");
  for (size_t i = 0; i < r.scores.size(); i++) {
    uint32_t b;
    std::memcpy(&b, &r.scores[i], 4);
    std::printf("%s%u", i ? ", " : "", b);
  }
  std::printf("], \"n\": [");
  for (size_t i = 0; i < r.n.size(); i++) std::printf("%s%u", i ? ", " : "", r.n[i]);
  std::printf("]}%s\n", last ? "" : ",");
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: abi_driver <scratch directory>\n");
    return 2;
  }
  const std::string dir = argv[1];
  int32_t ndev = 0;
  if (vdb_hip_device_count(&ndev) != VDB_OK || ndev <= 0) {  // no CPU fallback exists: the call says so
    std::printf("{\"no_device\": true, \"error\": \"%s\"}\n", vdb_hip_last_error());
    return 3;
  }
  std::vector<float> rows((size_t)kRows * kDim);
  std::vector<uint64_t> ids(kRows);
  for (uint32_t i = 0; i < kRows; i++) {
    ids[i] = 1000u + i;
    for (uint32_t j = 0; j < kDim; j++) rows[(size_t)i * kDim + j] = value(i, j);
  }
  std::vector<float> queries((size_t)kQueries * kDim);
  for (uint32_t q = 0; q < kQueries; q++)
    for (uint32_t j = 0; j < kDim; j++) queries[(size_t)q * kDim + j] = value(5000u + 37u * q, j);

  vdb_hip_index* ix = nullptr;
  CHECK(vdb_hip_index_create(kDim, VDB_COSINE, 8, 50, kRows, nullptr, 0, VDB_SHARD_REPLICA, &ix));
  uint64_t inserted = 0;
  CHECK(vdb_hip_index_insert_batch(ix, ids.data(), rows.data(), kRows - 1, &inserted));  // all but the last row ...
  if (inserted != kRows - 1) return 10;
  CHECK(vdb_hip_index_insert(ix, ids[kRows - 1], rows.data() + (size_t)(kRows - 1) * kDim, kDim));  // ... which arrives alone
  const int32_t dup = vdb_hip_index_insert(ix, ids[3], rows.data(), kDim);  // existing id: no-op (trait_impl.rs:23-25)
  if (dup != VDB_DUPLICATE_IGNORED) return 11;
  uint64_t len = 0;
  CHECK(vdb_hip_index_len(ix, &len));
  if (len != kRows) return 12;
  uint32_t dim = 0;
  int32_t metric = -1;
  CHECK(vdb_hip_index_dimension(ix, &dim));
  CHECK(vdb_hip_index_metric(ix, &metric));
  if (dim != kDim || metric != VDB_COSINE) return 13;
  // a wrong-sized query is an error code + message, never a crash (the shim panics with the reference's text)
  uint64_t one_id[kK];
  float one_sc[kK];
  uint32_t one_n = 0;
  const int32_t bad = vdb_hip_index_search(ix, queries.data(), kDim - 1, kK, 0, VDB_SEARCH_BRUTE, one_id, one_sc, &one_n);
  if (bad != VDB_ERR_DIM_MISMATCH || std::strlen(vdb_hip_last_error()) == 0) return 14;

  Result exact, graph, after_remove, reloaded;
  if (search(ix, queries, VDB_SEARCH_BRUTE, 0, &exact)) return 1;
  if (search(ix, queries, VDB_SEARCH_HNSW, 64, &graph)) return 1;
  int32_t removed = 0;
  CHECK(vdb_hip_index_remove(ix, exact.ids[0], &removed));
  if (removed != 1) return 15;
  CHECK(vdb_hip_index_remove(ix, exact.ids[0], &removed));  // again: not present any more
  if (removed != 0) return 16;
  CHECK(vdb_hip_index_len(ix, &len));
  if (len != kRows - 1) return 17;
  if (search(ix, queries, VDB_SEARCH_BRUTE, 0, &after_remove)) return 1;
  // the directory does not exist yet: HnswIndex::save creates it (constructors.rs:257 create_dir_all)
  CHECK(vdb_hip_index_save_dir(ix, dir.c_str()));
  vdb_hip_index_destroy(ix);
  ix = nullptr;
  CHECK(vdb_hip_index_load_dir(dir.c_str(), 0, &ix));
  CHECK(vdb_hip_index_len(ix, &len));
  if (len != kRows - 1) return 18;
  if (search(ix, queries, VDB_SEARCH_BRUTE, 0, &reloaded)) return 1;
  vdb_hip_index_destroy(ix);
  vdb_hip_index_destroy(nullptr);  // destroying nothing is allowed

  std::printf("{\"version\": \"%s\", \"devices\": %d,\n", vdb_hip_version(), (int)ndev);
  print_result("exact", exact, false);
  print_result("graph", graph, false);
  print_result("after_remove", after_remove, false);
  print_result("reloaded", reloaded, true);
  std::printf("}\n");
  return 0;
}
