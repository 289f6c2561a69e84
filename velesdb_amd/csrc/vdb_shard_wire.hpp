// vdb_shard_wire.hpp — the record that travels in the range-sharded search's all-gather and the rules of the merge behind it, as
// host / device inline functions: shard_group.hip's kernels (pack_shard_records, merge_shards_topk) are written over them, and
// tests/shard_wire_model.cpp compiles the very same text for the host so that the CPU tier (tests/test_sharded_cpu.py, gloo,
// world 2 / 4 / 8) runs the PRODUCT's packing and merge rule, not a restatement of them.
//   record = 3 x u32: id low, id high, score bits.  A slot past the shard's result count carries the sentinel (id ~0, score bits
//   0xFFFFFFFF); a query whose traversal list overflowed in a device-resident call (count 0xFFFFFFFF) is marked by score bits
//   0xFFFFFFFE in its first record and leaves the merge with count 0xFFFFFFFF again.
//   merged order = DistanceMetric::sort_results (core/distance.rs:95-103): IEEE total order of the score in the metric's direction,
//   equal scores in global row order = (shard, position in the shard's list).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define VDB_WIRE_FN __host__ __device__ inline
#else
#define VDB_WIRE_FN inline
#endif

namespace vdb {
namespace wire {

constexpr uint32_t kRecWords = 3;
constexpr uint32_t kRecEmpty = 0xFFFFFFFFu, kRecOverflow = 0xFFFFFFFEu;

// record `p` of a query for which the shard reports `count` results (0xFFFFFFFF: the overflow marker of a device-resident call)
VDB_WIRE_FN void pack(uint64_t id, uint32_t score_bits, uint32_t count, uint32_t p, uint32_t* rec) {
  uint32_t lo = 0xFFFFFFFFu, hi = 0xFFFFFFFFu, sb = kRecEmpty;
  if (count == 0xFFFFFFFFu) {
    if (p == 0) sb = kRecOverflow;
  } else if (p < count) {
    lo = (uint32_t)id;
    hi = (uint32_t)(id >> 32);
    sb = score_bits;
  }
  rec[0] = lo;
  rec[1] = hi;
  rec[2] = sb;
}
VDB_WIRE_FN bool is_empty(const uint32_t* rec) {
  return rec[0] == 0xFFFFFFFFu && rec[1] == 0xFFFFFFFFu && (rec[2] == kRecEmpty || rec[2] == kRecOverflow);
}
VDB_WIRE_FN bool is_overflow(const uint32_t* rec) { return is_empty(rec) && rec[2] == kRecOverflow; }
VDB_WIRE_FN uint64_t id_of(const uint32_t* rec) { return ((uint64_t)rec[1] << 32) | rec[0]; }
// u32 whose unsigned order is f32::total_cmp's (native/ordered_float.rs:31-36), inverted where higher is better: smaller = better;
// an empty slot sorts behind everything
VDB_WIRE_FN uint32_t select_key(const uint32_t* rec, bool higher_is_better) {
  if (is_empty(rec)) return 0xFFFFFFFFu;
  const uint32_t b = rec[2];
  const uint32_t k = b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
  return higher_is_better ? ~k : k;
}
// how many of shard t's n_t keys (ascending) precede a record of shard s with key `key`: strictly smaller for a later shard,
// smaller-or-equal for an earlier one (equal scores keep the global row order)
VDB_WIRE_FN uint32_t count_before(const uint32_t* keys_t, uint32_t n_t, uint32_t key, bool t_after_s) {
  uint32_t lo = 0, hi = n_t;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    const uint32_t v = keys_t[mid];
    const bool before = t_after_s ? v < key : v <= key;
    if (before) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// rank of record (s, p) among the S lists of one query (keys = [S][k] selection keys, ns = records per shard); stops counting at k
VDB_WIRE_FN uint32_t merged_rank(const uint32_t* keys, const uint32_t* ns, uint32_t S, uint32_t k, uint32_t s, uint32_t p) {
  const uint32_t key = keys[(uint64_t)s * k + p];
  uint32_t rank = p;
  for (uint32_t t = 0; t < S && rank < k; t++) {
    if (t == s) continue;
    rank += count_before(keys + (uint64_t)t * k, ns[t], key, t > s);
  }
  return rank;
}

}  // namespace wire
}  // namespace vdb
