// shard_wire_model.cpp — the range-sharded search's wire record and merge rule on the host, over the PRODUCT's own text
// (velesdb_amd/csrc/vdb_shard_wire.hpp: the inline functions shard_group.hip's pack_shard_records / merge_shards_topk kernels are
// written over).  Built by tests/test_sharded_cpu.py with g++ as a small shared library; no GPU, no HIP headers.
//   wire_pack_all   = the loop of pack_shard_records over [nq][k] results
//   wire_merge_all  = the body of merge_shards_topk, one query after the other (keys + counts per shard, merged_rank per record,
//                     filler behind the count, the overflow marker)
#include <cstdint>
#include <cstring>
#include <vector>

#include "vdb_shard_wire.hpp"

using namespace vdb;

extern "C" {

uint32_t wire_record_bytes() { return wire::kRecWords * 4; }

void wire_pack_all(const uint64_t* ids, const float* scores, const uint32_t* n, uint32_t* rec, uint32_t nq, uint32_t k) {
  const uint64_t total = (uint64_t)nq * k;
  for (uint64_t i = 0; i < total; i++) {
    const uint32_t q = (uint32_t)(i / k), p = (uint32_t)(i % k);
    const uint32_t c = n[q];
    const bool live = c != 0xFFFFFFFFu && p < c;
    uint32_t sb = 0;
    if (live) std::memcpy(&sb, &scores[i], 4);
    wire::pack(live ? ids[i] : 0ull, sb, c, p, rec + i * 3);
  }
}

void wire_merge_all(const uint32_t* rec, uint32_t S, uint32_t nq, uint32_t k, int hib, uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
  const uint32_t T = S * k;
  std::vector<uint32_t> keys(T), ns(S);
  for (uint32_t q = 0; q < nq; q++) {
    for (uint32_t s = 0; s < S; s++) ns[s] = 0;
    bool ovf = false;
    for (uint32_t i = 0; i < T; i++) {
      const uint32_t s = i / k, p = i - s * k;
      const uint32_t* r = rec + (((size_t)s * nq + q) * k + p) * 3;
      if (wire::is_overflow(r)) ovf = true;
      keys[i] = wire::select_key(r, hib != 0);
      if (!wire::is_empty(r)) ns[s]++;
    }
    uint32_t total = 0;
    for (uint32_t s = 0; s < S; s++) total += ns[s];
    for (uint32_t i = 0; i < T; i++) {
      const uint32_t s = i / k, p = i - s * k;
      if (p >= ns[s]) continue;
      const uint32_t rank = wire::merged_rank(keys.data(), ns.data(), S, k, s, p);
      if (rank < k) {
        const uint32_t* r = rec + (((size_t)s * nq + q) * k + p) * 3;
        out_ids[(size_t)q * k + rank] = wire::id_of(r);
        std::memcpy(&out_scores[(size_t)q * k + rank], &r[2], 4);
      }
    }
    const uint32_t cnt = total < k ? total : k;
    for (uint32_t e = cnt; e < k; e++) {
      out_ids[(size_t)q * k + e] = ~0ull;
      const uint32_t nan = 0x7FC00000u;
      std::memcpy(&out_scores[(size_t)q * k + e], &nan, 4);
    }
    out_n[q] = ovf ? 0xFFFFFFFFu : cnt;
  }
}

}  // extern "C"
