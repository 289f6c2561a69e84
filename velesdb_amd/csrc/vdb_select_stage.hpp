// vdb_select_stage.hpp — what index.hip (the dispatch of HnswIndex::search_brute_force) and select_stage.hip (the batches that
// select on the matrix cores and score exactly) share.  Internal to the library.
#pragma once
#include "vdb_index.hpp"
#include "vdb_kernels.hpp"

namespace vdb {

// index.hip
EventPair* next_sel_events(vdb_hip_index* ix);                         // HIP events around one launch of the selection kernel (timing on)
int blocks_for(const vdb_hip_index* ix, int B, uint32_t ngroups);      // grid of the vector-ALU sweep for a B-query tile

// select_stage.hip
// exact sweep over the bf16 copy of the rows (VDB_SEARCH_BRUTE_BF16)
int32_t brute_bf16_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k, uint64_t* d_ids, float* d_scores,
                       uint32_t* d_n, hipStream_t st);
// which selection level serves the next chunk of an exact batch (0: none) and how many queries it takes
int select_level(vdb_hip_index* ix, uint32_t nq_left, uint32_t k);
int select_level_l2(vdb_hip_index* ix, uint32_t nq_left, uint32_t k);
uint32_t select_chunk(uint32_t nq_left, uint32_t min_queries = 0);
// 10 < k <= kWideMaxK (sweep_wide.hip): 4 when the WIDE selection serves the next chunk of an exact Cosine / DotProduct batch
int select_level_wide(vdb_hip_index* ix, uint32_t nq_left, uint32_t k, bool sq8 = false);
int32_t brute_wide_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nqg, uint32_t k, uint64_t* d_ids, float* d_scores,
                       uint32_t* d_n, hipStream_t st, bool sq8 = false);
constexpr uint32_t kSelectMinQueriesSq8 = 6;  // (see select_stage.hip)
// the four-bit image of the packed bit rows (Hamming / Jaccard batches on the matrix cores, bits_gemm.hip)
int32_t ensure_bits_image(vdb_hip_index* ix, hipStream_t st);
// the residual-ratio scalar of a bf16 copy that is about to be (re)built from row 0
int32_t reset_bf16_rho(vdb_hip_index* ix, hipStream_t st);

}  // namespace vdb
