// vdb_wide.hpp — argument blocks and launchers of the selection stage for 10 < k <= kWideMaxK (sweep_wide.hip; the WIDE instance of the
// selection kernel: sweep_gemm_bf16.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vdb_gemm_schedule.hpp"

namespace vdb {

constexpr uint32_t kWideMaxK = 128;        // HnswIndex::search_brute_force's k on this path (the reference benches 10 / 50 / 100)
constexpr uint32_t kWideCap = 4096;        // entries of a query's global list (what passes the bounds of one batch's launches)
constexpr uint32_t kWidePoolMax = 1024;    // candidates one block re-scores exactly (a larger final list: the query is unproven)
constexpr uint32_t kWideSeedRows = 16384;  // rows of the seed sample (one key per 16 rows: seed_scores_bf16)
constexpr uint32_t kWideSeedGroups = kWideSeedRows / 16;
constexpr uint32_t kWideSmallSeedMaxK = 32; // up to this k the k <= 10 stage's 4 096-row sample seeds the batch (256 keys: k-th best of them)
constexpr uint32_t kWideGivenUp = 1u;      // state bit: no bound exists or the list overflowed — the exact fallback answers the query

struct WideArgs {
  uint64_t* keys;                // [nq][cap] the queries' lists: (approximate score, row) keys, any order
  uint32_t* cnt;                 // [nq] entries appended (may exceed cap: the surplus was dropped)
  uint32_t* state;               // [nq] kWideGivenUp
  uint64_t* tau;                 // [nq] the bound the next launch runs under (key of row 0 at the bound's score)
  float* delta;                  // [nq] error bound of an approximate score
  const float* qnorms;           // [nq] canonical norms of the f32 queries
  const float* rho_q;            // [nq] rounding residual ratios of the batch (nullable: constant bound)
  const uint32_t* rho_max_bits;  // device scalar: largest residual ratio of the row image (nullable)
  const uint32_t* norm_max_bits; // device scalar: max |v| (DotProduct, Euclidean)
  float* extra;                  // [nq] Euclidean only (nullable): what a bound is lowered by on top of 2 delta — the canonical chain's own distance from the truth
  uint32_t cap, k, dim;
  float eps_extra;               // added to the relative error bound (SQ8 storage mode: the reference chain's own distance, select_eps level 3)
};
struct WideOutArgs {
  const float* rows;     // f32 rows of the index
  const float* norms;    // canonical row norms (cosine)
  const float* queries;  // original f32 queries
  const uint64_t* ext_ids;
  uint64_t* out_ids;     // [nq][k]
  float* out_scores;     // [nq][k] exact scores (the exact kernels' bits)
  uint32_t* out_n;       // [nq]
  uint32_t* flags;       // [nq] 1 = not proven: the gathered exact pass answers
  uint32_t* qcount;      // device word, zero before the launch
  uint32_t* qmap;        // [nq] slot -> query
  uint32_t* qslot;       // [nq] query -> slot
  uint64_t row_stride, q_stride;
  uint32_t dim_pad;
  // SQ8 storage mode (wide_rerank_sq8): the candidates are re-scored with the reference's asymmetric chain over the codes
  const uint8_t* sq8_codes;
  const float* sq8_min;
  const float* sq8_max;
  const float* sq8_nsq;
  uint64_t sq8_stride;
  // probe builds (VELESDB_WIDE_STAMPS=1): 8 wall-clock stamps per block of wide_rerank_verify, nullptr otherwise
  unsigned long long* stamps;
};

void launch_wide_seed(int metric, const WideArgs& a, const uint64_t* seed_keys, uint32_t ngrp, uint32_t nq, hipStream_t st);
void launch_wide_reseed(const WideArgs& a, uint32_t nq, hipStream_t st);
// fuse_final_reseed: the kernel takes wide_reseed's step itself (final bound from the whole list, pool = the entries that pass it)
void launch_wide_rerank(int metric, const WideArgs& a, const WideOutArgs& o, uint32_t nq, bool fuse_final_reseed, hipStream_t st);
// Euclidean batches (the augmented DotProduct form s = q.v - |v|^2 / 2 of sweep_split.hip): seed with the form's own error bound, re-scoring
// with the canonical (q - v)^2 lane chain, proof in the squared-distance domain.  dim_a = dim + 64 (the augmented image's width)
void launch_wide_seed_l2(const WideArgs& a, const uint64_t* seed_keys, uint32_t ngrp, uint32_t dim_a, uint32_t nq, hipStream_t st);
void launch_wide_rerank_l2(const WideArgs& a, const WideOutArgs& o, uint32_t nq, hipStream_t st);
// SQ8 storage mode (Cosine / DotProduct): selection over the dequantised bf16 image, candidates re-scored with dot_product_quantized_simd /
// cosine_similarity_quantized_simd's chain (core/quantization.rs:410-554)
void launch_wide_rerank_sq8(int metric, const WideArgs& a, const WideOutArgs& o, uint32_t nq, hipStream_t st);
// the WIDE instance of the 256 x 256 selection kernel over one launch of a schedule (sweep_gemm_bf16.hip)
hipError_t launch_sweep_gemm_bf16_wide(int metric, const Bf16GemmPlan& p, const uint16_t* rows16, uint64_t row_stride, const float* norms,
                                       const uint8_t* alive, const uint16_t* queries16, uint64_t q_stride, const uint64_t* tau0,
                                       uint64_t* wide_keys, uint32_t* wide_cnt, uint32_t wide_cap, uint32_t dim, uint32_t nq, hipStream_t st,
                                       const float* qnorms);

}  // namespace vdb
