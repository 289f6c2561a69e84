// vdb_kernels.hpp — argument blocks and host-callable launchers of the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vdb {

struct SweepArgs {
  const float* rows;       // [n_rows][row_stride] f32, row_stride % 4 == 0, 16-B aligned
  const float* norms;      // [n_rows] canonical sqrt(sum sq) (cosine only)
  const uint8_t* alive;    // [n_rows] 0 = soft-deleted (nullable = all alive)
  const float* queries;    // [nq][q_stride]
  uint64_t* part_keys;     // [nq][n_blocks][k], unused slots = invalid key
  uint32_t* part_cnt;      // unused (kept for ABI stability of the arg block)
  uint64_t row_stride;     // floats
  uint64_t q_stride;       // floats
  uint32_t n_rows;
  uint32_t dim;
  uint32_t nq;             // queries in this pass (<= B)
  uint32_t k;
};

struct MergeArgs {
  const uint64_t* part_keys;  // [nq][n_lists][k], unused slots = invalid key
  const uint32_t* part_cnt;   // unused
  const uint64_t* ext_ids;    // [n_rows] external ids (nullable -> row + row_base)
  uint64_t* out_ids;          // [nq][k]
  float* out_scores;          // [nq][k]
  uint32_t* out_n;            // [nq]
  uint64_t row_base;
  uint32_t n_lists;
  uint32_t k;
};

struct BitsArgs {
  const uint32_t* bits;     // [n_rows][words]
  const uint32_t* qbits;    // [nq][words]
  const uint8_t* alive;
  uint64_t* part_keys;      // [nq][n_waves][k]
  uint32_t* part_cnt;
  uint32_t n_rows;
  uint32_t words;           // multiple of 4
  uint32_t k;
};

struct PrepArgs {
  const float* rows;
  float* norms;      // nullable
  uint32_t* bits;    // nullable
  uint64_t row_stride;
  uint32_t row0;
  uint32_t n_rows;
  uint32_t dim;
  uint32_t words;
};

struct ScoreArgs {
  const float* query;
  const float* rows;  // [n_rows][dim]
  float* out;
  uint64_t n_rows;
  uint32_t dim;
  int32_t kind;       // 0 engine distance, 1 raw
  int32_t aligned16;  // rows, query 16-B aligned and dim % 4 == 0
};

size_t sweep_lds_bytes(int B, uint32_t k, uint32_t dim, int cpl);
int sweep_cpl_for_dim(uint32_t dim);
void launch_sweep_f32(int metric, int B, const SweepArgs& a, int blocks, hipStream_t st);
void launch_merge(bool higher_is_better, const MergeArgs& m, uint32_t nq, hipStream_t st);
void launch_sweep_bits(int metric, const BitsArgs& a, int blocks, uint32_t nq, hipStream_t st);
void launch_prep_rows(const PrepArgs& a, hipStream_t st);
void launch_score_rows(int metric, const ScoreArgs& a, hipStream_t st);

}  // namespace vdb
