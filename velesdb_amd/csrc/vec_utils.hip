// vec_utils.hip — the remaining free functions of the reference's SIMD module that sit on §8(a) rows a16 / a18,
// as batch entry points (one launch for n vectors):
//   simd::norm / simd_explicit::norm_simd            (simd.rs:240-242, simd_explicit.rs:194-215)  -> vdb_hip_batch_norm
//   simd::normalize_inplace / normalize_inplace_simd (simd.rs:217-219, simd_explicit.rs:638-664)  -> vdb_hip_normalize_rows
//   simd::squared_l2_distance                        (simd.rs:207-211)  -> vdb_hip_batch_distance(kind = VDB_KIND_SQUARED)
//   cosine_similarity_normalized / batch_cosine_normalized (simd_avx512.rs:390-422): a plain dot product
//                                                                       -> vdb_hip_batch_distance(VDB_DOT, VDB_KIND_RAW)
//   simd_explicit::batch_dot_product                 (simd_explicit.rs:519-560)  -> vdb_hip_batch_dot_product
//   hamming_distance_binary(_fast), jaccard_similarity_binary over packed u64 words (simd_explicit.rs:308-360,457-500)
//                                                                       -> vdb_hip_batch_hamming_binary / _jaccard_binary
// (batch_similarity_top_k, simd_explicit.rs:583-634, is the exact search of sweep.hip / sweep_gemm.hip.)
// f32 arithmetic is the canonical mode C of vdb_device.hpp (the reference's own variants of norm differ from each
// other in summation order; its tests pin them to 1e-5); the packed-word functions are integer-exact.
#include <algorithm>
#include <memory>

#include "vdb_device.hpp"
#include "vdb_index.hpp"
#include "vdb_kernels.hpp"

namespace vdb {

// one wave per row: canonical sqrt(sum x^2); NORMALIZE: x * (1 / norm), a zero norm leaves the row unchanged
template <bool NORMALIZE>
__global__ __launch_bounds__(256) void rows_norm_kernel(float* rows, uint64_t n, uint32_t dim, float* out) {
  const int lane = lane_id();
  const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint64_t nwaves = (uint64_t)gridDim.x * 4;
  for (uint64_t r = wave; r < n; r += nwaves) {
    float* p = rows + (size_t)r * dim;
    float acc = 0.0f;
    for (uint32_t c = lane; c * 4 < dim; c += 64)  // element i -> chunk i/4 -> lane (i/4) % 64, increasing i
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint32_t i = c * 4 + e;
        if (i < dim) acc = __builtin_fmaf(p[i], p[i], acc);
      }
    const float nrm = sqrtf(butterfly_all(acc));
    if (NORMALIZE) {
      if (nrm != 0.0f) {  // simd_explicit.rs:641-643
        const float inv = 1.0f / nrm;
        for (uint32_t i = lane; i < dim; i += 64) p[i] = p[i] * inv;
      }
    } else if (lane == 0) {
      out[r] = nrm;
    }
  }
}

// lane per row: popcounts over the packed words
template <bool JACCARD>
__global__ __launch_bounds__(256) void binary_words_kernel(const uint64_t* q, const uint64_t* rows, uint64_t n, uint32_t words,
                                                           uint32_t* out_h, float* out_j) {
  const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  const uint64_t* p = rows + (size_t)r * words;
  uint32_t ham = 0, inter = 0, uni = 0;
  for (uint32_t w = 0; w < words; w++) {
    const uint64_t x = p[w], y = q[w];
    if (JACCARD) {
      inter += (uint32_t)__popcll(x & y);
      uni += (uint32_t)__popcll(x | y);
    } else {
      ham += (uint32_t)__popcll(x ^ y);
    }
  }
  if (JACCARD)
    out_j[r] = uni == 0 ? 1.0f : (float)inter / (float)uni;  // J(empty, empty) = 1 (simd_explicit.rs:489-492)
  else
    out_h[r] = ham;
}

static int32_t device_ready(int32_t device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(VDB_ERR_NO_DEVICE, "no HIP device visible (hipGetDeviceCount)");
  }
  if (device < 0 || device >= n) return fail(VDB_ERR_INVALID_ARG, "bad device ordinal");
  VDB_HIP(hipSetDevice(device));
  return VDB_OK;
}

struct TmpBuf {
  void* p = nullptr;
  ~TmpBuf() {
    if (p) (void)hipFree(p);
  }
};

}  // namespace vdb

using namespace vdb;

extern "C" {

int32_t vdb_hip_batch_norm(int32_t device, const float* vecs_rowmajor, uint64_t n, uint32_t dim, float* out) {
  return vdb::guarded([&]() -> int32_t {
  if (n == 0) return VDB_OK;
  if (!vecs_rowmajor || !out || dim == 0) return fail(VDB_ERR_INVALID_ARG, "null argument");
  int32_t rc = device_ready(device);
  if (rc != VDB_OK) return rc;
  TmpBuf d_rows, d_out;
  VDB_HIP(hipMalloc(&d_rows.p, (size_t)n * dim * 4));
  VDB_HIP(hipMalloc(&d_out.p, (size_t)n * 4));
  VDB_HIP(hipMemcpy(d_rows.p, vecs_rowmajor, (size_t)n * dim * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((rows_norm_kernel<false>), dim3((unsigned)std::min<uint64_t>((n + 3) / 4, 4096)), dim3(256), 0, 0,
                     (float*)d_rows.p, n, dim, (float*)d_out.p);
  VDB_HIP(hipGetLastError());
  VDB_HIP(hipMemcpy(out, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return VDB_OK;
  });
}

int32_t vdb_hip_normalize_rows(int32_t device, float* vecs_rowmajor, uint64_t n, uint32_t dim) {
  return vdb::guarded([&]() -> int32_t {
  if (n == 0) return VDB_OK;
  if (!vecs_rowmajor || dim == 0) return fail(VDB_ERR_INVALID_ARG, "null argument");
  int32_t rc = device_ready(device);
  if (rc != VDB_OK) return rc;
  TmpBuf d_rows;
  VDB_HIP(hipMalloc(&d_rows.p, (size_t)n * dim * 4));
  VDB_HIP(hipMemcpy(d_rows.p, vecs_rowmajor, (size_t)n * dim * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((rows_norm_kernel<true>), dim3((unsigned)std::min<uint64_t>((n + 3) / 4, 4096)), dim3(256), 0, 0,
                     (float*)d_rows.p, n, dim, (float*)nullptr);
  VDB_HIP(hipGetLastError());
  VDB_HIP(hipMemcpy(vecs_rowmajor, d_rows.p, (size_t)n * dim * 4, hipMemcpyDeviceToHost));
  return VDB_OK;
  });
}

static int32_t binary_words(int32_t device, const uint64_t* query_words, const uint64_t* rows_words, uint64_t n,
                            uint32_t words, uint32_t* out_h, float* out_j) {
  if (n == 0) return VDB_OK;
  if (!query_words || !rows_words || words == 0 || (!out_h && !out_j)) return fail(VDB_ERR_INVALID_ARG, "null argument");
  int32_t rc = device_ready(device);
  if (rc != VDB_OK) return rc;
  TmpBuf d_q, d_rows, d_out;
  VDB_HIP(hipMalloc(&d_q.p, (size_t)words * 8));
  VDB_HIP(hipMalloc(&d_rows.p, (size_t)n * words * 8));
  VDB_HIP(hipMalloc(&d_out.p, (size_t)n * 4));
  VDB_HIP(hipMemcpy(d_q.p, query_words, (size_t)words * 8, hipMemcpyHostToDevice));
  VDB_HIP(hipMemcpy(d_rows.p, rows_words, (size_t)n * words * 8, hipMemcpyHostToDevice));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (out_j)
    hipLaunchKernelGGL((binary_words_kernel<true>), dim3(blocks), dim3(256), 0, 0, (const uint64_t*)d_q.p,
                       (const uint64_t*)d_rows.p, n, words, (uint32_t*)nullptr, (float*)d_out.p);
  else
    hipLaunchKernelGGL((binary_words_kernel<false>), dim3(blocks), dim3(256), 0, 0, (const uint64_t*)d_q.p,
                       (const uint64_t*)d_rows.p, n, words, (uint32_t*)d_out.p, (float*)nullptr);
  VDB_HIP(hipGetLastError());
  VDB_HIP(hipMemcpy(out_j ? (void*)out_j : (void*)out_h, d_out.p, (size_t)n * 4, hipMemcpyDeviceToHost));
  return VDB_OK;
}

int32_t vdb_hip_batch_hamming_binary(int32_t device, const uint64_t* query_words, const uint64_t* rows_words, uint64_t n,
                                     uint32_t words, uint32_t* out) {
  return vdb::guarded([&]() -> int32_t {
  return binary_words(device, query_words, rows_words, n, words, out, nullptr);
  });
}

int32_t vdb_hip_batch_jaccard_binary(int32_t device, const uint64_t* query_words, const uint64_t* rows_words, uint64_t n,
                                     uint32_t words, float* out) {
  return vdb::guarded([&]() -> int32_t {
  return binary_words(device, query_words, rows_words, n, words, nullptr, out);
  });
}

// batch_dot_product(queries, vectors) -> out[i * n + j] = dot(queries[i], vectors[j]) (simd_explicit.rs:519-560)
int32_t vdb_hip_batch_dot_product(int32_t device, const float* queries_rowmajor, uint32_t nq, const float* vecs_rowmajor,
                                  uint64_t n, uint32_t dim, float* out) {
  return vdb::guarded([&]() -> int32_t {
  if (nq == 0 || n == 0) return VDB_OK;
  if (!queries_rowmajor || !vecs_rowmajor || !out || dim == 0) return fail(VDB_ERR_INVALID_ARG, "null argument");
  int32_t rc = device_ready(device);
  if (rc != VDB_OK) return rc;
  TmpBuf d_q, d_rows, d_out;
  VDB_HIP(hipMalloc(&d_q.p, (size_t)nq * dim * 4));
  VDB_HIP(hipMalloc(&d_rows.p, (size_t)n * dim * 4));
  VDB_HIP(hipMalloc(&d_out.p, (size_t)nq * n * 4));
  VDB_HIP(hipMemcpy(d_q.p, queries_rowmajor, (size_t)nq * dim * 4, hipMemcpyHostToDevice));
  VDB_HIP(hipMemcpy(d_rows.p, vecs_rowmajor, (size_t)n * dim * 4, hipMemcpyHostToDevice));
  for (uint32_t i = 0; i < nq; i++) {
    ScoreArgs sa{};
    sa.query = (const float*)d_q.p + (size_t)i * dim;
    sa.rows = (const float*)d_rows.p;
    sa.out = (float*)d_out.p + (size_t)i * n;
    sa.n_rows = n;
    sa.dim = dim;
    sa.kind = VDB_KIND_RAW;
    sa.aligned16 = (dim % 4 == 0) ? 1 : 0;
    launch_score_rows(VDB_DOT, sa, 0);
  }
  VDB_HIP(hipGetLastError());
  VDB_HIP(hipMemcpy(out, d_out.p, (size_t)nq * n * 4, hipMemcpyDeviceToHost));
  return VDB_OK;
  });
}

}  // extern "C"
