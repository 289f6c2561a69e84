// sweep_split.hip — exact f32 Cosine / DotProduct search for LARGE query batches at bf16 matrix-core speed
// (HnswIndex::search_brute_force over a batch, index/hnsw/index/search.rs:176-219; BASELINE configs[1]).
//
// The exact f32 contraction runs on v_mfma_f32_16x16x4_f32 at 1/16 of the bf16 matrix rate (sweep_gemm.hip: 0.77 of that
// pipe = 79 K queries/s at 1 M x 768).  Here the matrix cores only SELECT:
//   1. every row (once, at insert) and every query (per batch) is split x = hi + lo + e, hi = bf16(x), lo = bf16(x - hi),
//      |e| <= 2^-17 |x|; the selection kernel (sweep_gemm_bf16.hip, SPLIT instance) accumulates hi.hi + hi.lo + lo.hi in
//      f32: an approximation A(x, q) of x.q with |A - x.q| <= eps |x| |q| (select_eps: the split remainders, the dropped
//      lo.lo term, a 4x padded worst case for the f32 accumulation inside and between the MFMAs, whose internal order is
//      not documented).  LEVEL 2 selects on hi.hi alone — the plain bf16 instance over the bf16 copy of the rows, a third of
//      the matrix work and half the bytes, eps ~ 2^-7: enough whenever the k-th best score stands clear of the pool's last
//      by that much (it does on the benchmark data and on typical embeddings; near-duplicate clusters go to level 1).  It keeps, per block, the k' = min(10, k + 3) best rows by approximate score under thresholds
//      seeded by an EXACT sweep of the first rows (threshold lowered by the error bound).  (k' > k: the bound a block ends
//      with is the score of its k'-th best row; with k' = k the block that holds the best row of a k = 1 search would
//      report that very score as the bound of what it left out, and no proof could succeed.)
//   2. the per-block lists are merged to the K2 = 32 best by approximate score; split_rerank_verify re-scores those with
//      the canonical chain of the exact kernels (oracle mode M: one fmaf chain per pair in the matrix-core order
//      k = 128 U + 16 m + 4 kk + c) — the reported ids, ranks and score bits are the exact kernel's, bit for bit;
//   3. and PROVES the answer per query: every row that was left out — by a block's threshold or by the cut at K2 — has
//      an approximate score <= B, hence an exact score <= B + delta; if the k-th best exact score is above that, nothing
//      outside can belong to the top k.  A query without proof (near-ties closer than the bound, non-finite data) is
//      flagged; the last block of that launch lists the flagged queries (selection_batch_tail), and the exact passes that follow
//      decide on the device whether they have work — a gathered pass of the streaming matrix-core kernel for a few, the exact
//      matrix-core GEMM kernel for the 128-query tiles that contain one (tile_needed) for many — no host synchronisation;
//      select_finish hands each flagged query its exact result.
#include <algorithm>

#include "vdb_device.hpp"
#include "vdb_kernels.hpp"
#include "vdb_block_select.hpp"

namespace vdb {

__device__ __forceinline__ uint16_t bf16_rne(float x) {  // round to nearest even; NaN stays (quiet) NaN
  uint32_t u = __float_as_uint(x);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// One wave per vector: the split image (per 32 elements: 32 x hi then 32 x lo, 128 bytes) and, optionally, the canonical
// norm (the chain of prep_rows: the value the exact kernels divide by).  dim % 32 == 0.
__global__ __launch_bounds__(256) void split_vectors_kernel(const float* src, uint64_t src_stride, uint16_t* out, float* norms,
                                                            uint32_t row0, uint32_t n, uint32_t dim) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t r = wave; r < n; r += nwaves) {
    const uint32_t row = row0 + r;
    const float* p = src + (size_t)row * src_stride;
    uint16_t* o = out + (size_t)row * dim * 2;
    float acc = 0.0f;
    for (uint32_t c = lane; c * 4 < dim; c += 64) {  // chunk c = elements 4c .. 4c + 3 (canonical lane assignment)
      const float4 x = ld4(p + c * 4);
      acc = chain4<kOpDot>(acc, x, x);
      const float xs[4] = {x.x, x.y, x.z, x.w};
      uint16_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        hi[e] = bf16_rne(xs[e]);
        lo[e] = bf16_rne(xs[e] - __uint_as_float((uint32_t)hi[e] << 16));  // the difference is exact in f32
      }
      const uint32_t k0 = c * 4, u = k0 >> 5, j = k0 & 31u;
      uint16_t* oh = o + u * 64 + j;
      *reinterpret_cast<uint2*>(oh) = make_uint2((uint32_t)hi[0] | ((uint32_t)hi[1] << 16), (uint32_t)hi[2] | ((uint32_t)hi[3] << 16));
      *reinterpret_cast<uint2*>(oh + 32) = make_uint2((uint32_t)lo[0] | ((uint32_t)lo[1] << 16), (uint32_t)lo[2] | ((uint32_t)lo[3] << 16));
    }
    if (norms) {
      const float nn = sqrtf(butterfly_all(acc));
      if (lane == 0) norms[row] = nn;
    }
  }
}
void launch_split_vectors(const float* src, uint64_t src_stride, uint16_t* out, float* norms, uint32_t row0, uint32_t n,
                          uint32_t dim, hipStream_t st) {
  if (n == 0) return;
  const int blocks = (int)std::min<uint64_t>(((uint64_t)n + 3) / 4, 4096);
  hipLaunchKernelGGL(split_vectors_kernel, dim3(blocks), dim3(256), 0, st, src, src_stride, out, norms, row0, n, dim);
}

// Error bound of a selection score, relative to |x| |q|.  bf16 keeps 8 significant bits: |x - hi| <= 2^-8 |x| (round to
// nearest even), and the remainder's own rounding |x - hi - lo| <= 2^-17 |x|.
//   level 1 (split: hi.hi + hi.lo + lo.hi): the dropped lo.lo term <= 2^-16, the two remainders 2 * 2^-17  => 2^-15 (8.2 * 2^-18
//            with slack);
//   level 2 (plain bf16: hi.hi only): hi.lo + lo.hi + lo.lo <= 2 * 2^-8 + 2^-16;
// plus 16 dim 2^-24 for the f32 accumulation inside and between the MFMAs (order undocumented; 4x padded worst case).
// (Cauchy-Schwarz over the per-element errors: sum |e_i q_i| <= |e| |q|.)
__host__ __device__ inline float select_eps(uint32_t dim, int level) {
  const float acc = 16.0f * (float)dim * 5.9604645e-8f;
  // level 3 = level 2 over the SQ8 storage mode's dequantised rows: + the distance between the matrix-core seed scores and
  // the reference's left-to-right chain (dim 2^-24 each way, padded)
  return (level >= 2 ? 2.0f * 3.90625e-3f * 1.002f + 1.6e-5f : 8.2f * 3.8146973e-6f) + acc + (level >= 3 ? 1.5e-4f : 0.0f);
}

// Level 2 with MEASURED rounding residuals (round 3).  The constant above takes every element at bf16's worst case (2^-8 of its
// magnitude); with ex = x - bf16(x), eq = q - bf16(q) the dropped terms are ex.hq + hx.eq + ex.eq, and by Cauchy-Schwarz
//   |x.q - hx.hq| <= (rho_x + rho_q + 3 rho_x rho_q) |x| |q|,   rho_x = |ex| / |x|, rho_q = |eq| / |q|   (rho <= 1),
// where rho_x is bounded by the largest residual ratio of any row of the image (prep_bf16_rows keeps it: a device scalar,
// rows of typical data sit at 0.4 x 2^-8) and rho_q is the batch's own (query_round_error).  Residuals and norms are summed in
// f64 (no underflow for any f32 row, rounding error far below the 1.002 pad).  Half the constant bound on the benchmark data:
// a third of the candidates in front of the selection kernel's epilogue.
__device__ __forceinline__ float select_eps_q(uint32_t dim, int level, const float* rho_q, const uint32_t* rho_max_bits, uint32_t q) {
  if (level < 2 || !rho_q || !rho_max_bits) return select_eps(dim, level);
  const float rm = __uint_as_float(*rho_max_bits), rq = rho_q[q];
  // (level 3 = level 2 over the SQ8 storage mode's dequantised rows: + select_eps's extra term for the reference's own chain)
  // (+ 4e-6: the Cosine batches' NORMALISED images divide by a canonical f32 norm whose own rounding is ~6e-7 relative, both sides — seln_rows_kernel)
  return (rm + rq + 3.0f * rm * rq) * 1.002f + 16.0f * (float)dim * 5.9604645e-8f + 4e-6f + (level >= 3 ? 1.5e-4f : 0.0f);  // (NaN query: NaN -> no bound, no proof)
}
// rho_q per query: one wave per query
__global__ __launch_bounds__(256) void query_round_error_kernel(const float* q, uint64_t q_stride, float* rho_q, uint32_t nq, uint32_t dim) {
  const uint32_t lane = threadIdx.x & 63u, b = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (b >= nq) return;
  const float* p = q + (size_t)b * q_stride;
  double se = 0.0, sx = 0.0;
  for (uint32_t i = lane; i < dim; i += 64) {
    const float x = p[i];
    const float e = x - __uint_as_float((uint32_t)bf16_rne(x) << 16);  // exact in f32
    se += (double)e * (double)e;
    sx += (double)x * (double)x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    se += __shfl_xor(se, o, 64);
    sx += __shfl_xor(sx, o, 64);
  }
  if (lane == 0) rho_q[b] = sx > 0.0 ? (float)(sqrt(se / sx) * 1.0000002) : (sx == 0.0 ? 0.0f : __uint_as_float(0x7FC00000u));
}
// The front of a level-2 / level-3 batch in ONE launch (it was four: round_queries_bf16, prep_rows for the canonical norms,
// query_round_error, a fill of the flag words): per query — one wave — the bf16 image row (round to nearest even, zero-padded to
// the image stride), the canonical f32 norm (prep_rows' chain: float4 chunk c in lane c mod 64, fmaf chains, xor butterfly, sqrt)
// and, rho_q != nullptr, the rounding residual ratio; the first threads clear the batch's flag words.  dim % 4 == 0.
__global__ __launch_bounds__(256) void sel16_prep_queries_kernel(const float* q, uint64_t q_stride, uint16_t* img, uint64_t img_stride,
                                                                 float* qnorms, float* rho_q, uint32_t* zero_words, uint32_t n_zero,
                                                                 uint32_t nq, uint32_t dim) {
  const uint32_t lane = threadIdx.x & 63u, b = blockIdx.x * 4u + (threadIdx.x >> 6);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_zero; i += gridDim.x * 256u) zero_words[i] = 0u;
  if (b >= nq) return;
  const float* p = q + (size_t)b * q_stride;
  uint16_t* o = img + (size_t)b * img_stride;
  const uint32_t d4 = dim / 4, s4 = (uint32_t)(img_stride / 4);
  float acc = 0.0f;
  double se = 0.0, sx = 0.0;
  for (uint32_t c = lane; c < s4; c += 64) {
    if (c < d4) {
      const float4 x = ld4(p + c * 4);
      acc = chain4<kOpDot>(acc, x, x);
      const float xs[4] = {x.x, x.y, x.z, x.w};
      uint16_t h[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        h[e] = bf16_rne(xs[e]);
        if (rho_q) {
          const float d = xs[e] - __uint_as_float((uint32_t)h[e] << 16);  // exact in f32
          se += (double)d * (double)d;
          sx += (double)xs[e] * (double)xs[e];
        }
      }
      *reinterpret_cast<uint2*>(o + c * 4) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    } else {
      *reinterpret_cast<uint2*>(o + c * 4) = make_uint2(0u, 0u);
    }
  }
  const float n = sqrtf(butterfly_all(acc));
  if (rho_q) {
#pragma unroll
    for (int s2 = 32; s2 > 0; s2 >>= 1) {
      se += __shfl_xor(se, s2, 64);
      sx += __shfl_xor(sx, s2, 64);
    }
  }
  if (lane == 0) {
    qnorms[b] = n;
    if (rho_q) rho_q[b] = sx > 0.0 ? (float)(sqrt(se / sx) * 1.0000002) : (sx == 0.0 ? 0.0f : __uint_as_float(0x7FC00000u));
  }
}
void launch_sel16_prep_queries(const float* q, uint64_t q_stride, uint16_t* img, uint64_t img_stride, float* qnorms, float* rho_q,
                               uint32_t* zero_words, uint32_t n_zero, uint32_t nq, uint32_t dim, hipStream_t st) {
  hipLaunchKernelGGL(sel16_prep_queries_kernel, dim3((nq + 3) / 4), dim3(256), 0, st, q, q_stride, img, img_stride, qnorms, rho_q,
                     zero_words, n_zero, nq, dim);
}
void launch_query_round_error(const float* q, uint64_t q_stride, float* rho_q, uint32_t nq, uint32_t dim, hipStream_t st) {
  hipLaunchKernelGGL(query_round_error_kernel, dim3((nq + 3) / 4), dim3(256), 0, st, q, q_stride, rho_q, nq, dim);
}

// ---- Cosine batches over NORMALISED images (round 6) ------------------------------------------------------------------------------
// cos(q, v) = (q / |q|) . (v / |v|): with both sides normalised BEFORE the bf16 rounding the selection is a DotProduct of unit
// vectors — the selection kernel's DotProduct instance runs unchanged over these images, its quick test needs no row-norm statistics
// and its bound is as tight for every element as for the lane's best row (the Cosine instance bounds a lane's 32 rows by the
// smallest / largest of their norms: on N(0,1) data +-2.5 % of the cut, as much slack again as the error bound itself => two to three
// times the hot lanes).  Error of a selection score against the exact cosine: (rho_v + rho_q + 3 rho_v rho_q) for the two bf16
// roundings (measured residual ratios of the NORMALISED vectors: select_eps_q), the f32 accumulation term, and 4e-6 for the
// division by the canonical f32 norms (their relative error ~6e-7 each).  A zero norm gives a zero image row — the exact cosine is
// 0 by definition (simd_avx512.rs:344-351) and so is the approximation; a NaN / inf / huge norm still makes its lane hot in the
// kernel (the real norms are handed to it: g16_quicktest_dense.inc `force`).
__global__ __launch_bounds__(256) void seln_rows_kernel(const float* rows, uint64_t row_stride, const float* norms, uint16_t* out, uint64_t out_stride,
                                                        uint32_t row0, uint32_t n_rows, uint32_t dim, uint32_t* rho_max_bits) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t r = wave; r < n_rows; r += nwaves) {
    const uint32_t row = row0 + r;
    const float* p = rows + (size_t)row * row_stride;
    uint16_t* o = out + (size_t)row * out_stride;
    const float nn = norms[row];
    double se = 0.0, sx = 0.0;
    for (uint32_t i = lane * 4; i < dim; i += 256) {  // dim % 64 == 0
      const float4 x = ld4(p + i);
      const float xs[4] = {x.x, x.y, x.z, x.w};
      uint16_t h[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float u = nn == 0.0f ? 0.0f : xs[e] / nn;
        h[e] = bf16_rne(u);
        const float d = u - __uint_as_float((uint32_t)h[e] << 16);  // exact in f32
        se += (double)d * (double)d;
        sx += (double)u * (double)u;
      }
      *reinterpret_cast<uint2*>(o + i) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    }
#pragma unroll
    for (int s2 = 32; s2 > 0; s2 >>= 1) {
      se += __shfl_xor(se, s2, 64);
      sx += __shfl_xor(sx, s2, 64);
    }
    if (lane == 0 && sx > 0.0) {
      const float rho = (float)(sqrt(se / sx) * 1.0000002);
      if (rho == rho && rho < __uint_as_float(0x7F800000u)) atomicMax(rho_max_bits, __float_as_uint(rho));
    }
  }
}
void launch_seln_rows(const float* rows, uint64_t row_stride, const float* norms, uint16_t* out, uint64_t out_stride, uint32_t row0, uint32_t n,
                      uint32_t dim, uint32_t* rho_max_bits, hipStream_t st) {
  if (n == 0) return;
  const int blocks = (int)std::min<uint64_t>(((uint64_t)n + 3) / 4, 4096);
  hipLaunchKernelGGL(seln_rows_kernel, dim3(blocks), dim3(256), 0, st, rows, row_stride, norms, out, out_stride, row0, n, dim, rho_max_bits);
}
// the front of such a batch in one launch (sel16_prep_queries_kernel's counterpart): per query — one wave — the canonical f32 norm
// (prep_rows' chain), the bf16 image row of q / |q| (zero-padded to the image stride), its residual ratio; the first threads clear the
// batch's flag words.  dim % 4 == 0.
__global__ __launch_bounds__(256) void seln_prep_queries_kernel(const float* q, uint64_t q_stride, uint16_t* img, uint64_t img_stride, float* qnorms,
                                                                float* rho_q, uint32_t* zero_words, uint32_t n_zero, uint32_t nq, uint32_t dim) {
  const uint32_t lane = threadIdx.x & 63u, b = blockIdx.x * 4u + (threadIdx.x >> 6);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_zero; i += gridDim.x * 256u) zero_words[i] = 0u;
  if (b >= nq) return;
  const float* p = q + (size_t)b * q_stride;
  uint16_t* o = img + (size_t)b * img_stride;
  const uint32_t d4 = dim / 4, s4 = (uint32_t)(img_stride / 4);
  float acc = 0.0f;
  for (uint32_t c = lane; c < d4; c += 64) {
    const float4 x = ld4(p + c * 4);
    acc = chain4<kOpDot>(acc, x, x);
  }
  const float n = sqrtf(butterfly_all(acc));
  double se = 0.0, sx = 0.0;
  for (uint32_t c = lane; c < s4; c += 64) {
    if (c < d4) {
      const float4 x = ld4(p + c * 4);
      const float xs[4] = {x.x, x.y, x.z, x.w};
      uint16_t h[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float u = n == 0.0f ? 0.0f : xs[e] / n;
        h[e] = bf16_rne(u);
        const float d = u - __uint_as_float((uint32_t)h[e] << 16);  // exact in f32
        se += (double)d * (double)d;
        sx += (double)u * (double)u;
      }
      *reinterpret_cast<uint2*>(o + c * 4) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
    } else {
      *reinterpret_cast<uint2*>(o + c * 4) = make_uint2(0u, 0u);
    }
  }
#pragma unroll
  for (int s2 = 32; s2 > 0; s2 >>= 1) {
    se += __shfl_xor(se, s2, 64);
    sx += __shfl_xor(sx, s2, 64);
  }
  if (lane == 0) {
    qnorms[b] = n;
    // (a zero query: every cosine is 0 and so is every approximation; a NaN query: NaN -> no bound, no proof)
    if (rho_q) rho_q[b] = sx > 0.0 ? (float)(sqrt(se / sx) * 1.0000002) : (sx == 0.0 ? 0.0f : __uint_as_float(0x7FC00000u));
  }
}
void launch_seln_prep_queries(const float* q, uint64_t q_stride, uint16_t* img, uint64_t img_stride, float* qnorms, float* rho_q,
                              uint32_t* zero_words, uint32_t n_zero, uint32_t nq, uint32_t dim, hipStream_t st) {
  hipLaunchKernelGGL(seln_prep_queries_kernel, dim3((nq + 3) / 4), dim3(256), 0, st, q, q_stride, img, img_stride, qnorms, rho_q, zero_words, n_zero, nq, dim);
}

// Seed from the EXACT sweep of the first rows (merged to rows + raw scores, best first): list slot 0 of the candidate pool
// = its top k as keys (they are exact scores: re-scoring them later reproduces them), the query's error bound
// delta (cosine: eps; dot: eps |q| max|v|), and the selection kernel's starting bound = k-th best exact score lowered by
// delta (a row whose APPROXIMATE score is below that has an exact score below k rows of the prefix).
template <int METRIC>
__global__ __launch_bounds__(256) void split_seed_kernel(const uint64_t* ids, const float* scores, const uint32_t* n,
                                                         const float* qnorms, const uint32_t* norm_max_bits, uint64_t* tau0,
                                                         float* delta, uint64_t* list, uint64_t* blk_tau, uint32_t list_stride,
                                                         uint32_t nq, uint32_t k, uint32_t klist, uint32_t dim, int level,
                                                         const float* rho_q, const uint32_t* rho_max_bits) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const float eps = select_eps_q(dim, level, rho_q, rho_max_bits, q);
  const float d = METRIC == kCosine ? eps * 1.001f + 4e-7f : eps * 1.001f * qnorms[q] * __uint_as_float(*norm_max_bits) + 1e-30f;
  delta[q] = d;
  const uint32_t c = min(n[q], k);
  uint64_t t = kKeyInvalid;
  if (c >= k && k > 0) {
    const float s = scores[(size_t)q * k + k - 1];
    const float lowered = s - d * 1.01f - fabsf(s) * 1e-6f;
    t = lowered == lowered ? make_key<true>(lowered, 0u) : kKeyInvalid;  // NaN: no bound
  }
  tau0[q] = t;
  for (uint32_t e = 0; e < klist; e++)  // pool lists hold klist >= k entries
    list[(size_t)q * list_stride * klist + e] = e < c ? make_key<true>(scores[(size_t)q * k + e], (uint32_t)ids[(size_t)q * k + e]) : kKeyInvalid;
  blk_tau[(size_t)q * list_stride] = kKeyInvalid;  // slot 0 = the seed rows: excluded exactly, no bound needed
}
void launch_split_seed(int metric, const uint64_t* ids, const float* scores, const uint32_t* n, const float* qnorms,
                       const uint32_t* norm_max_bits, uint64_t* tau0, float* delta, uint64_t* list, uint64_t* blk_tau,
                       uint32_t list_stride, uint32_t nq, uint32_t k, uint32_t klist, uint32_t dim, int level, hipStream_t st,
                       const float* rho_q, const uint32_t* rho_max_bits) {
  if (metric == kCosine)
    hipLaunchKernelGGL((split_seed_kernel<kCosine>), dim3((nq + 255) / 256), dim3(256), 0, st, ids, scores, n, qnorms, norm_max_bits,
                       tau0, delta, list, blk_tau, list_stride, nq, k, klist, dim, level, rho_q, rho_max_bits);
  else
    hipLaunchKernelGGL((split_seed_kernel<kDot>), dim3((nq + 255) / 256), dim3(256), 0, st, ids, scores, n, qnorms, norm_max_bits,
                       tau0, delta, list, blk_tau, list_stride, nq, k, klist, dim, level, rho_q, rho_max_bits);
}

// ---- level 2's seed on the bf16 pipe (round 3) -----------------------------------------------------------------------------
// The exact f32 seed sweep cost 235 us of a 2.5 ms batch: one 128 x 128 tile per block at 1/16 of the bf16 rate, and — no
// threshold exists yet — every one of its 16 384 products a candidate for the fused top-k.  A bound does not need exact scores:
// with A_k the k-th best APPROXIMATE score over any set of rows and delta the bound of |approximate - exact|, the k rows
// behind it have exact scores >= A_k - delta, so the exact k-th best over the corpus is >= A_k - delta, and a row of the exact
// top k has an approximate score >= A_k - 2 delta: tau = A_k - 2 delta (what split_reseed_kernel already uses between
// launches).  So: seed_scores_bf16 = a plain bf16 GEMM of the first rows x the batch over the selection's own images.  The seed is
// a SAMPLE: any k rows with approximate scores >= A' prove tau = A' - 2 delta, so the kernel keeps the best key of every 16 rows
// ([nq][seed_rows / 16]: 2 MiB instead of 32), merge_topk_select picks the ks best per query, split_seed_approx_kernel (or
// l2_seed_kernel in its approximate mode) turns their k-th into tau, and — kSeedIsSample — pool slot 0 stays empty: the selection
// launches sweep the seed rows again (0.4 % of a 1 M corpus), which is what lets the seed drop rows.
// One wave = 64 rows x 64 queries; both operands straight from L2 in fragment shape (16 B per lane per fragment: lane (i = l &
// 15, kk = l >> 4) holds elements 32 s + 8 kk .. + 7 of row / query i), two 32-deep steps in flight.
typedef float f32x4_s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_s __attribute__((ext_vector_type(8)));
template <int METRIC>
__global__ __launch_bounds__(256) void seed_scores_bf16(const uint16_t* rows16, uint64_t row_stride, const float* norms, const uint8_t* alive,
                                                        const uint16_t* q16, uint64_t q_stride, const float* qnorms, uint64_t* keys,
                                                        uint32_t seed_rows, uint32_t nq, uint32_t dim) {
  // (Round 4, measured and rejected: one 64 x 64 tile per BLOCK, its four waves splitting the k-extent and adding up through LDS —
  // three request rounds per wave instead of twelve and four times the waves per CU: 81.6 us against 40 for a wave per tile.
  // profiles/r04seed_split_k_seed_rejected.txt)
  const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
  // a wave = 64 rows x 64 queries (round 3: 16 x 64 — every query fragment fed ONE row fragment, 7.4 TB/s through L2 for 4 096
  // seed rows, 66 us; now four): row blocks rb = 0..3 of 16 rows
  const uint32_t row0 = (blockIdx.x * 4u + wib) * 64u;
  const uint32_t qb = blockIdx.y * 64u;
  if (row0 >= seed_rows) return;
  const uint32_t i = lane & 15u, kk = lane >> 4;
  const uint16_t* ap[4];
#pragma unroll
  for (int rb = 0; rb < 4; rb++) ap[rb] = rows16 + (size_t)min(row0 + (uint32_t)rb * 16u + i, seed_rows - 1u) * row_stride + kk * 8u;
  const uint16_t* bp[4];
#pragma unroll
  for (int t = 0; t < 4; t++) bp[t] = q16 + (size_t)min(qb + (uint32_t)t * 16u + i, nq - 1u) * q_stride + kk * 8u;
  f32x4_s acc[4][4];
#pragma unroll
  for (int rb = 0; rb < 4; rb++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[rb][t] = f32x4_s{0.f, 0.f, 0.f, 0.f};
  // Every 64-deep step touches the NEXT 128-byte line of the wave's 64 rows — lines the previous batch's sweep evicted long ago —
  // and a CU holds one such wave per SIMD: a chain of dim / 64 memory round trips (measured, round 4: 23 of the kernel's 40 us
  // at dim 768, the epilogue 12; profiles/r04seedvar_*).  So the fragments are double-buffered in registers: step i + 1 is
  // requested before step i multiplies.
  bf16x8_s av[2][4][2], bv[2][2][4];
  auto request = [&](int buf, uint32_t k0) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const bool in = k0 + (uint32_t)s * 32u < dim;
#pragma unroll
      for (int rb = 0; rb < 4; rb++) av[buf][rb][s] = in ? *reinterpret_cast<const bf16x8_s*>(ap[rb] + k0 + s * 32) : bf16x8_s{};
#pragma unroll
      for (int t = 0; t < 4; t++) bv[buf][s][t] = in ? *reinterpret_cast<const bf16x8_s*>(bp[t] + k0 + s * 32) : bf16x8_s{};
    }
  };
  auto multiply = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int t = 0; t < 4; t++) acc[rb][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[buf][rb][s], bv[buf][s][t], acc[rb][t], 0, 0, 0);
  };
  request(0, 0);
  // (the epilogue's norms ride along with the first request instead of starting a round trip of their own behind the loop)
  float vn[4][4], qn_t[4];
#pragma unroll
  for (int rb = 0; rb < 4; rb++)
#pragma unroll
    for (int r = 0; r < 4; r++)
      vn[rb][r] = METRIC == kCosine ? norms[min(row0 + (uint32_t)rb * 16u + 4u * kk + (uint32_t)r, seed_rows - 1u)] : 1.0f;
#pragma unroll
  for (int t = 0; t < 4; t++) qn_t[t] = METRIC == kCosine ? qnorms[min(qb + (uint32_t)t * 16u + i, nq - 1u)] : 1.0f;
  for (uint32_t k0 = 0; k0 < dim; k0 += 128) {  // dim % 32 == 0: steps past dim are zero fragments
    if (k0 + 64 < dim) request(1, k0 + 64);
    multiply(0);
    if (k0 + 64 < dim) {
      if (k0 + 128 < dim) request(0, k0 + 128);
      multiply(1);
    }
  }
  // lane holds rows row0 + 16 rb + 4 kk + r (rb, r = 0..3) of query column qb + 16 t + i: the best of those 16 as ONE key —
  // keys[q][group], group = 4 (row0 / 64) + kk (the seed is a sample: sweep_split.hip file header, select_stage.hip brute_split_dev)
  const uint32_t ngrp = (seed_rows + 15u) / 16u;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const uint32_t q = qb + (uint32_t)t * 16u + i;
    if (q >= nq) continue;
    const float qn = qn_t[t];
    uint64_t best = kKeyInvalid;
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t row = row0 + (uint32_t)rb * 16u + 4u * kk + (uint32_t)r;
        if (row >= seed_rows) continue;
        const float sc = finish_score<METRIC>(acc[rb][t][r], qn, vn[rb][r]);
        const bool live = !alive || alive[row] != 0;
        const uint64_t key = live ? make_key<true>(sc, row) : kKeyInvalid;
        best = key < best ? key : best;
      }
    keys[(size_t)q * ngrp + (row0 / 64u) * 4u + kk] = best;
  }
}
void launch_seed_scores_bf16(int metric, const uint16_t* rows16, uint64_t row_stride, const float* norms, const uint8_t* alive,
                             const uint16_t* q16, uint64_t q_stride, const float* qnorms, uint64_t* keys, uint32_t seed_rows,
                             uint32_t nq, uint32_t dim, hipStream_t st) {
  const dim3 grid((seed_rows + 255) / 256, (nq + 63) / 64);
  if (metric == kCosine)
    hipLaunchKernelGGL((seed_scores_bf16<kCosine>), grid, dim3(256), 0, st, rows16, row_stride, norms, alive, q16, q_stride, qnorms, keys, seed_rows, nq, dim);
  else
    hipLaunchKernelGGL((seed_scores_bf16<kDot>), grid, dim3(256), 0, st, rows16, row_stride, norms, alive, q16, q_stride, qnorms, keys, seed_rows, nq, dim);
}
// the ks best approximate seed scores of every query (merged, best first) -> pool slot 0, its bound, delta, tau
template <int METRIC>
__global__ __launch_bounds__(256) void split_seed_approx_kernel(const uint64_t* ids, const float* scores, const uint32_t* n, const float* qnorms,
                                                                const uint32_t* norm_max_bits, uint64_t* tau0, float* delta, uint64_t* list,
                                                                uint64_t* blk_tau, uint32_t list_stride, uint32_t nq, uint32_t k, uint32_t klist,
                                                                uint32_t seed_rows, uint32_t dim, int level, const float* rho_q,
                                                                const uint32_t* rho_max_bits) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const float eps = select_eps_q(dim, level, rho_q, rho_max_bits, q);
  const float d = METRIC == kCosine ? eps * 1.001f + 4e-7f : eps * 1.001f * qnorms[q] * __uint_as_float(*norm_max_bits) + 1e-30f;
  delta[q] = d;
  const uint32_t c = min(n[q], klist);
  uint64_t t = kKeyInvalid;
  if (c >= k && k > 0) {
    const float s = scores[(size_t)q * klist + k - 1];
    const float lowered = s - 2.0f * d * 1.01f - fabsf(s) * 1e-6f;  // approximate scores on both sides: 2 delta
    t = lowered == lowered ? make_key<true>(lowered, 0u) : kKeyInvalid;  // NaN: no bound
  }
  tau0[q] = t;
  if (seed_rows == kSeedIsSample) {  // the seed rows are swept again: slot 0 holds nothing and excluded nothing
    for (uint32_t e = 0; e < klist; e++) list[(size_t)q * list_stride * klist + e] = kKeyInvalid;
    blk_tau[(size_t)q * list_stride] = kKeyInvalid;
    return;
  }
  for (uint32_t e = 0; e < klist; e++)
    list[(size_t)q * list_stride * klist + e] = e < c ? make_key<true>(scores[(size_t)q * klist + e], (uint32_t)ids[(size_t)q * klist + e]) : kKeyInvalid;
  // what slot 0 left out: every other seed row has a key >= its klist-th (nothing when the seed region had no more rows)
  blk_tau[(size_t)q * list_stride] = (c == klist && seed_rows > klist) ? make_key<true>(scores[(size_t)q * klist + klist - 1], (uint32_t)ids[(size_t)q * klist + klist - 1])
                                                                      : kKeyInvalid;
}
void launch_split_seed_approx(int metric, const uint64_t* ids, const float* scores, const uint32_t* n, const float* qnorms,
                              const uint32_t* norm_max_bits, uint64_t* tau0, float* delta, uint64_t* list, uint64_t* blk_tau,
                              uint32_t list_stride, uint32_t nq, uint32_t k, uint32_t klist, uint32_t seed_rows, uint32_t dim, int level,
                              hipStream_t st, const float* rho_q, const uint32_t* rho_max_bits) {
  if (metric == kCosine)
    hipLaunchKernelGGL((split_seed_approx_kernel<kCosine>), dim3((nq + 255) / 256), dim3(256), 0, st, ids, scores, n, qnorms, norm_max_bits,
                       tau0, delta, list, blk_tau, list_stride, nq, k, klist, seed_rows, dim, level, rho_q, rho_max_bits);
  else
    hipLaunchKernelGGL((split_seed_approx_kernel<kDot>), dim3((nq + 255) / 256), dim3(256), 0, st, ids, scores, n, qnorms, norm_max_bits,
                       tau0, delta, list, blk_tau, list_stride, nq, k, klist, seed_rows, dim, level, rho_q, rho_max_bits);
}

// The same for a seed that is a SAMPLE (kSeedIsSample: one key per 16 seed rows, [nq][ngrp <= 256]), straight from the sample keys: the
// k-th best of them by radix selection (vdb_block_select.hpp) instead of a merge to the ks best + the kernel above — one launch for two
// (round 6; same tau / delta bits: the bound is a function of the k-th best score alone, pool slot 0 stays empty either way).
template <int METRIC>
__global__ __launch_bounds__(256) void split_seed_sample_kernel(const uint64_t* keys, uint32_t ngrp, const float* qnorms, const uint32_t* norm_max_bits,
                                                                uint64_t* tau0, float* delta, uint64_t* list, uint64_t* blk_tau, uint32_t list_stride,
                                                                uint32_t k, uint32_t klist, uint32_t dim, int level, const float* rho_q,
                                                                const uint32_t* rho_max_bits) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t ctl[2];
  __shared__ uint32_t total;
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  uint64_t key1[1];
  key1[0] = tid < ngrp ? keys[(size_t)q * ngrp + tid] : kKeyInvalid;
  if (tid == 0) total = 0;
  __syncthreads();
  if (key1[0] != kKeyInvalid) atomicAdd(&total, 1u);
  __syncthreads();
  const uint32_t valid = total;
  const float eps = select_eps_q(dim, level, rho_q, rho_max_bits, q);
  const float d = METRIC == kCosine ? eps * 1.001f + 4e-7f : eps * 1.001f * qnorms[q] * __uint_as_float(*norm_max_bits) + 1e-30f;
  uint64_t t = kKeyInvalid;
  if (valid >= k && k > 0) {  // (block-uniform)
    const uint32_t hi = block_kth_hi<1>(key1, k, hist, ctl);
    const float s = key_score<true>((uint64_t)hi << 32);
    const float lowered = s - 2.0f * d * 1.01f - fabsf(s) * 1e-6f;  // approximate scores on both sides: 2 delta
    t = lowered == lowered ? make_key<true>(lowered, 0u) : kKeyInvalid;  // NaN: no bound
  }
  if (tid == 0) {
    delta[q] = d;
    tau0[q] = t;
    blk_tau[(size_t)q * list_stride] = kKeyInvalid;  // the seed rows are swept again: slot 0 holds nothing and excluded nothing
  }
  for (uint32_t e = tid; e < klist; e += 256) list[(size_t)q * list_stride * klist + e] = kKeyInvalid;
}
void launch_split_seed_sample(int metric, const uint64_t* keys, uint32_t ngrp, const float* qnorms, const uint32_t* norm_max_bits, uint64_t* tau0,
                              float* delta, uint64_t* list, uint64_t* blk_tau, uint32_t list_stride, uint32_t nq, uint32_t k, uint32_t klist, uint32_t dim,
                              int level, hipStream_t st, const float* rho_q, const uint32_t* rho_max_bits) {
  if (metric == kCosine)
    hipLaunchKernelGGL((split_seed_sample_kernel<kCosine>), dim3(nq), dim3(256), 0, st, keys, ngrp, qnorms, norm_max_bits, tau0, delta, list, blk_tau,
                       list_stride, k, klist, dim, level, rho_q, rho_max_bits);
  else
    hipLaunchKernelGGL((split_seed_sample_kernel<kDot>), dim3(nq), dim3(256), 0, st, keys, ngrp, qnorms, norm_max_bits, tau0, delta, list, blk_tau,
                       list_stride, k, klist, dim, level, rho_q, rho_max_bits);
}

// Between two selection launches: the next launch's bound = k-th best POOL score so far (approximate scores, exact ones for
// slot 0), lowered by delta when it is an exact score's turn to bound approximate ones — simply always (it costs a sliver
// of tightness).
__global__ __launch_bounds__(256) void split_reseed_kernel(const uint64_t* ids, const float* scores, const uint32_t* n,
                                                           const float* delta, uint64_t* tau0, uint32_t nq, uint32_t k, uint32_t kout) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  uint64_t t = kKeyInvalid;
  if (n[q] >= k && k > 0) {
    const float s = scores[(size_t)q * kout + k - 1];
    const float lowered = s - 2.0f * delta[q] * 1.01f - fabsf(s) * 1e-6f;  // pool scores may err by delta either way
    t = lowered == lowered ? make_key<true>(lowered, 0u) : kKeyInvalid;
  }
  tau0[q] = t;
}
void launch_split_reseed(const uint64_t* ids, const float* scores, const uint32_t* n, const float* delta, uint64_t* tau0,
                         uint32_t nq, uint32_t k, uint32_t kout, hipStream_t st) {
  hipLaunchKernelGGL(split_reseed_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, ids, scores, n, delta, tau0, nq, k, kout);
}

// An unproven query lists itself (the former one-block collect_flagged launch, ~4.6 us on every batch's critical path): slot j from
// a device-scope counter, qmap[j] = the query, qslot[query] = j.  The order of the list is whatever the blocks' finishing order made
// it — the gathered exact passes that read it answer every listed query on its own, and select_finish hands query q the result in
// slot qslot[q], so no result depends on it.  Plain stores: the readers are later launches.
__device__ __forceinline__ void list_unproven(const SplitRerankArgs& a, uint32_t qi) {
  if (!a.qcount) return;
  const uint32_t j = atomicAdd(a.qcount, 1u);
  a.qmap[j] = qi;
  a.qslot[qi] = j;
}

// One block per query: exact re-scoring of the K2 best of the pool, ranking, proof.  See the file header.
// SQ8: the exact score is the reference's asymmetric distance over the row's SQ8 code (core/quantization.rs:410-554) — one
// left-to-right chain per (query, row), multiplies and adds rounded separately (storage_modes.hip sweep_topk_sq8 computes
// the same bits for the whole corpus; here only for the candidates).
template <int METRIC, bool SQ8>
__global__ __launch_bounds__(256) void split_rerank_verify(SplitRerankArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* qs = reinterpret_cast<float*>(smem);                    // [dim padded to 128]
  uint64_t* keys = reinterpret_cast<uint64_t*>(qs + a.dim_pad);  // [K2]
  unsigned long long* bmin = reinterpret_cast<unsigned long long*>(keys + a.k2);
  float* qred = reinterpret_cast<float*>(bmin + 1);              // SQ8: sum(q), sum(q*q), left to right
  const uint32_t tid = threadIdx.x, qi = blockIdx.x;
  const uint32_t n = min(a.cand_n[qi], a.k2);
  const float* q = a.queries + (size_t)qi * a.q_stride;
  for (uint32_t i = tid; i < a.dim_pad; i += 256) qs[i] = i < a.dim ? q[i] : 0.0f;
  if (tid == 0) *bmin = ~0ull;
  __syncthreads();
  // the bound the blocks ended with: the best (smallest key) of them; slot 0 (exact seed rows) holds kKeyInvalid
  {
    unsigned long long m = ~0ull;
    for (uint32_t g = tid; g < a.lists; g += 256) m = min(m, (unsigned long long)a.blk_tau[(size_t)qi * a.lists + g]);
    if (m != ~0ull) atomicMin(bmin, m);
  }
  if (SQ8) {
    if (tid == 64) {  // :330-334 sum(q); :528 sum(q*q) — two chains, two waves
      float s1 = 0.0f;
      for (uint32_t d = 0; d < a.dim; d++) s1 = __fadd_rn(s1, qs[d]);
      qred[0] = s1;
    }
    if (tid == 128) {
      float s2 = 0.0f;
      for (uint32_t d = 0; d < a.dim; d++) s2 = __fadd_rn(s2, __fmul_rn(qs[d], qs[d]));
      qred[1] = s2;
    }
    __syncthreads();
    if (tid < n) {
      const uint32_t row = (uint32_t)a.cand_rows[(size_t)qi * a.k2 + tid];
      const float mn = a.sq8_min[row], range = __fsub_rn(a.sq8_max[row], mn);
      const uint32_t* c = reinterpret_cast<const uint32_t*>(a.sq8_codes + (size_t)row * a.sq8_stride);
      float acc = 0.0f;
      if (range < 1.1920929e-07f) {
        acc = __fmul_rn(qred[0], mn);
      } else {
        const float scale = __fdiv_rn(range, 255.0f);
        for (uint32_t i = 0; i < a.dim; i += 4) {
          const uint32_t w = c[i >> 2];
#pragma unroll
          for (int e = 0; e < 4; e++)
            if (i + e < a.dim) {
              const float dq = __fadd_rn(__fmul_rn((float)((w >> (8 * e)) & 0xFFu), scale), mn);
              acc = __fadd_rn(acc, __fmul_rn(qs[i + e], dq));
            }
        }
      }
      float score = acc;
      if (METRIC == kCosine) {
        const float denom = sqrtf(__fmul_rn(qred[1], a.sq8_nsq[row]));
        score = denom < 1.1920929e-07f ? 0.0f : __fdiv_rn(acc, denom);
      }
      keys[tid] = make_key<true>(score, row);
    }
  } else {
  // oracle mode M (vdb_oracle.cpp dotM; sweep_topk_gemm_f32): ONE fmaf chain per candidate over k = 128 U + 16 m + 4 kk + c in
  // the order U, m, c, kk, the vector zero-padded to a multiple of 128 (the padding steps are part of the chain).  The chain is
  // serial by definition and belongs to one lane; the MEMORY side is the block's: the candidates' rows travel through LDS in
  // steps of 64 elements, every load instruction a run of whole rows' 256-byte pieces (a lane reading its own row 16 bytes at
  // a time touched 64 lines per instruction and, with 8 blocks per CU, evicted them before their other 48 bytes were used:
  // 2.1 x the rows' bytes from HBM, 124 us per 1 024-query batch), double-buffered: the next step's loads are in flight while
  // the chains run.
  const float qn = METRIC == kCosine ? a.qnorms[qi] : 0.0f;
  {
    constexpr uint32_t kStep = 64, kStride = kStep + 4;  // floats per row and step; + 4: 16-B aligned, rows on distinct banks
    float* stage = qred + 2;                             // [2][k2][kStride], 16-byte aligned (bmin + qred = 16 bytes)
    uint32_t* crow = reinterpret_cast<uint32_t*>(stage + 2 * (size_t)a.k2 * kStride);  // [k2] candidate rows
    if (tid < n) crow[tid] = (uint32_t)a.cand_rows[(size_t)qi * a.k2 + tid];
    __syncthreads();
    const uint32_t nf4 = n * (kStep / 4);  // float4 per step
    float4 v[4];                           // k2 <= 64: <= 1 024 float4 per step, 4 per thread
    auto fetch = [&](uint32_t U) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t f = tid + 256u * (uint32_t)i;
        const uint32_t r = f / (kStep / 4), c4 = f % (kStep / 4);
        v[i] = (f < nf4 && U + 4 * c4 < a.dim) ? ld4(a.rows + (size_t)crow[r] * a.row_stride + U + 4 * c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    };
    auto park = [&](uint32_t buf) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t f = tid + 256u * (uint32_t)i;
        const uint32_t r = f / (kStep / 4), c4 = f % (kStep / 4);
        if (f < nf4) *reinterpret_cast<float4*>(stage + ((size_t)buf * a.k2 + r) * kStride + 4 * c4) = v[i];
      }
    };
    fetch(0);
    park(0);
    __syncthreads();
    float acc = 0.0f;
    for (uint32_t U = 0, buf = 0; U < a.dim_pad; U += kStep, buf ^= 1u) {
      const bool more = U + kStep < a.dim_pad;
      if (more) fetch(U + kStep);
      if (tid < n) {
        const float* x = stage + ((size_t)buf * a.k2 + tid) * kStride;
#pragma unroll
        for (int m = 0; m < 4; m++) {
          float xr[16];
#pragma unroll
          for (int e = 0; e < 16; e += 4) {
            const float4 w = *reinterpret_cast<const float4*>(x + 16 * m + e);
            xr[e] = w.x; xr[e + 1] = w.y; xr[e + 2] = w.z; xr[e + 3] = w.w;
          }
          const uint32_t base = U + 16 * m;
#pragma unroll
          for (int c = 0; c < 4; c++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc = __builtin_fmaf(xr[4 * kk + c], qs[base + 4 * kk + c], acc);
        }
      }
      if (more) park(buf ^ 1u);
      __syncthreads();
    }
    if (tid < n) {
      const uint32_t row = crow[tid];
      const float score = finish_score<METRIC>(acc, qn, METRIC == kCosine ? a.norms[row] : 1.0f);
      keys[tid] = make_key<true>(score, row);
    }
  }
  }
  __syncthreads();
  if (tid >= 64) return;
  const int lane = (int)tid;
  const uint64_t key = (uint32_t)lane < n ? keys[lane] : kKeyInvalid;
  uint32_t rank = 0;
  for (uint32_t j = 0; j < n; j++) rank += keys[j] < key ? 1u : 0u;
  // lane e picks up the candidate of rank e (ranks are a permutation of 0..n-1: keys are unique)
  uint32_t mine = 0;
  for (uint32_t j = 0; j < n; j++) {
    const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)rank, (int)j);
    if (rj == (uint32_t)lane) mine = j;
  }
  const uint32_t kk = min(a.k, n);
  bool ok = n >= a.k && a.k > 0;
  if (ok) {
    const float neg_inf = __uint_as_float(0xFF800000u);
    const unsigned long long bm = *bmin;
    const float a_blocks = bm == ~0ull ? neg_inf : key_score<true>((uint64_t)bm);
    const float a_cut = n == a.k2 ? a.cand_scores[(size_t)qi * a.k2 + a.k2 - 1] : neg_inf;  // pool score of the last one kept
    const float A = fmaxf(a_blocks, a_cut) == fmaxf(a_blocks, a_cut) ? fmaxf(a_blocks, a_cut) : __uint_as_float(0x7FC00000u);
    const bool anan = a_blocks != a_blocks || a_cut != a_cut;
    const uint64_t kth = keys[__builtin_amdgcn_readlane((int)mine, (int)(a.k - 1))];
    const double Ek = (double)key_score<true>(kth);
    ok = !anan && Ek > (double)A + (double)a.delta[qi];  // false for NaN anywhere
  }
  if (lane == 0) {
    a.flags[qi] = ok ? 0u : 1u;
    if (!ok && a.tile_needed) a.tile_needed[qi / a.fb_qper] = 1u;
    if (!ok) list_unproven(a, qi);
    a.out_n[qi] = kk;
  }
  for (uint32_t e = lane; e < a.k; e += 64) {
    if (e < kk) {
      const uint64_t ke = keys[mine];
      const uint32_t row = key_row(ke);
      a.out_ids[(size_t)qi * a.k + e] = a.ext_ids ? a.ext_ids[row] : (uint64_t)row;
      a.out_scores[(size_t)qi * a.k + e] = key_score<true>(ke);
    } else {
      a.out_ids[(size_t)qi * a.k + e] = ~0ull;
      a.out_scores[(size_t)qi * a.k + e] = __uint_as_float(0x7FC00000u);
    }
  }
}
void launch_split_rerank(int metric, const SplitRerankArgs& a, uint32_t nq, hipStream_t st) {
  // query | keys | bmin | qred (16 B) | row stage [2][k2][68] f32 | candidate rows [k2]   (k2 <= 64)
  const size_t lds = ((size_t)a.dim_pad * 4 + (size_t)a.k2 * 8 + 16 + 16 + 2 * (size_t)a.k2 * 68 * 4 + (size_t)a.k2 * 4 + 15) & ~(size_t)15;
  if (a.sq8_codes) {
    if (metric == kCosine)
      hipLaunchKernelGGL((split_rerank_verify<kCosine, true>), dim3(nq), dim3(256), lds, st, a);
    else
      hipLaunchKernelGGL((split_rerank_verify<kDot, true>), dim3(nq), dim3(256), lds, st, a);
  } else if (metric == kCosine) {
    hipLaunchKernelGGL((split_rerank_verify<kCosine, false>), dim3(nq), dim3(256), lds, st, a);
  } else {
    hipLaunchKernelGGL((split_rerank_verify<kDot, false>), dim3(nq), dim3(256), lds, st, a);
  }
}


// the unproven queries take the exact passes' results; block 0 posts the batch's counts (vdb_kernels.hpp launch_select_finish)
__global__ __launch_bounds__(64) void select_finish_kernel(SelectFinishArgs a) {
  const uint32_t q = blockIdx.x, lane = threadIdx.x;
  if (q == 0 && lane == 0 && a.stats_host) {  // (was a launch of its own: select_stats)
    a.stats_host[0] = a.qcount ? *a.qcount : 0u;
    a.stats_host[1] = a.nq;
    a.stats_host[3] = a.stats_level;
    __threadfence_system();
    a.stats_host[2] = a.stats_seq;  // last: a reader that sees the new sequence number sees the counts
  }
  if (!a.flags[q]) return;
  const bool gathered = a.g_ids && (a.max_listed == 0u || *a.qcount <= a.max_listed);
  if (!gathered && !a.fb_ids) return;
  const uint32_t slot = gathered ? a.qslot[q] : q;
  const uint64_t* s_ids = gathered ? a.g_ids : a.fb_ids;
  const float* s_sc = gathered ? a.g_scores : a.fb_scores;
  for (uint32_t e = lane; e < a.k; e += 64) {
    a.out_ids[(size_t)q * a.k + e] = s_ids[(size_t)slot * a.k + e];
    a.out_scores[(size_t)q * a.k + e] = s_sc[(size_t)slot * a.k + e];
  }
  if (lane == 0) a.out_n[q] = (gathered ? a.g_n : a.fb_n)[slot];
}

void launch_select_finish(const SelectFinishArgs& a, hipStream_t st) { hipLaunchKernelGGL(select_finish_kernel, dim3(a.nq), dim3(64), 0, st, a); }

// ---- Euclidean batches through the same selection stage ----------------------------------------------------------------
// |q - v|^2 = |q|^2 - 2 (q.v - |v|^2 / 2): the nearest rows are the rows with the largest s = q.v - h, h = |v|^2 / 2, and s is a
// DOT PRODUCT of augmented vectors v' = (v, -h_hi, -h_lo, 0...), q' = (q, 1, 1, 0...) (h split into two bf16 so that the extra
// term is exact to 2^-17 h).  The bf16 selection kernel (DotProduct instance) runs unchanged over the augmented images; the
// thresholds come from the exact f32 matrix-core kernel (DotProduct) over an augmented f32 prefix; the candidates are re-scored
// with the canonical (q - v)^2 lane chain of the small-batch kernels (l2_rerank_verify) and proven in the squared-distance
// domain.  Error of a selection score: eps2 |q| max|v| for the q.v part (the bf16 roundings) + (2^-16 + acc) max h for the
// augmentation and the f32 accumulation over it.
__global__ __launch_bounds__(256) void l2_augment_rows_kernel(const float* rows, uint64_t row_stride, const float* norms, uint16_t* img,
                                                              uint32_t dim_a, float* seed, uint32_t dim_s, uint32_t seed_rows,
                                                              uint32_t row0, uint32_t n, uint32_t dim, uint32_t* rho_max_bits) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t r = wave; r < n; r += nwaves) {
    const uint32_t row = row0 + r;
    const float* p = rows + (size_t)row * row_stride;
    uint16_t* o = img + (size_t)row * dim_a;
    double se = 0.0, sx = 0.0;  // the row's rounding residual ratio (select_eps_q), as prep_bf16_rows keeps it for the bf16 copy
    for (uint32_t i = lane * 4; i < dim; i += 256) {  // dim % 64 == 0
      const float4 x = ld4(p + i);
      const float xs[4] = {x.x, x.y, x.z, x.w};
      uint16_t h[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        h[e] = bf16_rne(xs[e]);
        const float d = xs[e] - __uint_as_float((uint32_t)h[e] << 16);  // exact in f32
        se += (double)d * (double)d;
        sx += (double)xs[e] * (double)xs[e];
      }
      *reinterpret_cast<uint2*>(o + i) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
      if (row < seed_rows) *reinterpret_cast<float4*>(seed + (size_t)row * dim_s + i) = x;
    }
    if (rho_max_bits) {
#pragma unroll
      for (int s2 = 32; s2 > 0; s2 >>= 1) {
        se += __shfl_xor(se, s2, 64);
        sx += __shfl_xor(sx, s2, 64);
      }
      if (lane == 0 && sx > 0.0) {
        const float rho = (float)(sqrt(se / sx) * 1.0000002);
        if (rho == rho && rho < __uint_as_float(0x7F800000u)) atomicMax(rho_max_bits, __float_as_uint(rho));
      }
    }
    const float nn = norms[row];
    const float h = -0.5f * nn * nn;
    const uint16_t hi = bf16_rne(h);
    const uint16_t lo = bf16_rne(h - __uint_as_float((uint32_t)hi << 16));
    if (lane < 64u && dim + lane < dim_a) o[dim + lane] = lane == 0 ? hi : (lane == 1 ? lo : (uint16_t)0);
    if (row < seed_rows && lane < 4) seed[(size_t)row * dim_s + dim + lane] = lane == 0 ? h : 0.0f;
  }
}
void launch_l2_augment_rows(const float* rows, uint64_t row_stride, const float* norms, uint16_t* img, uint32_t dim_a, float* seed,
                            uint32_t dim_s, uint32_t seed_rows, uint32_t row0, uint32_t n, uint32_t dim, hipStream_t st,
                            uint32_t* rho_max_bits) {
  if (n == 0) return;
  const int blocks = (int)std::min<uint64_t>(((uint64_t)n + 3) / 4, 4096);
  hipLaunchKernelGGL(l2_augment_rows_kernel, dim3(blocks), dim3(256), 0, st, rows, row_stride, norms, img, dim_a, seed, dim_s, seed_rows,
                     row0, n, dim, rho_max_bits);
}
__global__ __launch_bounds__(256) void l2_augment_queries_kernel(const float* q, uint64_t q_stride, uint16_t* img, uint32_t dim_a,
                                                                 float* qaug, uint32_t dim_s, uint32_t nq, uint32_t dim) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= nq) return;
  const float* p = q + (size_t)wave * q_stride;
  uint16_t* o = img + (size_t)wave * dim_a;
  for (uint32_t i = lane * 4; i < dim; i += 256) {
    const float4 x = ld4(p + i);
    *reinterpret_cast<uint2*>(o + i) = make_uint2((uint32_t)bf16_rne(x.x) | ((uint32_t)bf16_rne(x.y) << 16),
                                                  (uint32_t)bf16_rne(x.z) | ((uint32_t)bf16_rne(x.w) << 16));
    *reinterpret_cast<float4*>(qaug + (size_t)wave * dim_s + i) = x;
  }
  if (dim + lane < dim_a) o[dim + lane] = lane < 2 ? (uint16_t)0x3F80 : (uint16_t)0;  // bf16(1.0)
  if (lane < 4) qaug[(size_t)wave * dim_s + dim + lane] = lane == 0 ? 1.0f : 0.0f;
}
void launch_l2_augment_queries(const float* q, uint64_t q_stride, uint16_t* img, uint32_t dim_a, float* qaug, uint32_t dim_s, uint32_t nq,
                               uint32_t dim, hipStream_t st) {
  hipLaunchKernelGGL(l2_augment_queries_kernel, dim3((nq + 3) / 4), dim3(256), 0, st, q, q_stride, img, dim_a, qaug, dim_s, nq, dim);
}

// seed from the exact DotProduct sweep of the augmented prefix (merged: rows + s values, best first): delta, starting bound,
// pool slot 0, and the bound of what the seed's cut left out (its k-th s: those rows were excluded by a NEARLY exact value,
// d_seed wide instead of delta)
__global__ __launch_bounds__(256) void l2_seed_kernel(const uint64_t* ids, const float* scores, const uint32_t* n, const float* qnorms,
                                                      const uint32_t* norm_max_bits, uint64_t* tau0, float* delta, uint64_t* list,
                                                      uint64_t* blk_tau, uint32_t list_stride, uint32_t nq, uint32_t k, uint32_t klist,
                                                      uint32_t dim_a, float extra_rel, const float* rho_q, const uint32_t* rho_max_bits,
                                                      uint32_t approx_seed_rows) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const float nmax = __uint_as_float(*norm_max_bits), hmax = 0.5f * nmax * nmax;
  const float acc = 16.0f * (float)dim_a * 5.9604645e-8f;
  // the q.v part: the constant bound of bf16's worst case, or — the image's largest residual ratio and the batch's own are
  // known — the measured one (select_eps_q; the augmentation columns are exact to 2^-16 h and stay in the h term below)
  float eps_r = 2.0f * 3.90625e-3f * 1.002f + 1.6e-5f + acc + extra_rel;                // select_eps(dim_a, 2) (+ SQ8: level 3's extra)
  if (rho_q && rho_max_bits) {
    const float rm = __uint_as_float(*rho_max_bits), rq = rho_q[q];
    eps_r = (rm + rq + 3.0f * rm * rq) * 1.002f + acc + extra_rel;
  }
  const float d = eps_r * 1.001f * qnorms[q] * nmax + (1.6e-5f + acc + extra_rel) * hmax + 1e-30f;  // bf16 roundings of q.v + the augmentation
  const float d_seed = 4.0f * acc * (qnorms[q] * nmax + hmax) + 1e-30f;                 // f32 matrix-core s against the true s
  delta[q] = d;
  if (approx_seed_rows) {
    // the seed ran on the bf16 pipe over the selection's own image (seed_scores_bf16: every seed score as a key, the klist best
    // merged): APPROXIMATE s on both sides — bound = s_k - 2 delta, slot 0 = the klist best as they are, and what slot 0 left
    // out is bounded by its klist-th key, as a selection block reports it (split_seed_approx_kernel's rules)
    const uint32_t c = min(n[q], klist);
    uint64_t t = kKeyInvalid;
    if (c >= k && k > 0) {
      const float s = scores[(size_t)q * klist + k - 1];
      const float lowered = s - 2.0f * d * 1.01f - fabsf(s) * 1e-6f;
      t = lowered == lowered ? make_key<true>(lowered, 0u) : kKeyInvalid;
    }
    tau0[q] = t;
    if (approx_seed_rows == kSeedIsSample) {  // the seed rows are swept again: slot 0 holds nothing and excluded nothing
      for (uint32_t e = 0; e < klist; e++) list[(size_t)q * list_stride * klist + e] = kKeyInvalid;
      blk_tau[(size_t)q * list_stride] = kKeyInvalid;
      return;
    }
    for (uint32_t e = 0; e < klist; e++)
      list[(size_t)q * list_stride * klist + e] = e < c ? make_key<true>(scores[(size_t)q * klist + e], (uint32_t)ids[(size_t)q * klist + e]) : kKeyInvalid;
    blk_tau[(size_t)q * list_stride] = (c == klist && approx_seed_rows > klist)
                                           ? make_key<true>(scores[(size_t)q * klist + klist - 1], (uint32_t)ids[(size_t)q * klist + klist - 1])
                                           : kKeyInvalid;
    return;
  }
  const uint32_t c = min(n[q], k);
  uint64_t t = kKeyInvalid, bt = kKeyInvalid;
  if (c >= k && k > 0) {
    const float s = scores[(size_t)q * k + k - 1];
    const float lowered = s - d * 1.01f - fabsf(s) * 1e-6f;
    t = lowered == lowered ? make_key<true>(lowered, 0u) : kKeyInvalid;
    const float b = s - d + d_seed;  // A + delta then reads s + d_seed
    bt = b == b ? make_key<true>(b, 0u) : make_key<true>(__uint_as_float(0x7F800000u), 0u);
  }
  tau0[q] = t;
  for (uint32_t e = 0; e < klist; e++)
    list[(size_t)q * list_stride * klist + e] = e < c ? make_key<true>(scores[(size_t)q * k + e], (uint32_t)ids[(size_t)q * k + e]) : kKeyInvalid;
  blk_tau[(size_t)q * list_stride] = bt;
}
void launch_l2_seed(const uint64_t* ids, const float* scores, const uint32_t* n, const float* qnorms, const uint32_t* norm_max_bits,
                    uint64_t* tau0, float* delta, uint64_t* list, uint64_t* blk_tau, uint32_t list_stride, uint32_t nq, uint32_t k,
                    uint32_t klist, uint32_t dim_a, float extra_rel, hipStream_t st, const float* rho_q, const uint32_t* rho_max_bits,
                    uint32_t approx_seed_rows) {
  hipLaunchKernelGGL(l2_seed_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, ids, scores, n, qnorms, norm_max_bits, tau0, delta, list,
                     blk_tau, list_stride, nq, k, klist, dim_a, extra_rel, rho_q, rho_max_bits, approx_seed_rows);
}

// One block per query (the Euclidean sibling of split_rerank_verify): every candidate is re-scored with the canonical
// (q - v)^2 lane chain + butterfly + sqrt (the arithmetic of the small-batch kernels: euclid_rerank_verify, sweep.hip), ranked by
// (distance, row), and the answer is proven: a row outside the pool has a selection score s <= A, a true s <= A + delta, a true
// squared distance >= |q|^2 - 2 (A + delta); if that is above the k-th best canonical squared sum (plus the canonical chain's own
// distance from the truth), nothing outside belongs to the top k.
__global__ __launch_bounds__(256) void l2_rerank_verify(SplitRerankArgs a) {
  __shared__ uint64_t keys[64];
  __shared__ float sums[64];
  __shared__ unsigned long long bmin;
  const int lane = lane_id();
  const int wib = (int)(threadIdx.x >> 6);
  const uint32_t qi = blockIdx.x;
  const uint32_t n = min(a.cand_n[qi], a.k2);  // k2 <= 64
  const float* q = a.queries + (size_t)qi * a.q_stride;
  const int d4 = (int)((a.dim + 3) / 4);
  if (threadIdx.x == 0) bmin = ~0ull;
  __syncthreads();
  {
    unsigned long long m = ~0ull;
    for (uint32_t g = threadIdx.x; g < a.lists; g += 256) m = min(m, (unsigned long long)a.blk_tau[(size_t)qi * a.lists + g]);
    if (m != ~0ull) atomicMin(&bmin, m);
  }
  if (a.sq8_codes) {
    // SQ8 storage mode: euclidean_squared_quantized_simd (core/quantization.rs:469-516) over the candidate's code, one thread per
    // candidate — groups of four: sum += ((f0^2 + f1^2) + f2^2) + f3^2, the remainder one by one; the SQUARED distance is the score
    if (threadIdx.x < n) {
      const uint32_t c = threadIdx.x;
      const uint32_t row = (uint32_t)a.cand_rows[(size_t)qi * a.k2 + c];
      const float mn = a.sq8_min[row], range = __fsub_rn(a.sq8_max[row], mn);
      const uint32_t* cw = reinterpret_cast<const uint32_t*>(a.sq8_codes + (size_t)row * a.sq8_stride);
      float sum = 0.0f;
      if (range < 1.1920929e-07f) {  // constant vector (:478-481): sum((q - value)^2), left to right
        for (uint32_t d = 0; d < a.dim; d++) {
          const float f = __fsub_rn(q[d], mn);
          sum = __fadd_rn(sum, __fmul_rn(f, f));
        }
      } else {
        const float scale = __fdiv_rn(range, 255.0f);
        const uint32_t full = a.dim & ~3u;
        for (uint32_t i = 0; i < full; i += 4) {
          const uint32_t w = cw[i >> 2];
          float f[4];
#pragma unroll
          for (int e = 0; e < 4; e++) f[e] = __fsub_rn(q[i + e], __fadd_rn(__fmul_rn((float)((w >> (8 * e)) & 0xFFu), scale), mn));
          const float t = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(f[0], f[0]), __fmul_rn(f[1], f[1])), __fmul_rn(f[2], f[2])), __fmul_rn(f[3], f[3]));
          sum = __fadd_rn(sum, t);
        }
        for (uint32_t i = full; i < a.dim; i++) {
          const float dq = __fadd_rn(__fmul_rn((float)((cw[i >> 2] >> (8 * (i & 3u))) & 0xFFu), scale), mn);
          const float f = __fsub_rn(q[i], dq);
          sum = __fadd_rn(sum, __fmul_rn(f, f));
        }
      }
      sums[c] = sum;
      keys[c] = make_key<false>(sum, row);
    }
  } else
  // (a wave's candidates wib, wib + 4, ...: FOUR at a time — their rows' chunks are requested together; one after the other the
  // sixteen of a wave were sixteen dependent round trips, 57 us per 1 024-query batch)
  for (uint32_t c0 = wib; c0 < n; c0 += 16) {
    const float* p[4];
    uint32_t rowv[4];
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t c = min(c0 + 4u * (uint32_t)u, n - 1u);  // (a clamped duplicate: computed, not stored)
      rowv[u] = (uint32_t)a.cand_rows[(size_t)qi * a.k2 + c];
      p[u] = a.rows + (size_t)rowv[u] * a.row_stride;
    }
    for (int ch = lane; ch < d4; ch += 64) {
      const int nv = (int)a.dim - ch * 4;
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; u++) x[u] = ld4(p[u] + ch * 4);
      float4 qq;
      if (nv >= 4) {
        qq = make_float4(q[ch * 4], q[ch * 4 + 1], q[ch * 4 + 2], q[ch * 4 + 3]);
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = chain4<kOpL2>(acc[u], qq, x[u]);
      } else {
        qq = make_float4(q[ch * 4], nv > 1 ? q[ch * 4 + 1] : 0.f, nv > 2 ? q[ch * 4 + 2] : 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = chain4_tail<kOpL2>(acc[u], qq, x[u], nv);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t c = c0 + 4u * (uint32_t)u;
      const float sum = butterfly_all(acc[u]);
      if (lane == 0 && c < n) {
        sums[c] = sum;
        keys[c] = make_key<false>(finish_score<kEuclidean>(sum, 0.f, 0.f), rowv[u]);
      }
    }
  }
  __syncthreads();
  if (wib != 0) return;
  const uint64_t key = (uint32_t)lane < n ? keys[lane] : kKeyInvalid;
  uint32_t rank = 0;
  for (uint32_t j = 0; j < n; j++) rank += keys[j] < key ? 1u : 0u;
  uint32_t mine = 0;
  for (uint32_t j = 0; j < n; j++) {
    const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)rank, (int)j);
    if (rj == (uint32_t)lane) mine = j;
  }
  const uint32_t kk = min(a.k, n);
  bool ok = n >= a.k && a.k > 0;
  if (ok) {
    const float neg_inf = __uint_as_float(0xFF800000u);
    const float a_blocks = bmin == ~0ull ? neg_inf : key_score<true>((uint64_t)bmin);
    const float a_cut = n == a.k2 ? a.cand_scores[(size_t)qi * a.k2 + a.k2 - 1] : neg_inf;
    const bool anan = a_blocks != a_blocks || a_cut != a_cut;
    const double A = (double)fmaxf(a_blocks, a_cut);
    const double qn = (double)a.qnorms[qi], nmax = (double)__uint_as_float(*a.norm_max_bits);
    const double Ek = (double)sums[__builtin_amdgcn_readlane((int)mine, (int)(a.k - 1))];
    // the canonical chain and |q|^2 against the truth: 8 n eps (|q|^2 + max|v|^2), as in euclid_rerank_verify
    const double slack = 8.0 * (double)a.dim * 5.9604645e-8 * (qn * qn + nmax * nmax) + 1e-6 * Ek;
    ok = !anan && (Ek + slack) < qn * qn - 2.0 * (A + (double)a.delta[qi]);  // false for NaN anywhere
  }
  if (lane == 0) {
    a.flags[qi] = ok ? 0u : 1u;
    if (!ok) list_unproven(a, qi);
    a.out_n[qi] = kk;
  }
  for (uint32_t e = lane; e < a.k; e += 64) {
    if (e < kk) {
      const uint64_t ke = keys[mine];
      const uint32_t row = key_row(ke);
      a.out_ids[(size_t)qi * a.k + e] = a.ext_ids ? a.ext_ids[row] : (uint64_t)row;
      a.out_scores[(size_t)qi * a.k + e] = key_score<false>(ke);
    } else {
      a.out_ids[(size_t)qi * a.k + e] = ~0ull;
      a.out_scores[(size_t)qi * a.k + e] = __uint_as_float(0x7FC00000u);
    }
  }
}
void launch_l2_rerank(const SplitRerankArgs& a, uint32_t nq, hipStream_t st) {
  hipLaunchKernelGGL(l2_rerank_verify, dim3(nq), dim3(256), 0, st, a);
}

}  // namespace vdb
