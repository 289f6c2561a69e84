// pool_select.hip — what the selection stage (select_stage.hip brute_split_dev) needs of its candidate pool between and behind the
// launches of a batch, without merging it: the pool is [nq][list_stride][ks] keys — one list of ks (approximate score, row) keys per
// (launch, row group), kKeyInvalid in unused slots.
//   * between two launches only the BOUND of the next one matters: the k-th best pool score so far, lowered by twice the error bound
//     (sweep_split.hip split_reseed_kernel's rule) — pool_kth_reseed: a radix select over the keys in registers (vdb_block_select.hpp);
//   * behind the last launch the K2 best of the pool go to the exact re-scoring, best first — pool_topk: the K2-th smallest key by the
//     same selection (all 64 bits: keys of equal score are told apart by their row, as the merge does), the keys under it ranked by
//     counting.
// Rounds 2-5 ran merge_topk_select for both (one barrier per key bit over the whole pool in LDS): 4 x ~28 us of a 1.59-ms headline
// step (profiles/r05final2_headline_kernel_stats.csv).  Same outputs, bit for bit: the bound is a function of the k-th best score alone,
// the K2 best keys and their order are unique.
#include <algorithm>

#include "vdb_block_select.hpp"
#include "vdb_kernels.hpp"

namespace vdb {

template <int NPT>
__device__ __forceinline__ uint32_t pool_load(const uint64_t* pool, uint32_t q, uint32_t n_lists, uint32_t list_stride, uint32_t ks, uint64_t (&keys)[NPT]) {
  const uint32_t total = n_lists * ks;
  const uint64_t* base = pool + (size_t)q * list_stride * ks;  // (the first n_lists lists of the query are contiguous)
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < NPT; j++) {
    const uint32_t i = threadIdx.x + 256u * (uint32_t)j;
    keys[j] = i < total ? base[i] : kKeyInvalid;
    mine += keys[j] != kKeyInvalid ? 1u : 0u;
  }
  return mine;
}
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* word) {
  if (threadIdx.x == 0) *word = 0;
  __syncthreads();
  if (v) atomicAdd(word, v);
  __syncthreads();
  const uint32_t t = *word;
  __syncthreads();
  return t;
}

template <int NPT>
__global__ __launch_bounds__(256) void pool_kth_reseed_kernel(const uint64_t* pool, uint32_t n_lists, uint32_t list_stride, uint32_t ks, uint32_t k,
                                                              const float* delta, uint64_t* tau0) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t ctl[2];
  __shared__ uint32_t total;
  const uint32_t q = blockIdx.x;
  uint64_t keys[NPT];
  const uint32_t valid = block_sum(pool_load<NPT>(pool, q, n_lists, list_stride, ks, keys), &total);
  if (valid < k || k == 0) {  // (block-uniform) fewer than k keys so far: no bound
    if (threadIdx.x == 0) tau0[q] = kKeyInvalid;
    return;
  }
  const uint32_t hi = block_kth_hi<NPT>(keys, k, hist, ctl);
  if (threadIdx.x == 0) {
    const float s = key_score<true>((uint64_t)hi << 32);
    const float lowered = s - 2.0f * delta[q] * 1.01f - fabsf(s) * 1e-6f;  // (sweep.hip reseed_key: pool scores may err by delta either way)
    tau0[q] = lowered == lowered ? make_key<true>(lowered, 0u) : kKeyInvalid;
  }
}

template <int NPT>
__global__ __launch_bounds__(256) void pool_topk_kernel(const uint64_t* pool, uint32_t n_lists, uint32_t list_stride, uint32_t ks, uint32_t k2,
                                                        uint64_t* out_ids, float* out_scores, uint32_t* out_n) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t ctl[2];
  __shared__ uint32_t total;
  __shared__ uint64_t kept[128];
  __shared__ uint32_t nkept;
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  uint64_t keys[NPT];
  const uint32_t valid = block_sum(pool_load<NPT>(pool, q, n_lists, list_stride, ks, keys), &total);
  const uint32_t n = min(valid, k2);
  uint64_t cut = kKeyInvalid;  // keys <= cut are kept (everything valid when the pool holds no more than k2)
  if (valid > k2) cut = block_kth_key<NPT>(keys, k2, hist, ctl);
  if (tid == 0) nkept = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < NPT; j++)
    if (keys[j] != kKeyInvalid && keys[j] <= cut) kept[atomicAdd(&nkept, 1u)] = keys[j];  // (exactly n of them: keys are unique)
  __syncthreads();
  if (tid < n) {
    const uint64_t key = kept[tid];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; j++) rank += kept[j] < key ? 1u : 0u;
    out_ids[(size_t)q * k2 + rank] = (uint64_t)key_row(key);
    out_scores[(size_t)q * k2 + rank] = key_score<true>(key);
  }
  if (tid >= n && tid < k2) {  // the filler of merge_topk (sweep.hip)
    out_ids[(size_t)q * k2 + tid] = ~0ull;
    out_scores[(size_t)q * k2 + tid] = __uint_as_float(0x7FC00000u);
  }
  if (tid == 0) out_n[q] = n;
}

bool pool_select_supported(uint32_t n_lists, uint32_t ks, uint32_t k2) { return (uint64_t)n_lists * ks <= 12288u && k2 <= 128u; }

void launch_pool_kth_reseed(const uint64_t* pool, uint32_t n_lists, uint32_t list_stride, uint32_t ks, uint32_t k, const float* delta, uint64_t* tau0,
                            uint32_t nq, hipStream_t st) {
  const uint32_t total = n_lists * ks;
  if (total <= 1024)
    hipLaunchKernelGGL((pool_kth_reseed_kernel<4>), dim3(nq), dim3(256), 0, st, pool, n_lists, list_stride, ks, k, delta, tau0);
  else if (total <= 3072)
    hipLaunchKernelGGL((pool_kth_reseed_kernel<12>), dim3(nq), dim3(256), 0, st, pool, n_lists, list_stride, ks, k, delta, tau0);
  else
    hipLaunchKernelGGL((pool_kth_reseed_kernel<48>), dim3(nq), dim3(256), 0, st, pool, n_lists, list_stride, ks, k, delta, tau0);
}
void launch_pool_topk(const uint64_t* pool, uint32_t n_lists, uint32_t list_stride, uint32_t ks, uint32_t k2, uint64_t* out_ids, float* out_scores,
                      uint32_t* out_n, uint32_t nq, hipStream_t st) {
  const uint32_t total = n_lists * ks;
  if (total <= 1024)
    hipLaunchKernelGGL((pool_topk_kernel<4>), dim3(nq), dim3(256), 0, st, pool, n_lists, list_stride, ks, k2, out_ids, out_scores, out_n);
  else if (total <= 3072)
    hipLaunchKernelGGL((pool_topk_kernel<12>), dim3(nq), dim3(256), 0, st, pool, n_lists, list_stride, ks, k2, out_ids, out_scores, out_n);
  else
    hipLaunchKernelGGL((pool_topk_kernel<48>), dim3(nq), dim3(256), 0, st, pool, n_lists, list_stride, ks, k2, out_ids, out_scores, out_n);
}

}  // namespace vdb
