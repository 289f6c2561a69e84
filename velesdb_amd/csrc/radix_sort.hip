// radix_sort.hip — hand-written stable LSD radix sort of (u64 key, u64 value) pairs on the device.
//
// Used by the batch-synchronous graph construction (hnsw_build.hip): the back-link requests of a batch — key = (layer, target
// node, batch index), value = (source, distance) — must reach the link kernel grouped by (layer, target) and, inside a group, in
// the order the reference's sequential insert would have produced them (batch index = insertion order; add_bidirectional_
// connection, native/graph.rs:592-639).  north_star asks for hand-authored radix / bitonic sorting; this replaces the one vendor
// primitive the library used (rocprim::radix_sort_pairs).
//
// One pass = three launches over tiles of 2 048 pairs: (1) per-tile histogram of the pass's digit (<= 8 bits) into a
// digit-major table, (2) exclusive scan of the table by one block, (3) stable scatter — a tile is walked in index order, 256
// pairs per round, one wave after the other: a lane's rank among the lanes of its wave with the same digit comes from one ballot
// per digit bit (the match-any idiom), the running offset of every digit sits in LDS.  The caller lists the digit positions that
// can differ (hnsw_build.hip: only as many bits of the batch index / node id as the batch / the graph has), so a 1 M-node build
// sorts 35-bit keys in five passes.  Bound: none that matters — a batch's requests are a few hundred thousand pairs, the passes
// are launch-bound (~15 launches of a few microseconds per batch against a 10 ms insert kernel).
#include <algorithm>

#include "vdb_device.hpp"
#include "vdb_kernels.hpp"

namespace vdb {

constexpr uint32_t kRsTile = 2048, kRsThreads = 256, kRsRounds = kRsTile / kRsThreads;

__global__ __launch_bounds__(256) void rs_hist_kernel(const uint64_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t mask,
                                                      uint32_t* __restrict__ hist, uint32_t nb) {
  __shared__ uint32_t h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kRsTile;
#pragma unroll
  for (uint32_t r = 0; r < kRsRounds; r++) {
    const uint32_t i = base + r * kRsThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & mask], 1u);
  }
  __syncthreads();
  if (threadIdx.x <= mask) hist[(size_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

// exclusive scan of `total` counters in place (one block: a chunk per thread, then a scan of the chunk sums through LDS)
__global__ __launch_bounds__(1024) void rs_scan_kernel(uint32_t* __restrict__ hist, uint32_t total) {
  __shared__ uint32_t sums[1024];
  const uint32_t t = threadIdx.x;
  const uint32_t per = (total + 1023u) / 1024u;
  const uint32_t lo = min(t * per, total), hi = min(lo + per, total);
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; i++) s += hist[i];
  sums[t] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
    const uint32_t v = t >= d ? sums[t - d] : 0u;
    __syncthreads();
    sums[t] += v;
    __syncthreads();
  }
  uint32_t run = sums[t] - s;  // exclusive prefix of this thread's chunk
  for (uint32_t i = lo; i < hi; i++) {
    const uint32_t c = hist[i];
    hist[i] = run;
    run += c;
  }
}

__global__ __launch_bounds__(256) void rs_scatter_kernel(const uint64_t* __restrict__ keys_in, const uint64_t* __restrict__ vals_in,
                                                         uint64_t* __restrict__ keys_out, uint64_t* __restrict__ vals_out, uint32_t n,
                                                         uint32_t shift, uint32_t mask, uint32_t nbits, const uint32_t* __restrict__ hist,
                                                         uint32_t nb) {
  __shared__ uint32_t off[256];  // where the next pair of each digit goes
  const uint32_t t = threadIdx.x, lane = t & 63u, wib = t >> 6;
  if (t <= mask) off[t] = hist[(size_t)t * nb + blockIdx.x];
  __syncthreads();
  const uint32_t base = blockIdx.x * kRsTile;
  for (uint32_t r = 0; r < kRsRounds; r++) {
    const uint32_t i = base + r * kRsThreads + t;
    const bool valid = i < n;
    const uint64_t key = valid ? keys_in[i] : 0ull;
    const uint64_t val = valid ? vals_in[i] : 0ull;
    const uint32_t d = (uint32_t)(key >> shift) & mask;
    // the lanes of this wave that hold the same digit (invalid lanes match nobody)
    uint64_t eq = __ballot(valid);
    for (uint32_t b = 0; b < nbits; b++) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __ballot(bit);
      eq &= bit ? m : ~m;
    }
    const uint32_t rank = (uint32_t)__popcll(eq & ((1ull << lane) - 1ull));
    const uint32_t cnt = (uint32_t)__popcll(eq);
    for (uint32_t w = 0; w < 4; w++) {  // index order: wave 0's 64 pairs, then wave 1's, ...
      if (wib == w && valid) {
        const uint32_t p = off[d] + rank;  // (every lane reads before the group's first lane writes: one wave, program order)
        keys_out[p] = key;
        vals_out[p] = val;
        if (rank == 0) off[d] += cnt;
      }
      __syncthreads();
    }
  }
}

size_t radix_sort_scratch_bytes(uint32_t n) {
  const uint32_t nb = (n + kRsTile - 1) / kRsTile;
  return (size_t)256 * std::max<uint32_t>(nb, 1) * 4;
}

// Sorts n pairs by the listed digits, least significant first (stable); the result lands in (keys_b, vals_b) when the number of
// passes is odd, in (keys_a, vals_a) when it is even: *result_in_b tells.  scratch >= radix_sort_scratch_bytes(n).
hipError_t radix_sort_pairs_u64(uint64_t* keys_a, uint64_t* vals_a, uint64_t* keys_b, uint64_t* vals_b, uint32_t n, const RadixDigit* digits,
                                int n_digits, void* scratch, bool* result_in_b, hipStream_t st) {
  *result_in_b = false;
  if (n == 0) return hipSuccess;
  const uint32_t nb = (n + kRsTile - 1) / kRsTile;
  uint32_t* hist = static_cast<uint32_t*>(scratch);
  uint64_t *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
  for (int p = 0; p < n_digits; p++) {
    const uint32_t nbits = digits[p].bits, mask = (1u << nbits) - 1u;
    hipLaunchKernelGGL(rs_hist_kernel, dim3(nb), dim3(256), 0, st, ki, n, digits[p].shift, mask, hist, nb);
    hipLaunchKernelGGL(rs_scan_kernel, dim3(1), dim3(1024), 0, st, hist, (mask + 1u) * nb);
    hipLaunchKernelGGL(rs_scatter_kernel, dim3(nb), dim3(256), 0, st, ki, vi, ko, vo, n, digits[p].shift, mask, nbits, hist, nb);
    std::swap(ki, ko);
    std::swap(vi, vo);
    *result_in_b = !*result_in_b;
  }
  return hipGetLastError();
}

}  // namespace vdb
