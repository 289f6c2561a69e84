// vdb_block_select.hpp — selection of the k-th smallest key inside one 256-thread block whose keys sit in REGISTERS (device code;
// sweep_wide.hip's lists, pool_select.hip's pools): MSB-first radix select, 8 bits a pass, one 256-bin histogram in LDS — 3 barriers a
// pass, whatever the number of keys.  (The merge kernels of sweep.hip select bit by bit or extract key by key: one barrier per key
// bit or per extracted key — 28 us for the 2 570 keys between two launches of a headline step, profiles/r05final2_*.)
#pragma once
#include "vdb_device.hpp"

namespace vdb {

// one pass: the bin (of the byte at `shift`, among the keys whose word matches `prefix` under `mask`) that holds the rem-th smallest,
// and what is left of rem inside that bin.  word(j) = the 32-bit word of key j the pass works on; live(j) = key j takes part.
template <int NPT, class Word, class Live>
__device__ __forceinline__ void block_select_pass(Word&& word, Live&& live, uint32_t mask, uint32_t prefix, int shift, uint32_t& rem, uint32_t& bin_out,
                                                  uint32_t* hist, uint32_t* ctl) {
  const uint32_t tid = threadIdx.x;
  hist[tid] = 0;
  __syncthreads();
  // (scores of a pool share their leading bytes: in the first passes every key of a wave falls into ONE bin, and 3 000 atomics on
  // one LDS word are served one after the other — ~5 us of a 10-us kernel.  A wave whose taking lanes agree adds their count once.)
#pragma unroll
  for (int j = 0; j < NPT; j++) {
    const uint32_t w = word(j);
    const bool take = live(j) && (w & mask) == prefix;
    const uint32_t bin = (w >> shift) & 255u;
    bool mine = take;
    uint64_t tm = __ballot(mine);
    // (up to two rounds of "the lanes that share the first taker's bin add their count once", the rest one atomic per lane: the first
    // passes see one or two bins per wave, the last ones 64 different bins)
#pragma unroll
    for (int it = 0; it < 2; it++) {
      if (tm == 0ull) break;  // (wave-uniform)
      const int first = __builtin_ctzll(tm);
      const uint32_t lead = (uint32_t)__builtin_amdgcn_readlane((int)bin, first);
      const uint64_t same = __ballot(mine && bin == lead);
      if ((int)(threadIdx.x & 63u) == first) atomicAdd(&hist[lead], (uint32_t)__popcll(same));
      mine = mine && bin != lead;
      tm &= ~same;
    }
    if (mine) atomicAdd(&hist[bin], 1u);
  }
  __syncthreads();
  if (tid < 64) {  // lane l: bins 4 l .. 4 l + 3; inclusive prefix over the lanes
    const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
    const uint32_t s = c0 + c1 + c2 + c3;
    uint32_t incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = __shfl_up(incl, o, 64);
      if ((int)tid >= o) incl += up;
    }
    const uint32_t excl = incl - s;
    if (excl < rem && rem <= incl) {  // exactly one lane (rem <= the number of matching keys)
      uint32_t r = rem - excl, bin = 4 * tid;
      if (r > c0) { r -= c0; bin++; if (r > c1) { r -= c1; bin++; if (r > c2) { r -= c2; bin++; } } }
      ctl[0] = bin;
      ctl[1] = r;
    }
  }
  __syncthreads();
  bin_out = ctl[0];
  rem = ctl[1];
  __syncthreads();  // (hist and ctl are rewritten by the next pass)
}

// The k-th smallest of the block's keys by their HIGH words (the score keys; smaller = better).  Every thread holds NPT keys
// (kKeyInvalid = none); 1 <= k <= the number of valid keys; blockDim = 256; hist = 256 words, ctl = 2 words of LDS.  Returns the high
// word of the k-th smallest key; *rem_out (nullable) = the k-th key's position among the keys that share that high word (1-based).
template <int NPT>
__device__ uint32_t block_kth_hi(const uint64_t (&keys)[NPT], uint32_t k, uint32_t* hist, uint32_t* ctl, uint32_t* rem_out = nullptr) {
  uint32_t prefix = 0, mask = 0, rem = k;
#pragma unroll 1
  for (int shift = 24; shift >= 0; shift -= 8) {
    uint32_t bin;
    block_select_pass<NPT>([&](int j) { return (uint32_t)(keys[j] >> 32); }, [&](int j) { return keys[j] != kKeyInvalid; }, mask, prefix, shift, rem, bin, hist, ctl);
    prefix |= bin << shift;
    mask |= 255u << shift;
  }
  if (rem_out) *rem_out = rem;
  return prefix;
}
// The same for at most 256 keys, one per thread (kKeyInvalid = none; keys are unique): every thread counts the keys smaller than its
// own in LDS — two barriers instead of sixteen.  s256 = 256 keys of LDS, out = one word.  Returns the high word of the k-th smallest key
// (1 <= k <= the number of valid keys).
__device__ __forceinline__ uint32_t block_kth_hi_256(uint64_t key, uint32_t k, uint64_t* s256, uint32_t* out, uint32_t n = 256) {
  const uint32_t tid = threadIdx.x;
  s256[tid] = key;
  __syncthreads();
  uint32_t rank = 0;
#pragma unroll 8
  for (uint32_t j = 0; j < n; j++) rank += s256[j] < key ? 1u : 0u;  // (n = the threads that may hold a valid key: block-uniform)
  if (key != kKeyInvalid && rank + 1 == k) *out = (uint32_t)(key >> 32);
  __syncthreads();
  const uint32_t hi = *out;
  __syncthreads();  // (s256 / out may be rewritten by the caller)
  return hi;
}

// the k-th smallest KEY, all 64 bits (keys are unique: the low word is the row): four more passes over the low words of the keys that
// share the k-th key's high word — skipped when that high word belongs to one key only
template <int NPT>
__device__ uint64_t block_kth_key(const uint64_t (&keys)[NPT], uint32_t k, uint32_t* hist, uint32_t* ctl) {
  uint32_t rem;
  const uint32_t hi = block_kth_hi<NPT>(keys, k, hist, ctl, &rem);
  uint32_t prefix = 0, mask = 0;
#pragma unroll 1
  for (int shift = 24; shift >= 0; shift -= 8) {
    uint32_t bin;
    block_select_pass<NPT>([&](int j) { return (uint32_t)keys[j]; }, [&](int j) { return keys[j] != kKeyInvalid && (uint32_t)(keys[j] >> 32) == hi; }, mask, prefix,
                           shift, rem, bin, hist, ctl);
    prefix |= bin << shift;
    mask |= 255u << shift;
  }
  return ((uint64_t)hi << 32) | prefix;
}

}  // namespace vdb
