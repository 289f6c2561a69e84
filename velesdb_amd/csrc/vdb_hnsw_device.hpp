// vdb_hnsw_device.hpp — device building blocks shared by the traversal (hnsw_kernels.hip) and
// construction (hnsw_build.hip) kernels: the sorted candidate/result list in LDS, the block-wide
// distance phase (DistanceEngine::distance, native/distance.rs:75-85, canonical arithmetic) and small
// wave-uniform helpers.
#pragma once
#include "vdb_device.hpp"
#include "vdb_kernels.hpp"

namespace vdb {

// what the distance phase needs to know about the vector storage
struct DistCtx {
  const float* rows;
  const float* norms;
  const uint32_t* bits;
  uint64_t row_stride;
  uint32_t dim, words;
};

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float rflf(float v) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v)));
}
__device__ __forceinline__ uint64_t lt_mask(int lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ float key_dist(uint64_t key) { return asc_key_inv((uint32_t)(key >> 32)); }

// R per-lane partials per lane in a[0..R); afterwards lane l holds in a[0] the canonical 64-lane sum
// of partial index l % R (stages 32..R are plain butterflies, stages R/2..1 are transposed).
template <int R, int S>
struct PlainStages {  // butterfly stages S, S/2, ... R applied to all R values (every lane keeps every index)
  static __device__ __forceinline__ void run(float* a) {
#pragma unroll
    for (int i = 0; i < R; i++) a[i] = add_xor<S>(a[i]);
    if (S > R) PlainStages<R, (S > R ? S / 2 : S)>::run(a);
  }
};
template <int R>
__device__ __forceinline__ void reduce_rows(float* a, int lane) {
  PlainStages<R, 32>::run(a);
  TReduce<R, R / 2>::run(a, lane);
}

// Sorted list insert with per-entry flags.  Wave-uniform arguments, all 64 lanes participate.
// If the list is at capacity its last entry is dropped and reported (key + flag).
__device__ __forceinline__ void list_insert(volatile uint64_t* keys, volatile uint8_t* flags, uint32_t& cnt,
                                            uint32_t cap, uint64_t key, int lane, uint64_t& dropped,
                                            uint32_t& dropped_flag) {
  dropped = kKeyInvalid;
  dropped_flag = 1;
  uint32_t pos = 0;
  for (uint32_t c = 0; c < cnt; c += 64) {
    const uint32_t e = c + lane;
    const bool less = e < cnt && keys[e] < key;
    pos += (uint32_t)__popcll(__ballot(less));
  }
  if (pos >= cap) {
    dropped = key;
    dropped_flag = 0;
    return;
  }
  if (cnt == cap) {
    dropped = keys[cap - 1];
    dropped_flag = flags[cap - 1];
  }
  const uint32_t newcnt = cnt < cap ? cnt + 1 : cap;
  if (newcnt - 1 > pos) {
    const uint32_t span = newcnt - 1 - pos;
    for (int32_t c = (int32_t)((span - 1) / 64) * 64; c >= 0; c -= 64) {
      const uint32_t e = pos + (uint32_t)c + lane;
      const bool mv = e < newcnt - 1;
      const uint64_t v = mv ? keys[e] : 0;
      const uint8_t f = mv ? flags[e] : (uint8_t)0;
      if (mv) {
        keys[e + 1] = v;
        flags[e + 1] = f;
      }
    }
  }
  if (lane == 0) {
    keys[pos] = key;
    flags[pos] = 0;
  }
  cnt = newcnt;
}

// entries past ef stay only up to the last one the termination test could still expand
__device__ __forceinline__ void list_truncate(volatile uint64_t* keys, uint32_t& cnt, uint32_t ef, int lane) {
  if (cnt <= ef) return;
  const float wd = key_dist(keys[ef - 1]);
  uint32_t last = ef - 1;
  for (uint32_t c = ef; c < cnt; c += 64) {
    const uint32_t e = c + lane;
    const bool alive = e < cnt && !(key_dist(keys[e]) > wd);  // negation of graph.rs:474's raw compare
    const uint64_t mask = __ballot(alive);
    if (mask) last = c + 63u - (uint32_t)__clzll((long long)mask);
  }
  cnt = last + 1;
}

__device__ __forceinline__ float transform_score_dev(int metric, float d) {  // backend_adapter.rs:160-168
  if (metric == kCosine) {
    float s = 1.0f - d;
    if (s < 0.0f) s = 0.0f;
    if (s > 1.0f) s = 1.0f;
    return s;
  }
  if (metric == kDot) return -d;
  return d;
}

// ---- distance evaluation of nb_id[0..m) -> nb_d[0..m): DistanceEngine::distance (native/distance.rs:75-85)
template <int METRIC, int CPL>
__device__ __forceinline__ void dist_phase_f32(const DistCtx& a, const float4* q, float qnorm,
                                               const float* qgen, uint32_t m, volatile uint32_t* nb_id,
                                               volatile float* nb_d, int lane, int wib, bool raw = false) {
  constexpr int OP = (METRIC == kEuclidean) ? kOpL2 : kOpDot;
  constexpr int R = 8;
  const int d4 = (int)((a.dim + 3) / 4);
  for (uint32_t j0 = (uint32_t)wib * R; j0 < m; j0 += 4 * R) {
    float acc[R];
    if (CPL > 0) {
      float4 v[R][CPL > 0 ? CPL : 1];
#pragma unroll
      for (int r = 0; r < R; r++) {
        const uint32_t j = j0 + r < m ? j0 + r : m - 1;
        const float* p = a.rows + (size_t)nb_id[j] * a.row_stride + (size_t)lane * 4;
#pragma unroll
        for (int c = 0; c < CPL; c++) v[r][c] = ld4(p + c * 256);
      }
#pragma unroll
      for (int r = 0; r < R; r++) {
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < CPL; c++) s = chain4<OP>(s, q[c], v[r][c]);
        acc[r] = s;
      }
    } else {
      const float* rp[R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        const uint32_t j = j0 + r < m ? j0 + r : m - 1;
        rp[r] = a.rows + (size_t)nb_id[j] * a.row_stride;
        acc[r] = 0.0f;
      }
      for (int c = lane; c < d4; c += 64) {
        const float4 qq = ld4(qgen + c * 4);
        const int nv = (int)a.dim - c * 4;
#pragma unroll
        for (int r = 0; r < R; r++) {
          const float4 x = ld4(rp[r] + c * 4);
          acc[r] = nv >= 4 ? chain4<OP>(acc[r], qq, x) : chain4_tail<OP>(acc[r], qq, x, nv);
        }
      }
    }
    reduce_rows<R>(acc, lane);
    const uint32_t j = j0 + (uint32_t)(lane & (R - 1));
    if (lane < R && j < m) {
      float vnorm = 1.0f;
      if (METRIC == kCosine) vnorm = a.norms[nb_id[j]];
      const float s = finish_score<METRIC>(acc[0], qnorm, vnorm);
      // raw = HnswIndex::compute_distance (search.rs:30-38), otherwise DistanceEngine::distance
      nb_d[j] = raw ? s : ((METRIC == kCosine) ? 1.0f - s : ((METRIC == kDot) ? -s : s));
    }
  }
}

template <int METRIC>
__device__ __forceinline__ void dist_phase_bits(const DistCtx& a, const uint32_t* qbits, uint32_t m,
                                                volatile uint32_t* nb_id, volatile float* nb_d, bool raw = false) {
  const uint32_t W = a.words;
  for (uint32_t t = threadIdx.x; t < m; t += 256) {
    const uint4* p = reinterpret_cast<const uint4*>(a.bits + (size_t)nb_id[t] * W);
    uint32_t ham = 0, inter = 0, uni = 0;
    for (uint32_t w = 0; w < W; w += 4) {
      const uint4 x = p[w / 4];
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint32_t qq = qbits[w + e];
        if (METRIC == kHamming) {
          ham += __popc(xs[e] ^ qq);
        } else {
          inter += __popc(xs[e] & qq);
          uni += __popc(xs[e] | qq);
        }
      }
    }
    if (METRIC == kHamming) {
      nb_d[t] = (float)ham;  // simd_explicit.rs:234-287 on the exact re-encoding bit = (x > 0.5)
    } else {
      const float sim = (uni == 0) ? 1.0f : (float)inter / (float)uni;  // simd_explicit.rs:431-442
      nb_d[t] = raw ? sim : 1.0f - sim;                                  // native/distance.rs:83
    }
  }
}


}  // namespace vdb
