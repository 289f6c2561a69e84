// vdb_device.hpp — device-side building blocks shared by every kernel of libvelesdb_hip.
// gfx950 (MI355X, CDNA4) only: 64-wide wavefronts are assumed everywhere.
//
// CANONICAL ARITHMETIC ("mode C" of the oracle, oracle/vdb_oracle.cpp reduceC/butterfly64)
//   For a pair of f32 vectors of length n
//     * element i belongs to float4-chunk c = i/4; chunk c belongs to lane c % 64
//     * every lane runs ONE fmaf chain from +0.0f over its elements in increasing i
//       (elements i >= n do not exist: tails are predicated, never zero-padded into the chain)
//     * lanes are combined with the xor butterfly 32,16,8,4,2,1 : t[l] = t[l] + t[l^s]
//   f32 add is commutative, so a kernel may run the butterfly "transposed" (64 values per lane
//   in, one finished value per lane out) and still produce the same bits.
//   No fast-math anywhere; fma only where written; sqrt and divide correctly rounded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vdb {

constexpr int kWave = 64;
constexpr uint64_t kKeyInvalid = ~0ull;

enum Metric : int { kCosine = 0, kEuclidean = 1, kDot = 2, kHamming = 3, kJaccard = 4 };

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ float shx(float v, int s) { return __shfl_xor(v, s, 64); }

// ---- cross-lane exchange without the LDS crossbar ------------------------------------------------
// gfx950 has v_permlane32_swap / v_permlane16_swap (swap the odd 32/16-lane rows of one register with
// the even rows of another) and DPP row rotations / quad permutes: every stage of the xor butterfly maps
// onto one of them, so reductions cost no LDS cycles (ds_bpermute shares the LDS pipe with the query
// reads of the sweep).  Lane mappings verified on hardware with tools/probes/xlane_check.hip.
//   swap32(X,Y): X' = {X[0..31], Y[0..31]},  Y' = {X[32..63], Y[32..63]}
//   swap16(X,Y): X' = {X[0..15], Y[0..15], X[32..47], Y[32..47]},  Y' = the other four 16-lane rows
template <int S>
__device__ __forceinline__ float lane_xor(float v) {  // value of lane (l ^ S), S in {8,4,2,1}
  const uint32_t u = __float_as_uint(v);
  uint32_t r;
  if (S == 8) {
    r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xf, 0xf, false);  // row_ror:8
  } else if (S == 4) {
    r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x104, 0xf, 0x5, false);       // row_shl:4 -> banks 0,2
    r = (uint32_t)__builtin_amdgcn_update_dpp((int)r, (int)u, 0x114, 0xf, 0xa, false);  // row_shr:4 -> banks 1,3
  } else if (S == 2) {
    r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
  } else {
    r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
  }
  return __uint_as_float(r);
}
// t[l] + t[l ^ S] in every lane (one butterfly stage)
template <int S>
__device__ __forceinline__ float add_xor(float v) {
  if (S == 32) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else if (S == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
  } else {
    return v + lane_xor<S>(v);
  }
}

// every lane ends with the canonical sum of the 64 per-lane partials
__device__ __forceinline__ float butterfly_all(float t) {
  t = add_xor<32>(t);
  t = add_xor<16>(t);
  t = add_xor<8>(t);
  t = add_xor<4>(t);
  t = add_xor<2>(t);
  t = add_xor<1>(t);
  return t;
}

// Transposed butterfly: N partials per lane in a[0..N); lane l finishes with the canonical 64-lane sum of
// partial index (l mod N) in a[0] (N = 64: index l).  Stage S pairs lane bit S with index bit S: the lane
// keeps one half of its values and receives the partner's copy of the same half.
template <int N, int S>
struct TReduce {
  static __device__ __forceinline__ void run(float* a, int lane) {
    constexpr int H = N / 2;
    if (S == 32 || S == 16) {
#pragma unroll
      for (int i = 0; i < H; i++) {
        // after the swap the first register holds, in every lane, the value the lane keeps and the second
        // the partner's value of the same index (see the row maps above)
        if (S == 32) {
          auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[i + H]), false, false);
          a[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        } else {
          auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a[i]), __float_as_uint(a[i + H]), false, false);
          a[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
      }
    } else {
      const bool up = (lane & S) != 0;
#pragma unroll
      for (int i = 0; i < H; i++) {
        const float keep = up ? a[i + H] : a[i];
        const float send = up ? a[i] : a[i + H];
        a[i] = keep + lane_xor<S>(send);
      }
    }
    TReduce<H, S / 2>::run(a, lane);
  }
};
template <int S>
struct TReduce<1, S> {
  static __device__ __forceinline__ void run(float*, int) {}
};
__device__ __forceinline__ void treduce64(float* a, int lane) { TReduce<64, 32>::run(a, lane); }

// ---- IEEE total order keys (f32::total_cmp; native/ordered_float.rs:31-36) --------------
// asc_key: u32 whose unsigned order equals total_cmp order.
__device__ __forceinline__ uint32_t asc_key(float f) {
  uint32_t b = __float_as_uint(f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float asc_key_inv(uint32_t k) {
  uint32_t b = (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k;
  return __uint_as_float(b);
}
// 64-bit selection key: smaller = better; ties broken by row index ascending.
template <bool HIGHER_IS_BETTER>
__device__ __forceinline__ uint64_t make_key(float score, uint32_t row) {
  uint32_t k = asc_key(score);
  if (HIGHER_IS_BETTER) k = ~k;
  return ((uint64_t)k << 32) | row;
}
template <bool HIGHER_IS_BETTER>
__device__ __forceinline__ float key_score(uint64_t key) {
  uint32_t k = (uint32_t)(key >> 32);
  if (HIGHER_IS_BETTER) k = ~k;
  return asc_key_inv(k);
}
__device__ __forceinline__ uint32_t key_row(uint64_t key) { return (uint32_t)key; }

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int src) {
  uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, src);
  uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
  return ((uint64_t)hi << 32) | lo;
}

// LDS state that waves of a block — or lanes of one wave — hand to each other (the walk / construction kernels' lists and control
// words, the sweeps' top-k lists and counters) is accessed through volatile pointers, so that no access is cached in a register or
// moved across another.  Typed as LDS pointers: a volatile access through a
// GENERIC pointer stays a FLAT instruction (the address-space inference pass leaves volatile accesses alone) — `flat_store … sc0
// sc1` + `s_waitcnt vmcnt(0)`, i.e. the LDS access queues behind every global load in flight (round 4: the walk's neighbour-list prefetch was
// waited for by the first LDS store behind it; a sweep's per-tile threshold read waited for the next tile's rows).  Through these types the same accesses are `ds_read` / `ds_write`.
typedef __attribute__((address_space(3))) void* lds_void_p;
typedef volatile __attribute__((address_space(3))) uint64_t lds_vu64;
typedef volatile __attribute__((address_space(3))) uint32_t lds_vu32;
typedef volatile __attribute__((address_space(3))) float lds_vf32;
typedef volatile __attribute__((address_space(3))) uint8_t lds_vu8;

// ---- wave-owned sorted top-k list in LDS -------------------------------------------------
// list[0..*cnt) ascending u64 keys, capacity k.  All 64 lanes call together with a
// wave-uniform key.  Returns nothing; *cnt is updated by lane 0 semantics (uniform value).
__device__ __forceinline__ void wave_list_insert(lds_vu64* list, uint32_t& cnt, uint32_t k,
                                                 uint64_t key, int lane) {
  if (cnt == k && key >= list[k - 1]) return;  // uniform
  // position = number of elements < key
  uint32_t pos = 0;
  for (uint32_t c = 0; c < cnt; c += 64) {
    uint32_t e = c + lane;
    bool less = e < cnt && list[e] < key;
    pos += (uint32_t)__popcll(__ballot(less));
  }
  const uint32_t newcnt = cnt < k ? cnt + 1 : k;
  // shift [pos, newcnt-1) up by one, top chunk first
  if (newcnt - 1 > pos) {
    const uint32_t span = newcnt - 1 - pos;  // elements to move: pos .. newcnt-2
    for (int32_t c = (int32_t)((span - 1) / 64) * 64; c >= 0; c -= 64) {
      uint32_t e = pos + (uint32_t)c + lane;
      bool mv = e < newcnt - 1;
      uint64_t v = mv ? list[e] : 0;
      if (mv) list[e + 1] = v;
    }
  }
  if (lane == 0) list[pos] = key;
  cnt = newcnt;
}

// ---- block-shared sorted top-k list in LDS, one per query, guarded by a per-list lock ----------
// A wave streams only n_rows / #waves rows, so a private list per wave costs ~k*ln(rows_per_wave/k)
// insertions per wave and query; sharing the list between the waves of a block tightens the threshold
// (#waves in the block) times faster.  The final content is the k smallest keys offered, whatever the
// interleaving (keys are unique), so results stay deterministic.  Lane 0 spins; a wave never holds two
// locks and never reaches a barrier while holding one.
__device__ __forceinline__ void shared_list_offer(lds_vu64* list, lds_vu32* cnt, uint32_t* lock,
                                                  uint32_t k, uint64_t key, int lane) {
  if (*cnt == k && key >= list[k - 1]) return;  // unlocked pre-check: the k-th best only ever improves
  if (lane == 0) {
    while (atomicCAS(lock, 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  uint32_t c = *cnt;
  wave_list_insert(list, c, k, key, lane);  // re-checks against the current k-th under the lock
  if (lane == 0) *cnt = c;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (lane == 0) atomicExch(lock, 0u);
}

// k-th best key of a block-shared list for the UNLOCKED pre-filter of a sweep's epilogue (kKeyInvalid while the list is not full).
// Any value the list has held is a safe bound — the k-th best only ever improves and shared_list_offer re-checks under the lock —
// so these are PLAIN LDS reads: the compiler may issue a tile's reads together and wait once (the volatile form waits for every
// one of them; 16 per 64-row group in the SQ8 sweep).  The caller puts `asm volatile("" ::: "memory")` in front of a tile's reads,
// which is what makes them fresh per tile.
__device__ __forceinline__ uint64_t list_tau_relaxed(lds_vu64* lists, lds_vu32* cnts, uint32_t b, uint32_t k) {
  const __attribute__((address_space(3))) uint32_t* c = (const __attribute__((address_space(3))) uint32_t*)cnts;
  const __attribute__((address_space(3))) uint64_t* l = (const __attribute__((address_space(3))) uint64_t*)lists;
  return c[b] == k ? l[(size_t)b * k + (k - 1)] : kKeyInvalid;
}

// ---- canonical per-lane chains (shared by the sweep, traversal and construction kernels) ----
enum Op : int { kOpDot = 0, kOpL2 = 1 };

template <int OP>
__device__ __forceinline__ float chain4(float acc, const float4& q, const float4& v) {
  if (OP == kOpL2) {
    float d0 = q.x - v.x, d1 = q.y - v.y, d2 = q.z - v.z, d3 = q.w - v.w;
    acc = __builtin_fmaf(d0, d0, acc);
    acc = __builtin_fmaf(d1, d1, acc);
    acc = __builtin_fmaf(d2, d2, acc);
    acc = __builtin_fmaf(d3, d3, acc);
  } else {
    acc = __builtin_fmaf(q.x, v.x, acc);
    acc = __builtin_fmaf(q.y, v.y, acc);
    acc = __builtin_fmaf(q.z, v.z, acc);
    acc = __builtin_fmaf(q.w, v.w, acc);
  }
  return acc;
}
// two chains (two queries) advanced by one element with ONE packed instruction (v_pk_fma_f32): each half is
// the same IEEE fmaf as chain4's, so results are bit-identical to the unpacked form
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__device__ __forceinline__ f32x2 pk_step(f32x2 acc, f32x2 qpair, float v) {
  const f32x2 vv = {v, v};
  if (OP == kOpL2) {
    const f32x2 d = qpair - vv;
    return __builtin_elementwise_fma(d, d, acc);
  }
  return __builtin_elementwise_fma(qpair, vv, acc);
}

// tail chunk: only elements with index < dim exist
template <int OP>
__device__ __forceinline__ float chain4_tail(float acc, const float4& q, const float4& v, int nvalid) {
  const float qa[4] = {q.x, q.y, q.z, q.w};
  const float va[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; e++) {
    if (e < nvalid) {
      if (OP == kOpL2) {
        float d = qa[e] - va[e];
        acc = __builtin_fmaf(d, d, acc);
      } else {
        acc = __builtin_fmaf(qa[e], va[e], acc);
      }
    }
  }
  return acc;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// a - b on this hardware is a + (-b): a NaN in b comes out with its sign flipped (x86 returns the operand's NaN as it
// is).  In the IEEE total order that moves a row with a NaN component from the very end of an ascending ranking to its
// very front.  Scores that went through a subtraction are therefore canonicalised: any NaN -> +qNaN, which is what
// the reference computes for the standard NaN (f32::NAN, 0x7FC00000).  (A negative NaN in the INPUT keeps its sign on
// x86 and does not here: documented deviation)
__device__ __forceinline__ float canon_nan(float x) { return x != x ? __uint_as_float(0x7FC00000u) : x; }

// final score of one (row, query) pair from the canonical sums
template <int METRIC>
__device__ __forceinline__ float finish_score(float sum, float qnorm, float vnorm) {
  if (METRIC == kCosine) {
    // simd_avx512.rs:344-351: dot / (sqrt(na) * sqrt(nb)); 0.0 if either norm is 0
    if (qnorm == 0.0f || vnorm == 0.0f) return 0.0f;
    return sum / (qnorm * vnorm);
  } else if (METRIC == kEuclidean) {
    // canonicalised BEFORE the root: hipcc drops an isnan test on sqrtf's result, and the root returns a NaN operand as it is
    return sqrtf(canon_nan(sum));  // simd_avx512.rs:119-121
  } else {
    return sum;
  }
}

// the same for the bf16 result kernels: half_precision::cosine_similarity on VectorData::BF16 (half_precision.rs:237-254)
// returns 0.0 when a norm is below f32::EPSILON (not only when it is zero); dot_product is the plain sum
constexpr float kHalfNormEps = 1.1920929e-7f;
template <int METRIC>
__device__ __forceinline__ float finish_score_half(float sum, float qnorm, float vnorm) {
  if (METRIC == kCosine) {
    if (qnorm < kHalfNormEps || vnorm < kHalfNormEps) return 0.0f;
    return sum / (qnorm * vnorm);
  }
  return sum;
}

constexpr bool higher_is_better(int metric) {  // core/distance.rs:76-82
  return metric == kCosine || metric == kDot || metric == kJaccard;
}

}  // namespace vdb
