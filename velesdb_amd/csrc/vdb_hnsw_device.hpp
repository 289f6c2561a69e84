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
__device__ __forceinline__ void list_insert(lds_vu64* keys, lds_vu8* flags, uint32_t& cnt,
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

// ---- exact visited set in LDS (round 3): open addressing over node id + 1 (0 = empty), linear probing, one LDS compare-
// and-swap per probe.  The HBM bitmap costs a dependent memory round trip per expansion (atomicOr on a line that is rarely
// in L2: 125 KB of bitmap per query in flight) between the neighbour ids and their rows; this costs ~100 cycles.  Exact: a
// node is reported new exactly once.  The kernels stop a query (overflow flag -> the caller re-runs it on the bitmap) before
// the table passes 3/4 of its entries, so a probe sequence always ends.
struct VisSet {
  uint32_t* tab;   // LDS, `mask + 1` entries
  uint32_t mask;   // entries - 1 (a power of two)
  uint32_t shift;  // 32 - log2(entries)
  __device__ __forceinline__ bool test_and_set(uint32_t id) const {  // true: newly inserted (HashSet::insert, graph.rs:499)
    const uint32_t key = id + 1u;
    uint32_t h = (id * 0x9E3779B1u) >> shift;
    for (;;) {
      const uint32_t old = atomicCAS(&tab[h], 0u, key);
      if (old == 0u) return true;
      if (old == key) return false;
      h = (h + 1u) & mask;
    }
  }
  __device__ __forceinline__ void clear(uint32_t tid, uint32_t nthreads) const {  // every thread of the block
    uint4* t4 = reinterpret_cast<uint4*>(tab);
    for (uint32_t i = tid; i < (mask + 1u) / 4u; i += nthreads) t4[i] = uint4{0u, 0u, 0u, 0u};
  }
};

// distance domain of a key: f32 (total-order bits, compared as raw floats like the reference does) or u32 (the
// integer L2^2 of the int8 traversal, dual_precision.rs:336)
template <bool UD>
__device__ __forceinline__ bool key_dist_gt(uint64_t a, uint64_t b) {
  if (UD) return (uint32_t)(a >> 32) > (uint32_t)(b >> 32);
  return key_dist(a) > key_dist(b);
}

// entries past ef stay only up to the last one the termination test could still expand
template <bool UD = false>
__device__ __forceinline__ void list_truncate(lds_vu64* keys, uint32_t& cnt, uint32_t ef, int lane) {
  if (cnt <= ef) return;
  const uint64_t wk = keys[ef - 1];
  uint32_t last = ef - 1;
  for (uint32_t c = ef; c < cnt; c += 64) {
    const uint32_t e = c + lane;
    const bool alive = e < cnt && !key_dist_gt<UD>(keys[e], wk);  // negation of graph.rs:474's raw compare
    const uint64_t mask = __ballot(alive);
    if (mask) last = c + 63u - (uint32_t)__clzll((long long)mask);
  }
  cnt = last + 1;
}

// ---- int8 traversal distances (DualPrecisionHnsw, native/quantization.rs:42-91): integer L2^2 between u8 codes
// = qsq + rsq[row] - 2 * sum(q_i * r_i), every term an exact integer (v_dot4_u32_u8), so the result is the
// reference's u32 bit for bit whatever the summation order.
template <int S>
__device__ __forceinline__ uint32_t add_xor_u32(uint32_t v) {
  if (S == 32) {
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return r[0] + r[1];
  } else if (S == 16) {
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    return r[0] + r[1];
  } else {
    return v + __float_as_uint(lane_xor<S>(__uint_as_float(v)));
  }
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
  v = add_xor_u32<32>(v);
  v = add_xor_u32<16>(v);
  v = add_xor_u32<8>(v);
  v = add_xor_u32<4>(v);
  v = add_xor_u32<2>(v);
  v = add_xor_u32<1>(v);
  return v;
}
struct CodeCtx {
  const uint32_t* codes;  // [n_rows][code_words] packed u8 codes, zero padded
  const uint32_t* rsq;    // [n_rows] sum of squared codes
  uint32_t code_words;    // words per row (multiple of 4)
};
// nb_id[0..m) -> nb_d[0..m) as u32 bit patterns.  qw: this lane's query words (word w = lane + 64*j), QW of them.
template <int QW, int WAVES = 4>
__device__ __forceinline__ void dist_phase_int8(const CodeCtx& c, const uint32_t (&qw)[QW], uint32_t qsq, uint32_t m,
                                                lds_vu32* nb_id, lds_vf32* nb_d, int lane, int wib) {
  constexpr int R = 8;
  for (uint32_t j0 = (uint32_t)wib * R; j0 < m; j0 += WAVES * R) {
    uint32_t dot[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint32_t j = j0 + r < m ? j0 + r : m - 1;
      const uint32_t* p = c.codes + (size_t)nb_id[j] * c.code_words + lane;
      uint32_t w[QW];
#pragma unroll
      for (int t = 0; t < QW; t++) w[t] = ((uint32_t)lane + 64u * t < c.code_words) ? p[64 * t] : 0u;
      uint32_t acc = 0;
#pragma unroll
      for (int t = 0; t < QW; t++) acc = __builtin_amdgcn_udot4(qw[t], w[t], acc, false);
      dot[r] = acc;
    }
#pragma unroll
    for (int r = 0; r < R; r++) dot[r] = wave_sum_u32(dot[r]);
    // lane r finishes row j0 + r
    uint32_t mine = 0;
#pragma unroll
    for (int r = 0; r < R; r++) mine = (lane == r) ? dot[r] : mine;
    const uint32_t j = j0 + (uint32_t)lane;
    if (lane < R && j < m) {
      const uint32_t d = qsq + c.rsq[nb_id[j]] - 2u * mine;
      nb_d[j] = __uint_as_float(d);
    }
  }
}

// ---- the candidate/result list behind one interface ---------------------------------------------
// Two implementations with identical semantics (the reference's two heaps, see hnsw_kernels.hip):
//   CandList<0>   sorted keys + flags in LDS (any capacity that fits LDS)
//   CandList<NS>  NS*64 entries held in REGISTERS of the leader wave: entry e lives in lane e%64, slot e/64,
//                 sorted ascending, empty slots = ~0.  An insert is one DPP wave_shr:1 per slot (the lane
//                 below hands its entry up; mapping verified with tools/probes/wave_shr_check.hip) + selects:
//                 ~12 VALU instructions per slot, no LDS traffic and no loops — the LDS version spends
//                 hundreds of cycles per admitted neighbour, and admission is the serial part of a step.
// External key = (total-order(dist) << 32 | node).  The register form shifts the node up by one bit and keeps
// the "expanded" flag in bit 0 (node ids < 2^31).
constexpr uint32_t kNoIndex = 0xFFFFFFFFu;

template <int NS, bool UD = false>
struct CandList {
  uint64_t k[NS];
  uint32_t cnt;
  static constexpr uint32_t CAP = NS * 64;

  static __device__ __forceinline__ uint64_t enc(uint64_t ext) {
    return (ext & 0xFFFFFFFF00000000ull) | ((ext & 0xFFFFFFFFull) << 1);
  }
  static __device__ __forceinline__ uint64_t dec(uint64_t in) {
    return (in & 0xFFFFFFFF00000000ull) | ((in & 0xFFFFFFFFull) >> 1);
  }
  // lane l receives v of lane l-1; lane 0 receives `carry` (wave-uniform)
  static __device__ __forceinline__ uint64_t shr1(uint64_t v, uint64_t carry) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)carry, (int)(uint32_t)v, 0x138, 0xf, 0xf, false);
    const uint32_t hi =
        (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(carry >> 32), (int)(uint32_t)(v >> 32), 0x138, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
  }
  __device__ __forceinline__ void init(lds_vu64*, lds_vu8*, uint32_t) { reset(); }
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int s = 0; s < NS; s++) k[s] = ~0ull;
    cnt = 0;
  }
  __device__ __forceinline__ uint32_t size() const { return cnt; }
  __device__ __forceinline__ void insert(uint64_t ext, int lane, uint32_t& overflow) {
    const uint64_t key = enc(ext);
    const uint64_t last = readlane64(k[NS - 1], 63);
    uint64_t nk[NS];
#pragma unroll
    for (int s = NS - 1; s >= 0; s--) {
      const uint64_t carry = s > 0 ? readlane64(k[s > 0 ? s - 1 : 0], 63) : 0ull;
      const uint64_t prev = shr1(k[s], carry);
      const bool ge = !(k[s] < key);
      const bool pl = (s == 0 && lane == 0) ? true : (prev < key);
      nk[s] = ge ? (pl ? key : prev) : k[s];
    }
#pragma unroll
    for (int s = 0; s < NS; s++) k[s] = nk[s];
    if (last != ~0ull) {  // the list was full: its last entry (or the key itself, if it is the largest) fell off
      const uint64_t dropped = (last < key) ? key : last;
      if ((dropped & 1ull) == 0) overflow = 1;
    } else {
      cnt += 1;
    }
  }
  // Admission of a whole chunk of evaluated neighbours at once (round 3; the sequential loop of P_Z_ADMIT cost 2.6 us of a
  // 7-us expansion: ~0.4 us per admitted neighbour).  `mask` = the lanes whose neighbour passed the chunk-start test of
  // graph.rs:503 (d < furthest || len < ef), d / nb = the lane's distance and node.  Without exact distance ties the
  // sequential process ends with the ef smallest keys of (list U candidates): a candidate among them is below the furthest
  // distance at its turn whatever came before it (the furthest only falls, and removing one of the ef smallest leaves an
  // ef-th that is strictly larger), and every other one is cut by the truncation behind its insert or rejected outright.
  // So: every accepted key's rank among the old entries (two ballots) and among the candidates, every old entry shifted by
  // the number of candidates below it, one scatter through LDS, one truncate.  Returns false — nothing changed — when the
  // caller has to walk the chunk one by one: an exact tie of distances between a candidate and anything (the reference's
  // strict compare decides those in arrival order; equal distances are neighbours in the merged order, so the test looks at
  // neighbours after the scatter), a NaN distance, or more keys than the list holds.
  __device__ __forceinline__ bool admit_batch(uint64_t mask, float d, uint32_t nb, int lane, uint32_t ef, lds_vu64* scratch_v,
                                              lds_vu8*) {
    const uint32_t n_acc = (uint32_t)__popcll(mask);
    const uint32_t total = cnt + n_acc;
    if (total > CAP) return false;
    const bool mine = ((mask >> lane) & 1ull) != 0;
    if (__ballot(mine && !(d == d))) return false;
    const uint64_t mykey = mine ? enc(make_key<false>(d, nb)) : ~0ull;
    uint32_t shift[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) shift[s] = 0;
    uint32_t mypos = 0;
    for (uint64_t rest = mask; rest; rest &= rest - 1) {
      const int j = __ffsll((long long)rest) - 1;
      const uint64_t kj = readlane64(mykey, j);
      uint32_t r = (uint32_t)__popcll(__ballot(mine && mykey < kj));
#pragma unroll
      for (int s = 0; s < NS; s++)
        if ((uint32_t)s * 64u < cnt) {  // (wave-uniform: slots past the list hold nothing)
          const bool lt = k[s] < kj;    // keys are distinct (a node enters the list once): kj < k[s] is its negation
          r += (uint32_t)__popcll(__ballot(lt));
          shift[s] += lt ? 0u : 1u;     // (empty places, ~0, are shifted too: they are not written)
        }
      if (lane == j) mypos = r;
    }
    // Scatter through LDS — plain accesses (a volatile one waits for its own round trip: 20 of them were 3 us), ordered by
    // the wave's in-order LDS queue and the compiler barriers: old entries behind the candidates below them, candidates at old
    // rank + candidate rank; then everything is read back in one go, and every candidate looks at its two neighbours in the
    // merged order (equal distances are neighbours there).
    __attribute__((address_space(3))) uint64_t* scratch = (__attribute__((address_space(3))) uint64_t*)scratch_v;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < NS; s++)
      if ((uint32_t)s * 64u < cnt && k[s] != ~0ull) scratch[(uint32_t)s * 64u + (uint32_t)lane + shift[s]] = k[s];  // < total <= CAP
    if (mine) scratch[mypos] = mykey;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    uint64_t nk[NS];
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const uint32_t e = (uint32_t)s * 64u + (uint32_t)lane;
      nk[s] = ((uint32_t)s * 64u < total && e < total) ? scratch[e] : ~0ull;
    }
    const uint64_t below = (mine && mypos > 0) ? scratch[mypos - 1] : ~0ull;
    const uint64_t above = (mine && mypos + 1 < total) ? scratch[mypos + 1] : ~0ull;
    asm volatile("" ::: "memory");
    const bool tie = mine && ((below != ~0ull && key_dist(dec(below)) == d) || (above != ~0ull && key_dist(dec(above)) == d));
    if (__ballot(tie)) return false;  // (k, cnt untouched)
#pragma unroll
    for (int s = 0; s < NS; s++) k[s] = nk[s];
    cnt = total;
    truncate(ef, lane);
    return true;
  }
  __device__ __forceinline__ uint32_t first_unexpanded(int) const {
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const uint64_t un = __ballot((k[s] & 1ull) == 0);  // empty slots are ~0: bit 0 set
      if (un) return (uint32_t)s * 64 + (uint32_t)__ffsll((long long)un) - 1;
    }
    return kNoIndex;
  }
  __device__ __forceinline__ uint64_t key_at(uint32_t idx, int) const {  // idx wave-uniform
    // (mask-and-or instead of a select chain: the compiler turns the chain into a dynamically indexed load,
    // which forces the whole list into scratch memory)
    const uint32_t slot = idx >> 6;
    uint64_t v = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) v |= k[s] & ((slot == (uint32_t)s) ? ~0ull : 0ull);
    return dec(readlane64(v, (int)(idx & 63)));
  }
  __device__ __forceinline__ void mark_expanded(uint32_t idx, int lane) {
    const uint32_t slot = idx >> 6, l = idx & 63;
#pragma unroll
    for (int s = 0; s < NS; s++) k[s] |= (slot == (uint32_t)s && (uint32_t)lane == l) ? 1ull : 0ull;
  }
  __device__ __forceinline__ void truncate(uint32_t ef, int lane) {
    if (cnt <= ef) return;
    const uint64_t wk = enc(key_at(ef - 1, lane));
    uint32_t last = ef - 1;
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const uint32_t e = (uint32_t)s * 64 + lane;
      const bool alive = e >= ef && e < cnt && !key_dist_gt<UD>(k[s], wk);  // negation of graph.rs:474's raw compare
      const uint64_t mask = __ballot(alive);
      if (mask) last = (uint32_t)s * 64 + 63u - (uint32_t)__clzll((long long)mask);
    }
    cnt = last + 1;
#pragma unroll
    for (int s = 0; s < NS; s++)
      if ((uint32_t)s * 64 + lane >= cnt) k[s] = ~0ull;
  }
  // external key of entry base + lane (base a multiple of 64, wave-uniform); meaningless beyond size()
  __device__ __forceinline__ uint64_t chunk_key(uint32_t base, int) const {
    const uint32_t slot = base >> 6;
    uint64_t v = 0;
#pragma unroll
    for (int s = 0; s < NS; s++) v |= k[s] & ((slot == (uint32_t)s) ? ~0ull : 0ull);
    return dec(v);
  }
  // copy the first n entries to LDS as external keys (construction: select_neighbors works on LDS arrays)
  __device__ __forceinline__ void dump(lds_vu64* keys, uint32_t n, int lane) const {
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const uint32_t e = (uint32_t)s * 64 + lane;
      if (e < n) keys[e] = dec(k[s]);
    }
  }
};

template <bool UD>
struct CandList<0, UD> {
  lds_vu64* keys;
  lds_vu8* flags;
  uint32_t cnt, cap;
  __device__ __forceinline__ void init(lds_vu64* k_, lds_vu8* f_, uint32_t cap_) {
    keys = k_;
    flags = f_;
    cap = cap_;
    cnt = 0;
  }
  __device__ __forceinline__ void reset() { cnt = 0; }
  __device__ __forceinline__ uint32_t size() const { return cnt; }
  __device__ __forceinline__ void insert(uint64_t ext, int lane, uint32_t& overflow) {
    uint64_t dr;
    uint32_t df;
    list_insert(keys, flags, cnt, cap, ext, lane, dr, df);
    if (dr != kKeyInvalid && df == 0) overflow = 1;  // an unexpanded candidate fell off the list
  }
  __device__ __forceinline__ bool admit_batch(uint64_t, float, uint32_t, int, uint32_t, lds_vu64*, lds_vu8*) { return false; }  // (LDS list: one by one)
  __device__ __forceinline__ uint32_t first_unexpanded(int lane) const {
    for (uint32_t c = 0; c < cnt; c += 64) {
      const uint32_t e = c + lane;
      const uint64_t un = __ballot(e < cnt && flags[e] == 0);
      if (un) return c + (uint32_t)__ffsll((long long)un) - 1;
    }
    return kNoIndex;
  }
  __device__ __forceinline__ uint64_t key_at(uint32_t idx, int) const { return keys[idx]; }
  __device__ __forceinline__ void mark_expanded(uint32_t idx, int lane) {
    if (lane == 0) flags[idx] = 1;
  }
  __device__ __forceinline__ void truncate(uint32_t ef, int lane) { list_truncate<UD>(keys, cnt, ef, lane); }
  __device__ __forceinline__ uint64_t chunk_key(uint32_t base, int lane) const { return keys[base + lane]; }
  __device__ __forceinline__ void dump(lds_vu64*, uint32_t, int) const {}
};

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
template <int METRIC, int CPL, int WAVES = 4, int R = 8>
__device__ __forceinline__ void dist_phase_f32(const DistCtx& a, const float4* q, float qnorm,
                                               const float* qgen, uint32_t m, lds_vu32* nb_id,
                                               lds_vf32* nb_d, int lane, int wib, bool raw = false) {
  constexpr int OP = (METRIC == kEuclidean) ? kOpL2 : kOpDot;
  const int d4 = (int)((a.dim + 3) / 4);
  for (uint32_t j0 = (uint32_t)wib * R; j0 < m; j0 += WAVES * R) {
    float acc[R];
    // the row's norm (cosine) travels with the rows: requested behind the reduction it was a second dependent round trip
    const uint32_t jn = j0 + (uint32_t)(lane & (R - 1));
    float vnorm_early = 1.0f;
    if (METRIC == kCosine && lane < R && jn < m) vnorm_early = a.norms[nb_id[jn]];
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
      const float vnorm = vnorm_early;
      const float s = finish_score<METRIC>(acc[0], qnorm, vnorm);
      // raw = HnswIndex::compute_distance (search.rs:30-38), otherwise DistanceEngine::distance
      nb_d[j] = raw ? s : ((METRIC == kCosine) ? 1.0f - s : ((METRIC == kDot) ? -s : s));
    }
  }
}

template <int METRIC>
__device__ __forceinline__ void dist_phase_bits(const DistCtx& a, const uint32_t* qbits, uint32_t m,
                                                lds_vu32* nb_id, lds_vf32* nb_d, bool raw = false) {
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
