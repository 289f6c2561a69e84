// sweep_gemm_bf16.hip — the bf16 GEMM-distance sweep for large query batches (BASELINE configs[3]: 10 M x 768 bf16,
// 1 024 queries per batch): half_precision::dot_product / cosine_similarity on VectorData::BF16
// (crates/velesdb-core/src/half_precision.rs:199-255: bf16 operands, f32 accumulation) for a whole batch, with the
// top-k selection of HnswIndex::search_brute_force (index/hnsw/index/search.rs:176-219) fused into the epilogue.
//
// Bound: the bf16 matrix pipe (v_mfma_f32_16x16x32_bf16, 2.5 PFLOP/s dense); algorithmic flop = 2 * rows * dim * queries.
//
// What differs from the f32-structured kernel of sweep_gemm.hip (whose BF16 instance this replaces for big batches):
//   * 256-row x 256-query block tile, eight waves as 2 (rows) x 4 (queries), wave tile 128 x 64 = 8 x 4 accumulator
//     tiles (128 registers), ONE block per CU, k-tiles of 64 bf16 (one 128-B line per row).
//   * NO register staging: both operands go HBM/L2 -> LDS with `global_load_lds_dwordx4` (LDS-DMA).  The LDS image of
//     a wave instruction is lane-linear (8 rows x 128 B), so the bank swizzle of the fragment reads
//     (slot ^= (row >> 1) & 7, conflict-free ds_read_b128) is applied on the SOURCE address: lane (r, p) of an
//     instruction fetches logical slot p ^ ((row >> 1) & 7) of its row — the eight lanes of a row still cover one
//     whole 128-B line.  No ds_write pass, no staging registers; the loads of k-tile i + 1 are in flight while
//     k-tile i is multiplied, two LDS buffers, ONE barrier per k-tile.
//   * an epilogue that costs (almost) nothing once the thresholds are warm: per row tile a lane reduces its 32
//     accumulators per query column with v_max3 and compares the maximum with a conservative per-lane bound
//     (k-th best score of the query x smallest / largest row norm of the lane's rows); only a wave in which some lane
//     passes builds the exact per-element mask.  Survivors are finished exactly (IEEE divide), checked against the
//     query's k-th best KEY and parked in a small per-wave queue; after the k-tile's barrier they are appended to the
//     query's candidate buffer (LDS atomics).  Compaction (rank by counting, as in sweep_gemm.hip) runs only when a
//     buffer went past k, between two barriers, at the next synchronisation point.  A wave whose survivors do not fit
//     its queue makes the whole block repeat the round — correct for any data, fast for data that is not adversarial.
//   * thresholds are SEEDED: the host first runs the 128 x 128 kernel over the first rows of the corpus and hands
//     every block the k-th best key found there (+ 1), so the first row tile of a block passes a few dozen candidates
//     instead of 65 536.
// Arithmetic: products of bf16 values are exact in f32; the order in which one instruction adds its 32 products is not
// documented => parity with the oracle at the stated tolerance (tests/test_gpu_bf16.py), exactly as the kernel replaced.
#include <algorithm>

#include "vdb_device.hpp"
#include "vdb_kernels.hpp"

namespace vdb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

constexpr int kG16BM = 256, kG16BN = 256, kG16Waves = 8;
constexpr int kG16Cap = 12;     // candidate buffer entries per query (k <= kGemmBf16MaxK = 10)
constexpr int kG16Queue = 40;   // per-wave queue of finished survivors
// LDS map (bytes): two tile buffers (A 32 KiB + B 32 KiB each), candidate buffers, k-th best keys, counters, query norms,
// row-tile norms, flags, per-wave queues (keys + query slots)
constexpr size_t kOffCand = 131072, kOffTauk = kOffCand + (size_t)kG16BN * kG16Cap * 8, kOffCnts = kOffTauk + kG16BN * 8,
                 kOffQn = kOffCnts + kG16BN * 4, kOffVns = kOffQn + kG16BN * 4, kOffFlags = kOffVns + kG16BM * 4,
                 kOffQueue = kOffFlags + 16, kQueueBytes = (size_t)kG16Queue * 8 + 48,
                 kG16Lds = kOffQueue + kG16Waves * kQueueBytes;
static_assert(kG16Lds <= 160 * 1024, "LDS budget");

struct Bf16GemmArgs {
  const uint16_t* rows;     // [n_rows + slack][row_stride] bf16
  const float* norms;       // [n_rows + slack] norms of the rounded rows
  const uint8_t* alive;     // nullable
  const uint16_t* queries;  // [nq][q_stride] bf16 (round_queries_bf16)
  const uint64_t* tau0;     // [nq] seed: (k-th best key over a prefix of the rows) + 1, or kKeyInvalid
  uint64_t* part_keys;      // [nq][G][k]
  uint64_t row_stride, q_stride;  // elements
  uint32_t n_rows, dim, nq, k;
  uint32_t KT, G, nqt, qper;
};

// acc[rf][t][r] for a per-lane element index e = (rf * 4 + t) * 4 + r without dynamic register indexing
template <int LO, int N>
struct Acc128 {
  static __device__ __forceinline__ float get(const f32x4 (&acc)[8][4], uint32_t e) {
    const float lo = Acc128<LO, N / 2>::get(acc, e);
    const float hi = Acc128<LO + N / 2, N / 2>::get(acc, e);
    return (e & (uint32_t)(N / 2)) ? hi : lo;
  }
};
template <int LO>
struct Acc128<LO, 1> {
  static __device__ __forceinline__ float get(const f32x4 (&acc)[8][4], uint32_t) { return acc[LO / 16][(LO / 4) % 4][LO % 4]; }
};

template <int METRIC>
__global__ __launch_bounds__(512, 2) void sweep_topk_gemm_bf16_glds(Bf16GemmArgs a) {
  constexpr bool HIB = true;  // Cosine / DotProduct
  constexpr int BM = kG16BM, BN = kG16BN, WAVES = kG16Waves, CAP = kG16Cap, QCAP = kG16Queue;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* cand = reinterpret_cast<uint64_t*>(smem + kOffCand);   // [BN][CAP]
  uint64_t* tauk = reinterpret_cast<uint64_t*>(smem + kOffTauk);   // [BN] k-th best key (kKeyInvalid: none)
  uint32_t* cnts = reinterpret_cast<uint32_t*>(smem + kOffCnts);   // [BN]
  float* qn = reinterpret_cast<float*>(smem + kOffQn);             // [BN] query norms (cosine)
  float* vns = reinterpret_cast<float*>(smem + kOffVns);           // [BM] norms of the current row tile
  volatile uint32_t* flags = reinterpret_cast<volatile uint32_t*>(smem + kOffFlags);  // [0] again, [1] need, [2] last-again

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wib >> 2, wq = wib & 3;
  uint64_t* wq_keys = reinterpret_cast<uint64_t*>(smem + kOffQueue + (size_t)wib * kQueueBytes);
  uint8_t* wq_qs = reinterpret_cast<uint8_t*>(wq_keys + QCAP);

  // block -> (query tile, row group): the query tiles of a row group sit on one XCD in adjacent dispatch slots, run in
  // lock-step and share every row tile through that XCD's L2 (measured: HBM traffic = 1.04 x the corpus)
  const uint32_t bid = blockIdx.x;
  const uint32_t xcd = bid & 7u, slot_id = bid >> 3;
  const uint32_t qt = slot_id % a.nqt;
  const uint32_t g = (slot_id / a.nqt) * 8u + xcd;
  const uint32_t q0 = qt * a.qper;
  const uint32_t nq_t = min(a.qper, a.nq - q0);
  const uint32_t k = a.k;
  const uint16_t* queries = a.queries + (size_t)q0 * a.q_stride;

  if (tid < BN) {
    cnts[tid] = 0;
    tauk[tid] = ((uint32_t)tid < nq_t && a.tau0) ? a.tau0[q0 + tid] : kKeyInvalid;
    qn[tid] = 0.0f;
  }
  if (tid < 4) flags[tid] = 0u;
  __syncthreads();
  {  // norm of the ROUNDED query, canonical lane-chain order (as sweep_topk_mfma_bf16); DotProduct keeps it for the
     // overflow guard of the quick test only
    for (uint32_t b = wib; b < nq_t; b += WAVES) {
      const uint16_t* qp = queries + (size_t)b * a.q_stride;
      float nacc = 0.0f;
      for (uint32_t c = lane; c * 4 < a.dim; c += 64)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint32_t i = c * 4 + e;
          if (i < a.dim) {
            const float x = __uint_as_float((uint32_t)qp[i] << 16);
            nacc = __builtin_fmaf(x, x, nacc);
          }
        }
      const float n = sqrtf(butterfly_all(nacc));
      if (lane == 0) qn[b] = n;
    }
  }
  __syncthreads();
  float qn_t[4];
#pragma unroll
  for (int t = 0; t < 4; t++) qn_t[t] = qn[wq * 64 + t * 16 + (lane & 15)];

  const uint32_t ntiles = (a.n_rows + BM - 1) / BM;
  const uint32_t my_tiles = g < ntiles ? (ntiles - g + a.G - 1) / a.G : 0;
  const uint32_t total = my_tiles * a.KT;

  // ---- LDS-DMA staging: wave w, instruction j fills the 1 KiB row block rb = 8 j + w (rows 8 rb .. 8 rb + 7) ----
  // lane (r = l >> 3, p = l & 7) lands at row 8 rb + r, physical slot p, and fetches logical slot p ^ ((row >> 1) & 7);
  // (row >> 1) & 7 does not depend on j (64 j >> 1 is a multiple of 8): one per-lane offset, four uniform bases
  const uint32_t st_row = (uint32_t)wib * 8u + ((uint32_t)lane >> 3);
  const uint32_t st_slot = ((uint32_t)lane & 7u) ^ ((st_row >> 1) & 7u);
  const uint32_t voff_a = st_row * (uint32_t)a.row_stride * 2u + st_slot * 16u;
  uint32_t voff_b[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t q = st_row + 64u * (uint32_t)j;
    voff_b[j] = (q < nq_t ? q : nq_t - 1) * (uint32_t)a.q_stride * 2u + st_slot * 16u;  // padded slots repeat a query
  }
  const unsigned char* rows_b = reinterpret_cast<const unsigned char*>(a.rows);
  const unsigned char* queries_b = reinterpret_cast<const unsigned char*>(queries);
  uint32_t ld_rt = g, ld_kt = 0;  // (row tile, k-tile) of the NEXT step to request
#define VDB_G16_ISSUE(BUF) do { \
    const unsigned char* base_a = rows_b + ((size_t)ld_rt * BM * a.row_stride + (size_t)ld_kt * 64) * 2; \
    const unsigned char* base_b = queries_b + (size_t)ld_kt * 128; \
    unsigned char* la = smem + (size_t)(BUF) * 65536 + (size_t)wib * 1024; \
_Pragma("unroll") \
    for (int j = 0; j < 4; j++) \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base_a + (size_t)j * 64 * a.row_stride * 2 + voff_a), \
                                       (lds_ptr_t)(la + j * 8192), 16, 0, 0); \
_Pragma("unroll") \
    for (int j = 0; j < 4; j++) \
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base_b + voff_b[j]), (lds_ptr_t)(la + 32768 + j * 8192), 16, 0, 0); \
    if (++ld_kt == a.KT) { \
      ld_kt = 0; \
      ld_rt += a.G; \
    } \
  } while (0)

  f32x4 acc[8][4];
#pragma unroll
  for (int rf = 0; rf < 8; rf++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[rf][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment reads: lane (i = l & 15, kk = l >> 4) reads slot (4 m + kk) ^ ((i >> 1) & 7) of row i (+ 16 rows per fragment)
  const int sw_i = ((lane & 15) >> 1) & 7;
  const int rd_off = (lane & 15) * 128 + ((((lane >> 4) ^ sw_i) & 3) << 4);
  const int rd_x = (sw_i & 4) << 4;
  const int a_rd0 = wr * 128 * 128 + rd_off;           // + buffer base
  const int b_rd0 = 32768 + wq * 64 * 128 + rd_off;

  // ---- compaction of the candidate buffers this wave owns (queries wib, wib + 8, ...): one buffer per 16 lanes ----
  auto compact = [&]() __attribute__((always_inline)) {
    const uint32_t bq = (uint32_t)wib + (uint32_t)WAVES * (uint32_t)lane;  // lane l looks at query wib + 8 l (l < 32)
    const uint32_t cq = (lane < 32 && bq < nq_t) ? cnts[bq] : 0u;
    uint64_t need = __ballot(cq > k);
    const uint32_t li = (uint32_t)lane & 15u, grp = (uint32_t)lane >> 4;
    while (need) {
      int src[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        src[u] = -1;
        if (need) {
          src[u] = __ffsll((long long)need) - 1;
          need &= need - 1;
        }
      }
      const int mysrc = grp == 0 ? src[0] : (grp == 1 ? src[1] : (grp == 2 ? src[2] : src[3]));
      const bool active = mysrc >= 0;
      const uint32_t b = (uint32_t)wib + (uint32_t)WAVES * (uint32_t)(active ? mysrc : 0);
      const uint32_t n = min(cnts[b], (uint32_t)CAP);
      uint64_t* cb = cand + (size_t)b * CAP;
      const bool mine = active && li < n;
      const uint64_t key = mine ? cb[li] : kKeyInvalid;
      uint32_t rank = 0;
#pragma unroll
      for (int j = 0; j < CAP; j++) {
        const uint64_t kj = cb[j];
        rank += ((uint32_t)j < n && kj < key) ? 1u : 0u;
      }
      // all reads of the group precede its writes (same wave: program order; LDS ops complete in order)
      if (mine && rank < k) cb[rank] = key;
      if (mine && rank == k - 1) tauk[b] = key;
      if (active && li == 0) cnts[b] = k;
    }
  };

  uint32_t qcnt = 0;   // entries in this wave's queue (carried over while their candidate buffer is full)
  uint32_t epoch = 0;  // block-uniform: ++ per synchronisation point of the epilogue protocol

  if (total) VDB_G16_ISSUE(0);
  __syncthreads();  // (drains the LDS-DMA: the compiler puts vmcnt(0) in front of a barrier while one is in flight)
  uint32_t kt = 0, rt = g;
  for (uint32_t it = 0; it < total; it++) {
    const int buf = (int)(it & 1u);
    const bool more = it + 1 < total;
    if (more) VDB_G16_ISSUE(buf ^ 1);
    if (kt == 0 && wib == 0)  // the row tile's norms: 1 KiB = one instruction; vns was last read before the previous barrier
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(reinterpret_cast<const unsigned char*>(a.norms) + (size_t)rt * BM * 4 + lane * 16),
                                       (lds_ptr_t)(smem + kOffVns), 16, 0, 0);
    {  // ---- multiply k-tile `it` out of LDS: two 32-deep halves, 12 fragment reads + 32 MFMAs each ----
      const unsigned char* tb = smem + (size_t)buf * 65536;
#pragma unroll
      for (int m = 0; m < 2; m++) {
        float4 av[8], bv[4];
#pragma unroll
        for (int t = 0; t < 4; t++) bv[t] = *reinterpret_cast<const float4*>(tb + b_rd0 + ((m * 64) ^ rd_x) + t * 2048);
#pragma unroll
        for (int rf = 0; rf < 8; rf++) av[rf] = *reinterpret_cast<const float4*>(tb + a_rd0 + ((m * 64) ^ rd_x) + rf * 2048);
#pragma unroll
        for (int rf = 0; rf < 8; rf++)
#pragma unroll
          for (int t = 0; t < 4; t++)
            acc[rf][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[rf]),
                                                                __builtin_bit_cast(bf16x8, bv[t]), acc[rf][t], 0, 0, 0);
      }
    }
    if (++kt < a.KT) {
      __syncthreads();  // every wave is done with this buffer; the next one has landed
      continue;
    }
    // =====================================================================================================
    // last k-tile of a row tile: the accumulators hold 128 rows x 64 queries of dot products per wave
    // =====================================================================================================
    const bool last = !more;
    uint32_t pm[4] = {0u, 0u, 0u, 0u};  // per-lane mask of pending elements, element e at bit 31 - e % 32 of word e / 32
    {
      float cutq[4];
      bool hot = false;
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const uint32_t b = wq * 64 + t * 16 + (lane & 15);
        const uint64_t tkb = tauk[b];
        const float tf = tkb == kKeyInvalid ? __uint_as_float(0xFF800000u) : key_score<HIB>(tkb);
        const float cut = tf - (fabsf(tf) * 1.9073486e-6f + 1e-37f);  // 16-ulp margin
        cutq[t] = b < nq_t ? (METRIC == kCosine ? cut * qn_t[t] : cut) : __uint_as_float(0x7F800000u);
        hot |= !(qn_t[t] < 1e18f);  // a query norm that is NaN / inf / huge: no bound holds
      }
      // norms of the lane's 32 rows (rows 16 rf + 4 (l >> 4) + r of the wave's 128)
      f32x4 vn[8];
#pragma unroll
      for (int rf = 0; rf < 8; rf++) vn[rf] = *reinterpret_cast<const f32x4*>(vns + wr * 128 + rf * 16 + 4 * (lane >> 4));
      float vmin = vn[0][0], vmax = vn[0][0], vsum = 0.0f;
#pragma unroll
      for (int rf = 0; rf < 8; rf++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          vmin = fminf(vmin, vn[rf][r]);
          vmax = fmaxf(vmax, vn[rf][r]);
          vsum += vn[rf][r];  // NaN / inf / overflow-prone norms show up in the sum (min / max drop NaNs)
        }
      hot |= !(vsum < 1e18f);
      // quick test: can ANY of the lane's 32 elements of column t reach the query's k-th best?
#pragma unroll
      for (int t = 0; t < 4; t++) {
        float mx = acc[0][t][0];
#pragma unroll
        for (int rf = 0; rf < 8; rf++)
#pragma unroll
          for (int r = 0; r < 4; r++) mx = fmaxf(mx, acc[rf][t][r]);
        // cosine: score = acc / (|q| |v|) >= cut  <=>  acc >= cutq |v|: the smallest |v| of the lane bounds it for cutq > 0,
        // the largest for cutq <= 0 (rounding slack: the 16-ulp margin of cut)
        const float thr = METRIC == kCosine ? (cutq[t] > 0.0f ? cutq[t] * vmin : cutq[t] * vmax) : cutq[t];
        hot |= !(mx < thr);
      }
      if (__ballot(hot)) {
        // exact per-element filter (one bit per accumulator element); NaN / inf / zero-norm cases compare false = pass
#pragma unroll
        for (int rf = 0; rf < 8; rf++) {
          f32x4 rvn = f32x4{1.f, 1.f, 1.f, 1.f};
          if (METRIC == kCosine)
#pragma unroll
            for (int r = 0; r < 4; r++) rvn[r] = __builtin_amdgcn_rcpf(vn[rf][r]);
#pragma unroll
          for (int t = 0; t < 4; t++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int e = (rf * 4 + t) * 4 + r;
              const bool pass = !((METRIC == kCosine ? acc[rf][t][r] * rvn[r] : acc[rf][t][r]) < cutq[t]);
              pm[e / 32] = (pm[e / 32] << 1) | (pass ? 1u : 0u);
            }
        }
      }
    }
    const uint32_t lane_rl = (uint32_t)(wr * 128 + 4 * (lane >> 4));  // row in tile for rf = r = 0
    const uint32_t lane_b = (uint32_t)(wq * 64 + (lane & 15));         // query in tile for t = 0
    for (;;) {
      ++epoch;
      // ---- (1) drain the masks: every pass finishes each lane's lowest pending element exactly and parks the ones that
      //      beat their query's k-th best key in the wave's queue; stops when the queue is full ----
      bool pend = false;
      for (;;) {
        const uint32_t any = pm[0] | pm[1] | pm[2] | pm[3];
        if (__ballot(any != 0u) == 0) break;
        uint32_t wi = 3, word = pm[3];
#pragma unroll
        for (int w = 2; w >= 0; w--) {
          const bool nz = pm[w] != 0u;
          word = nz ? pm[w] : word;
          wi = nz ? (uint32_t)w : wi;
        }
        const bool has = any != 0u;
        const uint32_t lz = (uint32_t)__builtin_clz(word | 1u);
        const uint32_t e = lz + 32u * wi;
        const float dv = Acc128<0, 128>::get(acc, e);
        const uint32_t r = e & 3u, t = (e >> 2) & 3u, rf = e >> 4;
        const uint32_t rl = lane_rl + rf * 16u + r, b = lane_b + t * 16u;
        const uint32_t row = rt * BM + rl;
        const float score = finish_score<METRIC>(dv, qn[b], METRIC == kCosine ? vns[rl] : 1.0f);
        const uint64_t key = make_key<HIB>(score, row);
        bool take = has & (b < nq_t) & (row < a.n_rows) & (key < tauk[b]);
        if (take && a.alive) take = a.alive[row] != 0;  // soft-deleted rows are filtered where it is rare
        const uint64_t mt = __ballot(take);
        const uint32_t nt = (uint32_t)__popcll(mt);
        if (qcnt + nt > (uint32_t)QCAP) {  // does not fit: nothing of this pass is committed
          pend = true;
          break;
        }
        const uint32_t cleared = word & ~(0x80000000u >> lz);
#pragma unroll
        for (int w = 0; w < 4; w++) pm[w] = (has && wi == (uint32_t)w) ? cleared : pm[w];
        if (take) {
          const uint32_t slot = qcnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mt >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mt, 0u));
          wq_keys[slot] = key;
          wq_qs[slot] = (uint8_t)b;
        }
        qcnt += nt;
      }
      if (pend || (last && qcnt)) flags[0] = epoch;
      __syncthreads();  // sync point `epoch` (first round: also the k-tile's closing barrier)
      // ---- (2) buffers past k since the last sync point: compact them (everybody, between two barriers) ----
      if (flags[1] == epoch) {
        compact();
        __syncthreads();
      }
      // ---- (3) append the queue to the candidate buffers; an entry whose buffer is full stays queued ----
      bool want = false;
      {
        const uint64_t key = (uint32_t)lane < qcnt ? wq_keys[lane] : kKeyInvalid;
        const uint32_t b = (uint32_t)lane < qcnt ? wq_qs[lane] : 0u;
        bool full = false;
        if (key != kKeyInvalid && key < tauk[b]) {
          const uint32_t idx = atomicAdd(&cnts[b], 1u);
          if (idx < (uint32_t)CAP) {
            cand[(size_t)b * CAP + idx] = key;
            want = idx >= k;  // past k: the k-th best can be tightened
          } else {
            full = true;
            want = true;
          }
        }
        const uint64_t mf = __ballot(full);
        qcnt = (uint32_t)__popcll(mf);
        if (full) {
          const uint32_t slot = __builtin_amdgcn_mbcnt_hi((uint32_t)(mf >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mf, 0u));
          wq_keys[slot] = key;  // slot <= lane and every lane has read its entry: no entry is overwritten before it is read
          wq_qs[slot] = (uint8_t)b;
        }
      }
      if (__ballot(want)) flags[1] = epoch + 1;
      bool again = flags[0] == epoch;
      if (last) {  // the block's last row tile: everything still queued must get in before the lists are written out
        if (qcnt) flags[2] = epoch;
        __syncthreads();
        again |= flags[2] == epoch;
      }
      if (!again) break;
    }
#pragma unroll
    for (int rf = 0; rf < 8; rf++)
#pragma unroll
      for (int t = 0; t < 4; t++) acc[rf][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    kt = 0;
    rt += a.G;
  }
#undef VDB_G16_ISSUE
  __syncthreads();
  compact();  // every buffer still holding more than k keys
  __syncthreads();
  for (uint32_t b = wib; b < nq_t; b += WAVES) {
    const uint32_t c = min(cnts[b], k);  // <= k entries, whatever order (the merge kernel scans them all)
    uint64_t* out = a.part_keys + ((size_t)(q0 + b) * a.G + g) * k;
    for (uint32_t e = lane; e < k; e += 64) out[e] = e < c ? cand[(size_t)b * CAP + e] : kKeyInvalid;
  }
}

// k-th best key of a query over the seed rows (+ 1: the key itself must still pass `key < tauk`)
__global__ __launch_bounds__(256) void seed_tau_kernel(const uint64_t* ids, const float* scores, const uint32_t* n, uint64_t* tau0,
                                                       uint32_t nq, uint32_t k) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  uint64_t t = kKeyInvalid;
  if (n[q] >= k && k > 0) t = make_key<true>(scores[(size_t)q * k + k - 1], (uint32_t)ids[(size_t)q * k + k - 1]) + 1ull;
  tau0[q] = t;
}
void launch_seed_tau(const uint64_t* ids, const float* scores, const uint32_t* n, uint64_t* tau0, uint32_t nq, uint32_t k,
                     hipStream_t st) {
  hipLaunchKernelGGL(seed_tau_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, ids, scores, n, tau0, nq, k);
}

// ---- host side -------------------------------------------------------------------------------------------
void sweep_gemm_bf16_plan(uint32_t nq, uint32_t n_rows, int n_cus, Bf16GemmPlan* p) {
  p->nqt = (nq + kG16BN - 1) / kG16BN;
  p->qper = (nq + p->nqt - 1) / p->nqt;
  const uint32_t ntiles = (n_rows + kG16BM - 1) / kG16BM;
  // row groups: whole XCD rounds, never more blocks than the chip holds at once (one block per CU)
  uint32_t G = (uint32_t)std::max(8, n_cus / (int)p->nqt / 8 * 8);
  G = std::min(G, (ntiles + 7) / 8 * 8);
  p->G = G;
  p->blocks = (int)(G * p->nqt);
}

hipError_t launch_sweep_gemm_bf16_glds(int metric, const Bf16GemmPlan& p, const uint16_t* rows16, uint64_t row_stride,
                                       const float* norms, const uint8_t* alive, const uint16_t* queries16, uint64_t q_stride,
                                       const uint64_t* tau0, uint64_t* part_keys, uint32_t n_rows, uint32_t dim, uint32_t nq,
                                       uint32_t k, hipStream_t st) {
  Bf16GemmArgs a{};
  a.rows = rows16;
  a.norms = norms;
  a.alive = alive;
  a.queries = queries16;
  a.tau0 = tau0;
  a.part_keys = part_keys;
  a.row_stride = row_stride;
  a.q_stride = q_stride;
  a.n_rows = n_rows;
  a.dim = dim;
  a.nq = nq;
  a.k = k;
  a.KT = dim / 64;
  a.G = p.G;
  a.nqt = p.nqt;
  a.qper = p.qper;
  static bool done[2] = {false, false};
  const int mi = metric == kCosine ? 0 : 1;
  if (!done[mi]) {
    hipError_t e = metric == kCosine
                       ? hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_gemm_bf16_glds<kCosine>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
                       : hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_gemm_bf16_glds<kDot>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done[mi] = true;
  }
  if (metric == kCosine)
    hipLaunchKernelGGL((sweep_topk_gemm_bf16_glds<kCosine>), dim3(p.blocks), dim3(512), kG16Lds, st, a);
  else
    hipLaunchKernelGGL((sweep_topk_gemm_bf16_glds<kDot>), dim3(p.blocks), dim3(512), kG16Lds, st, a);
  return hipGetLastError();
}

}  // namespace vdb
