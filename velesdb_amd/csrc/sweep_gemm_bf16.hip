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
  uint64_t* part_keys;      // [nq][list_stride][k]: this launch fills lists list_off .. list_off + G - 1
  uint64_t row_stride, q_stride;  // elements
  uint32_t n_rows, dim, nq, k;    // n_rows: end of the row range of this launch
  uint32_t KT, G, nqt, qper;
  uint32_t row_tile0;       // first 256-row tile of the row range
  uint32_t list_stride, list_off;
  // selection stage of the exact f32 search only (sweep_split.hip; SPLIT instance, or the bf16 instance as its first level)
  const float* qnorms;      // [nq] canonical f32 norms of the ORIGINAL queries (cosine); norms = those of the f32 rows
  uint64_t* blk_tau;        // [nq][list_stride]: the bound this block ends with per query (kKeyInvalid: it excluded nothing)
};

// The lane id, re-derived where it is needed: a value computed from threadIdx before the main loop stays live across it,
// and with 128 accumulators + 48 fragment registers the allocator answers every such value with a scratch spill — whose
// reload is a VMEM operation that queues behind the LDS-DMA loads of the next k-tile (measured: it serialises the
// pipeline).  `asm volatile` keeps the two v_mbcnt inside the block that uses them.
__device__ __forceinline__ uint32_t lane_now() {
  uint32_t l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// acc += A x B on the matrix cores, IN PLACE.  Through the builtin hipcc (ROCm 7.2) writes every product of this
// kernel to a second register set (vdst != srcC, the two sets swapping roles every half k-tile): 128 accumulators then
// occupy ~240 registers and the rest of the kernel lives in scratch.  The tied "+v" operand pins vdst = srcC.
// Hazards the compiler no longer sees (cdna_hip_programming.md 5.7): the accumulators are read by vector-ALU code only
// in the epilogue, behind mfma_drain(); two MFMAs on one accumulator are always >= 31 MFMAs apart.
// `asm volatile`: the MFMAs keep their program order among themselves and against mfma_drain() / lane_now(); fragment
// reads (plain LDS loads) still move freely around them.
__device__ __forceinline__ void mfma_bf16_inplace(f32x4& c, const f32x4& a, const f32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// first product of a row tile: srcC = 0, no zeroing pass over the 128 accumulators
__device__ __forceinline__ void mfma_bf16_first(f32x4& c, const f32x4& a, const f32x4& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

// One LDS-DMA instruction (`buffer_load_dwordx4 ... offen lds`: 64 lanes x 16 B land at M0 + 16 lane).  s_nop 3: with the
// s_mov that is 5 wait states between a vector-ALU write of one of the SGPR operands (v_readlane restoring a spilled SGPR)
// and the load — "VALU writes SGPR -> VMEM reads that SGPR" is a hazard the compiler's recogniser cannot see inside inline
// asm (an experiment with asm global loads next to a v_readlane faulted on exactly this).  Pinned in the
// instruction stream (`asm volatile` keeps its place among the MFMAs).  The compiler does not count these on vmcnt: every k-tile step ends with an explicit `s_waitcnt vmcnt(0)` in front of its barrier.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds_b128(const i32x4& rsrc, uint32_t voff, uint32_t soff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 3\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :
               : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr)
               : "m0");  // no "memory" clobber: fragment reads of the OTHER buffer may move across a request
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base) {  // raw buffer, byte-addressed, no bounds in reach
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  return i32x4{(int)(uint32_t)b, (int)((uint32_t)(b >> 32) & 0xFFFFu), 0x7FFFFFFF, 0x00020000};
}
__device__ __forceinline__ void wait_glds() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// SPLIT: the same kernel as the SELECTION stage of the exact f32 sweep.  Rows and queries arrive as split bf16 — per 32
// elements 64 B of `hi` (the value rounded to bf16) followed by 64 B of `lo` (the remainder rounded to bf16), i.e. the same
// 128 B per row and k-tile — and a k-tile contributes hi.hi + hi.lo + lo.hi (three MFMAs per accumulator tile instead of
// two halves of one): x.q to ~2^-16 relative to |x||q| at 3/16 of the f32 matrix pipe's cost.  Scores are approximate;
// sweep_split.hip re-scores the survivors exactly and proves the selection (or sends the query to the exact kernel).
template <int METRIC, bool SPLIT>
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
  if (a.qnorms) {  // selection for the exact f32 search (SPLIT, or plain bf16 as its first level): the caller computed the
                   // canonical norms of the f32 queries — what the exact score divides by
    if ((uint32_t)tid < nq_t) qn[tid] = a.qnorms[q0 + tid];
  } else {  // norm of the ROUNDED query, canonical lane-chain order (as sweep_topk_mfma_bf16); DotProduct keeps it for the
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
  const uint32_t ntiles = (a.n_rows + BM - 1) / BM;  // tiles row_tile0 .. ntiles - 1 belong to this launch
  const uint32_t rt_first = a.row_tile0 + g;
  const uint32_t my_tiles = rt_first < ntiles ? (ntiles - rt_first + a.G - 1) / a.G : 0;
  const uint32_t total = my_tiles * a.KT;

  // ---- LDS-DMA staging: wave w, instruction j fills the 1 KiB row block rb = 8 j + w (rows 8 rb .. 8 rb + 7) ----
  // lane (r = l >> 3, p = l & 7) lands at row 8 rb + r, physical slot p, and fetches logical slot p ^ ((row >> 1) & 7);
  // (row >> 1) & 7 does not depend on j (64 j >> 1 is a multiple of 8).  The tile's base address sits in a buffer
  // descriptor (4 SGPRs, rebuilt per step with scalar adds), the per-lane offset is one VGPR per operand (rebuilt from the
  // lane id, see lane_now), the row block of instruction j a scalar offset.
  // (The query buffer is zero-padded to whole 256-query tiles by the host, so B needs no clamping.)
  const uint32_t soff_a = 64u * (uint32_t)a.row_stride * 2u, soff_b = 64u * (uint32_t)a.q_stride * 2u;
  const unsigned char* rows_b = reinterpret_cast<const unsigned char*>(a.rows);
  const unsigned char* queries_b = reinterpret_cast<const unsigned char*>(queries);
  uint32_t ld_rt = rt_first, ld_kt = 0;  // (row tile, k-tile) of the NEXT step to request
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;  // LDS byte address of the dynamic block (0)
  // operands of the eight requests of one step, for buffer BUF
#define VDB_G16_REQ(BUF) \
    const uint32_t ln_ = lane_now(); \
    const uint32_t st_row = (uint32_t)wib * 8u + (ln_ >> 3); \
    const uint32_t st_slot = (ln_ & 7u) ^ ((st_row >> 1) & 7u); \
    const uint32_t voff_a = st_row * (uint32_t)a.row_stride * 2u + st_slot * 16u; \
    const uint32_t voff_b = st_row * (uint32_t)a.q_stride * 2u + st_slot * 16u; \
    const i32x4 ra = make_rsrc(rows_b + ((size_t)ld_rt * BM * a.row_stride + (size_t)ld_kt * 64) * 2); \
    const i32x4 rb = make_rsrc(queries_b + (size_t)ld_kt * 128); \
    const uint32_t la = lds0 + (uint32_t)(BUF) * 65536u + (uint32_t)wib * 1024u;
  // request J (0..3: row tile, 4..7: query tile)
#define VDB_G16_GLDS(J) do { \
    if ((J) < 4) glds_b128(ra, voff_a, (uint32_t)(J) * soff_a, la + (uint32_t)(J) * 8192u); \
    else glds_b128(rb, voff_b, (uint32_t)((J) - 4) * soff_b, la + 32768u + (uint32_t)((J) - 4) * 8192u); \
  } while (0)
  // after a step's requests: the step after it, unless there is none (the last step is then simply requested again)
#define VDB_G16_ADVANCE() do { \
    if (it + 2 < total && ++ld_kt == a.KT) { \
      ld_kt = 0; \
      ld_rt += a.G; \
    } \
  } while (0)

  f32x4 acc[8][4];

  // ---- compaction of the candidate buffers this wave owns (queries wib, wib + 8, ...): one buffer per 16 lanes ----
  auto compact = [&]() __attribute__((always_inline)) {
    const int lane = (int)lane_now();
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

  // One k-tile step: the current buffer is multiplied — two 32-deep halves, 12 fragment reads + 32 MFMAs each — with the
  // eight LDS-DMA requests of the NEXT step (the other buffer) spread over the FIRST half, one per 4 MFMAs.  The eight
  // waves' 64 requests of a step are 64 KiB through the CU's 64 B/clk vector-memory path = 1 024 cycles during which a
  // wave that has a request pending issues nothing else: sent as one burst at the head of the step they stall every
  // wave at once (measured 0.5 us per step); sent between MFMA groups they hide behind the other wave's MFMAs; sent over
  // the WHOLE step the last ones have no time left to land before the step's barrier (measured 5 % slower than the burst).
  // The row tile's norms travel with its first step.  Fragment reads: lane (i = l & 15, kk = l >> 4) reads slot
  // (4 m + kk) ^ ((i >> 1) & 7) of row i (+ 16 rows per fragment); the address terms are rebuilt from the lane id every step.
#define VDB_G16_STEP(KT_NOW, FIRST) do { \
    const int buf = (int)(it & 1u); \
    VDB_G16_REQ(buf ^ 1) \
    if ((KT_NOW) == 0 && wib == 0) /* vns was last read before the previous barrier */ \
      glds_b128(make_rsrc(reinterpret_cast<const unsigned char*>(a.norms) + (size_t)rt * BM * 4), ln_ * 16u, 0u, lds0 + (uint32_t)kOffVns); \
    const int sw_i = (int)((ln_ & 15u) >> 1) & 7; \
    const int rd_off = (int)(ln_ & 15u) * 128 + (((int)(ln_ >> 4) ^ sw_i) & 3) * 16; \
    const int rd_x = (sw_i & 4) << 4; \
    const int a_rd0 = wr * 128 * 128 + rd_off; \
    const int b_rd0 = 32768 + wq * 64 * 128 + rd_off; \
    const unsigned char* tb = smem + (size_t)buf * 65536; \
    if (!SPLIT) { \
_Pragma("unroll") \
      for (int m = 0; m < 2; m++) { \
        f32x4 av[8], bv[4]; \
_Pragma("unroll") \
        for (int t = 0; t < 4; t++) bv[t] = *reinterpret_cast<const f32x4*>(tb + b_rd0 + ((m * 64) ^ rd_x) + t * 2048); \
_Pragma("unroll") \
        for (int rf = 0; rf < 8; rf++) av[rf] = *reinterpret_cast<const f32x4*>(tb + a_rd0 + ((m * 64) ^ rd_x) + rf * 2048); \
_Pragma("unroll") \
        for (int rf = 0; rf < 8; rf++) { \
_Pragma("unroll") \
          for (int t = 0; t < 4; t++) { \
            if ((FIRST) && m == 0) mfma_bf16_first(acc[rf][t], av[rf], bv[t]); \
            else mfma_bf16_inplace(acc[rf][t], av[rf], bv[t]); \
          } \
          if (m == 0) VDB_G16_GLDS(rf); \
        } \
      } \
    } else { /* half 0 of the line = hi, half 1 = lo; per pair of row fragments: hi.hi, hi.lo, lo.hi (8 MFMAs each) */ \
      f32x4 bh[4], bl[4]; \
_Pragma("unroll") \
      for (int t = 0; t < 4; t++) { \
        bh[t] = *reinterpret_cast<const f32x4*>(tb + b_rd0 + (0 ^ rd_x) + t * 2048); \
        bl[t] = *reinterpret_cast<const f32x4*>(tb + b_rd0 + (64 ^ rd_x) + t * 2048); \
      } \
_Pragma("unroll") \
      for (int rp = 0; rp < 4; rp++) { \
        f32x4 ah[2], al[2]; \
_Pragma("unroll") \
        for (int i = 0; i < 2; i++) { \
          ah[i] = *reinterpret_cast<const f32x4*>(tb + a_rd0 + (0 ^ rd_x) + (rp * 2 + i) * 2048); \
          al[i] = *reinterpret_cast<const f32x4*>(tb + a_rd0 + (64 ^ rd_x) + (rp * 2 + i) * 2048); \
        } \
_Pragma("unroll") \
        for (int i = 0; i < 2; i++) \
_Pragma("unroll") \
          for (int t = 0; t < 4; t++) { \
            if (FIRST) mfma_bf16_first(acc[rp * 2 + i][t], ah[i], bh[t]); \
            else mfma_bf16_inplace(acc[rp * 2 + i][t], ah[i], bh[t]); \
          } \
        VDB_G16_GLDS(rp * 2); \
_Pragma("unroll") \
        for (int i = 0; i < 2; i++) \
_Pragma("unroll") \
          for (int t = 0; t < 4; t++) mfma_bf16_inplace(acc[rp * 2 + i][t], ah[i], bl[t]); \
        VDB_G16_GLDS(rp * 2 + 1); \
_Pragma("unroll") \
        for (int i = 0; i < 2; i++) \
_Pragma("unroll") \
          for (int t = 0; t < 4; t++) mfma_bf16_inplace(acc[rp * 2 + i][t], al[i], bh[t]); \
      } \
    } \
    VDB_G16_ADVANCE(); \
    it++; \
    wait_glds(); /* this wave's requests have landed; the barrier that follows makes that true for every wave */ \
  } while (0)

  uint32_t it = 0;
  if (total) {  // prologue: the first step's tiles
    VDB_G16_REQ(0)
#pragma unroll
    for (int j = 0; j < 8; j++) VDB_G16_GLDS(j);
    if (total > 1 && ++ld_kt == a.KT) {
      ld_kt = 0;
      ld_rt += a.G;
    }
    wait_glds();
  }
  __syncthreads();
  // Row tiles outside, k-tiles inside: the accumulators are a loop-carried value of the INNER loop only, updated in place
  // on its single back edge.  (One flat loop with a `continue` made hipcc copy all 128 of them on every back edge.)
  for (uint32_t rt = rt_first; rt < ntiles; rt += a.G) {
    VDB_G16_STEP(0, true);  // KT >= 2 (host)
    __syncthreads();        // every wave is done with this buffer; the next one has landed
    for (uint32_t kt = 1; kt + 1 < a.KT; kt++) {
      VDB_G16_STEP(kt, false);
      __syncthreads();
    }
    VDB_G16_STEP(a.KT - 1, false);  // the last k-tile: its closing barrier is the first sync point of the epilogue
    const bool more = it < total;
    // =====================================================================================================
    // last k-tile of a row tile: the accumulators hold 128 rows x 64 queries of dot products per wave
    // =====================================================================================================
    const bool last = !more;
    mfma_drain();  // the matrix pipe has written every accumulator before the vector ALU reads one
    const int lane = (int)lane_now();  // (shadows the kernel-scope lane: see lane_now)
    // result mode (VDB_SEARCH_BRUTE_BF16): half_precision.rs's cosine — 0.0 when a norm is below f32::EPSILON; selection
    // mode approximates the exact f32 cosine (0.0 only for a zero norm)
    const bool halfp = a.qnorms == nullptr;
    // Quick test: per query column t a lane reduces its 32 accumulators (rows 16 rf + 4 (l >> 4) + r of the wave's 128)
    // with max and compares with a bound no element that matters can miss.  hm[t] = the lanes that MAY hold a survivor
    // (wave-uniform masks: they are also the state carried through the rounds of the protocol below).
    uint64_t hm[4];
    {
      const float* vnl = vns + wr * 128 + 4 * (lane >> 4);
      float vmin = vnl[0], vmax = vmin, vsum = 0.0f;
#pragma unroll
      for (int rf = 0; rf < 8; rf++) {
        const f32x4 vn = *reinterpret_cast<const f32x4*>(vnl + rf * 16);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          vmin = fminf(vmin, vn[r]);
          vmax = fmaxf(vmax, vn[r]);
          vsum += vn[r];  // NaN / inf / overflow-prone norms show up in the sum (min / max drop NaNs)
        }
      }
      // (result mode: a row norm below f32::EPSILON means score 0.0 whatever the accumulator — no bound holds for the lane)
      const bool force = !(vsum < 1e18f) | (halfp & (METRIC == kCosine) & (vmin < kHalfNormEps));
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const uint32_t b = wq * 64 + t * 16 + (lane & 15);
        const uint64_t tkb = tauk[b];
        const float tf = tkb == kKeyInvalid ? __uint_as_float(0xFF800000u) : key_score<HIB>(tkb);
        const float cut = tf - (fabsf(tf) * 1.9073486e-6f + 1e-37f);  // 16-ulp margin
        const float qnb = qn[b];
        const float cutq = METRIC == kCosine ? cut * qnb : cut;
        float mx = acc[0][t][0];
#pragma unroll
        for (int rf = 0; rf < 8; rf++)
#pragma unroll
          for (int r = 0; r < 4; r++) mx = fmaxf(mx, acc[rf][t][r]);
        // cosine: score = acc / (|q| |v|) >= cut  <=>  acc >= cutq |v|: the smallest |v| of the lane bounds it for cutq > 0,
        // the largest for cutq <= 0 (rounding slack: the 16-ulp margin of cut).  A norm that is NaN / inf / huge (query
        // or row): no bound holds, the lane is looked at.  fmaxf drops NaN accumulators: they only arise from such norms.
        const float thr = METRIC == kCosine ? (cutq > 0.0f ? cutq * vmin : cutq * vmax) : cutq;
        const bool hot = (b < nq_t) & (force | !(qnb < 1e18f) | (halfp & (METRIC == kCosine) & (qnb < kHalfNormEps)) | !(mx < thr));
        hm[t] = __ballot(hot);
      }
    }
    for (;;) {
      ++epoch;
      // ---- (1) look at the hot lanes, one at a time: its 32 elements of the column are spread over lanes 0..31
      //      (v_readlane with a uniform source lane) and finished densely — exact score, key, test against the query's
      //      k-th best key — and the survivors are parked in the wave's queue.  Cost per hot lane ~100 instructions,
      //      whatever the number of survivors; a lane whose survivors do not fit the queue waits for the next round. ----
      bool pend = false;
#define VDB_G16_LOOK(T) \
      while (hm[T] && !pend) { \
        const int src = __ffsll((long long)hm[T]) - 1; \
        float x = 0.0f; \
_Pragma("unroll") \
        for (int i = 0; i < 32; i++) { \
          const float v = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(acc[i / 4][T][i % 4]), src)); \
          x = lane == i ? v : x; \
        } \
        const uint32_t rl = (uint32_t)(wr * 128) + ((uint32_t)lane >> 2) * 16u + 4u * ((uint32_t)src >> 4) + ((uint32_t)lane & 3u); \
        const uint32_t b = (uint32_t)(wq * 64 + (T) * 16) + ((uint32_t)src & 15u); \
        const uint32_t row = rt * BM + (rl & 255u); \
        const float score = halfp ? finish_score_half<METRIC>(x, qn[b], METRIC == kCosine ? vns[rl & 255u] : 1.0f) \
                                  : finish_score<METRIC>(x, qn[b], METRIC == kCosine ? vns[rl & 255u] : 1.0f); \
        const uint64_t key = make_key<HIB>(score, row); \
        bool take = (lane < 32) & (row < a.n_rows) & (key < tauk[b]); \
        if (take && a.alive) take = a.alive[row] != 0; /* soft-deleted rows are filtered where it is rare */ \
        const uint64_t mt = __ballot(take); \
        const uint32_t nt = (uint32_t)__popcll(mt); \
        if (qcnt + nt > (uint32_t)QCAP) { /* does not fit: the lane stays hot for the next round */ \
          pend = true; \
          break; \
        } \
        hm[T] &= hm[T] - 1; \
        if (take) { \
          const uint32_t slot = qcnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(mt >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mt, 0u)); \
          wq_keys[slot] = key; \
          wq_qs[slot] = (uint8_t)b; \
        } \
        qcnt += nt; \
      }
      VDB_G16_LOOK(0)
      VDB_G16_LOOK(1)
      VDB_G16_LOOK(2)
      VDB_G16_LOOK(3)
#undef VDB_G16_LOOK
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
  }
#undef VDB_G16_STEP
#undef VDB_G16_ADVANCE
#undef VDB_G16_GLDS
#undef VDB_G16_REQ
  __syncthreads();
  compact();  // every buffer still holding more than k keys
  __syncthreads();
  const uint32_t lane_o = lane_now();
  for (uint32_t b = wib; b < nq_t; b += WAVES) {
    const uint32_t c = min(cnts[b], k);  // <= k entries, whatever order (the merge kernel scans them all)
    uint64_t* out = a.part_keys + ((size_t)(q0 + b) * a.list_stride + a.list_off + g) * k;
    for (uint32_t e = lane_o; e < k; e += 64) out[e] = e < c ? cand[(size_t)b * CAP + e] : kKeyInvalid;
    if (a.blk_tau && lane_o == 0) a.blk_tau[(size_t)(q0 + b) * a.list_stride + a.list_off + g] = tauk[b];
  }
}

// From a merged prefix top-k (internal rows + raw scores, merge_topk with ext_ids = nullptr): the query's bound for the
// next launch = k-th best key + 1 (the key itself must still pass `key < tauk`), and — list != nullptr — the top-k as a
// key list in slot 0 of the launch-spanning list array (the seed rows are swept by a different kernel).
__global__ __launch_bounds__(256) void seed_tau_kernel(const uint64_t* ids, const float* scores, const uint32_t* n, uint64_t* tau0,
                                                       uint64_t* list, uint32_t list_stride, uint32_t nq, uint32_t k) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const uint32_t c = min(n[q], k);
  uint64_t t = kKeyInvalid;
  if (c >= k && k > 0) t = make_key<true>(scores[(size_t)q * k + k - 1], (uint32_t)ids[(size_t)q * k + k - 1]) + 1ull;
  tau0[q] = t;
  if (list)
    for (uint32_t e = 0; e < k; e++)
      list[(size_t)q * list_stride * k + e] = e < c ? make_key<true>(scores[(size_t)q * k + e], (uint32_t)ids[(size_t)q * k + e]) : kKeyInvalid;
}
void launch_seed_tau(const uint64_t* ids, const float* scores, const uint32_t* n, uint64_t* tau0, uint64_t* list,
                     uint32_t list_stride, uint32_t nq, uint32_t k, hipStream_t st) {
  hipLaunchKernelGGL(seed_tau_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, ids, scores, n, tau0, list, list_stride, nq, k);
}

// ---- host side -------------------------------------------------------------------------------------------
void sweep_gemm_bf16_plan(uint32_t nq, uint32_t row_lo, uint32_t row_hi, int n_cus, Bf16GemmPlan* p) {
  p->nqt = (nq + kG16BN - 1) / kG16BN;
  p->qper = (nq + p->nqt - 1) / p->nqt;
  p->row_lo = row_lo;
  p->row_hi = row_hi;
  const uint32_t ntiles = (row_hi - row_lo + kG16BM - 1) / kG16BM;
  // row groups: whole XCD rounds, never more blocks than the chip holds at once (one block per CU)
  uint32_t G = (uint32_t)std::max(8, n_cus / (int)p->nqt / 8 * 8);
  G = std::min(G, (ntiles + 7) / 8 * 8);
  p->G = G;
  p->blocks = (int)(G * p->nqt);
}

template <int METRIC, bool SPLIT>
static hipError_t launch_g16(const Bf16GemmArgs& a, int blocks, hipStream_t st) {
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_gemm_bf16_glds<METRIC, SPLIT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL((sweep_topk_gemm_bf16_glds<METRIC, SPLIT>), dim3(blocks), dim3(512), kG16Lds, st, a);
  return hipGetLastError();
}

// split == false: rows16 / queries16 are bf16, strides in elements, k-tiles of 64 (dim % 64 == 0, dim >= 128).
// split == true: split-bf16 images (sweep_split.hip), strides = 2 dim, k-tiles of 32 elements (dim % 32 == 0, dim >= 64).
hipError_t launch_sweep_gemm_bf16_glds(int metric, const Bf16GemmPlan& p, const uint16_t* rows16, uint64_t row_stride,
                                       const float* norms, const uint8_t* alive, const uint16_t* queries16, uint64_t q_stride,
                                       const uint64_t* tau0, uint64_t* part_keys, uint32_t list_stride, uint32_t list_off,
                                       uint32_t dim, uint32_t nq, uint32_t k, hipStream_t st, bool split, const float* qnorms,
                                       uint64_t* blk_tau) {
  Bf16GemmArgs a{};
  a.rows = rows16;
  a.norms = norms;
  a.alive = alive;
  a.queries = queries16;
  a.tau0 = tau0;
  a.part_keys = part_keys;
  a.row_stride = row_stride;
  a.q_stride = q_stride;
  a.n_rows = p.row_hi;
  a.row_tile0 = p.row_lo / kG16BM;  // row_lo is a multiple of the tile height (host)
  a.list_stride = list_stride;
  a.list_off = list_off;
  a.dim = dim;
  a.nq = nq;
  a.k = k;
  a.KT = split ? dim / 32 : dim / 64;
  a.G = p.G;
  a.nqt = p.nqt;
  a.qper = p.qper;
  a.qnorms = qnorms;
  a.blk_tau = blk_tau;
  if (split)
    return metric == kCosine ? launch_g16<kCosine, true>(a, p.blocks, st) : launch_g16<kDot, true>(a, p.blocks, st);
  return metric == kCosine ? launch_g16<kCosine, false>(a, p.blocks, st) : launch_g16<kDot, false>(a, p.blocks, st);
}

}  // namespace vdb
