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

#include "vdb_probe_env.hpp"
#include "vdb_device.hpp"
#include "vdb_kernels.hpp"

namespace vdb {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;

constexpr int kG16BM = 256, kG16BN = 256, kG16Waves = 8;
constexpr int kG16Cap = 12;     // candidate buffer entries per query (k <= kGemmBf16MaxK = 10)
constexpr int kG16Queue = 39;   // per-wave queue of finished survivors (and, transiently, raw ones: g16_protocol.inc)
#ifndef VDB_G16_DUMP_MAX_LANES
#define VDB_G16_DUMP_MAX_LANES 6
#endif
constexpr int kG16DumpMaxLanes = VDB_G16_DUMP_MAX_LANES;  // hot lanes of a wave tile up to which the look phase skips the scan (g16_protocol.inc)
static_assert(kG16Queue == 39, "g16_protocol.inc keeps two hot lanes' 64 accumulators in queue entries 8 .. 39");
static_assert(kG16Queue <= 64 && kG16Queue >= 32, "the finish pass takes one queue entry per lane; one hot lane's 32 elements fit");
// LDS map (bytes): two tile buffers (A 32 KiB + B 32 KiB each), candidate buffers, k-th best keys, counters, query norms,
// row-tile norms, flags, per-wave queues (keys + query slots)
constexpr size_t kOffCand = 131072, kOffTauk = kOffCand + (size_t)kG16BN * kG16Cap * 8, kOffCnts = kOffTauk + kG16BN * 8,
                 kOffQn = kOffCnts + kG16BN * 4, kOffVns = kOffQn + kG16BN * 4, kOffFlags = kOffVns + kG16BM * 4,
                 kOffQueue = kOffFlags + 16,
                 kQueueBytes = 368,  // keys [kG16Queue] + one slot for pushes past the end, query columns [kG16Queue]
                
                 kG16Lds = kOffQueue + kG16Waves * kQueueBytes;
static_assert(kG16Lds <= 160 * 1024, "LDS budget");
static_assert((size_t)(kG16Queue + 1) * 8 + kG16Queue + 4 <= 364 + 4 && (size_t)(kG16Queue + 1) * 8 + kG16Queue <= 364, "queue layout");

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
  // result mode (VDB_SEARCH_BRUTE_BF16): [nq] norms of the ROUNDED queries (query_norms_bf16), or nullptr: computed by every block
  const float* qnorms_half;
  unsigned long long* dbg;  // (-DVDB_PP_STAMP variant builds only; nullptr otherwise)
  // WIDE instance (k beyond the candidate buffers: sweep_wide.hip): no list is kept in the block — every row that passes the query's
  // launch-constant bound tau0 is appended to the query's global list [nq][wide_cap] (a counter per query; entries past the capacity
  // are dropped and show as a count above it)
  uint64_t* wide_keys;
  uint32_t* wide_cnt;
  uint32_t wide_cap;
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
// Descriptor of a raw buffer, byte-addressed.  num_records = the bytes that EXIST behind base: a lane whose offset lies past
// them gets zeros (the hardware's range check of raw buffers), which is how the corpus' ragged last tile is read — its rows past
// the end come out as zero rows with zero norms instead of whatever the allocation holds (NaN patterns make every lane of the
// tile hot; the quick test ignores zero norms, a zero product passes no positive bound).
__device__ __forceinline__ i32x4 make_rsrc(const void* base, uint32_t num_records = 0x7FFFFFFFu) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  return i32x4{(int)(uint32_t)b, (int)((uint32_t)(b >> 32) & 0xFFFFu), (int)num_records, 0x00020000};
}
// the same for a base the compiler may have parked in a vector register (loop-carried uniform values under scalar-register
// pressure: an "s" operand fed from one came out as v[0:3] — not an encodable descriptor)
__device__ __forceinline__ i32x4 make_rsrc_uniform(const void* base, uint32_t num_records = 0x7FFFFFFFu) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
  return i32x4{(int)lo, (int)(hi & 0xFFFFu), __builtin_amdgcn_readfirstlane((int)num_records), 0x00020000};
}
// bytes of the row image that exist behind the first byte of k-tile kt of row tile rt (the launch's rows end at n_rows)
__device__ __forceinline__ uint32_t a_tile_records(uint32_t n_rows, uint32_t rt, uint32_t kt, uint64_t row_stride) {
  const uint64_t left = (uint64_t)(n_rows - rt * (uint32_t)kG16BM) * row_stride * 2u - (uint64_t)kt * 128u;
  return (uint32_t)(left < 0x7FFFFFFFull ? left : 0x7FFFFFFFull);
}
__device__ __forceinline__ uint32_t norm_records(uint32_t n_rows, uint32_t rt) {
  return min(n_rows - rt * (uint32_t)kG16BM, (uint32_t)kG16BM) * 4u;
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
  constexpr bool WIDE = false;  // (the epilogue text asks: g16_protocol.inc)
  constexpr int BM = kG16BM, BN = kG16BN, WAVES = kG16Waves, CAP = kG16Cap, QCAP = kG16Queue;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* cand = reinterpret_cast<uint64_t*>(smem + kOffCand);   // [BN][CAP]
  uint64_t* tauk = reinterpret_cast<uint64_t*>(smem + kOffTauk);   // [BN] k-th best key (kKeyInvalid: none)
  uint32_t* cnts = reinterpret_cast<uint32_t*>(smem + kOffCnts);   // [BN]
  float* qn = reinterpret_cast<float*>(smem + kOffQn);             // [BN] query norms (cosine)
  float* vns = reinterpret_cast<float*>(smem + kOffVns);           // [BM] norms of the current row tile
  // [0] again (the latest round marked), [1] need, [2] last-again.  An LDS-typed pointer: a volatile access through a generic pointer is a FLAT instruction,
  // which counts on vmcnt as well — its wait drained every LDS-DMA request in flight, once per flag read
  volatile lds_u32_t* flags = (volatile lds_u32_t*)(lds_ptr_t)(smem + kOffFlags);

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wib >> 2, wq = wib & 3;
  uint64_t* wq_keys = reinterpret_cast<uint64_t*>(smem + kOffQueue + (size_t)wib * kQueueBytes);
  uint8_t* wq_qs = reinterpret_cast<uint8_t*>(wq_keys + QCAP + 1);

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
  } else if (a.qnorms_half) {
    if ((uint32_t)tid < nq_t) qn[tid] = a.qnorms_half[q0 + tid];
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
    const i32x4 ra = make_rsrc(rows_b + ((size_t)ld_rt * BM * a.row_stride + (size_t)ld_kt * 64) * 2, a_tile_records(a.n_rows, ld_rt, ld_kt, a.row_stride)); \
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

#include "g16_compact.inc"
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
      glds_b128(make_rsrc(reinterpret_cast<const unsigned char*>(a.norms) + (size_t)rt * BM * 4, norm_records(a.n_rows, rt)), ln_ * 16u, 0u, lds0 + (uint32_t)kOffVns); \
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
#define VDB_G16_ACC_F(V) (V)
#include "g16_quicktest.inc"
    if constexpr (METRIC == kHamming || METRIC == kJaccard) {
#include "g16_quicktest_bits.inc"
    } else {
#include "g16_quicktest_dense.inc"
    }
#define VDB_G16_ACC_ELEM(X, A) asm volatile("v_mov_b32 %0, %1" : "=v"(X) : "v"(A))
#define VDB_G16_NO_DUMP 1
#include "g16_protocol.inc"
#undef VDB_G16_NO_DUMP
#undef VDB_G16_ACC_ELEM
#undef VDB_G16_ACC_F
  }
#undef VDB_G16_STEP
#undef VDB_G16_ADVANCE
#undef VDB_G16_GLDS
#undef VDB_G16_REQ
#include "g16_writeout.inc"
}

// =====================================================================================================================
// sweep_topk_gemm_bf16_pp — the plain-bf16 instance as a PING-PONG pipeline (round 3).  Same tile (256 rows x 256 queries,
// eight waves as 2 x 4, wave tile 128 x 64), same LDS image (two 64-KiB stages of 128-B lines, swizzle on the DMA's source
// address), same epilogue text (g16_quicktest.inc, g16_protocol.inc) and therefore the same results; what changes is WHEN things happen:
//   * (round 3-4: the accumulators in the accumulation registers, "+a"; round 5: everything in vector registers, see
//     mfma_accv) the wave's other ~100 registers hold fragments: the B fragments of a whole k-tile (64 queries x 64 k = 32 registers) and BOTH halves of the A
//     fragments (2 x 64 rows x 64 k = 2 x 32 registers) are resident, so fragment reads can be placed a phase or more ahead
//     of the products that use them;
//   * a k-tile is four PHASES of 16 products — one quadrant (64 rows x 32 queries x 64 k) of the wave tile each — and the
//     two wave rows run one barrier apart: while waves 0-3 multiply, waves 4-7 read fragments and issue LDS-DMA requests,
//     and vice versa.  A SIMD holds one wave of each row, so its matrix pipe is fed by one wave while the other does
//     everything else (MI355X_MICROARCH.md "Two waves per SIMD"; cdna_hip_programming.md, the 256^2 8-phase template);
//   * the stage buffers are recycled per HALF-TILE (A rows 0-127 / 128-255, B queries 0-127 / 128-255: 16 KiB = two
//     requests per wave), each half-tile of k-tile c + 2 requested into the slot of k-tile c as soon as its last reader is
//     done, 1.5 k-tiles ahead of its first reader; waves wait with vmcnt(6) — three half-tiles stay in flight across every
//     barrier, nothing ever waits for vmcnt(0) in the loop.
// Schedule of k-tile c (buffer c & 1), per wave, "read" = ds_read_b128 into fragment registers, "req" = 2 LDS-DMA requests:
//   phase 1: read B(c)            req B1(c+1)   | products (rows 0-63,   queries 0-31)
//   phase 2: read A rows 64-127(c) req B0(c+2)  | products (rows 0-63,   queries 32-63)
//   phase 3:                      req A0(c+2), vmcnt(6): A(c+1) has landed        | products (rows 64-127, queries 32-63)
//   phase 4: read A rows 0-63(c+1) req A1(c+2), vmcnt(6): B(c+1) has landed       | products (rows 64-127, queries 0-31)
// Every phase is [reads, requests, lgkmcnt(0), barrier, 16 products, barrier]; waves 4-7 start one barrier late.  A buffer
// is read one phase after the wait + barrier that retire its requests, and re-requested at least one barrier after the
// lgkmcnt(0) that retired its last reads (lifetimes: B half-tiles are read in phase 1 only, A half-tile wr in phases 4 (of
// the k-tile before) and 2).  At the end of a row tile the two wave rows re-align (one extra barrier for waves 0-3), run
// the epilogue protocol together, and waves 4-7 fall one barrier behind again.
// What it buys, measured (profiles/r03c_*, r03e_*; 4 M x 768 x 1 024 queries): the matrix pipe is busy 0.56 of the cycles
// instead of 0.51, wave-cycles fall 12 % — and the kernel is 3-4 % faster, because the board sits at its power limit under
// both kernels (1 205-1 230 W; rocm-smi beside a sustained run) and answers the busier pipe with a lower shader clock
// (2.13 -> 1.95 GHz).  Ablations of this kernel (no epilogue 5.03 ms; + no requests 3.75; + no fragment reads 3.39 = 0.74
// of the bf16 peak with the matrix pipe saturated) say the same: under the cap time follows ENERGY — products, bytes
// through L2 -> LDS -> registers — not how well the two overlap.  A second version with two 32-product phases per k-tile
// (half the barriers, requests 4-5 segments ahead) measured within 1 % of this one.  The vendor GEMM (hipBLASLt through
// torch.matmul) on the same box: 1 121-1 230 TFLOP/s for K = 768 shapes, 1 666 for 8192^3 (profiles/r03d_gemm_ceiling.log).
// Idle time is not what the clock governor charges for either: an artificial bubble of 8 x 1 024 cycles per row tile
// (+40 % cycles) cost 9 % of time — the shader clock rose from 2.01 to 2.30 GHz (profiles/r03g_idle_bubble_experiment.log);
// conversely removing real bubbles (the scratch reloads and FLAT flag reads that used to drain the DMA queue in every
// epilogue, the quick test of waves 0-3 moved beside the last products of waves 4-7) changed nothing measurable.
// =====================================================================================================================
// FP4: the same 16-byte fragments hold 32 four-bit values (E2M1: 0, +-1 are all the bit metrics need) instead of 8 bf16 —
// v_mfma_scale_f32_16x16x128_f8f6f4 with both block scales 2^0 (E8M0 byte 0x7F): FOUR times the k-extent per instruction at the
// same issue cost (MI355X_MICROARCH.md: ~10 PFLOP/s dense), f32 accumulators that hold the exact integer dot products (< 2^24).
// Both operands are read with the same slot -> lane mapping, so whatever order the instruction assigns the k-values of a fragment
// is the same for rows and queries: a dot product does not care.  (Round 4 ran this path on v_mfma_i32_16x16x64_i8 first —
// byte images, twice the bytes and twice the products: 0.96 / 1.02 ms per 1 024 queries at 1 M x 768 against 0.72 / 0.82 here,
// profiles/r04p_bit_metrics_fp4_vs_i8.log.)
// Register layout (round 5): accumulators AND fragments in the vector registers (128 + 96 of the wave's 256; no accumulation
// registers at all, so the compiler grants the kernel the whole unified file as vector registers — with ANY "a" operand it splits
// the file 128 / 128 and the 128 accumulators + addresses no longer fit the vector half).  Round 4 kept the accumulators in the
// accumulation registers ("+a"): the epilogue's quick test then paid one v_accvgpr_read per accumulator, 128 per wave and row tile,
// its floor.  Same k-loop, instruction for instruction; measured on one box, same run (profiles/r05b_accv_ab.log): selection
// launches of a 1 024-query step 1.534-1.547 -> 1.463-1.474 ms, step 1.753 -> 1.671 ms, the 10 M bf16 batch 13.87-13.95 -> 13.26-13.30 ms.
template <bool FP4>
__device__ __forceinline__ void mfma_accv(f32x4& c, const f32x4& a, const f32x4& b) {
  if (FP4)
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "+v"(c) : "v"(a), "v"(b), "v"(0x7F7F7F7Fu));
  else
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <bool FP4>
__device__ __forceinline__ void mfma_accv_first(f32x4& c, const f32x4& a, const f32x4& b) {
  if (FP4)
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, 0, %3, %3 op_sel_hi:[0,0,0] cbsz:4 blgp:4" : "=&v"(c) : "v"(a), "v"(b), "v"(0x7F7F7F7Fu));
  else
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
}
// one fragment read: LDS byte address in a vector register + an immediate.  `asm volatile`: it keeps its
// place among the barriers, requests and products of the phase; its completion is the explicit lgkmcnt(0) of pp_barrier_reads_done
// (the compiler does not count it — every use of a fragment sits behind that wait)
template <int OFF>
__device__ __forceinline__ void lds_frag_read(f32x4& dst, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
// the barrier in front of a phase's products: this wave's fragment reads have completed (their stage slots may be
// re-requested by anybody who has passed the barrier); "memory": no LDS access moves across
__device__ __forceinline__ void pp_barrier_reads_done() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void pp_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void pp_wait_dma6() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }

// FP4 instance (Hamming / Jaccard batches on four-bit images, bits_gemm.hip): rows / queries are nibble images addressed through
// the same arguments — row_stride / q_stride / dim in units of TWO bytes, k-tiles of 128 bytes = 256 values — and `norms` /
// `qnorms_half` carry the bit counts |v|, |q| as floats (Hamming: dim); the accumulators hold the exact integer dot products.
// WIDE: the instance for k beyond the candidate buffers (11 .. kWideMaxK, sweep_wide.hip): the same k-loop and quick test under a bound
// that stays what the launch was given (tau0: the k-th best approximate score seen so far, lowered by twice the error bound); the
// epilogue appends every survivor to the query's global list instead of a block-local top-k, and nothing is written out at the end.
template <int METRIC, bool FP4 = false, bool WIDE = false>
__global__ __launch_bounds__(512, 2) void sweep_topk_gemm_bf16_pp(Bf16GemmArgs a) {
  static_assert(FP4 == (METRIC == kHamming || METRIC == kJaccard), "the four-bit instance serves the bit metrics, the bf16 instance Cosine / DotProduct");
  constexpr bool HIB = METRIC != kHamming;  // Cosine / DotProduct / Jaccard: higher is better; Hamming: a distance
  constexpr int BM = kG16BM, BN = kG16BN, WAVES = kG16Waves, CAP = kG16Cap, QCAP = kG16Queue;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* cand = reinterpret_cast<uint64_t*>(smem + kOffCand);   // [BN][CAP]
  uint64_t* tauk = reinterpret_cast<uint64_t*>(smem + kOffTauk);   // [BN] k-th best key (kKeyInvalid: none)
  uint32_t* cnts = reinterpret_cast<uint32_t*>(smem + kOffCnts);   // [BN]
  float* qn = reinterpret_cast<float*>(smem + kOffQn);             // [BN] query norms (cosine)
  float* vns = reinterpret_cast<float*>(smem + kOffVns);           // [BM] norms of the current row tile
  // [0] again (the latest round marked), [1] need, [2] last-again.  An LDS-typed pointer: a volatile access through a generic pointer is a FLAT instruction,
  // which counts on vmcnt as well — its wait drained every LDS-DMA request in flight, once per flag read
  volatile lds_u32_t* flags = (volatile lds_u32_t*)(lds_ptr_t)(smem + kOffFlags);

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wib >> 2, wq = wib & 3;
  uint64_t* wq_keys = reinterpret_cast<uint64_t*>(smem + kOffQueue + (size_t)wib * kQueueBytes);
  uint8_t* wq_qs = reinterpret_cast<uint8_t*>(wq_keys + QCAP + 1);

  // block -> (query tile, row group), as in the kernel above
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
  if (a.qnorms) {
    if ((uint32_t)tid < nq_t) qn[tid] = a.qnorms[q0 + tid];
  } else if (a.qnorms_half) {  // (32 dependent row reads per wave and launch otherwise: 50 us before the first product)
    if ((uint32_t)tid < nq_t) qn[tid] = a.qnorms_half[q0 + tid];
  } else {
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
  const uint32_t ntiles = (a.n_rows + BM - 1) / BM;
  const uint32_t rt_first = a.row_tile0 + g;
  const uint32_t my_tiles = rt_first < ntiles ? (ntiles - rt_first + a.G - 1) / a.G : 0;
  const uint32_t total = my_tiles * a.KT;  // k-tiles of this block's flat (row tile, k-tile) stream; KT >= 2 (host)

  // ---- LDS-DMA requests: wave w, request J fills the 1 KiB row block 8 J + w of the A (J < 4) or B (J >= 4) image; a
  // ---- half-tile is two consecutive J.  Lane (r = l >> 3, p = l & 7) lands at row 8 rb + r, physical slot p, and fetches
  // ---- logical slot p ^ ((row >> 1) & 7) (see the kernel above).
  // (the per-lane offsets live in registers only while the k-loop runs: they are re-derived from the lane id behind every
  // epilogue — VDB_PP_LANE — instead of being carried across it, where they ended in scratch memory whose reloads are VMEM
  // operations: the wait for them drained the whole LDS-DMA queue once per row tile)
  uint32_t voff_a, voff_b;
  int a_rd0, b_rd0, rd_x;
#define VDB_PP_LANE() do { \
    const uint32_t ln_ = lane_now(); \
    const uint32_t st_row_ = (uint32_t)wib * 8u + (ln_ >> 3); \
    const uint32_t st_slot_ = (ln_ & 7u) ^ ((st_row_ >> 1) & 7u); \
    voff_a = st_row_ * (uint32_t)a.row_stride * 2u + st_slot_ * 16u; \
    voff_b = st_row_ * (uint32_t)a.q_stride * 2u + st_slot_ * 16u; \
    const int sw_i_ = (int)((ln_ & 15u) >> 1) & 7; \
    const int rd_off_ = (int)(ln_ & 15u) * 128 + (((int)(ln_ >> 4) ^ sw_i_) & 3) * 16; \
    rd_x = (sw_i_ & 4) << 4; \
    a_rd0 = wr * 128 * 128 + rd_off_; \
    b_rd0 = 32768 + wq * 64 * 128 + rd_off_; \
  } while (0)
  VDB_PP_LANE();
  const uint32_t soff_a = 64u * (uint32_t)a.row_stride * 2u, soff_b = 64u * (uint32_t)a.q_stride * 2u;
  const unsigned char* rows_b = reinterpret_cast<const unsigned char*>(a.rows);
  const unsigned char* queries_b = reinterpret_cast<const unsigned char*>(queries);
  const uint32_t lds0 = (uint32_t)(size_t)(lds_ptr_t)smem;
  const uint32_t lds_w = lds0 + (uint32_t)wib * 1024u;
  // half-tile HALF (0, 1) of the A image of k-tile (RT, KT_) -> stage BUF; the same for B
#define VDB_PP_REQ_A(HALF, RT, KT_, BUF) do { \
    const i32x4 ra_ = make_rsrc_uniform(rows_b + ((size_t)(RT) * BM * a.row_stride + (size_t)(KT_) * 64) * 2, a_tile_records(a.n_rows, (RT), (KT_), a.row_stride)); \
    glds_b128(ra_, voff_a, (uint32_t)(2 * (HALF)) * soff_a, lds_w + (uint32_t)(BUF) * 65536u + (uint32_t)(2 * (HALF)) * 8192u); \
    glds_b128(ra_, voff_a, (uint32_t)(2 * (HALF) + 1) * soff_a, lds_w + (uint32_t)(BUF) * 65536u + (uint32_t)(2 * (HALF) + 1) * 8192u); \
  } while (0)
#define VDB_PP_REQ_B(HALF, KT_, BUF) do { \
    const i32x4 rb_ = make_rsrc_uniform(queries_b + (size_t)(KT_) * 128); \
    glds_b128(rb_, voff_b, (uint32_t)(2 * (HALF)) * soff_b, lds_w + (uint32_t)(BUF) * 65536u + 32768u + (uint32_t)(2 * (HALF)) * 8192u); \
    glds_b128(rb_, voff_b, (uint32_t)(2 * (HALF) + 1) * soff_b, lds_w + (uint32_t)(BUF) * 65536u + 32768u + (uint32_t)(2 * (HALF) + 1) * 8192u); \
  } while (0)

  // ---- fragment reads: lane (i = l & 15, kk = l >> 4) reads slot (4 m + kk) ^ ((i >> 1) & 7) of row i (+ 16 rows per fragment)
#define VDB_PP_READ_A(DST, RF0, BUF) do { \
    const uint32_t ad0_ = lds0 + (uint32_t)(BUF) * 65536u + (uint32_t)(a_rd0 + rd_x), ad1_ = lds0 + (uint32_t)(BUF) * 65536u + (uint32_t)(a_rd0 + (64 ^ rd_x)); \
    lds_frag_read<((RF0) + 0) * 2048>(DST[0][0], ad0_); lds_frag_read<((RF0) + 0) * 2048>(DST[0][1], ad1_); \
    lds_frag_read<((RF0) + 1) * 2048>(DST[1][0], ad0_); lds_frag_read<((RF0) + 1) * 2048>(DST[1][1], ad1_); \
    lds_frag_read<((RF0) + 2) * 2048>(DST[2][0], ad0_); lds_frag_read<((RF0) + 2) * 2048>(DST[2][1], ad1_); \
    lds_frag_read<((RF0) + 3) * 2048>(DST[3][0], ad0_); lds_frag_read<((RF0) + 3) * 2048>(DST[3][1], ad1_); \
  } while (0)
#define VDB_PP_READ_B(BUF) do { \
    const uint32_t bd0_ = lds0 + (uint32_t)(BUF) * 65536u + (uint32_t)(b_rd0 + rd_x), bd1_ = lds0 + (uint32_t)(BUF) * 65536u + (uint32_t)(b_rd0 + (64 ^ rd_x)); \
    lds_frag_read<0 * 2048>(bv[0][0], bd0_); lds_frag_read<0 * 2048>(bv[0][1], bd1_); \
    lds_frag_read<1 * 2048>(bv[1][0], bd0_); lds_frag_read<1 * 2048>(bv[1][1], bd1_); \
    lds_frag_read<2 * 2048>(bv[2][0], bd0_); lds_frag_read<2 * 2048>(bv[2][1], bd1_); \
    lds_frag_read<3 * 2048>(bv[3][0], bd0_); lds_frag_read<3 * 2048>(bv[3][1], bd1_); \
  } while (0)
  // 16 products: rows RF0 .. RF0 + 3 (fragments of AV) x queries T0, T0 + 1 x both 32-deep halves
#define VDB_PP_MFMA(AV, RF0, T0, FIRST) do { \
_Pragma("unroll") \
    for (int m_ = 0; m_ < 2; m_++) \
_Pragma("unroll") \
      for (int rf_ = 0; rf_ < 4; rf_++) \
_Pragma("unroll") \
        for (int t_ = 0; t_ < 2; t_++) { \
          if ((FIRST) && m_ == 0) mfma_accv_first<FP4>(acc[(RF0) + rf_][(T0) + t_], AV[rf_][m_], bv[(T0) + t_][m_]); \
          else mfma_accv<FP4>(acc[(RF0) + rf_][(T0) + t_], AV[rf_][m_], bv[(T0) + t_][m_]); \
        } \
  } while (0)

  f32x4 acc[8][4];
  f32x4 bv[4][2], a0v[4][2], a1v[4][2];

#include "g16_compact.inc"

  uint32_t qcnt = 0;   // entries in this wave's queue (carried over while their candidate buffer is full)
  uint32_t epoch = 0;  // block-uniform: ++ per synchronisation point of the epilogue protocol

  // positions in the flat stream: (rt1, kt1) = k-tile c + 1, (rt2, kt2) = k-tile c + 2, both clamped to the last k-tile (a
  // request past the end re-fetches the last k-tile into a slot nobody reads again: the request count per phase stays
  // uniform, which is what the vmcnt(6) waits count on)
  uint32_t rt1 = rt_first, kt1 = 0, rt2 = rt_first, kt2 = 0, n2 = 0;
#define VDB_PP_NEXT2() do { \
    rt1 = rt2; \
    kt1 = kt2; \
    if (n2 + 1 < total) { \
      n2++; \
      if (++kt2 == a.KT) { \
        kt2 = 0; \
        rt2 += a.G; \
      } \
    } \
  } while (0)
  // (Round 5, measured and rejected: the two requests of a phase issued BEHIND its 16 products — the multiplying wave row pays for
  // them instead of the reading one, waits at vmcnt(4).  3-4 % slower: 1.497-1.504 against 1.454 ms per step's launches, the 10 M bf16
  // batch 13.96-14.09 against 13.11 ms; profiles/r05e_requests_after_mfma_ab.log, tools/probes/pp_requests_after_mfma_experiment.patch.)
  uint32_t c = 0;  // k-tiles done
  if (total) {
    // prologue: k-tile 0 complete (stage 0), of k-tile 1 (stage 1) everything but B1 — the request order of the loop
    VDB_PP_REQ_A(0, rt_first, 0, 0);
    VDB_PP_REQ_A(1, rt_first, 0, 0);
    VDB_PP_REQ_B(0, 0, 0);
    VDB_PP_REQ_B(1, 0, 0);
    VDB_PP_NEXT2();  // (rt2, kt2) = k-tile 1
    VDB_PP_REQ_B(0, kt2, 1);
    VDB_PP_REQ_A(0, rt2, kt2, 1);
    VDB_PP_REQ_A(1, rt2, kt2, 1);
    VDB_PP_NEXT2();  // (rt1, kt1) = k-tile 1, (rt2, kt2) = k-tile 2 (or the last)
    pp_wait_dma6();  // k-tile 0 has landed (this wave's share; the barrier: everybody's)
  }
  pp_barrier();
  // (a static s_setprio 1 for either wave row — MI355X_MICROARCH.md "static priority for the younger half" — measured nothing here:
  // step 1.602 / 1.586-1.594 / 1.588-1.592 ms for none / waves 4-7 / waves 0-3, profiles/r05c_setprio_ab.log; the phases of this
  // loop hold no vector-ALU work the two rows could take from each other)
  if (total) VDB_PP_READ_A(a0v, 0, 0);  // A rows 0-63 of k-tile 0 (in the loop: read in phase 4 of the k-tile before)
  if (wr == 1) pp_barrier();  // waves 4-7 run one barrier behind
  // ---- wall-clock stamps of the phases (variant build only: -DVDB_PP_STAMP=1 = boundaries of every phase, =2 = also behind the
  // ---- issue of a phase's reads / requests; tools/probes/pp_stamp_probe.py).  One wave of each row of block 8 of the bf16 cosine
  // ---- instance reads the shader clock (s_memtime, waited for at once: ~60 cycles each, which the numbers include) and adds the time
  // ---- since its previous stamp to the slot of the segment that just ended: slot = 4 x (phase pair: 0 = phases 1-2, 1 = phases 3-4) +
  // ---- {0 reads + requests + waits + opening barrier, 1 the 16 products, 2 closing barrier, 3 (=2 only) issue of reads + requests};
  // ---- slot 8 = epilogue, 9 = k-tiles counted.
#ifndef VDB_PP_STAMP
#define VDB_PP_STAMP 0
#endif
#ifndef VDB_PP_STAMP_METRIC
#define VDB_PP_STAMP_METRIC kCosine
#endif
#if VDB_PP_STAMP
  const bool st_on = a.dbg != nullptr && blockIdx.x == 8u && (wib == 0 || wib == 4) && METRIC == VDB_PP_STAMP_METRIC;  // (-DVDB_PP_STAMP_METRIC=3: Hamming, the four-bit instance)
  uint32_t st_acc[28] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  uint32_t st_last = 0;
#define VDB_PP_STAMP_AT(SLOT) do { \
    if (st_on) { \
      uint64_t t_; \
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
      st_acc[SLOT] += (uint32_t)t_ - st_last; \
      st_last = (uint32_t)t_; \
    } \
  } while (0)
#define VDB_PP_STAMP_ISSUE(SLOT) do { if (VDB_PP_STAMP >= 2) VDB_PP_STAMP_AT(SLOT); } while (0)
#define VDB_PP_STAMP_COUNT(SLOT) do { if (st_on) st_acc[SLOT] += 1u; } while (0)
#define VDB_PP_STAMP_ADD(SLOT, N) do { if (st_on) st_acc[SLOT] += (uint32_t)(N); } while (0)
#else
#define VDB_PP_STAMP_AT(SLOT) do { } while (0)
#define VDB_PP_STAMP_ISSUE(SLOT) do { } while (0)
#endif
  // The reads-and-requests part of a phase is the longest segment of the loop (stamped timeline, profiles/r05g_*: "issue of reads +
  // requests" 270-356 of a phase's ~870 cycles against 311 for its 16 products): an LDS-DMA request costs the issuing wave ~120-150
  // cycles here whatever surrounds it.  Round 5 tried to move that cost, twice, and both forms are SLOWER: the requests behind the
  // products (3-4 %, profiles/r05e_*) and the odd query columns of a row issuing requests first, the even ones reads first
  // (4-5 %, the issue segment grows to 414 cycles: profiles/r05h_*; tools/probes/pp_stagger_experiment.patch).
#define VDB_PP_LPART(READS, REQS) do { \
    READS; \
    REQS; \
  } while (0)
  // one k-tile: four phases (see the schedule above)
#define VDB_PP_KTILE(FIRST, LAST) do { \
    const uint32_t buf = c & 1u; \
    /* phase 1 */ \
    VDB_PP_LPART(VDB_PP_READ_B(buf), VDB_PP_REQ_B(1, kt1, buf ^ 1u)); \
    if ((FIRST) && wib == 0) /* the row tile's norms (vns was last read in the epilogue before) */ \
      glds_b128(make_rsrc_uniform(reinterpret_cast<const unsigned char*>(a.norms) + (size_t)rt * BM * 4, norm_records(a.n_rows, rt)), lane_now() * 16u, 0u, lds0 + (uint32_t)kOffVns); \
    VDB_PP_STAMP_ISSUE(3); \
    pp_barrier_reads_done(); \
    VDB_PP_STAMP_AT(0); \
    VDB_PP_MFMA(a0v, 0, 0, FIRST); \
    VDB_PP_STAMP_AT(1); \
    pp_barrier(); \
    VDB_PP_STAMP_AT(2); \
    /* phase 2 */ \
    VDB_PP_LPART(VDB_PP_READ_A(a1v, 4, buf), VDB_PP_REQ_B(0, kt2, buf)); \
    VDB_PP_STAMP_ISSUE(3); \
    pp_barrier_reads_done(); \
    VDB_PP_STAMP_AT(0); \
    VDB_PP_MFMA(a0v, 0, 2, FIRST); \
    VDB_PP_STAMP_AT(1); \
    pp_barrier(); \
    VDB_PP_STAMP_AT(2); \
    /* phase 3 */ \
    VDB_PP_REQ_A(0, rt2, kt2, buf); \
    VDB_PP_STAMP_ISSUE(7); \
    pp_wait_dma6(); \
    pp_barrier_reads_done(); \
    VDB_PP_STAMP_AT(4); \
    VDB_PP_MFMA(a1v, 4, 2, FIRST); \
    VDB_PP_STAMP_AT(5); \
    pp_barrier(); \
    VDB_PP_STAMP_AT(6); \
    /* phase 4 */ \
    /* (the last k-tile of a row tile: read behind the epilogue, which gets the registers) */ \
    VDB_PP_LPART(if (!(LAST)) VDB_PP_READ_A(a0v, 0, buf ^ 1u), VDB_PP_REQ_A(1, rt2, kt2, buf)); \
    VDB_PP_STAMP_ISSUE(7); \
    pp_wait_dma6(); \
    pp_barrier_reads_done(); \
    VDB_PP_STAMP_AT(4); \
    VDB_PP_MFMA(a1v, 4, 0, FIRST); \
    VDB_PP_STAMP_AT(5); \
    pp_barrier(); \
    VDB_PP_STAMP_AT(6); \
    VDB_PP_NEXT2(); \
    c++; \
  } while (0)

#if VDB_PP_STAMP
  if (st_on) {  // the first stamp: everything before the loop is nobody's segment
    uint64_t t_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
    st_last = (uint32_t)t_;
  }
#endif
  for (uint32_t rt = rt_first; rt < ntiles; rt += a.G) {
    VDB_PP_KTILE(true, false);
    for (uint32_t kt = 1; kt + 1 < a.KT; kt++) VDB_PP_KTILE(false, false);
    VDB_PP_KTILE(false, true);
    const bool more = c < total;
    // waves 0-3 wait for the last products of waves 4-7: the block is aligned again, and both wave rows run their quick tests at the same
    // time.  (Rounds 3-5 had waves 0-3 test first, beside those last products, and align behind the test: waves 4-7 then waited at their
    // closing barrier for that test and ran their own after it — the two tests of a SIMD in series, ~1 800 + ~2 200 cycles of the
    // stamped timeline.  profiles/r05q_*: step 1.665-1.675 -> 1.646 ms, Jaccard batches 0.722-0.724 -> 0.677-0.687 ms.)
    if (wr == 0) pp_barrier();
    VDB_PP_STAMP_AT(11);
#define VDB_G16_ACC_F(V) (V)
#include "g16_quicktest.inc"
    if constexpr (METRIC == kHamming || METRIC == kJaccard) {
#include "g16_quicktest_bits.inc"
    } else {
#include "g16_quicktest_dense.inc"
    }
    VDB_PP_STAMP_AT(10);  // (epilogue legs, stamped builds: 11 alignment barrier, 10 quick test, 12-24 inside g16_protocol.inc, 8 the rest)
#define VDB_G16_ACC_ELEM(X, A) asm volatile("v_mov_b32 %0, %1" : "=v"(X) : "v"(A))
#include "g16_protocol.inc"
#undef VDB_G16_ACC_ELEM
#undef VDB_G16_ACC_F
    // A rows 0-63 of the next row tile's first k-tile (landed: waited for in phase 3 above).  Unconditional — behind the last
    // row tile it reads a stage nobody uses: a conditional read would keep the OLD fragments alive across the epilogue
    VDB_PP_LANE();
    VDB_PP_READ_A(a0v, 0, c & 1u);
    if (more && wr == 1) pp_barrier();  // ... and waves 4-7 fall one barrier behind again
    VDB_PP_STAMP_AT(8);
  }
#undef VDB_PP_KTILE
#undef VDB_PP_LANE
#undef VDB_PP_NEXT2
#undef VDB_PP_MFMA
#undef VDB_PP_READ_B
#undef VDB_PP_READ_A
#undef VDB_PP_REQ_B
#undef VDB_PP_REQ_A
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // requests past the end must not land in LDS that is no longer ours
#if VDB_PP_STAMP
  if (st_on && lane_now() == 0) {  // [launch-size class][wave row][10]: the LARGEST launch of the batch is what the probe reads
    st_acc[9] = c;
    unsigned long long* d = a.dbg + (size_t)(wib == 0 ? 0 : 1) * 28;
    for (int i = 0; i < 28; i++) d[i] = st_acc[i];
  }
#endif
#include "g16_writeout.inc"
}

// Norms of the rounded queries of a result-mode batch, once per batch instead of once per block: one wave per query, the
// chain the kernels above run when qnorms_half is absent (canonical lane-chain order, as sweep_topk_mfma_bf16).
__global__ __launch_bounds__(256) void query_norms_bf16_kernel(const uint16_t* q16, uint64_t q_stride, float* out, uint32_t nq, uint32_t dim) {
  const uint32_t lane = threadIdx.x & 63u, b = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (b >= nq) return;
  const uint16_t* qp = q16 + (size_t)b * q_stride;
  float nacc = 0.0f;
  for (uint32_t c = lane; c * 4 < dim; c += 64)
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const uint32_t i = c * 4 + e;
      if (i < dim) {
        const float x = __uint_as_float((uint32_t)qp[i] << 16);
        nacc = __builtin_fmaf(x, x, nacc);
      }
    }
  const float n = sqrtf(butterfly_all(nacc));
  if (lane == 0) out[b] = n;
}
void launch_query_norms_bf16(const uint16_t* q16, uint64_t q_stride, float* out, uint32_t nq, uint32_t dim, hipStream_t st) {
  hipLaunchKernelGGL(query_norms_bf16_kernel, dim3((nq + 3) / 4), dim3(256), 0, st, q16, q_stride, out, nq, dim);
}

// From a merged prefix top-k (internal rows + raw scores, merge_topk with ext_ids = nullptr): the query's bound for the
// next launch = k-th best key + 1 (the key itself must still pass `key < tauk`), and — list != nullptr — the top-k as a
// key list in slot 0 of the launch-spanning list array (the seed rows are swept by a different kernel).
template <bool HIB>
__global__ __launch_bounds__(256) void seed_tau_kernel(const uint64_t* ids, const float* scores, const uint32_t* n, uint64_t* tau0,
                                                       uint64_t* list, uint32_t list_stride, uint32_t nq, uint32_t k) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const uint32_t c = min(n[q], k);
  uint64_t t = kKeyInvalid;
  if (c >= k && k > 0) t = make_key<HIB>(scores[(size_t)q * k + k - 1], (uint32_t)ids[(size_t)q * k + k - 1]) + 1ull;
  tau0[q] = t;
  if (list)
    for (uint32_t e = 0; e < k; e++)
      list[(size_t)q * list_stride * k + e] = e < c ? make_key<HIB>(scores[(size_t)q * k + e], (uint32_t)ids[(size_t)q * k + e]) : kKeyInvalid;
}
void launch_seed_tau(const uint64_t* ids, const float* scores, const uint32_t* n, uint64_t* tau0, uint64_t* list,
                     uint32_t list_stride, uint32_t nq, uint32_t k, hipStream_t st, bool hib) {
  if (hib) hipLaunchKernelGGL(seed_tau_kernel<true>, dim3((nq + 255) / 256), dim3(256), 0, st, ids, scores, n, tau0, list, list_stride, nq, k);
  else hipLaunchKernelGGL(seed_tau_kernel<false>, dim3((nq + 255) / 256), dim3(256), 0, st, ids, scores, n, tau0, list, list_stride, nq, k);
}

#if VDB_PP_STAMP
static unsigned long long* g_pp_stamp_buf = nullptr;
}  // namespace vdb
extern "C" int32_t vdb_hip_debug_pp_stamps(unsigned long long* out /* [2][20] */) {
  if (!vdb::g_pp_stamp_buf) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  return hipMemcpy(out, vdb::g_pp_stamp_buf, 448, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -3;
}
namespace vdb {
#endif
// ---- host side -------------------------------------------------------------------------------------------
// (sweep_gemm_bf16_plan, gemm_schedule: vdb_gemm_schedule.hpp — host arithmetic only, checked on the CPU by tests/gemm_schedule_model.cpp)
static_assert(kG16BM == (int)kGemmTileRows && kG16BN == (int)kGemmTileQueries, "the schedule's tile is the kernel's");

// VELESDB_BF16_PP=0: the lock-step kernel (A / B probes)
static bool pingpong_enabled() {
  static const bool on = [] {
    const char* e = probe_env("VELESDB_BF16_PP");
    return !(e && e[0] == '0');
  }();
  return on;
}
template <int METRIC>
static hipError_t launch_g16_pp(const Bf16GemmArgs& a, int blocks, hipStream_t st) {
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_gemm_bf16_pp<METRIC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL((sweep_topk_gemm_bf16_pp<METRIC>), dim3(blocks), dim3(512), kG16Lds, st, a);
  return hipGetLastError();
}

template <int METRIC>
static hipError_t launch_g16_wide(const Bf16GemmArgs& a, int blocks, hipStream_t st) {
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_gemm_bf16_pp<METRIC, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL((sweep_topk_gemm_bf16_pp<METRIC, false, true>), dim3(blocks), dim3(512), kG16Lds, st, a);
  return hipGetLastError();
}

template <int METRIC>
static hipError_t launch_g16_fp4(const Bf16GemmArgs& a, int blocks, hipStream_t st) {
  static bool done = false;
  if (!done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_gemm_bf16_pp<METRIC, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL((sweep_topk_gemm_bf16_pp<METRIC, true>), dim3(blocks), dim3(512), kG16Lds, st, a);
  return hipGetLastError();
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
// WIDE instance over one launch of a schedule (sweep_wide.hip): bounds tau0 (launch-constant), survivors to the queries' global lists
hipError_t launch_sweep_gemm_bf16_wide(int metric, const Bf16GemmPlan& p, const uint16_t* rows16, uint64_t row_stride, const float* norms,
                                       const uint8_t* alive, const uint16_t* queries16, uint64_t q_stride, const uint64_t* tau0,
                                       uint64_t* wide_keys, uint32_t* wide_cnt, uint32_t wide_cap, uint32_t dim, uint32_t nq, hipStream_t st,
                                       const float* qnorms) {
  Bf16GemmArgs a{};
  a.rows = rows16;
  a.norms = norms;
  a.alive = alive;
  a.queries = queries16;
  a.tau0 = tau0;
  a.row_stride = row_stride;
  a.q_stride = q_stride;
  a.n_rows = p.row_hi;
  a.row_tile0 = p.row_lo / kG16BM;
  a.dim = dim;
  a.nq = nq;
  a.k = kGemmBf16MaxK;  // (unused by the WIDE epilogue; the candidate buffers stay empty)
  a.KT = dim / 64;
  a.G = p.G;
  a.nqt = p.nqt;
  a.qper = p.qper;
  a.qnorms = qnorms;
  a.wide_keys = wide_keys;
  a.wide_cnt = wide_cnt;
  a.wide_cap = wide_cap;
  return metric == kCosine ? launch_g16_wide<kCosine>(a, p.blocks, st) : launch_g16_wide<kDot>(a, p.blocks, st);
}

hipError_t launch_sweep_gemm_bf16_glds(int metric, const Bf16GemmPlan& p, const uint16_t* rows16, uint64_t row_stride,
                                       const float* norms, const uint8_t* alive, const uint16_t* queries16, uint64_t q_stride,
                                       const uint64_t* tau0, uint64_t* part_keys, uint32_t list_stride, uint32_t list_off,
                                       uint32_t dim, uint32_t nq, uint32_t k, hipStream_t st, bool split, const float* qnorms,
                                       uint64_t* blk_tau, const float* qnorms_half) {
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
  a.KT = split ? dim / 32 : dim / 64;  // (four-bit instance: dim in units of two bytes => k-tiles of 128 bytes)
  a.G = p.G;
  a.nqt = p.nqt;
  a.qper = p.qper;
  a.qnorms = qnorms;
  a.blk_tau = blk_tau;
  a.qnorms_half = qnorms_half;
#if VDB_PP_STAMP
  {
    static unsigned long long* dbg = [] {
      void* p = nullptr;
      if (hipMalloc(&p, 512) == hipSuccess) (void)hipMemset(p, 0, 512);
      return static_cast<unsigned long long*>(p);
    }();
    g_pp_stamp_buf = dbg;
    a.dbg = dbg;  // (every launch of the batch writes it; the last — the largest — launch's numbers stay)
  }
#endif
  if (metric == kHamming) return launch_g16_fp4<kHamming>(a, p.blocks, st);
  if (metric == kJaccard) return launch_g16_fp4<kJaccard>(a, p.blocks, st);
  if (split)
    return metric == kCosine ? launch_g16<kCosine, true>(a, p.blocks, st) : launch_g16<kDot, true>(a, p.blocks, st);
  if (pingpong_enabled()) return metric == kCosine ? launch_g16_pp<kCosine>(a, p.blocks, st) : launch_g16_pp<kDot>(a, p.blocks, st);
  return metric == kCosine ? launch_g16<kCosine, false>(a, p.blocks, st) : launch_g16<kDot, false>(a, p.blocks, st);
}

}  // namespace vdb
