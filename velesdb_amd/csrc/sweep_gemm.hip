// sweep_gemm.hip — exact Cosine / DotProduct sweep for LARGE query batches, structured as a tiled f32 GEMM on
// the matrix cores with the top-k selection fused into the epilogue (HnswIndex::search_brute_force,
// index/hnsw/index/search.rs:176-219, over a whole batch of queries; simd_explicit.rs:583-634
// batch_similarity_top_k is the reference's only batched-query x corpus routine).
//
// Why a second matrix-core kernel: sweep_topk_mfma_f32 (sweep.hip) streams each row straight from HBM into the A
// operand of ONE wave, so every B fragment a wave reads from LDS feeds a single 16-row tile and the kernel runs
// out of steam at 48 queries per corpus pass (0.90 ms, 82 TFLOP/s, both HBM and the f32 matrix pipe half used).
// From ~64 queries up the work is bound by the exact-f32 matrix pipe (2*N*D flop per query against 157 TFLOP/s),
// so the right shape is a GEMM: a 128-row x 32*NQF-query block tile, both operands staged through LDS in 32-deep
// k-tiles, each wave owning 64 rows x 16*NQF queries (4 x NQF accumulator tiles of v_mfma_f32_16x16x4_f32), every
// LDS fragment used 4 (B) or NQF (A) times.  The corpus is read once per <=128 queries.
//
//   * block = 4 waves as 2 (rows) x 2 (queries); persistent over the row tiles g, g+G, g+2G, ... of its row group
//     and ONE query tile; the (row tile, k-tile) sequence is one software-pipelined stream: the global loads of
//     step i+1 are in flight (registers) while step i is multiplied out of LDS; two barriers per step (one tile
//     buffer: the LDS saved pays for the candidate buffers below at 2 blocks per CU, and the second block of the
//     CU fills the matrix pipe across the barriers).
//   * LDS tiles are [rows][32 floats] with the 16-B slot index XOR-ed by (row>>1)&7: a 16-lane ds_read_b128 group
//     (16 consecutive rows, one slot) covers all 16 slots of the 256-B bank row — conflict-free, and so are the
//     staging writes (8 lanes = one 128-B row).
//   * arithmetic = oracle mode M, bit for bit the same chain as sweep_topk_mfma_f32: for every (row, query) ONE
//     fmaf chain over k = 128U + 16m + 4kk + c (U; m = 0..7; c = 0..3; kk = 0..3 inside the instruction), the
//     dimension zero-padded to a multiple of 128.  Lane (i = l&15, kk = l>>4) reads the float4 at k = 16m' + 4kk of
//     its row: component c is the operand of the c-th MFMA of the group.
//   * top-k without locks: per query a candidate buffer cand[CAP] (u64 keys) + counter in LDS.  The epilogue of
//     a row tile filters the 16 x NQF accumulators of a lane against the query's k-th best (reciprocal multiply,
//     16-ulp margin), finishes survivors exactly and APPENDS them (one LDS atomic add + one store per candidate,
//     all lanes in parallel).  Between the two barriers of the step, wave w compacts the queries w, w+4, ... whose
//     buffer holds more than k keys: one key per lane, rank = number of smaller keys (readlane sweep), keys of
//     rank < k written back in order, k-th best published.  A full buffer makes the block repeat the
//     append/compact round for the candidates that did not fit (only the very first row tile does).  The
//     locked one-key-at-a-time insertion of sweep.hip costs ~1000 cycles per candidate with the other waves
//     waiting at the barrier: 6.7 K candidates per block made it 2/3 of this kernel's time.
//   * blockIdx -> (query tile, row group) keeps the query tiles of one row group on ONE XCD in adjacent dispatch
//     slots, so a row tile needed by several query tiles comes out of that XCD's L2 the second time.
// Bound: the f32 matrix pipe (157.3 TFLOP/s dense); algorithmic flop per launch = 2 * n_rows * dim * nq.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "vdb_probe_env.hpp"
#include "vdb_device.hpp"
#include "vdb_kernels.hpp"

// wave priority outside the MFMA stream (see the s_setprio note in the main loop)
#define VDB_GEMM_PRIO 2
namespace vdb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGemmBM = 128;  // rows per block tile
constexpr int kGemmBK = 32;   // floats per k-tile (one 128-B line per row)
constexpr int kGemmQueue = 256;  // per-wave queue of filter survivors (entries)

struct GemmSweepArgs {
  SweepArgs s;
  uint32_t KT;     // k-tiles per row: 4 * ceil(dim / 128)
  uint32_t G;      // row groups (= top-k lists per query), multiple of 8
  uint32_t nqt;    // query tiles
  uint32_t qper;   // queries per tile (<= 32 * NQF)
  uint32_t cap;    // candidate buffer entries per query (k < cap <= 64)
  const uint32_t* tile_needed;  // nullable: [nqt] a query tile with 0 here is skipped (device-driven fallback, sweep_split.hip)
};

// acc[rf][t][r] for a per-lane element index e = (rf*NQF + t)*4 + r, without dynamic register indexing: a binary
// select tree (one v_cndmask per inner node, the six bit tests shared by a level).
template <int RF, int NQF, int LO, int N>
struct AccSelect {
  static __device__ __forceinline__ float get(const f32x4 (&acc)[RF][NQF], uint32_t e) {
    if (LO >= 4 * RF * NQF) return 0.0f;  // past the last element (NQF = 3: 48 of 64)
    const float lo = AccSelect<RF, NQF, LO, N / 2>::get(acc, e);
    if (LO + N / 2 >= 4 * RF * NQF) return lo;
    const float hi = AccSelect<RF, NQF, LO + N / 2, N / 2>::get(acc, e);
    return (e & (uint32_t)(N / 2)) ? hi : lo;
  }
};
template <int RF, int NQF, int LO>
struct AccSelect<RF, NQF, LO, 1> {
  static __device__ __forceinline__ float get(const f32x4 (&acc)[RF][NQF], uint32_t) {
    if (LO >= 4 * RF * NQF) return 0.0f;
    return acc[(LO / 4) / NQF][(LO / 4) % NQF][LO % 4];
  }
};
template <int RF, int NQF>
__device__ __forceinline__ float select_acc(const f32x4 (&acc)[RF][NQF], uint32_t e) {
  return AccSelect<RF, NQF, 0, (RF == 8 ? 128 : 64)>::get(acc, e);
}

// FULL: dim % 128 == 0 and the queries are readable as aligned float4 — no zero-fill, and the tile loads are
// `uniform base (SGPR) + loop-invariant per-thread offset`: no vector ALU work at all for addressing.
// BF16: the same kernel over the bf16 copy of the rows and bf16-rounded queries (a.rows / a.queries then point at
// uint16 data, strides in elements): k-tiles of 64 bf16 = the same 128 bytes per row, so the LDS geometry, staging and
// epilogue are byte-for-byte the f32 kernel's; one 16-B fragment feeds ONE v_mfma_f32_16x16x32_bf16 (f32 accumulate:
// half_precision.rs:199-255 semantics) instead of four f32 MFMAs.  FULL only (dim % 64 == 0).
// RF / WAVES: accumulator fragments per wave along the rows (wave tile = 16 RF rows x 16 NQF queries) and waves per
// block (2 along the rows x WAVES/2 along the queries).  (4, 4) = the 128 x 32 NQF tile at two blocks per CU described
// above; (8, 8) = a 256 x 256 tile, ONE block of eight waves per CU (bf16 only): twice the MFMAs per LDS fragment
// read / staging instruction / barrier, half the row re-reads — the bf16 multiply is 16x cheaper per element than the
// exact-f32 one, so there the per-step instruction overhead is what bounds the kernel.  (The f32 instance of the big
// tile was measured too: bit-identical, 120.6 vs 122.8 TFLOP/s at 1 024 queries — not kept.)
template <int METRIC, int NQF, bool QVEC, bool FULL, bool BF16, int RF = 4, int WAVES = 4>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 1 : 2) void sweep_topk_gemm_f32(GemmSweepArgs ga) {
  static_assert(!BF16 || FULL, "the bf16 variant has no zero-fill path");
  static_assert((RF == 4 && WAVES == 4) || (RF == 8 && WAVES == 8 && NQF == 4 && BF16), "supported tile shapes");
  constexpr int ES = BF16 ? 2 : 4;        // bytes per element in HBM
  constexpr int BKE = 128 / ES;           // elements per k-tile (one 128-B line per row)
  constexpr int NT = WAVES * 64, WCOLS = WAVES / 2, WROWS = 16 * RF, SR = NT / 8;  // SR: rows staged per pass
  constexpr int BM = 2 * WROWS, BK = kGemmBK, BN = WCOLS * 16 * NQF;
  static_assert(BM == 4 * SR && BN == NQF * SR, "staging passes: 4 for the rows, NQF for the queries");
  constexpr int NW = RF * NQF * 4 / 32 > 2 ? RF * NQF * 4 / 32 : 2;  // 32-bit words of the per-lane pass mask
  // Euclidean instance: the accumulators are still q.v; the kernel selects by the APPROXIMATE squared distance
  // |v|^2 + |q|^2 - 2 q.v (lower is better) and keeps k' = k + slack candidates per query; the caller re-scores them
  // with the canonical (q - v)^2 chain and verifies that nothing outside the candidates can reach the top k
  // (euclid_rerank_verify, index.hip) — the identity alone loses all relative accuracy for near-duplicates.
  constexpr bool HIB = METRIC != kEuclidean;
  constexpr bool NORMS = METRIC == kCosine || METRIC == kEuclidean;  // per-row / per-query norms are needed
  const SweepArgs& a = ga.s;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* As = reinterpret_cast<float*>(smem);  // [BM][BK]
  float* Bs = As + BM * BK;                    // [BN][BK]
  unsigned char* tail = reinterpret_cast<unsigned char*>(Bs + BN * BK);
  const uint32_t k = a.k, CAP = ga.cap;
  uint64_t* cand = reinterpret_cast<uint64_t*>(tail);                           // [BN][CAP] keys, first k sorted after a compaction
  uint64_t* tauk = reinterpret_cast<uint64_t*>(tail + (size_t)BN * CAP * 8);    // [BN] k-th best key (invalid: none yet)
  uint32_t* cnts = reinterpret_cast<uint32_t*>(tail + (size_t)BN * CAP * 8 + (size_t)BN * 8);
  float* qn = reinterpret_cast<float*>(tail + (size_t)BN * CAP * 8 + (size_t)BN * 12);
  lds_vu32* ovf = (lds_vu32*)(lds_void_p)(tail + (size_t)BN * CAP * 8 + (size_t)BN * 16);  // overflow token
  float* vns = reinterpret_cast<float*>(tail + (size_t)BN * CAP * 8 + (size_t)BN * 16 + 16);  // [BM] norms of the row tile
  uint64_t* wqueue_all = reinterpret_cast<uint64_t*>(tail + (size_t)BN * CAP * 8 + (size_t)BN * 16 + 16 + BM * 4);  // [WAVES][kGemmQueue]

  __builtin_amdgcn_s_setprio(VDB_GEMM_PRIO);
  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wib / WCOLS, wq = wib % WCOLS;
  uint64_t* wqueue = wqueue_all + wib * kGemmQueue;  // this wave's survivors of the filter: (dot bits, row in tile, query)

  // block -> (query tile, row group): the nqt query tiles of a row group sit on one XCD (blockIdx % 8)
  const uint32_t bid = blockIdx.x;
  const uint32_t xcd = bid & 7u, slot_id = bid >> 3;
  const uint32_t qt = slot_id % ga.nqt;
  const uint32_t g = (slot_id / ga.nqt) * 8u + xcd;
  // device-driven fallback (sweep_split.hip): only the query tiles that hold an unproven query — and nothing at all when
  // the unproven queries are few enough for the gathered pass of the streaming kernel (a.qcount / a.qcount_max)
  if (ga.tile_needed && (ga.tile_needed[qt] == 0u || (a.qcount && *a.qcount <= a.qcount_max))) return;  // block-uniform, before the first barrier
  const uint32_t q0 = qt * ga.qper;
  const uint32_t nq_t = min(ga.qper, a.nq - q0);
  const float* queries = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(a.queries) + (size_t)q0 * a.q_stride * ES);

  if (tid < BN) {
    cnts[tid] = 0;
    tauk[tid] = kKeyInvalid;
    qn[tid] = 0.0f;
  }
  if (tid == 0) *ovf = 0u;
  __syncthreads();
  if (METRIC == kCosine && BF16) {  // norm of the ROUNDED query, canonical lane-chain order (as sweep_topk_mfma_bf16)
    for (uint32_t b = wib; b < nq_t; b += WAVES) {
      const uint16_t* qp = reinterpret_cast<const uint16_t*>(queries) + (size_t)b * a.q_stride;
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
  } else if (NORMS) {  // canonical query norms (same as every other kernel); Euclidean keeps the squared norm
    const int d4 = (int)((a.dim + 3) / 4);
    for (uint32_t b = wib; b < nq_t; b += WAVES) {
      const float* qp = queries + (size_t)b * a.q_stride;
      float nacc = 0.0f;
      for (int c = lane; c < d4; c += 64) {
        const int nv = (int)a.dim - c * 4;
        float4 x;
        if (nv >= 4) {
          x = make_float4(qp[c * 4], qp[c * 4 + 1], qp[c * 4 + 2], qp[c * 4 + 3]);
          nacc = chain4<kOpDot>(nacc, x, x);
        } else {
          x = make_float4(qp[c * 4], nv > 1 ? qp[c * 4 + 1] : 0.f, nv > 2 ? qp[c * 4 + 2] : 0.f, 0.f);
          nacc = chain4_tail<kOpDot>(nacc, x, x, nv);
        }
      }
      const float nsq = butterfly_all(nacc);
      const float n = METRIC == kEuclidean ? nsq : sqrtf(nsq);
      if (lane == 0) qn[b] = n;
    }
  }
  __syncthreads();
  float qn_t[NQF];
#pragma unroll
  for (int t = 0; t < NQF; t++) qn_t[t] = qn[wq * 16 * NQF + t * 16 + (lane & 15)];

  const uint32_t ntiles = (a.n_rows + BM - 1) / BM;
  const uint32_t my_tiles = g < ntiles ? (ntiles - g + ga.G - 1) / ga.G : 0;
  const uint32_t total = my_tiles * ga.KT;

  // ---- staging: thread t moves the 16-B slot (t & 7) of rows (t >> 3) + SR j ----
  const int st_slot = tid & 7, st_row = tid >> 3;
  float4 ra[4], rb[NQF];
  uint32_t ld_rt = g, ld_kt = 0;  // (row tile, k-tile) of the NEXT step to load
  uint32_t pend_kf = 0;           // k offset of the loads currently held in ra / rb
  // Branch-free: every load is issued unconditionally from a clamped (valid) address and zeroed by a select —
  // a conditional load makes hipcc branch around it and wait vmcnt(0) per element (serialised round trips).
  // FULL path: per-thread byte offsets inside a row tile / the query tile (loop-invariant)
  const uint32_t voff_a = (uint32_t)st_row * (uint32_t)a.row_stride * (uint32_t)ES + (uint32_t)st_slot * 16u;
  uint32_t voff_b[NQF];
#pragma unroll
  for (int j = 0; j < NQF; j++) {
    const uint32_t q = st_row + SR * j;
    voff_b[j] = (q < nq_t ? q : nq_t - 1) * (uint32_t)a.q_stride * (uint32_t)ES + (uint32_t)st_slot * 16u;  // padded slots repeat a query
  }
  // (macros, not lambdas: hipcc does not always promote arrays captured by a lambda to registers)
#define VDB_GEMM_GLOAD() do { \
    if (FULL) { \
 \
      const unsigned char* base_a = reinterpret_cast<const unsigned char*>(a.rows) + \
                                    ((size_t)ld_rt * BM * a.row_stride + (size_t)ld_kt * BKE) * ES; \
      const unsigned char* base_b = reinterpret_cast<const unsigned char*>(queries) + (size_t)ld_kt * BKE * ES; \
_Pragma("unroll") \
      for (int j = 0; j < 4; j++) \
        ra[j] = ld4(reinterpret_cast<const float*>(base_a + (size_t)(SR * j) * a.row_stride * ES + voff_a)); \
_Pragma("unroll") \
      for (int j = 0; j < NQF; j++) rb[j] = ld4(reinterpret_cast<const float*>(base_b + voff_b[j])); \
    } else { \
      const uint32_t kf = ld_kt * BK + st_slot * 4; \
      const bool kin = kf < (uint32_t)a.row_stride; \
      const uint32_t kfa = kin ? kf : 0u; \
_Pragma("unroll") \
      for (int j = 0; j < 4; j++) { \
        uint32_t row = ld_rt * BM + st_row + SR * j; \
        row = row < a.n_rows ? row : a.n_rows - 1; \
        ra[j] = ld4(a.rows + (size_t)row * a.row_stride + kfa); \
      } \
_Pragma("unroll") \
      for (int j = 0; j < NQF; j++) { \
        const uint32_t q = st_row + SR * j; \
        const float* qp = queries + (size_t)(q < nq_t ? q : 0u) * a.q_stride; \
        if (QVEC) { \
          rb[j] = ld4(qp + (kf < a.dim ? kf : 0u)); \
        } else { \
          const uint32_t dl = a.dim - 1; \
          rb[j] = make_float4(qp[min(kf, dl)], qp[min(kf + 1, dl)], qp[min(kf + 2, dl)], qp[min(kf + 3, dl)]); \
        } \
      } \
      pend_kf = kf; \
    } \
    if (++ld_kt == ga.KT) { \
      ld_kt = 0; \
      ld_rt += ga.G; \
    } \
  } while (0)
  // staging writes: rows SR (32 / 64) apart share the swizzle term: one address + immediate offsets
  unsigned char* const st_wr_a = reinterpret_cast<unsigned char*>(As) + st_row * (BK * 4) + ((st_slot ^ ((st_row >> 1) & 7)) << 4);
  unsigned char* const st_wr_b = reinterpret_cast<unsigned char*>(Bs) + st_row * (BK * 4) + ((st_slot ^ ((st_row >> 1) & 7)) << 4);
#define VDB_GEMM_LDS_STORE() do { \
    if (FULL) { \
_Pragma("unroll") \
      for (int j = 0; j < 4; j++) *reinterpret_cast<f32x4*>(st_wr_a + j * (SR * BK * 4)) = f32x4{ra[j].x, ra[j].y, ra[j].z, ra[j].w}; \
_Pragma("unroll") \
      for (int j = 0; j < NQF; j++) *reinterpret_cast<f32x4*>(st_wr_b + j * (SR * BK * 4)) = f32x4{rb[j].x, rb[j].y, rb[j].z, rb[j].w}; \
    } else { \
    const bool kin = pend_kf < (uint32_t)a.row_stride; \
    const bool k0 = pend_kf < a.dim, k1 = pend_kf + 1 < a.dim, k2 = pend_kf + 2 < a.dim, k3 = pend_kf + 3 < a.dim; \
_Pragma("unroll") \
    for (int j = 0; j < 4; j++) { \
      float4 v = ra[j]; \
      v.x = kin ? v.x : 0.f; \
      v.y = kin ? v.y : 0.f; \
      v.z = kin ? v.z : 0.f; \
      v.w = kin ? v.w : 0.f; \
      *reinterpret_cast<float4*>(st_wr_a + j * (SR * BK * 4)) = v; \
    } \
_Pragma("unroll") \
    for (int j = 0; j < NQF; j++) { \
      const int q = st_row + SR * j; \
      const bool qin = (uint32_t)q < nq_t; \
      float4 v = rb[j]; \
      v.x = (qin && k0) ? v.x : 0.f; \
      v.y = (qin && k1) ? v.y : 0.f; \
      v.z = (qin && k2) ? v.z : 0.f; \
      v.w = (qin && k3) ? v.w : 0.f; \
      *reinterpret_cast<float4*>(st_wr_b + j * (SR * BK * 4)) = v; \
    } \
    } \
  } while (0)

  f32x4 acc[RF][NQF];
#pragma unroll
  for (int rf = 0; rf < RF; rf++)
#pragma unroll
    for (int t = 0; t < NQF; t++) acc[rf][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- compaction of the candidate buffers this wave owns (queries wib, wib+WAVES, ...) ----
  // `force`: every buffer holding more than k keys (overflow rounds, end of the sweep); otherwise only the
  // buffers past the watermark — a query's k-th best then lags behind, which costs a few more (cheap) appends
  // and saves most of the (expensive) compactions: ~4 per query and block instead of one per row tile.
  const uint32_t watermark = (k + CAP) / 2;
  auto compact = [&](bool force) __attribute__((always_inline)) {
    // lane l looks at query wib + WAVES*l
    const uint32_t bq = (uint32_t)wib + (uint32_t)WAVES * (uint32_t)lane;
    const uint32_t cq = bq < nq_t ? cnts[bq] : 0u;
    uint64_t need = __ballot(cq > k && (force || cq >= watermark));
    // 64 / CAP buffers per pass (CAP = 32: the two halves of the wave rank one buffer each): a lane holds one key
    // and counts the smaller ones — the other keys come as LDS broadcast reads, four in flight (keys are unique)
    const uint32_t per = 64u / CAP, li = (uint32_t)lane & (CAP - 1u), half = (uint32_t)lane / CAP;
    while (need) {
      const int src0 = __ffsll((long long)need) - 1;
      need &= need - 1;
      int src1 = src0;
      if (per == 2 && need) {
        src1 = __ffsll((long long)need) - 1;
        need &= need - 1;
      }
      const bool active = half == 0 || src1 != src0;  // odd count: the second half idles
      const uint32_t b = (uint32_t)wib + (uint32_t)WAVES * (uint32_t)(half ? src1 : src0);
      const uint32_t n = min(cnts[b], CAP);
      uint64_t* cb = cand + (size_t)b * CAP;
      const uint64_t key = (active && li < n) ? cb[li] : kKeyInvalid;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < CAP; j += 4) {
        uint64_t kj[4];
#pragma unroll
        for (int u = 0; u < 4; u++) kj[u] = cb[j + u];
#pragma unroll
        for (int u = 0; u < 4; u++) rank += (j + u < n && kj[u] < key) ? 1u : 0u;
      }
      const bool mine = active && li < n;
      if (mine && rank < k) cb[rank] = key;
      if (mine && rank == k - 1) tauk[b] = key;
      if (active && li == 0) cnts[b] = k;
    }
  };

  // fragment read addresses: lane (i = l&15, kk = l>>4) reads slot (4m + kk) ^ ((i>>1)&7) of row i (+16 per fragment)
  const int sw_i = ((lane & 15) >> 1) & 7;
  const int rd_off = (lane & 15) * (BK * 4) + ((((lane >> 4) ^ sw_i) & 3) << 4);  // bits 0-1 of the slot
  const int rd_x = (sw_i & 4) << 4;                                                // bit 2 of the slot, as byte 64
  const unsigned char* a_rd = reinterpret_cast<const unsigned char*>(As) + wr * WROWS * (BK * 4) + rd_off;
  const unsigned char* b_rd = reinterpret_cast<const unsigned char*>(Bs) + wq * 16 * NQF * (BK * 4) + rd_off;
  const int a_rd_x = rd_x, b_rd_x = rd_x;

  if (total) {
    VDB_GEMM_GLOAD();
    VDB_GEMM_LDS_STORE();
  }
  __syncthreads();
  uint32_t kt = 0, rt = g;
  uint32_t token = 0;  // ++ per epilogue round, block-uniform: a value written to *ovf is never reused
  for (uint32_t it = 0; it < total; it++) {
    const bool more = it + 1 < total;
    if (more) VDB_GEMM_GLOAD();
    // the row tile's norms: loaded here every step, written to LDS by the epilogue step in front of its first barrier.
    // Loaded unconditionally every step (clamped address, L2 hit): a conditional load would make hipcc wait
    // vmcnt(0) in front of the branch, i.e. for the tile loads just issued.
    float vn_reg = 0.0f;
    if (NORMS) {
      const uint32_t row = rt * BM + (uint32_t)(tid & (BM - 1));
      vn_reg = a.norms[row < a.n_rows ? row : a.n_rows - 1];
      if (METRIC == kEuclidean) vn_reg *= vn_reg;  // |v|^2
    }
    {  // ---- multiply k-tile `it` out of LDS ----
      // rows 16 apart share the swizzle term, so the 4 / NQF fragment reads of a 16-deep group are one base +
      // immediate offsets; the second group flips bit 2 of the slot = byte 64 of the (swizzled) address.
      // Both groups are requested up front: the second one's LDS latency hides behind the first one's MFMAs.
      // (the 256 x 256 variant holds ONE 16-deep group at a time: 128 accumulator registers leave no room for two)
      constexpr int MB = RF == 8 ? 1 : 2;
#pragma unroll
      for (int m0 = 0; m0 < 2; m0 += MB) {
      float4 av[MB][RF], bv[MB][NQF];
#pragma unroll
      for (int m = 0; m < MB; m++) {
#pragma unroll
        for (int rf = 0; rf < RF; rf++) av[m][rf] = *reinterpret_cast<const float4*>(a_rd + (((m0 + m) * 64) ^ a_rd_x) + rf * (16 * BK * 4));
#pragma unroll
        for (int t = 0; t < NQF; t++) bv[m][t] = *reinterpret_cast<const float4*>(b_rd + (((m0 + m) * 64) ^ b_rd_x) + t * (16 * BK * 4));
      }
      // Wave priority: LOW while streaming MFMAs, HIGH for everything else.  The two blocks of a CU share each SIMD's
      // issue port; at equal priority the staging / epilogue instructions of one block queue behind the partner's
      // dense MFMA stream (8 ds_writes took ~1900 cycles).  With priority they slip through at once, the wave gets
      // back to feeding the matrix pipe sooner, and the partner's MFMAs fill the gaps anyway: +8 % (110 -> 118 TF).
      __builtin_amdgcn_s_setprio(0);
      if (BF16) {
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int m = 0; m < MB; m++)
#pragma unroll
          for (int rf = 0; rf < RF; rf++)
#pragma unroll
            for (int t = 0; t < NQF; t++)
              acc[rf][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[m][rf]),
                                                                  __builtin_bit_cast(bf16x8, bv[m][t]), acc[rf][t], 0, 0, 0);
      } else {
#pragma unroll
      for (int m = 0; m < MB; m++) {
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
          for (int rf = 0; rf < RF; rf++) {
            const float ax = c == 0 ? av[m][rf].x : (c == 1 ? av[m][rf].y : (c == 2 ? av[m][rf].z : av[m][rf].w));
#pragma unroll
            for (int t = 0; t < NQF; t++) {
              const float bx = c == 0 ? bv[m][t].x : (c == 1 ? bv[m][t].y : (c == 2 ? bv[m][t].z : bv[m][t].w));
              acc[rf][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bx, acc[rf][t], 0, 0, 0);
            }
          }
        }
      }
      }
      __builtin_amdgcn_s_setprio(VDB_GEMM_PRIO);
      }
    }
    const bool vn_step = NORMS && kt + 2 == ga.KT;  // the step before the epilogue step stages the norms
    if (++kt < ga.KT) {
      __syncthreads();  // every wave is done reading the tile
      if (more) VDB_GEMM_LDS_STORE();
      if (vn_step && tid < BM) vns[tid] = vn_reg;
      __syncthreads();
      continue;
    }
    // ---- last k-tile of a row tile: epilogue = filter, exact finish, append; compaction between the barriers ----
    if (NORMS && ga.KT < 2 && tid < BM) vns[tid] = vn_reg;  // single-k-tile rows: no earlier step to do it in
    __syncthreads();  // every wave is done reading the tile; norms visible
    if (more) VDB_GEMM_LDS_STORE();  // staging registers are dead from here on: the epilogue gets their 32 VGPRs
    // (1a) filter, branch-free: one bit per accumulator element in a per-lane 64-bit mask, element e = (rf*NQF + t)*4
    //      + r shifted in at the bottom (so e = 63 - bit index ... see clz below).  Pass unless clearly below the
    //      query's k-th best (16-ulp margin); cosine compares dot * (1/|v|) with cut * |q|: NaN / inf / zero-norm
    //      cases compare false and go to the exact path; padded query slots get +inf (nothing passes).  Rows past
    //      n_rows are weeded out by the dense pass (2).
    uint32_t pm[NW];  // pm[w]: elements 32w .. 32w+31 (element e at bit 31 - e % 32)
#pragma unroll
    for (int w = 0; w < NW; w++) pm[w] = 0u;
    {
      float cutq[NQF];
#pragma unroll
      for (int t = 0; t < NQF; t++) {
        const uint32_t b = wq * 16 * NQF + t * 16 + (lane & 15);
        const uint64_t tkb = tauk[b];
        if (METRIC == kEuclidean) {
          // pass iff |v|^2 + |q|^2 - 2 acc <= tau (+ margin)  <=>  !(acc < 0.5 |v|^2 (1 - 2^-18) + hq), with
          // hq = 0.5 (|q|^2 (1 - 2^-18) - tau'); the 2^-18 slack covers the rounding of both forms ~10x over
          const float tf = tkb == kKeyInvalid ? __uint_as_float(0x7F800000u) : key_score<HIB>(tkb);
          const float tfm = tf + (fabsf(tf) * 3.8146973e-6f + 1e-37f);
          cutq[t] = b < nq_t ? 0.5f * (qn_t[t] * 0.99999619f - tfm) : __uint_as_float(0x7F800000u);
        } else {
        const float tf = tkb == kKeyInvalid ? __uint_as_float(0xFF800000u) : key_score<HIB>(tkb);
        const float cut = tf - (fabsf(tf) * 1.9073486e-6f + 1e-37f);
        cutq[t] = b < nq_t ? (METRIC == kCosine ? cut * qn_t[t] : cut) : __uint_as_float(0x7F800000u);
        // bf16 results follow half_precision.rs: a norm below f32::EPSILON makes every score of the query 0.0 — nothing
        // may be filtered by the approximation (NaN compares false: everything passes to the exact finish)
        if (BF16 && METRIC == kCosine && b < nq_t && qn_t[t] < kHalfNormEps) cutq[t] = __uint_as_float(0x7FC00000u);
        }
      }
#pragma unroll
      for (int rf = 0; rf < RF; rf++) {
        f32x4 rvn = f32x4{1.f, 1.f, 1.f, 1.f};
        if (NORMS) {
          const f32x4 vn = *reinterpret_cast<const f32x4*>(vns + wr * WROWS + rf * 16 + 4 * (lane >> 4));
#pragma unroll
          for (int r = 0; r < 4; r++) {
            rvn[r] = METRIC == kEuclidean ? vn[r] * 0.49999809f : __builtin_amdgcn_rcpf(vn[r]);
            if (BF16 && METRIC == kCosine && vn[r] < kHalfNormEps) rvn[r] = __uint_as_float(0x7FC00000u);  // score 0.0: see above
          }
        }
#pragma unroll
        for (int t = 0; t < NQF; t++) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            constexpr int kHalf = 32;
            const int e = (rf * NQF + t) * 4 + r;
            const bool pass = METRIC == kEuclidean ? !(acc[rf][t][r] < rvn[r] + cutq[t])
                                                   : !((METRIC == kCosine ? acc[rf][t][r] * rvn[r] : acc[rf][t][r]) < cutq[t]);
            pm[e / kHalf] = (pm[e / kHalf] << 1) | (pass ? 1u : 0u);
          }
        }
      }
      // left-align both words: element e of a word sits at bit 31 - (e % 32)
      constexpr int kElems = 4 * RF * NQF;
      if (kElems < 32) pm[0] <<= (32 - kElems);
      if (kElems > 32 && kElems < 64) pm[1] <<= (64 - kElems);
    }
    uint32_t qcarry = 0;  // queue entries carried into the next round (their candidate buffer was full)
    // per-lane part of a queue entry's low word: (row in tile) << 8 | query in tile, for rf = t = r = 0
    const uint32_t lane_word = ((uint32_t)(wr * WROWS + 4 * (lane >> 4)) << 8) | (uint32_t)(wq * 16 * NQF + (lane & 15));
    for (;;) {
      bool failed = false;
      ++token;
      // (1b) drain the masks into the wave's queue as raw (dot, row, query): every pass takes each lane's lowest
      //      pending element (a 6-level select tree over the accumulators — no dynamic register indexing), slots
      //      from ballot/mbcnt, plain LDS stores.  Passes = the largest number of survivors in one lane (3-4).
      uint32_t qn_ent = qcarry;  // wave-uniform
      for (;;) {
        uint32_t pm_any = pm[0];
#pragma unroll
        for (int w = 1; w < NW; w++) pm_any |= pm[w];
        const bool has = pm_any != 0u;
        const uint64_t mh = __ballot(has);
        if (mh == 0) break;
        const uint32_t nh = (uint32_t)__popcll(mh);
        if (qn_ent + nh > (uint32_t)kGemmQueue) {
          failed = true;  // queue full: finish what is queued, then continue draining
          break;
        }
        uint32_t word, e;
        if (NW == 2) {
          const bool lo = pm[0] != 0u;
          word = lo ? pm[0] : pm[1];
          const uint32_t lz = (uint32_t)__builtin_clz(word | 1u);  // element inside the word
          e = lz + (lo ? 0u : 32u);
          const uint32_t cleared = word & ~(0x80000000u >> lz);
          if (has) {
            if (lo) pm[0] = cleared; else pm[1] = cleared;
          }
        } else {
          uint32_t wi = NW - 1;  // first non-empty word
          word = pm[NW - 1];
#pragma unroll
          for (int w = NW - 2; w >= 0; w--) {
            const bool nz = pm[w] != 0u;
            word = nz ? pm[w] : word;
            wi = nz ? (uint32_t)w : wi;
          }
          const uint32_t lz = (uint32_t)__builtin_clz(word | 1u);
          e = lz + 32u * wi;
          const uint32_t cleared = word & ~(0x80000000u >> lz);
#pragma unroll
          for (int w = 0; w < NW; w++) pm[w] = (has && wi == (uint32_t)w) ? cleared : pm[w];
        }
        const float dv = select_acc<RF, NQF>(acc, e);
        // e = (rf*NQF + t)*4 + r
        const uint32_t r = e & 3u, ft = e >> 2;
        const uint32_t rf = NQF == 4 ? ft >> 2 : (NQF == 2 ? ft >> 1 : (ft * 11u) >> 5);  // ft / NQF (NQF = 3: ft < 16)
        const uint32_t t = ft - rf * NQF;
        const uint32_t slot = qn_ent + __builtin_amdgcn_mbcnt_hi((uint32_t)(mh >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mh, 0u));
        if (has) wqueue[slot] = ((uint64_t)__float_as_uint(dv) << 32) | (lane_word + (((rf * 16u + r) << 8) + t * 16u));
        qn_ent += nh;
      }
      // (2) finish the queue densely, one entry per lane: exact score, key, candidate buffer of the entry's query
      qcarry = 0;
#pragma unroll 1
      for (uint32_t c = 0; c * 64 < qn_ent; c++) {
        // entries whose candidate buffer is full are re-queued at slot <= c*64 + lane: never ahead of the read position
        const uint64_t ent1 = (c * 64 + lane) < qn_ent ? wqueue[c * 64 + lane] : kKeyInvalid;
        const bool valid = ent1 != kKeyInvalid;
        const uint32_t b = valid ? (uint32_t)ent1 & 0xFFu : 0u;
        const uint32_t rl = valid ? ((uint32_t)ent1 >> 8) & 0xFFu : 0u;
        const uint32_t row = rt * BM + rl;
        const float dotv = __uint_as_float((uint32_t)(ent1 >> 32));
        const float score = METRIC == kEuclidean ? __builtin_fmaf(-2.0f, dotv, qn[b] + vns[rl])  // approximate |q - v|^2
                                                 : (BF16 ? finish_score_half<METRIC>(dotv, qn[b], METRIC == kCosine ? vns[rl] : 1.0f)
                                                         : finish_score<METRIC>(dotv, qn[b], METRIC == kCosine ? vns[rl] : 1.0f));
        const uint64_t key = make_key<HIB>(score, row);
        bool take = valid & (b < nq_t) & (row < a.n_rows) & (key < tauk[b]);
        if (take && a.alive) take = a.alive[row] != 0;  // soft-deleted rows are filtered where it is rare
        bool full = false;
        if (take) {
          const uint32_t idx = atomicAdd(&cnts[b], 1u);
          if (idx < CAP)
            cand[(size_t)b * CAP + idx] = key;
          else
            full = true;  // candidate buffer full: keep the entry queued for the round after the compaction
        }
        const uint64_t mf = __ballot(full);
        if (mf) {
          const uint32_t slot = qcarry + __builtin_amdgcn_mbcnt_hi((uint32_t)(mf >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mf, 0u));
          if (full) wqueue[slot] = ent1;
          qcarry += (uint32_t)__popcll(mf);
          failed = true;
        }
      }
      if (failed) *ovf = token;
      __syncthreads();  // appends visible
      const bool again = *ovf == token;
      compact(again || !more);
      __syncthreads();  // the compacted lists (and, first round, the next tile) visible
      if (!again) break;
    }
#pragma unroll
    for (int rf = 0; rf < RF; rf++)
#pragma unroll
      for (int t = 0; t < NQF; t++) acc[rf][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    kt = 0;
    rt += ga.G;
  }
  __syncthreads();
  for (uint32_t b = wib; b < nq_t; b += WAVES) {
    const uint32_t c = min(cnts[b], k);  // <= k entries: whatever order (the merge kernel scans them all)
    uint64_t* out = a.part_keys + ((size_t)(q0 + b) * ga.G + g) * k;
    for (uint32_t e = lane; e < k; e += 64) out[e] = e < c ? cand[(size_t)b * CAP + e] : kKeyInvalid;
  }
}

#undef VDB_GEMM_GLOAD
#undef VDB_GEMM_LDS_STORE

// ---- host side ---------------------------------------------------------------------------------
uint32_t sweep_gemm_cap(uint32_t k) { return k <= 16 ? 32u : 64u; }  // candidate buffer entries per query
size_t sweep_gemm_lds_bytes(int nqf, uint32_t k, bool big) {
  const size_t BM = big ? 256 : kGemmBM, BN = big ? 256 : (size_t)32 * nqf, waves = big ? 8 : 4;
  return (BM * kGemmBK * 4 + BN * kGemmBK * 4 + BN * sweep_gemm_cap(k) * 8 + BN * 16 + 16 + BM * 4 + waves * kGemmQueue * 8 + 15) & ~(size_t)15;
}

void sweep_gemm_plan(uint32_t nq, uint32_t n_rows, int n_cus, uint32_t k, GemmPlan* p, bool allow_big) {
  static const bool env_big = [] {  // VELESDB_GEMM_BIG=0: never pick the 256 x 256 tile (A/B probes)
    const char* e = probe_env("VELESDB_GEMM_BIG");
    return !(e && e[0] == '0');
  }();
  // the 256 x 256 tile wins by ~20 % per (padded) query slot: taken when the batch fills its query tiles to >= 7/8
  const uint32_t nqt_big = (nq + 255) / 256;
  p->big = allow_big && env_big && nq >= kGemmBigMinQueries && (uint64_t)nq * 8 >= (uint64_t)nqt_big * 256 * 7 &&
           sweep_gemm_lds_bytes(4, k, true) <= 160 * 1024;
  const uint32_t bn = p->big ? 256 : 128, bm = p->big ? 256 : kGemmBM;
  p->nqt = (nq + bn - 1) / bn;
  p->qper = (nq + p->nqt - 1) / p->nqt;
  p->nqf = p->big ? 4 : (int)((p->qper + 31) / 32);
  if (p->nqf < 2) p->nqf = 2;
  p->lds = sweep_gemm_lds_bytes(p->nqf, k, p->big);
  const int per_cu = (!p->big && p->lds * 2 <= 160 * 1024) ? 2 : 1;
  const uint32_t ntiles = (n_rows + bm - 1) / bm;
  // row groups: whole XCD rounds (multiples of 8; groups beyond the last row tile emit empty lists), and never more
  // blocks than the chip holds at once — a few blocks left over for a second round would double the launch's time
  uint32_t G = (uint32_t)std::max(8, n_cus * per_cu / (int)p->nqt / 8 * 8);
  G = std::min(G, (ntiles + 7) / 8 * 8);
  p->G = G;
  p->blocks = (int)(G * p->nqt);
}

template <int METRIC, int NQF, bool QVEC, bool FULL, bool BF16, int RF = 4, int WAVES = 4>
static hipError_t launch_gemm_v(const GemmSweepArgs& ga, int blocks, size_t lds, hipStream_t st) {
  static bool done = false;
  if (lds > 64 * 1024 && !done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_gemm_f32<METRIC, NQF, QVEC, FULL, BF16, RF, WAVES>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL((sweep_topk_gemm_f32<METRIC, NQF, QVEC, FULL, BF16, RF, WAVES>), dim3(blocks), dim3(WAVES * 64), lds, st, ga);
  return hipGetLastError();
}
template <int METRIC, int NQF>
static hipError_t launch_gemm_t(const GemmSweepArgs& ga, bool qvec, int blocks, size_t lds, hipStream_t st) {
  if (qvec && ga.s.dim % 128 == 0) return launch_gemm_v<METRIC, NQF, true, true, false>(ga, blocks, lds, st);
  return qvec ? launch_gemm_v<METRIC, NQF, true, false, false>(ga, blocks, lds, st)
              : launch_gemm_v<METRIC, NQF, false, false, false>(ga, blocks, lds, st);
}
hipError_t launch_sweep_gemm(int metric, const GemmPlan& p, const SweepArgs& a, hipStream_t st, const uint32_t* tile_needed) {
  GemmSweepArgs ga;
  ga.s = a;
  ga.tile_needed = tile_needed;
  ga.KT = (a.dim + 127) / 128 * 4;
  ga.G = p.G;
  ga.nqt = p.nqt;
  ga.qper = p.qper;
  ga.cap = sweep_gemm_cap(a.k);
  // queries readable as aligned float4?
  const bool qvec = a.dim % 4 == 0 && a.q_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.queries) & 15) == 0;
  if (p.big) return hipErrorInvalidValue;  // the 256 x 256 tile is a bf16 instance only (f32: measured no gain)
  if (metric == kCosine) {
    switch (p.nqf) {
      case 2: return launch_gemm_t<kCosine, 2>(ga, qvec, p.blocks, p.lds, st);
      case 3: return launch_gemm_t<kCosine, 3>(ga, qvec, p.blocks, p.lds, st);
      default: return launch_gemm_t<kCosine, 4>(ga, qvec, p.blocks, p.lds, st);
    }
  }
  if (metric == kEuclidean) {
    switch (p.nqf) {
      case 2: return launch_gemm_t<kEuclidean, 2>(ga, qvec, p.blocks, p.lds, st);
      case 3: return launch_gemm_t<kEuclidean, 3>(ga, qvec, p.blocks, p.lds, st);
      default: return launch_gemm_t<kEuclidean, 4>(ga, qvec, p.blocks, p.lds, st);
    }
  }
  switch (p.nqf) {
    case 2: return launch_gemm_t<kDot, 2>(ga, qvec, p.blocks, p.lds, st);
    case 3: return launch_gemm_t<kDot, 3>(ga, qvec, p.blocks, p.lds, st);
    default: return launch_gemm_t<kDot, 4>(ga, qvec, p.blocks, p.lds, st);
  }
}

// ---- bf16 variant: rows16 / queries16 are bf16 (uint16), strides in elements, dim % 64 == 0 ----
// round-to-nearest-even copy of the query batch (VectorData::from_f32_slice(.., BF16), half_precision.rs:94-101)
__global__ __launch_bounds__(256) void round_queries_bf16(const float* q, uint64_t q_stride, uint16_t* out, uint64_t out_stride,
                                                          uint32_t nq, uint32_t dim) {
  const uint64_t n = (uint64_t)nq * out_stride;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const uint32_t b = (uint32_t)(i / out_stride), d = (uint32_t)(i % out_stride);
    uint16_t h = 0;
    if (d < dim) {
      uint32_t u = __float_as_uint(q[(size_t)b * q_stride + d]);
      if ((u & 0x7FFFFFFFu) > 0x7F800000u) {
        h = (uint16_t)((u >> 16) | 0x0040u);  // NaN stays NaN (quiet)
      } else {
        u += 0x7FFFu + ((u >> 16) & 1u);
        h = (uint16_t)(u >> 16);
      }
    }
    out[i] = h;
  }
}
void launch_round_queries_bf16(const float* q, uint64_t q_stride, uint16_t* out, uint64_t out_stride, uint32_t nq,
                               uint32_t dim, hipStream_t st) {
  const uint64_t n = (uint64_t)nq * out_stride;
  hipLaunchKernelGGL(round_queries_bf16, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 4096)), dim3(256), 0, st, q,
                     q_stride, out, out_stride, nq, dim);
}

hipError_t launch_sweep_gemm_bf16(int metric, const GemmPlan& p, const uint16_t* rows16, uint64_t row_stride,
                                  const float* norms, const uint8_t* alive, const uint16_t* queries16, uint64_t q_stride,
                                  uint64_t* part_keys, uint32_t n_rows, uint32_t dim, uint32_t nq, uint32_t k,
                                  hipStream_t st) {
  GemmSweepArgs ga;
  ga.s = SweepArgs{};
  ga.s.rows = reinterpret_cast<const float*>(rows16);        // bytes; the kernel addresses them as ES = 2
  ga.s.norms = norms;
  ga.s.alive = alive;
  ga.s.queries = reinterpret_cast<const float*>(queries16);
  ga.s.part_keys = part_keys;
  ga.s.row_stride = row_stride;
  ga.s.q_stride = q_stride;
  ga.s.n_rows = n_rows;
  ga.s.dim = dim;
  ga.s.nq = nq;
  ga.s.k = k;
  ga.KT = dim / 64;
  ga.G = p.G;
  ga.nqt = p.nqt;
  ga.qper = p.qper;
  ga.cap = sweep_gemm_cap(k);
  ga.tile_needed = nullptr;
  if (p.big)
    return metric == kCosine ? launch_gemm_v<kCosine, 4, true, true, true, 8, 8>(ga, p.blocks, p.lds, st)
                             : launch_gemm_v<kDot, 4, true, true, true, 8, 8>(ga, p.blocks, p.lds, st);
  if (metric == kCosine) {
    switch (p.nqf) {
      case 2: return launch_gemm_v<kCosine, 2, true, true, true>(ga, p.blocks, p.lds, st);
      case 3: return launch_gemm_v<kCosine, 3, true, true, true>(ga, p.blocks, p.lds, st);
      default: return launch_gemm_v<kCosine, 4, true, true, true>(ga, p.blocks, p.lds, st);
    }
  }
  switch (p.nqf) {
    case 2: return launch_gemm_v<kDot, 2, true, true, true>(ga, p.blocks, p.lds, st);
    case 3: return launch_gemm_v<kDot, 3, true, true, true>(ga, p.blocks, p.lds, st);
    default: return launch_gemm_v<kDot, 4, true, true, true>(ga, p.blocks, p.lds, st);
  }
}

}  // namespace vdb
