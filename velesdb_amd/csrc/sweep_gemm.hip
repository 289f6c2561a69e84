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

#include "vdb_device.hpp"
#include "vdb_kernels.hpp"

namespace vdb {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kGemmBM = 128;  // rows per block tile
constexpr int kGemmBK = 32;   // floats per k-tile (one 128-B line per row)

struct GemmSweepArgs {
  SweepArgs s;
  uint32_t KT;     // k-tiles per row: 4 * ceil(dim / 128)
  uint32_t G;      // row groups (= top-k lists per query), multiple of 8
  uint32_t nqt;    // query tiles
  uint32_t qper;   // queries per tile (<= 32 * NQF)
  uint32_t cap;    // candidate buffer entries per query (k < cap <= 64)
};

template <int METRIC, int NQF, bool QVEC>
__global__ __launch_bounds__(256, 2) void sweep_topk_gemm_f32(GemmSweepArgs ga) {
  constexpr int BM = kGemmBM, BK = kGemmBK, BN = 32 * NQF;
  constexpr bool HIB = true;  // cosine and dot: higher is better
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
  volatile uint32_t* ovf = reinterpret_cast<volatile uint32_t*>(tail + (size_t)BN * CAP * 8 + (size_t)BN * 16);  // overflow token

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wib >> 1, wq = wib & 1;

  // block -> (query tile, row group): the nqt query tiles of a row group sit on one XCD (blockIdx % 8)
  const uint32_t bid = blockIdx.x;
  const uint32_t xcd = bid & 7u, slot_id = bid >> 3;
  const uint32_t qt = slot_id % ga.nqt;
  const uint32_t g = (slot_id / ga.nqt) * 8u + xcd;
  const uint32_t q0 = qt * ga.qper;
  const uint32_t nq_t = min(ga.qper, a.nq - q0);
  const float* queries = a.queries + (size_t)q0 * a.q_stride;

  if (tid < BN) {
    cnts[tid] = 0;
    tauk[tid] = kKeyInvalid;
    qn[tid] = 0.0f;
  }
  if (tid == 0) *ovf = 0u;
  __syncthreads();
  if (METRIC == kCosine) {  // canonical query norms (same as every other kernel)
    const int d4 = (int)((a.dim + 3) / 4);
    for (uint32_t b = wib; b < nq_t; b += 4) {
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
      const float n = sqrtf(butterfly_all(nacc));
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

  // ---- staging: thread t moves the 16-B slot (t & 7) of rows (t >> 3) + 32 j ----
  const int st_slot = tid & 7, st_row = tid >> 3;
  float4 ra[4], rb[NQF];
  uint32_t ld_rt = g, ld_kt = 0;  // (row tile, k-tile) of the NEXT step to load
  uint32_t pend_kf = 0;           // k offset of the loads currently held in ra / rb
  // Branch-free: every load is issued unconditionally from a clamped (valid) address and zeroed by a select —
  // a conditional load makes hipcc branch around it and wait vmcnt(0) per element (serialised round trips).
  auto gload = [&]() __attribute__((always_inline)) {
    const uint32_t kf = ld_kt * BK + st_slot * 4;
    const bool kin = kf < (uint32_t)a.row_stride;
    const uint32_t kfa = kin ? kf : 0u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      uint32_t row = ld_rt * BM + st_row + 32 * j;
      row = row < a.n_rows ? row : a.n_rows - 1;  // tail rows: re-read the last row, masked in the epilogue
      ra[j] = ld4(a.rows + (size_t)row * a.row_stride + kfa);
    }
#pragma unroll
    for (int j = 0; j < NQF; j++) {
      const uint32_t q = st_row + 32 * j;
      const float* qp = queries + (size_t)(q < nq_t ? q : 0u) * a.q_stride;
      if (QVEC) {
        rb[j] = ld4(qp + (kf < a.dim ? kf : 0u));
      } else {
        const uint32_t dl = a.dim - 1;
        rb[j] = make_float4(qp[min(kf, dl)], qp[min(kf + 1, dl)], qp[min(kf + 2, dl)], qp[min(kf + 3, dl)]);
      }
    }
    pend_kf = kf;  // the zero-fill selects run in lds_store, after the multiply, so nothing waits on the loads here
    if (++ld_kt == ga.KT) {
      ld_kt = 0;
      ld_rt += ga.G;
    }
  };
  auto lds_store = [&]() __attribute__((always_inline)) {
    float* Ab = As;
    float* Bb = Bs;
    const bool kin = pend_kf < (uint32_t)a.row_stride;
    const bool k0 = pend_kf < a.dim, k1 = pend_kf + 1 < a.dim, k2 = pend_kf + 2 < a.dim, k3 = pend_kf + 3 < a.dim;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int row = st_row + 32 * j;
      float4 v = ra[j];
      v.x = kin ? v.x : 0.f;
      v.y = kin ? v.y : 0.f;
      v.z = kin ? v.z : 0.f;
      v.w = kin ? v.w : 0.f;
      *reinterpret_cast<float4*>(Ab + row * BK + ((st_slot ^ ((row >> 1) & 7)) << 2)) = v;
    }
#pragma unroll
    for (int j = 0; j < NQF; j++) {
      const int q = st_row + 32 * j;
      const bool qin = (uint32_t)q < nq_t;
      float4 v = rb[j];
      v.x = (qin && k0) ? v.x : 0.f;
      v.y = (qin && k1) ? v.y : 0.f;
      v.z = (qin && k2) ? v.z : 0.f;
      v.w = (qin && k3) ? v.w : 0.f;
      *reinterpret_cast<float4*>(Bb + q * BK + ((st_slot ^ ((q >> 1) & 7)) << 2)) = v;
    }
  };

  f32x4 acc[4][NQF];
#pragma unroll
  for (int rf = 0; rf < 4; rf++)
#pragma unroll
    for (int t = 0; t < NQF; t++) acc[rf][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- compaction of the candidate buffers this wave owns (queries wib, wib+4, ...) ----
  auto compact = [&]() __attribute__((always_inline)) {
    // lane l looks at query wib + 4*l
    const uint32_t bq = (uint32_t)wib + 4u * (uint32_t)lane;
    uint64_t need = __ballot(bq < nq_t && cnts[bq < (uint32_t)BN ? bq : 0] > k);
    while (need) {
      const int src = __ffsll((long long)need) - 1;
      need &= need - 1;
      const uint32_t b = (uint32_t)wib + 4u * (uint32_t)src;
      const uint32_t n = min(cnts[b], CAP);
      uint64_t* cb = cand + (size_t)b * CAP;
      const uint64_t key = (uint32_t)lane < n ? cb[lane] : kKeyInvalid;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n; j++) rank += (readlane64(key, (int)j) < key) ? 1u : 0u;  // keys are unique
      if ((uint32_t)lane < n && rank < k) cb[rank] = key;
      if ((uint32_t)lane < n && rank == k - 1) tauk[b] = key;
      if (lane == 0) cnts[b] = k;
    }
  };

  if (total) {
    gload();
    lds_store();
  }
  __syncthreads();
  uint32_t kt = 0, rt = g;
  uint32_t token = 0;  // ++ per epilogue round, block-uniform: a value written to *ovf is never reused
  for (uint32_t it = 0; it < total; it++) {
    const bool more = it + 1 < total;
    if (more) gload();
    {  // ---- multiply k-tile `it` out of LDS ----
#pragma unroll
      for (int m = 0; m < 2; m++) {
        const int slot = m * 4 + (lane >> 4);
        float4 av[4], bv[NQF];
#pragma unroll
        for (int rf = 0; rf < 4; rf++) {
          const int row = wr * 64 + rf * 16 + (lane & 15);
          av[rf] = ld4(As + row * BK + ((slot ^ ((row >> 1) & 7)) << 2));
        }
#pragma unroll
        for (int t = 0; t < NQF; t++) {
          const int q = wq * 16 * NQF + t * 16 + (lane & 15);
          bv[t] = ld4(Bs + q * BK + ((slot ^ ((q >> 1) & 7)) << 2));
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
          for (int rf = 0; rf < 4; rf++) {
            const float ax = c == 0 ? av[rf].x : (c == 1 ? av[rf].y : (c == 2 ? av[rf].z : av[rf].w));
#pragma unroll
            for (int t = 0; t < NQF; t++) {
              const float bx = c == 0 ? bv[t].x : (c == 1 ? bv[t].y : (c == 2 ? bv[t].z : bv[t].w));
              acc[rf][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, bx, acc[rf][t], 0, 0, 0);
            }
          }
        }
      }
    }
    if (++kt < ga.KT) {
      __syncthreads();  // every wave is done reading the tile
      if (more) lds_store();
      __syncthreads();
      continue;
    }
    // ---- last k-tile of a row tile: epilogue = filter, exact finish, append; compaction between the barriers ----
    uint64_t done = 0;  // bit (rf*NQF + t)*4 + r: element already appended (overflow rounds must not append twice)
    for (bool first = true;; first = false) {
      bool failed = false;
      ++token;
      float cut[NQF];
      uint64_t tk[NQF];
#pragma unroll
      for (int t = 0; t < NQF; t++) {
        const uint32_t b = wq * 16 * NQF + t * 16 + (lane & 15);
        tk[t] = tauk[b];
        const float tf = tk[t] == kKeyInvalid ? __uint_as_float(0xFF800000u) : key_score<HIB>(tk[t]);
        cut[t] = tf - (fabsf(tf) * 1.9073486e-6f + 1e-37f);
      }
#pragma unroll
      for (int rf = 0; rf < 4; rf++) {
        __builtin_amdgcn_sched_barrier(0);  // one 16-row slab at a time: keeps the epilogue's live set small
        const uint32_t rbase = rt * BM + wr * 64 + rf * 16 + 4 * (lane >> 4);
        f32x4 vn;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const uint32_t row = rbase + r;
          vn[r] = (METRIC == kCosine) ? a.norms[row < a.n_rows ? row : a.n_rows - 1] : 1.0f;
        }
#pragma unroll
        for (int t = 0; t < NQF; t++) {
          const uint32_t b = wq * 16 * NQF + t * 16 + (lane & 15);
          const f32x4 d = acc[rf][t];
          // pass unless clearly below the k-th best; NaN / inf / zero-norm cases always pass to the exact path
          bool pass[4];
          bool any = false;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float rq = (METRIC == kCosine) ? __builtin_amdgcn_rcpf(qn_t[t] * vn[r]) : 1.0f;
            pass[r] = !(d[r] * rq < cut[t]) && rbase + r < a.n_rows && b < nq_t &&
                      !((done >> ((rf * NQF + t) * 4 + r)) & 1ull);
            any |= pass[r];
          }
          if (__ballot(any) == 0) continue;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            if (pass[r]) {
              const uint32_t row = rbase + r;
              float dv = d[r];
              asm volatile("" : "+v"(dv));  // pins the exact finish + key packing inside the (rare) branch: hipcc
                                            // otherwise speculates all 64 keys up front (128 VGPRs, spills)
              const float score = finish_score<METRIC>(dv, qn_t[t], vn[r]);
              const uint64_t key = make_key<HIB>(score, row);
              bool take = key < tk[t];
              if (take && a.alive) take = a.alive[row] != 0;  // soft-deleted rows are filtered where it is rare
              if (take) {
                const uint32_t idx = atomicAdd(&cnts[b], 1u);
                if (idx < CAP) {
                  cand[(size_t)b * CAP + idx] = key;
                  done |= 1ull << ((rf * NQF + t) * 4 + r);
                } else {
                  failed = true;  // buffer full: compact, then offer again
                }
              } else {
                done |= 1ull << ((rf * NQF + t) * 4 + r);
              }
            }
          }
        }
      }
      if (failed) *ovf = token;
      __syncthreads();  // barrier 1: tile reads done, appends visible
      const bool again = *ovf == token;
      if (first && more) lds_store();
      compact();
      __syncthreads();  // barrier 2: next tile and the compacted lists visible
      if (!again) break;
    }
#pragma unroll
    for (int rf = 0; rf < 4; rf++)
#pragma unroll
      for (int t = 0; t < NQF; t++) acc[rf][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    kt = 0;
    rt += ga.G;
  }
  __syncthreads();
  for (uint32_t b = wib; b < nq_t; b += 4) {
    const uint32_t c = min(cnts[b], k);  // <= k entries: whatever order (the merge kernel scans them all)
    uint64_t* out = a.part_keys + ((size_t)(q0 + b) * ga.G + g) * k;
    for (uint32_t e = lane; e < k; e += 64) out[e] = e < c ? cand[(size_t)b * CAP + e] : kKeyInvalid;
  }
}

// ---- host side ---------------------------------------------------------------------------------
uint32_t sweep_gemm_cap(uint32_t k) { return k <= 16 ? 32u : 64u; }  // candidate buffer entries per query
size_t sweep_gemm_lds_bytes(int nqf, uint32_t k) {
  const size_t BN = (size_t)32 * nqf;
  return ((size_t)kGemmBM * kGemmBK * 4 + BN * kGemmBK * 4 + BN * sweep_gemm_cap(k) * 8 + BN * 16 + 16 + 15) & ~(size_t)15;
}

void sweep_gemm_plan(uint32_t nq, uint32_t n_rows, int n_cus, uint32_t k, GemmPlan* p) {
  p->nqt = (nq + 127) / 128;
  p->qper = (nq + p->nqt - 1) / p->nqt;
  p->nqf = (int)((p->qper + 31) / 32);
  if (p->nqf < 2) p->nqf = 2;
  p->lds = sweep_gemm_lds_bytes(p->nqf, k);
  const int per_cu = p->lds * 2 <= 160 * 1024 ? 2 : 1;
  const uint32_t ntiles = (n_rows + kGemmBM - 1) / kGemmBM;
  uint32_t G = (uint32_t)std::max(1, n_cus * per_cu / (int)p->nqt);
  G = std::min(G, ntiles);
  G = (G + 7) / 8 * 8;  // whole XCD rounds (groups beyond the last row tile emit empty lists)
  p->G = G;
  p->blocks = (int)(G * p->nqt);
}

template <int METRIC, int NQF, bool QVEC>
static hipError_t launch_gemm_v(const GemmSweepArgs& ga, int blocks, size_t lds, hipStream_t st) {
  static bool done = false;
  if (lds > 64 * 1024 && !done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_gemm_f32<METRIC, NQF, QVEC>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL((sweep_topk_gemm_f32<METRIC, NQF, QVEC>), dim3(blocks), dim3(256), lds, st, ga);
  return hipGetLastError();
}
template <int METRIC, int NQF>
static hipError_t launch_gemm_t(const GemmSweepArgs& ga, bool qvec, int blocks, size_t lds, hipStream_t st) {
  return qvec ? launch_gemm_v<METRIC, NQF, true>(ga, blocks, lds, st)
              : launch_gemm_v<METRIC, NQF, false>(ga, blocks, lds, st);
}

hipError_t launch_sweep_gemm(int metric, const GemmPlan& p, const SweepArgs& a, hipStream_t st) {
  GemmSweepArgs ga;
  ga.s = a;
  ga.KT = (a.dim + 127) / 128 * 4;
  ga.G = p.G;
  ga.nqt = p.nqt;
  ga.qper = p.qper;
  ga.cap = sweep_gemm_cap(a.k);
  // queries readable as aligned float4?
  const bool qvec = a.dim % 4 == 0 && a.q_stride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.queries) & 15) == 0;
  if (metric == kCosine) {
    switch (p.nqf) {
      case 2: return launch_gemm_t<kCosine, 2>(ga, qvec, p.blocks, p.lds, st);
      case 3: return launch_gemm_t<kCosine, 3>(ga, qvec, p.blocks, p.lds, st);
      default: return launch_gemm_t<kCosine, 4>(ga, qvec, p.blocks, p.lds, st);
    }
  }
  switch (p.nqf) {
    case 2: return launch_gemm_t<kDot, 2>(ga, qvec, p.blocks, p.lds, st);
    case 3: return launch_gemm_t<kDot, 3>(ga, qvec, p.blocks, p.lds, st);
    default: return launch_gemm_t<kDot, 4>(ga, qvec, p.blocks, p.lds, st);
  }
}

}  // namespace vdb
