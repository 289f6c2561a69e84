// sweep.hip — exact "distance sweep + top-k" kernels (HnswIndex::search_brute_force,
// crates/velesdb-core/src/index/hnsw/index/search.rs:176-219; brute_force_search_parallel,
// batch.rs:223-244) and the score-all kernels behind DistanceEngine::batch_distance /
// GpuAccelerator::batch_* (native/distance.rs:21-24; gpu/gpu_backend.rs:157,355,397).
//
// Design (MI355X-first, HBM-bound):
//   * corpus rows are contiguous in HBM (row stride = dim rounded up to 4 floats), so a wave
//     reads a row as float4 per lane: 1 KiB per load instruction, fully coalesced;
//   * one wave owns whole rows: lane l keeps ONE fmaf chain per (row, query) over chunks
//     l, l+64, ... (the canonical order) — queries live in VGPRs, rows are streamed once;
//   * 64 (row,query) partials per lane are combined by a transposed xor-butterfly so each lane
//     ends with one finished score: 63 shuffles per 64 scores instead of 384;
//   * top-k is fused: each wave keeps a sorted k-list per query in LDS and only touches it
//     when a score beats the current k-th best (ballot-driven, rare after warm-up);
//   * the 4 waves of a block fold their lists in LDS; one k-list per block goes to HBM and a
//     one-block-per-query merge kernel finishes (same threshold + insert scheme).
// Algorithmic HBM bytes per launch: n_rows * dim * 4 (+ 4 B/row of norms for cosine).
#include <algorithm>

#include "vdb_probe_env.hpp"
#include "vdb_device.hpp"
#include "vdb_kernels.hpp"

namespace vdb {

// ------------------------------------------------------------------------------------------
// f32 sweep with fused top-k.  B = queries per pass (power of two <= 64), RPG = 64/B rows per
// group, CPL = float4 chunks per lane (dim == CPL*256) or 0 for any dim (generic, slower).
// grid.x * 4 waves; wave w owns row groups w, w+W, ...   LDS: lists[B][k] u64 | cnt[B] | lock[B] (block-shared
// top-k lists, vdb_device.hpp shared_list_offer) | generic-dim query scratch.
// ------------------------------------------------------------------------------------------
template <int METRIC, int B, int CPL>
__global__ __launch_bounds__(256) void sweep_topk_f32(SweepArgs a) {
  constexpr int OP = (METRIC == kEuclidean) ? kOpL2 : kOpDot;
  constexpr int RPG = 64 / B;
  constexpr bool HIB = higher_is_better(METRIC);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = (int)(threadIdx.x >> 6);
  const uint32_t wave = blockIdx.x * 4 + wib;
  const uint32_t nwaves = gridDim.x * 4;
  const uint32_t k = a.k;
  lds_vu64* lists = (lds_vu64*)(lds_void_p)(smem);
  lds_vu32* cnts = (lds_vu32*)(lds_void_p)(smem + (size_t)B * k * 8);
  uint32_t* locks = reinterpret_cast<uint32_t*>(smem + (size_t)B * k * 8 + (size_t)B * 4);
  float* qgen = reinterpret_cast<float*>(smem + ((((size_t)B * k * 8 + (size_t)B * 8) + 15) & ~(size_t)15));  // generic path only
  uint32_t nq_here = a.nq, slot0 = 0;
  if (a.qmap) {  // gathered mode (SweepArgs): this block row's share of the listed queries
    const uint32_t listed = *a.qcount;
    slot0 = blockIdx.y * (uint32_t)B;
    if ((a.qcount_max && listed > a.qcount_max) || slot0 >= listed) return;  // (uniform per block, before the first barrier)
    nq_here = min((uint32_t)B, listed - slot0);
  }
  if (threadIdx.x < B) {
    cnts[threadIdx.x] = 0;
    locks[threadIdx.x] = 0;
  }
  __syncthreads();

  const int d4 = (int)((a.dim + 3) / 4);  // chunks per row
  // ---- queries into registers (CPL>0) or LDS (generic), plus their canonical norms ----
  float4 q[B][CPL > 0 ? CPL : 1];
  float qnorm_mine = 0.0f;  // lane l keeps the norm of query (l % B)
  if (CPL > 0) {
#pragma unroll
    for (int b = 0; b < B; b++) {
      float nacc = 0.0f;
#pragma unroll
      for (int j = 0; j < CPL; j++) {
        q[b][j] = (b < (int)nq_here) ? ld4(a.queries + (size_t)(a.qmap ? a.qmap[slot0 + b] : (uint32_t)b) * a.q_stride +
                                           (size_t)(j * 64 + lane) * 4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        nacc = chain4<kOpDot>(nacc, q[b][j], q[b][j]);
      }
      if (METRIC == kCosine) {
        float n = sqrtf(butterfly_all(nacc));
        if ((lane % B) == b) qnorm_mine = n;
      }
    }
  } else {
    // generic: stage queries (zero padded to d4*4) in LDS, shared by the 4 waves
    const int qlen = d4 * 4;
    for (int i = threadIdx.x; i < B * qlen; i += 256) {
      int b = i / qlen, e = i % qlen;
      qgen[i] = (b < (int)nq_here && e < (int)a.dim) ? a.queries[(size_t)(a.qmap ? a.qmap[slot0 + b] : (uint32_t)b) * a.q_stride + e] : 0.0f;
    }
    __syncthreads();
    if (METRIC == kCosine) {
      for (int b = 0; b < B; b++) {
        float nacc = 0.0f;
        for (int c = lane; c < d4; c += 64) {
          float4 x = ld4(qgen + (size_t)b * qlen + c * 4);
          int nv = (int)a.dim - c * 4;
          nacc = nv >= 4 ? chain4<kOpDot>(nacc, x, x) : chain4_tail<kOpDot>(nacc, x, x, nv);
        }
        float n = sqrtf(butterfly_all(nacc));
        if ((lane % B) == b) qnorm_mine = n;
      }
    }
  }

  const uint32_t ngroups = (a.n_rows + RPG - 1) / RPG;
  for (uint32_t g = wave; g < ngroups; g += nwaves) {
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; i++) acc[i] = 0.0f;
    const uint32_t row0 = g * RPG;
    if (CPL > 0) {
      constexpr int RB = (RPG >= 4) ? 4 : RPG;  // rows loaded together
#pragma unroll
      for (int r = 0; r < RPG; r += RB) {
        float4 v[RB][CPL > 0 ? CPL : 1];
#pragma unroll
        for (int rr = 0; rr < RB; rr++) {
          uint32_t row = row0 + r + rr;
          row = row < a.n_rows ? row : a.n_rows - 1;  // tail rows: re-read the last row, masked later
          const float* p = a.rows + (size_t)row * a.row_stride + (size_t)lane * 4;
#pragma unroll
          for (int j = 0; j < CPL; j++) v[rr][j] = ld4(p + j * 256);
        }
#pragma unroll
        for (int rr = 0; rr < RB; rr++)
#pragma unroll
          for (int b = 0; b < B; b++) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < CPL; j++) s = chain4<OP>(s, q[b][j], v[rr][j]);
            acc[(r + rr) * B + b] = s;
          }
      }
    } else {
      const int qlen = d4 * 4;
#pragma unroll
      for (int r = 0; r < RPG; r++) {
        uint32_t row = row0 + r;
        row = row < a.n_rows ? row : a.n_rows - 1;
        const float* p = a.rows + (size_t)row * a.row_stride;
        for (int c = lane; c < d4; c += 64) {
          float4 x = ld4(p + c * 4);
          int nv = (int)a.dim - c * 4;
#pragma unroll
          for (int b = 0; b < B; b++) {
            float4 qq = ld4(qgen + (size_t)b * qlen + c * 4);
            acc[r * B + b] = nv >= 4 ? chain4<OP>(acc[r * B + b], qq, x) : chain4_tail<OP>(acc[r * B + b], qq, x, nv);
          }
        }
      }
    }
    treduce64(acc, lane);
    // lane l now owns pair idx = l: row r = l / B, query b = l % B
    const int b = lane % B;
    const uint32_t row = row0 + lane / B;
    const bool valid = row < a.n_rows && b < (int)nq_here;
    float vnorm = 1.0f;
    if (METRIC == kCosine && valid) vnorm = a.norms[row];
    const float score = finish_score<METRIC>(acc[0], qnorm_mine, vnorm);
    const uint64_t key = valid ? make_key<HIB>(score, row) : kKeyInvalid;
    const uint32_t c_b = cnts[b];
    const uint64_t tau = (c_b == k) ? lists[(size_t)b * k + (k - 1)] : kKeyInvalid;
    uint64_t mask = __ballot(key < tau);
    while (mask) {
      const int src = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const uint64_t kk = readlane64(key, src);
      if (a.alive && a.alive[key_row(kk)] == 0) continue;  // soft-deleted rows are filtered where it is rare
      const int bb = src % B;
      shared_list_offer(lists + (size_t)bb * k, cnts + bb, locks + bb, k, kk, lane);
    }
  }
  // ---- one list per query per block goes to HBM, padded with invalid keys ----
  __syncthreads();
  for (int b = wib; b < (int)nq_here && b < B; b += 4) {
    const uint32_t c = cnts[b];
    uint64_t* out = a.part_keys + ((size_t)(slot0 + b) * gridDim.x + blockIdx.x) * k;
    for (uint32_t e = lane; e < k; e += 64) out[e] = e < c ? lists[(size_t)b * k + e] : kKeyInvalid;
  }
}

// ------------------------------------------------------------------------------------------
// f32 sweep for LARGE query tiles: B = 16 or 32 queries per corpus pass.  Same arithmetic, same
// top-k scheme and the same output layout as sweep_topk_f32, but the queries live in LDS (B x dim
// floats, read back as ds_read_b128: lane l reads its own chunk, conflict-free) instead of VGPRs,
// so the per-lane state is only the 64 (row, query) partials + RPG = 64/B rows in flight.
// WAVES waves per block share the query tile.  dim == CPL*256 only.
// LDS: q[B][dim] f32 | lists[WAVES][B][k] u64 | cnt[WAVES][B] u32.
// VALU work per row is B*dim FMAs: at B = 32 that is ~0.3 ms per 1M x 768 rows, still under the
// ~0.5 ms the HBM stream needs, so the kernel stays bandwidth-bound while serving 4x the queries of
// the register-resident B = 8 tile.
// ------------------------------------------------------------------------------------------
template <int METRIC, int B, int CPL, int WAVES>
__global__ __launch_bounds__(WAVES * 64, (B == 32 ? 4 : 3)) void sweep_topk_f32_qlds(SweepArgs a) {
  constexpr int OP = (METRIC == kEuclidean) ? kOpL2 : kOpDot;
  constexpr int RPG = 64 / B;
  constexpr bool HIB = higher_is_better(METRIC);
  constexpr int DIM = CPL * 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform => scalar row addressing
  const uint32_t wave = blockIdx.x * WAVES + wib;
  const uint32_t nwaves = gridDim.x * WAVES;
  const uint32_t k = a.k;
  float* qs = reinterpret_cast<float*>(smem);
  unsigned char* lbase = smem + (size_t)B * DIM * 4;
  // ONE sorted k-list per query per block, shared by the block's waves under a per-query LDS lock: a wave
  // streams only n_rows / (#waves) rows, so per-wave lists would each need their own ~k*ln(rows/k) insertions
  // (x B queries x thousands of waves); shared, the threshold tightens WAVES times faster.
  lds_vu64* lists = (lds_vu64*)(lds_void_p)(lbase);
  lds_vu32* cnts = (lds_vu32*)(lds_void_p)(lbase + (size_t)B * k * 8);
  uint32_t* locks = reinterpret_cast<uint32_t*>(lbase + (size_t)B * k * 8 + (size_t)B * 4);
  if (threadIdx.x < B) {
    cnts[threadIdx.x] = 0;
    locks[threadIdx.x] = 0;
  }
  // query tile -> LDS, two queries interleaved element by element: slot ((p*CPL + j)*2 + h)*64 + lane holds
  // {q[2p][e], q[2p+1][e], q[2p][e+1], q[2p+1][e+1]} for e = 4*(j*64+lane) + 2h, so ONE v_pk_fma_f32 advances the
  // chains of two queries by one element (same fmaf chain per query as everywhere else) and every
  // ds_read_b128 is contiguous across lanes
  for (int i = threadIdx.x; i < B * (DIM / 4); i += WAVES * 64) {
    const int b = i / (DIM / 4), c = i % (DIM / 4);
    const float4 v = (b < (int)a.nq) ? ld4(a.queries + (size_t)b * a.q_stride + (size_t)c * 4)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    const int pr = b >> 1, sl = b & 1, j = c >> 6, l = c & 63;
    float* d0 = qs + ((size_t)((pr * CPL + j) * 2 + 0) * 64 + l) * 4;
    float* d1 = qs + ((size_t)((pr * CPL + j) * 2 + 1) * 64 + l) * 4;
    d0[sl] = v.x;
    d0[2 + sl] = v.y;
    d1[sl] = v.z;
    d1[2 + sl] = v.w;
  }
  __syncthreads();
  float qnorm_mine = 0.0f;  // lane l keeps the norm of query (l % B)
  if (METRIC == kCosine) {
#pragma unroll 1
    for (int b = 0; b < B; b++) {
      const int pr = b >> 1, sl = b & 1;
      float nacc = 0.0f;
#pragma unroll
      for (int j = 0; j < CPL; j++) {
        const float* d0 = qs + ((size_t)((pr * CPL + j) * 2 + 0) * 64 + lane) * 4;
        const float* d1 = qs + ((size_t)((pr * CPL + j) * 2 + 1) * 64 + lane) * 4;
        const float4 x = make_float4(d0[sl], d0[2 + sl], d1[sl], d1[2 + sl]);
        nacc = chain4<kOpDot>(nacc, x, x);
      }
      const float n = sqrtf(butterfly_all(nacc));
      if ((lane % B) == b) qnorm_mine = n;
    }
  }

  const uint32_t ngroups = (a.n_rows + RPG - 1) / RPG;
  // Software pipeline: the row chunks of step (g, j+1) are requested from HBM before the FMAs of step
  // (g, j) start, and the query chunk of b+1 is requested from LDS before the FMAs of b; only
  // 2*RPG float4 of row data and 2 float4 of query data are live next to the 64 partials.
  auto row_ptr = [&](uint32_t g, int r) -> const float* {
    uint32_t row = g * RPG + r;
    row = row < a.n_rows ? row : a.n_rows - 1;  // tail rows: re-read the last row, masked later
    return a.rows + (size_t)row * a.row_stride + (size_t)lane * 4;  // scalar base + lane offset
  };
  const int myb = lane % B;
  // row chunks are requested TWO steps ahead (cur = this step, n1 = next, n2 = the one after): with one step
  // of lookahead a CU has only #waves x RPG KiB in flight, below the latency-bandwidth product of HBM
  // Ring of three buffers: step s computes from ring[s % 3] and requests step s+2 into ring[(s+2) % 3].  With
  // CPL a multiple of 3 the roles repeat every group and the ring index is a compile-time constant (no
  // register moves); otherwise the buffers are rotated by moves.
  constexpr bool STATIC_RING = (CPL % 3) == 0;
  float4 ring[3][RPG];
  if (wave < ngroups) {
    const uint32_t g1 = (CPL > 1) ? wave : (wave + nwaves < ngroups ? wave + nwaves : wave);
#pragma unroll
    for (int r = 0; r < RPG; r++) {
      ring[0][r] = ld4(row_ptr(wave, r));
      ring[1][r] = ld4(row_ptr(g1, r) + (CPL > 1 ? 256 : 0));
    }
  }
  for (uint32_t g = wave; g < ngroups; g += nwaves) {
    f32x2 acc2[32];
#pragma unroll
    for (int i = 0; i < 32; i++) acc2[i] = f32x2{0.0f, 0.0f};
    const uint32_t row0 = g * RPG;
    const uint32_t gn = g + nwaves < ngroups ? g + nwaves : g;  // last group: harmless re-read
    const uint32_t gn2 = g + 2 * nwaves < ngroups ? g + 2 * nwaves : g;
    const uint32_t row = row0 + lane / B;
    const bool valid = row < a.n_rows && myb < (int)a.nq;
    float vnorm = 1.0f;
#pragma unroll
    for (int j = 0; j < CPL; j++) {
      constexpr int dummy = 0;
      (void)dummy;
      const int ic = STATIC_RING ? (j % 3) : 0, in2 = STATIC_RING ? ((j + 2) % 3) : 2;
      float4(&cur)[RPG] = ring[ic];
      float4(&n2)[RPG] = ring[in2];
      const float* qj = qs + (size_t)(j * 2 * 64 + lane) * 4;  // pair 0, half 0 of chunk j
      constexpr int PSTRIDE = CPL * 2 * 64 * 4;                 // floats between consecutive pairs
      float4 h0 = ld4(qj), h1 = ld4(qj + 64 * 4);
      // sched_barrier: keep the requests ahead of the FMAs that hide them (the scheduler otherwise sinks
      // every load to its first use to save registers)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pr = 0; pr < B / 2; pr++) {
        float4 n0 = h0, n1 = h1;
        if (pr + 1 < B / 2) {
          n0 = ld4(qj + (size_t)(pr + 1) * PSTRIDE);
          n1 = ld4(qj + (size_t)(pr + 1) * PSTRIDE + 64 * 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        // element-major order: the RPG accumulators touched between two steps of the same chain are independent,
        // so the packed FMAs issue back to back without dependency stalls
#pragma unroll
        for (int r = 0; r < RPG; r++)
          acc2[(r * B) / 2 + pr] = pk_step<OP>(acc2[(r * B) / 2 + pr], f32x2{h0.x, h0.y}, cur[r].x);
#pragma unroll
        for (int r = 0; r < RPG; r++)
          acc2[(r * B) / 2 + pr] = pk_step<OP>(acc2[(r * B) / 2 + pr], f32x2{h0.z, h0.w}, cur[r].y);
#pragma unroll
        for (int r = 0; r < RPG; r++)
          acc2[(r * B) / 2 + pr] = pk_step<OP>(acc2[(r * B) / 2 + pr], f32x2{h1.x, h1.y}, cur[r].z);
#pragma unroll
        for (int r = 0; r < RPG; r++)
          acc2[(r * B) / 2 + pr] = pk_step<OP>(acc2[(r * B) / 2 + pr], f32x2{h1.z, h1.w}, cur[r].w);
        h0 = n0;
        h1 = n1;
        if (pr == 0) {
          // HBM requests of the NEXT step go out here, behind the first FMAs of this step: the wait in front
          // of those FMAs (conservatively vmcnt(0) at the loop head) then only covers rows that are due anyway.
          // The row norm is requested first so that waiting for it later does not wait for the prefetches.
          if (j == 0 && METRIC == kCosine) vnorm = a.norms[row < a.n_rows ? row : a.n_rows - 1];
#pragma unroll
          for (int r = 0; r < RPG; r++) {
            // step s+2: chunk j+2 of this group, or of the next group; with ONE chunk per row (CPL = 1) it is
            // chunk 0 of the group after the next
            if (j + 2 < CPL)
              n2[r] = ld4(row_ptr(g, r) + (j + 2) * 256);
            else if (j + 2 - CPL < CPL)
              n2[r] = ld4(row_ptr(gn, r) + (j + 2 - CPL) * 256);
            else
              n2[r] = ld4(row_ptr(gn2, r) + (j + 2 - 2 * CPL) * 256);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!STATIC_RING) {
#pragma unroll
        for (int r = 0; r < RPG; r++) {
          ring[0][r] = ring[1][r];
          ring[1][r] = ring[2][r];
        }
      }
    }
    float acc[64];
#pragma unroll
    for (int i = 0; i < 32; i++) {
      acc[2 * i] = acc2[i].x;
      acc[2 * i + 1] = acc2[i].y;
    }
    treduce64(acc, lane);
    const int b = myb;
    const float score = finish_score<METRIC>(acc[0], qnorm_mine, vnorm);
    const uint64_t key = valid ? make_key<HIB>(score, row) : kKeyInvalid;
    const uint32_t c_b = cnts[b];
    const uint64_t tau = (c_b == k) ? lists[(size_t)b * k + (k - 1)] : kKeyInvalid;
    uint64_t mask = __ballot(key < tau);
    while (mask) {
      const int src = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const uint64_t kk = readlane64(key, src);
      if (a.alive && a.alive[key_row(kk)] == 0) continue;  // soft-deleted rows are filtered where it is rare
      const int bb = src % B;
      shared_list_offer(lists + (size_t)bb * k, cnts + bb, locks + bb, k, kk, lane);
    }
  }
  __syncthreads();
  for (int b = wib; b < (int)a.nq && b < B; b += WAVES) {
    const uint32_t c = cnts[b];
    uint64_t* out = a.part_keys + ((size_t)b * gridDim.x + blockIdx.x) * k;
    for (uint32_t e = lane; e < k; e += 64) out[e] = e < c ? lists[(size_t)b * k + e] : kKeyInvalid;
  }
}

// ------------------------------------------------------------------------------------------
// MFMA f32 sweep for Cosine / DotProduct: corpus rows x query tile as a true f32 contraction on the matrix
// cores (v_mfma_f32_16x16x4_f32: exact f32, every product rounded once, k-ordered chain — the f32 VALU rate
// without spending VALU issue slots, and no cross-lane reduction at all).  The VALU kernels above top out
// near 50 TFLOP/s, which makes a 32-query pass compute-bound; the matrix pipe sustains ~3x that.
//   * a wave owns 16 rows and NQT*16 queries; lane l = (i = l&15, kk = l>>4).
//   * A operand straight from HBM: load m of macro step U reads, for row i, the 64 contiguous bytes
//     k = 128U + 16m .. +15 (the four kk lanes take 16 B each), so a 128-B line is consumed by two
//     back-to-back load instructions and a macro step covers 512 contiguous bytes per row.
//   * B operand from LDS, stored once per block in fragment order [U][tile][m][lane][4] so every
//     ds_read_b128 is contiguous across lanes; 4*NQT MFMAs per NQT LDS reads.
//   * summation order ("mode M" of the oracle): for U, for m in 0..7, for c in 0..3, for kk in 0..3:
//     k = 128U + 16m + 4kk + c — one fmaf chain per (row, query), dim zero-padded to a multiple of 128.
//   * epilogue: D[i = 4*(l>>4)+r][j = l&15]; a cheap conservative filter (reciprocal multiply, 16-ulp margin)
//     in front of the exact finish + the block-shared top-k lists.
// LDS: q fragments KU*NQT*8 KiB | lists[B][k] u64 | cnt[B] | lock[B] | qnorm[B].
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int METRIC, int NQT, int WAVES, bool FULL128>
__global__ __launch_bounds__(WAVES * 64, 4) void sweep_topk_mfma_f32(SweepArgs a, uint32_t KU) {
  constexpr int B = NQT * 16;
  constexpr bool HIB = true;  // cosine and dot: higher is better
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t wave = blockIdx.x * WAVES + wib;
  const uint32_t nwaves = gridDim.x * WAVES;
  const uint32_t k = a.k;
  float* qs = reinterpret_cast<float*>(smem);
  const size_t qbytes = (size_t)KU * NQT * 8192;
  lds_vu64* lists = (lds_vu64*)(lds_void_p)(smem + qbytes);
  lds_vu32* cnts = (lds_vu32*)(lds_void_p)(smem + qbytes + (size_t)B * k * 8);
  uint32_t* locks = reinterpret_cast<uint32_t*>(smem + qbytes + (size_t)B * k * 8 + (size_t)B * 4);
  float* qn = reinterpret_cast<float*>(smem + qbytes + (size_t)B * k * 8 + (size_t)B * 8);
  const int d4 = (int)((a.dim + 3) / 4);
  uint32_t nq_here = a.nq, slot0 = 0;
  if (a.qmap) {  // gathered mode: this block row's share of the listed queries
    const uint32_t listed = *a.qcount;
    slot0 = blockIdx.y * (uint32_t)B;
    if (listed > a.qcount_max || slot0 >= listed) return;  // (uniform per block, before the first barrier)
    nq_here = min((uint32_t)B, listed - slot0);
  }

  for (uint32_t i = threadIdx.x; i < KU * NQT * 2048; i += WAVES * 64) qs[i] = 0.0f;
  if (threadIdx.x < B) {
    cnts[threadIdx.x] = 0;
    locks[threadIdx.x] = 0;
    qn[threadIdx.x] = 0.0f;
  }
  __syncthreads();
  for (uint32_t idx = threadIdx.x; idx < (uint32_t)B * a.dim; idx += WAVES * 64) {
    const uint32_t b = idx / a.dim, kx = idx % a.dim;
    if (b < nq_here) {
      const uint32_t t = b >> 4, j = b & 15, U = kx >> 7, m = (kx >> 4) & 7, kk = (kx >> 2) & 3, c = kx & 3;
      const uint32_t qrow = a.qmap ? a.qmap[slot0 + b] : b;
      qs[((((size_t)U * NQT + t) * 8 + m) * 64 + kk * 16 + j) * 4 + c] = a.queries[(size_t)qrow * a.q_stride + kx];
    }
  }
  if (METRIC == kCosine) {  // canonical query norms (same as every other kernel)
    for (uint32_t b = wib; b < (uint32_t)B && b < nq_here; b += WAVES) {
      const float* qp = a.queries + (size_t)(a.qmap ? a.qmap[slot0 + b] : b) * a.q_stride;
      float nacc = 0.0f;
      for (int c = lane; c < d4; c += 64) {
        const int nv = (int)a.dim - c * 4;
        float4 x;
        if (nv >= 4) {
          x = ld4(qp + c * 4);
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
  float qn_t[NQT];
#pragma unroll
  for (int t = 0; t < NQT; t++) qn_t[t] = qn[t * 16 + (lane & 15)];

  const uint32_t kk32 = (uint32_t)(lane >> 4) * 4;  // this lane's float offset inside every 16-float load group
  const uint32_t ntiles = (a.n_rows + 15) / 16;

  auto row_ptr = [&](uint32_t g) -> const float* {
    uint32_t row = g * 16 + (lane & 15);
    row = row < a.n_rows ? row : a.n_rows - 1;  // tail rows: re-read the last row, masked in the epilogue
    return a.rows + (size_t)row * a.row_stride + kk32;
  };
  auto load_a = [&](const float* rp, uint32_t U, float4(&dst)[8]) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const uint32_t k0 = U * 128 + m * 16 + kk32;
      if (FULL128 || k0 < (uint32_t)a.row_stride)
        dst[m] = ld4(rp + (size_t)U * 128 + m * 16);
      else
        dst[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  float4 A0[8], A1[8];
  if (wave < ntiles) load_a(row_ptr(wave), 0, A0);
  for (uint32_t g = wave; g < ntiles; g += nwaves) {
    const float* rp = row_ptr(g);
    const uint32_t gnext = g + nwaves < ntiles ? g + nwaves : g;
    const float* rpn = row_ptr(gnext);
    float vn[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t row = g * 16 + 4 * (lane >> 4) + r;
      vn[r] = (METRIC == kCosine) ? a.norms[row < a.n_rows ? row : a.n_rows - 1] : 1.0f;
    }
    f32x4 acc[NQT];
#pragma unroll
    for (int t = 0; t < NQT; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](uint32_t U, const float4(&av)[8]) {
      const float* qb = qs + ((size_t)U * NQT * 8 * 64 + lane) * 4;
      constexpr bool DBUF = NQT <= 2;  // three tiles: no register room for the look-ahead copy of the B fragments
      float4 bq[NQT], bn[NQT];
#pragma unroll
      for (int t = 0; t < NQT; t++) bq[t] = ld4(qb + (size_t)(t * 8) * 256);
#pragma unroll
      for (int m = 0; m < 8; m++) {
        if (DBUF && m + 1 < 8) {
#pragma unroll
          for (int t = 0; t < NQT; t++) bn[t] = ld4(qb + (size_t)(t * 8 + m + 1) * 256);
        }
        const float ax[4] = {av[m].x, av[m].y, av[m].z, av[m].w};
#pragma unroll
        for (int c = 0; c < 4; c++) {
#pragma unroll
          for (int t = 0; t < NQT; t++) {
            const float bx = c == 0 ? bq[t].x : (c == 1 ? bq[t].y : (c == 2 ? bq[t].z : bq[t].w));
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax[c], bx, acc[t], 0, 0, 0);
          }
        }
        if (DBUF) {
#pragma unroll
          for (int t = 0; t < NQT; t++) bq[t] = bn[t];
        } else if (m + 1 < 8) {
#pragma unroll
          for (int t = 0; t < NQT; t++) bq[t] = ld4(qb + (size_t)(t * 8 + m + 1) * 256);
        }
      }
    };
    // macro steps, two at a time so the ring roles are static; the next tile's first chunk is requested
    // during the last step of this one
    for (uint32_t U = 0; U < KU; U += 2) {
      if (U + 1 < KU) load_a(rp, U + 1, A1); else load_a(rpn, 0, A1);
      __builtin_amdgcn_sched_barrier(0);  // one macro step of lookahead, no more (registers)
      compute(U, A0);
      __builtin_amdgcn_sched_barrier(0);
      if (U + 1 < KU) {
        if (U + 2 < KU) load_a(rp, U + 2, A0); else load_a(rpn, 0, A0);
        __builtin_amdgcn_sched_barrier(0);
        compute(U + 1, A1);
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int m = 0; m < 8; m++) A0[m] = A1[m];
      }
    }
    // ---- epilogue: NQT x 4 raw dots per lane.  Almost none of them can enter a top-k list once the lists
    // have warmed up, so the exact finish (IEEE divide for cosine, key packing, LDS reads) runs only behind a
    // cheap conservative filter: an approximate score (multiply by a reciprocal, ~1 ulp) is compared with the
    // k-th best score of its query minus a 16-ulp margin; whatever passes is finished exactly and offered. ----
    float tau_f[NQT];  // k-th best score of this lane's query in tile t (-inf while the list is not full)
    asm volatile("" ::: "memory");  // (fresh, plain LDS reads issued together: vdb_device.hpp list_tau_relaxed)
#pragma unroll
    for (int t = 0; t < NQT; t++) {
      const uint32_t b = t * 16 + (lane & 15);
      const uint64_t tk = list_tau_relaxed(lists, cnts, b, k);
      tau_f[t] = tk != kKeyInvalid ? key_score<HIB>(tk) : __uint_as_float(0xFF800000u);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t row = g * 16 + 4 * (lane >> 4) + r;
#pragma unroll
      for (int t = 0; t < NQT; t++) {
        const uint32_t b = t * 16 + (lane & 15);
        const float dotv = acc[t][r];
        const float rq = (METRIC == kCosine) ? __builtin_amdgcn_rcpf(qn_t[t] * vn[r]) : 1.0f;
        const float approx = dotv * rq;
        // pass unless clearly below the threshold; NaN / inf / zero-norm cases always pass to the exact path
        const float margin = fabsf(tau_f[t]) * 1.9073486e-6f + 1e-37f;
        const bool maybe = !(approx < tau_f[t] - margin) && row < a.n_rows && b < nq_here;
        uint64_t mask = __ballot(maybe);
        if (mask == 0) continue;
        const float score = finish_score<METRIC>(dotv, qn_t[t], vn[r]);
        const uint64_t key = maybe ? make_key<HIB>(score, row) : kKeyInvalid;
        const uint64_t tau = (cnts[b] == k) ? lists[(size_t)b * k + (k - 1)] : kKeyInvalid;
        mask = __ballot(key < tau);
        while (mask) {
          const int src = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          const uint64_t kk = readlane64(key, src);
          if (a.alive && a.alive[key_row(kk)] == 0) continue;
          const int bb = t * 16 + (src & 15);
          shared_list_offer(lists + (size_t)bb * k, cnts + bb, locks + bb, k, kk, lane);
        }
      }
    }
  }
  __syncthreads();
  for (int b = wib; b < (int)nq_here && b < B; b += WAVES) {
    const uint32_t c = cnts[b];
    uint64_t* out = a.part_keys + ((size_t)(slot0 + b) * gridDim.x + blockIdx.x) * k;
    for (uint32_t e = lane; e < k; e += 64) out[e] = e < c ? lists[(size_t)b * k + e] : kKeyInvalid;
  }
}

// ------------------------------------------------------------------------------------------
// bf16 GEMM-distance sweep (BASELINE configs[3]: 10 M x 768 bf16, 1024 queries per batch): the corpus is kept as
// bf16 (round-to-nearest-even of the f32 rows, half the HBM bytes), the query tile is rounded the same way, and
// rows x queries is a true bf16 contraction on the matrix cores with f32 accumulation
// (v_mfma_f32_16x16x32_bf16) — the semantics of the reference's half_precision::dot_product /
// cosine_similarity on VectorData::BF16 (half_precision.rs:199-255: f32 accumulate over the bf16 values).
// Same structure as sweep_topk_mfma_f32: a wave owns 16 rows x NQT*16 queries; lane (i = l&15, kk = l>>4)
// reads 8 bf16 = 16 B per load, the four kk lanes 64 contiguous bytes of row i, eight loads 512 B (a macro
// step of 256 k-values); query fragments in LDS in operand order.  One MFMA retires 32 k-values, so the kernel
// is HBM-bound up to ~100 queries per pass: NQT = 6 (96 queries, 144 KiB of LDS) is the large tile.
// Products of bf16 values are exact in f32; the order in which one instruction adds its 32 products is not
// documented, so parity with the oracle is by tolerance (tests), not bit for bit.
// LDS: q fragments KU*NQT*8 KiB | lists[B][k] u64 | cnt[B] | lock[B] | qnorm[B].
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);  // NaN stays NaN (quiet)
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// rows f32 -> bf16 copy + norm of the ROUNDED row (canonical lane-chain order over the rounded values); rho_max_bits != nullptr:
// the row's rounding residual ratio |x - bf16(x)| / |x| (f64 sums: no f32 row underflows them) raises a device scalar — the
// measured error bound of the level-2 selection (sweep_split.hip select_eps_q).  Rows with non-finite elements contribute
// nothing: their scores are non-finite too and never proven.
__global__ __launch_bounds__(256) void prep_bf16_rows(const float* rows, uint64_t row_stride, uint16_t* out,
                                                      uint64_t out_stride, float* norms, uint32_t row0, uint32_t n_rows,
                                                      uint32_t dim, uint32_t* rho_max_bits) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t r = wave; r < n_rows; r += nwaves) {
    const uint32_t row = row0 + r;
    const float* p = rows + (size_t)row * row_stride;
    uint16_t* o = out + (size_t)row * out_stride;
    float acc = 0.0f;
    double se = 0.0, sx = 0.0;
    for (uint32_t c = lane; c * 4 < out_stride; c += 64) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint32_t i = c * 4 + e;
        if (i < out_stride) {
          const float v = i < dim ? p[i] : 0.0f;
          const uint16_t h = i < dim ? f32_to_bf16_rne(v) : (uint16_t)0;
          o[i] = h;
          if (i < dim) {
            const float x = bf16_to_f32(h);
            acc = __builtin_fmaf(x, x, acc);
            if (rho_max_bits) {
              const float d = v - x;  // exact in f32
              se += (double)d * (double)d;
              sx += (double)v * (double)v;
            }
          }
        }
      }
    }
    const float n = sqrtf(butterfly_all(acc));
    if (lane == 0 && norms) norms[row] = n;
    if (rho_max_bits) {
#pragma unroll
      for (int o2 = 32; o2 > 0; o2 >>= 1) {
        se += __shfl_xor(se, o2, 64);
        sx += __shfl_xor(sx, o2, 64);
      }
      if (lane == 0 && sx > 0.0) {
        const float rho = (float)(sqrt(se / sx) * 1.0000002);  // (rounded to f32: the pad covers it)
        if (rho == rho && rho < __uint_as_float(0x7F800000u)) atomicMax(rho_max_bits, __float_as_uint(rho));  // >= +0: bit order = value order
      }
    }
  }
}

struct Bf16SweepArgs {
  const uint16_t* rows;   // [n_rows][row_stride] bf16, row_stride % 8 == 0
  const float* norms;     // norm of the rounded rows (cosine)
  const uint8_t* alive;
  const float* queries;   // f32, rounded to bf16 while staging
  uint64_t* part_keys;
  uint64_t row_stride, q_stride;
  uint32_t n_rows, dim, nq, k, KU;
};

template <int METRIC, int NQT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, (WAVES == 16 ? 4 : 2)) void sweep_topk_mfma_bf16(Bf16SweepArgs a) {
  constexpr int B = NQT * 16;
  constexpr bool HIB = true;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t wave = blockIdx.x * WAVES + wib;
  const uint32_t nwaves = gridDim.x * WAVES;
  const uint32_t k = a.k, KU = a.KU;
  uint16_t* qs = reinterpret_cast<uint16_t*>(smem);
  const size_t qbytes = (size_t)KU * NQT * 8192;
  lds_vu64* lists = (lds_vu64*)(lds_void_p)(smem + qbytes);
  lds_vu32* cnts = (lds_vu32*)(lds_void_p)(smem + qbytes + (size_t)B * k * 8);
  uint32_t* locks = reinterpret_cast<uint32_t*>(smem + qbytes + (size_t)B * k * 8 + (size_t)B * 4);
  float* qn = reinterpret_cast<float*>(smem + qbytes + (size_t)B * k * 8 + (size_t)B * 8);

  for (uint32_t i = threadIdx.x; i < KU * NQT * 2048; i += WAVES * 64) reinterpret_cast<uint32_t*>(qs)[i] = 0u;
  if (threadIdx.x < B) {
    cnts[threadIdx.x] = 0;
    locks[threadIdx.x] = 0;
    qn[threadIdx.x] = 0.0f;
  }
  __syncthreads();
  // fragment order: slot ((U*NQT + t)*8 + m)*64 + (kk*16 + j) holds q[j][256U + 32m + 8kk + 0..7]
  for (uint32_t idx = threadIdx.x; idx < (uint32_t)B * a.dim; idx += WAVES * 64) {
    const uint32_t b = idx / a.dim, kx = idx % a.dim;
    if (b < a.nq) {
      const uint32_t t = b >> 4, j = b & 15, U = kx >> 8, m = (kx >> 5) & 7, kk = (kx >> 3) & 3, e = kx & 7;
      qs[((((size_t)U * NQT + t) * 8 + m) * 64 + kk * 16 + j) * 8 + e] = f32_to_bf16_rne(a.queries[(size_t)b * a.q_stride + kx]);
    }
  }
  if (METRIC == kCosine) {  // norm of the ROUNDED query, canonical lane-chain order
    for (uint32_t b = wib; b < (uint32_t)B && b < a.nq; b += WAVES) {
      const float* qp = a.queries + (size_t)b * a.q_stride;
      float nacc = 0.0f;
      for (uint32_t c = lane; c * 4 < a.dim; c += 64)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint32_t i = c * 4 + e;
          if (i < a.dim) {
            const float x = bf16_to_f32(f32_to_bf16_rne(qp[i]));
            nacc = __builtin_fmaf(x, x, nacc);
          }
        }
      const float n = sqrtf(butterfly_all(nacc));
      if (lane == 0) qn[b] = n;
    }
  }
  __syncthreads();
  float qn_t[NQT];
#pragma unroll
  for (int t = 0; t < NQT; t++) qn_t[t] = qn[t * 16 + (lane & 15)];

  const uint32_t kk8 = (uint32_t)(lane >> 4) * 8;  // element offset of this lane inside every 32-element load group
  const uint32_t ntiles = (a.n_rows + 15) / 16;
  auto row_ptr = [&](uint32_t g) -> const uint16_t* {
    uint32_t row = g * 16 + (lane & 15);
    row = row < a.n_rows ? row : a.n_rows - 1;
    return a.rows + (size_t)row * a.row_stride + kk8;
  };
  auto load_a = [&](const uint16_t* rp, uint32_t U, uint4(&dst)[8]) {
#pragma unroll
    for (int m = 0; m < 8; m++) {
      const uint32_t k0 = U * 256 + m * 32 + kk8;
      if (k0 < (uint32_t)a.row_stride)
        dst[m] = *reinterpret_cast<const uint4*>(rp + (size_t)U * 256 + m * 32);
      else
        dst[m] = make_uint4(0u, 0u, 0u, 0u);
    }
  };
  uint4 A0[8], A1[8];
  if (wave < ntiles) load_a(row_ptr(wave), 0, A0);
  for (uint32_t g = wave; g < ntiles; g += nwaves) {
    const uint16_t* rp = row_ptr(g);
    const uint32_t gnext = g + nwaves < ntiles ? g + nwaves : g;
    const uint16_t* rpn = row_ptr(gnext);
    float vn[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t row = g * 16 + 4 * (lane >> 4) + r;
      vn[r] = (METRIC == kCosine) ? a.norms[row < a.n_rows ? row : a.n_rows - 1] : 1.0f;
    }
    f32x4 acc[NQT];
#pragma unroll
    for (int t = 0; t < NQT; t++) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto compute = [&](uint32_t U, const uint4(&av)[8]) {
      const uint16_t* qb = qs + ((size_t)U * NQT * 8 * 64 + lane) * 8;
#pragma unroll
      for (int m = 0; m < 8; m++) {
        const bf16x8 afrag = __builtin_bit_cast(bf16x8, av[m]);
#pragma unroll
        for (int t = 0; t < NQT; t++) {
          const uint4 bu = *reinterpret_cast<const uint4*>(qb + (size_t)(t * 8 + m) * 512);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag, __builtin_bit_cast(bf16x8, bu), acc[t], 0, 0, 0);
        }
      }
    };
    for (uint32_t U = 0; U < KU; U += 2) {
      if (U + 1 < KU) load_a(rp, U + 1, A1); else load_a(rpn, 0, A1);
      __builtin_amdgcn_sched_barrier(0);
      compute(U, A0);
      __builtin_amdgcn_sched_barrier(0);
      if (U + 1 < KU) {
        if (U + 2 < KU) load_a(rp, U + 2, A0); else load_a(rpn, 0, A0);
        __builtin_amdgcn_sched_barrier(0);
        compute(U + 1, A1);
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int m = 0; m < 8; m++) A0[m] = A1[m];
      }
    }
    float tau_f[NQT];
    asm volatile("" ::: "memory");  // (fresh, plain LDS reads issued together: vdb_device.hpp list_tau_relaxed)
#pragma unroll
    for (int t = 0; t < NQT; t++) {
      const uint32_t b = t * 16 + (lane & 15);
      const uint64_t tk = list_tau_relaxed(lists, cnts, b, k);
      tau_f[t] = tk != kKeyInvalid ? key_score<HIB>(tk) : __uint_as_float(0xFF800000u);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t row = g * 16 + 4 * (lane >> 4) + r;
#pragma unroll
      for (int t = 0; t < NQT; t++) {
        const uint32_t b = t * 16 + (lane & 15);
        const float dotv = acc[t][r];
        const float rq = (METRIC == kCosine) ? __builtin_amdgcn_rcpf(qn_t[t] * vn[r]) : 1.0f;
        const float approx = dotv * rq;
        const float margin = fabsf(tau_f[t]) * 1.9073486e-6f + 1e-37f;
        // (a norm below f32::EPSILON makes the score 0.0 whatever the dot product: never filtered by the approximation)
        const bool tiny = METRIC == kCosine && (qn_t[t] < kHalfNormEps || vn[r] < kHalfNormEps);
        const bool maybe = (tiny || !(approx < tau_f[t] - margin)) && row < a.n_rows && b < a.nq;
        uint64_t mask = __ballot(maybe);
        if (mask == 0) continue;
        const float score = finish_score_half<METRIC>(dotv, qn_t[t], vn[r]);
        const uint64_t key = maybe ? make_key<HIB>(score, row) : kKeyInvalid;
        const uint64_t tau = (cnts[b] == k) ? lists[(size_t)b * k + (k - 1)] : kKeyInvalid;
        mask = __ballot(key < tau);
        while (mask) {
          const int src = __ffsll((long long)mask) - 1;
          mask &= mask - 1;
          const uint64_t kk = readlane64(key, src);
          if (a.alive && a.alive[key_row(kk)] == 0) continue;
          const int bb = t * 16 + (src & 15);
          shared_list_offer(lists + (size_t)bb * k, cnts + bb, locks + bb, k, kk, lane);
        }
      }
    }
  }
  __syncthreads();
  for (int b = wib; b < (int)a.nq && b < B; b += WAVES) {
    const uint32_t c = cnts[b];
    uint64_t* out = a.part_keys + ((size_t)b * gridDim.x + blockIdx.x) * k;
    for (uint32_t e = lane; e < k; e += 64) out[e] = e < c ? lists[(size_t)b * k + e] : kKeyInvalid;
  }
}

// ------------------------------------------------------------------------------------------
// merge: one wave per query scans the per-wave lists with the same threshold + insert scheme
// and writes ids / scores best-first.
// ------------------------------------------------------------------------------------------
// the lowered bound of the selection stage's next launch (pool scores may err by delta either way)
__device__ __forceinline__ uint64_t reseed_key(float s, float delta) {
  const float lowered = s - 2.0f * delta * 1.01f - fabsf(s) * 1e-6f;
  return lowered == lowered ? make_key<true>(lowered, 0u) : kKeyInvalid;  // NaN: no bound
}
template <bool HIB>
__global__ __launch_bounds__(256) void merge_topk(MergeArgs m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = (int)(threadIdx.x >> 6);
  const uint32_t qi = blockIdx.x;
  if (m.active && (qi >= *m.active || (m.active_max && *m.active > m.active_max))) return;  // (uniform per block)
  if (m.skip_cnt && *m.skip_cnt <= m.skip_le) return;
  if (m.gate && m.gate[qi] == 0u) return;
  const uint32_t kin = m.k;                     // entries per partial list
  const uint32_t k = m.k_out ? m.k_out : m.k;   // entries kept (k_out > k: a candidate pool for a re-scoring stage)
  lds_vu64* list = (lds_vu64*)(lds_void_p)(smem) + (size_t)wib * k;
  lds_vu32* wcnt = (lds_vu32*)(lds_void_p)(smem + (size_t)4 * k * 8);
  uint32_t cnt = 0;
  const uint64_t* keys = m.part_keys + (size_t)qi * (m.list_stride ? m.list_stride : m.n_lists) * kin;
  const uint32_t total = m.n_lists * kin;  // slots beyond a list's count hold kKeyInvalid
  // wave w scans the slice [lo, hi); 4 independent loads in flight per lane
  const uint32_t per = ((total + 3) / 4 + 255) / 256 * 256;
  const uint32_t lo = wib * per, hi = min(total, lo + per);
  for (uint32_t base = lo; base < hi; base += 256) {
    uint64_t key[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t i = base + u * 64 + lane;
      key[u] = i < hi ? keys[i] : kKeyInvalid;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint64_t tau = (cnt == k) ? list[k - 1] : kKeyInvalid;
      uint64_t mask = __ballot(key[u] < tau);
      while (mask) {
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        wave_list_insert(list, cnt, k, readlane64(key[u], src), lane);
      }
    }
  }
  if (lane == 0) wcnt[wib] = cnt;
  __syncthreads();
  if (wib != 0) return;
  for (int w = 1; w < 4; w++) {
    lds_vu64* src = (lds_vu64*)(lds_void_p)(smem) + (size_t)w * k;
    const uint32_t cs = wcnt[w];
    for (uint32_t e = 0; e < cs; e++) {
      const uint64_t kk = src[e];
      if (cnt == k && kk >= list[k - 1]) break;
      wave_list_insert(list, cnt, k, kk, lane);
    }
  }
  for (uint32_t e = lane; e < k; e += 64) {
    if (e < cnt) {
      const uint64_t key = list[e];
      const uint32_t row = key_row(key);
      const float s = key_score<HIB>(key);  // raw compute_distance value (search.rs:209)
      m.out_ids[(size_t)qi * k + e] = m.ext_ids ? m.ext_ids[row] : (uint64_t)row + m.row_base;
      m.out_scores[(size_t)qi * k + e] = s;
      if (m.reseed_delta && e + 1 == m.reseed_k) m.reseed_tau[qi] = reseed_key(s, m.reseed_delta[qi]);
    } else {
      m.out_ids[(size_t)qi * k + e] = ~0ull;
      m.out_scores[(size_t)qi * k + e] = __uint_as_float(0x7FC00000u);
    }
  }
  if (lane == 0) {
    m.out_n[qi] = cnt;
    if (m.reseed_delta && (cnt < m.reseed_k || m.reseed_k == 0)) m.reseed_tau[qi] = kKeyInvalid;
  }
}

// ------------------------------------------------------------------------------------------
// The same merge by SELECTION, for queries whose partial lists fit the LDS (<= kMergeSelectMaxKeys slots, k <= 256).
// merge_topk inserts one key at a time into a sorted list — ~1 000 cycles per insertion into a 64-entry list: the
// selection stage's final merge (a 64-candidate pool per query out of ~3 300 slots) took 200 us per 1 024 queries.  Here:
//   1. the valid keys are compacted into LDS (ballot + one LDS atomic per wave and 256 slots);
//   2. the k-th smallest key is built bit by bit from the top: with P the bits decided so far, c = #{keys <= P | low bits all
//      ones}; c < k means the k-th smallest has the bit set.  One ballot-count per wave and 256 keys, one LDS atomic per wave,
//      one barrier per bit; it stops at the first bound that holds exactly k keys (typically after the ~35 leading bits);
//   3. the <= k keys under the bound are compacted and ranked among themselves (one key per thread), and written in order.
// Keys are distinct (a row sits in exactly one partial list; equal keys would still get distinct places: ties by position).
// Output identical to merge_topk.
// (Round 4, measured and rejected: ranking every valid key against all of them out of LDS broadcasts — no barrier per bit — when few
// are valid.  256 keys x 1 024 queries: 15.1 us against 11.0 us for the bit-by-bit bound; 300 - 1 000 keys, four per thread: 32 - 71 us
// against 15 - 22 us.  profiles/r04z_merge_rank_variants.txt.)
// ------------------------------------------------------------------------------------------
constexpr uint32_t kMergeSelectMaxKeys = 7900;  // x 8 B + counters + a 256-key selection buffer: inside the default 64-KiB window
constexpr uint32_t kMergeSelectMaxK = 256;
constexpr uint32_t kMergeHeadsMaxQueries = 64;  // merge_topk_heads: calls of few queries whose sweeps left many partial lists
template <bool HIB>
__global__ __launch_bounds__(256) void merge_topk_select(MergeArgs m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const uint32_t tid = threadIdx.x;
  const uint32_t qi = blockIdx.x;
  if (m.active && (qi >= *m.active || (m.active_max && *m.active > m.active_max))) return;  // (uniform per block)
  if (m.skip_cnt && *m.skip_cnt <= m.skip_le) return;
  if (m.gate && m.gate[qi] == 0u) return;
  const uint32_t kin = m.k;
  const uint32_t k = m.k_out ? m.k_out : m.k;
  const uint32_t total = m.n_lists * kin;
  uint64_t* ks = reinterpret_cast<uint64_t*>(smem);                       // [total] compacted keys
  uint64_t* sel = ks + total;                                              // [kMergeSelectMaxK] the keys under the bound
  uint32_t* cnt_s = reinterpret_cast<uint32_t*>(sel + kMergeSelectMaxK);  // [64] one counter per bit + [64] n, n_sel
  const uint64_t* keys = m.part_keys + (size_t)qi * (m.list_stride ? m.list_stride : m.n_lists) * kin;
  if (tid < 66) cnt_s[tid] = 0;
  __syncthreads();
  auto mbcnt = [](uint64_t mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); };
  for (uint32_t base = 0; base < total; base += 256) {
    const uint32_t i = base + tid;
    const uint64_t key = i < total ? keys[i] : kKeyInvalid;
    const bool valid = key != kKeyInvalid;
    const uint64_t mask = __ballot(valid);
    uint32_t off = 0;
    if (lane == 0 && mask) off = atomicAdd(&cnt_s[64], (uint32_t)__popcll(mask));
    off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
    if (valid) ks[off + mbcnt(mask)] = key;
  }
  __syncthreads();
  const uint32_t n = cnt_s[64];
  const uint32_t cnt = min(n, k);
  uint64_t bound = kKeyInvalid;  // every key passes
  if (n > k) {
    uint64_t P = 0;
    for (int bit = 63; bit >= 0; bit--) {
      const uint64_t cand = P | ((1ull << bit) - 1ull);
      uint32_t wc = 0;
      for (uint32_t base = 0; base < n; base += 256) {
        const uint32_t i = base + tid;
        wc += (uint32_t)__popcll(__ballot(i < n && ks[i] <= cand));
      }
      if (lane == 0 && wc) atomicAdd(&cnt_s[bit], wc);
      __syncthreads();
      const uint32_t c = cnt_s[bit];
      if (c == k) {  // (block-uniform)
        bound = cand;
        break;
      }
      if (c < k) P |= 1ull << bit;
      bound = P;  // after the last bit: P is the k-th smallest key itself
    }
  }
  for (uint32_t base = 0; base < n; base += 256) {
    const uint32_t i = base + tid;
    const uint64_t key = i < n ? ks[i] : kKeyInvalid;
    const bool take = i < n && key <= bound;
    const uint64_t mask = __ballot(take);
    uint32_t off = 0;
    if (lane == 0 && mask) off = atomicAdd(&cnt_s[65], (uint32_t)__popcll(mask));
    off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
    if (take) {
      const uint32_t slot = off + mbcnt(mask);
      if (slot < kMergeSelectMaxK) sel[slot] = key;
    }
  }
  __syncthreads();
  const uint32_t ns = min(cnt_s[65], kMergeSelectMaxK);  // == cnt for distinct keys
  if (tid < ns) {
    const uint64_t key = sel[tid];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < ns; j++) {
      const uint64_t kj = sel[j];
      rank += (kj < key || (kj == key && j < tid)) ? 1u : 0u;
    }
    if (rank < k) {
      const uint32_t row = key_row(key);
      m.out_ids[(size_t)qi * k + rank] = m.ext_ids ? m.ext_ids[row] : (uint64_t)row + m.row_base;
      m.out_scores[(size_t)qi * k + rank] = key_score<HIB>(key);  // raw compute_distance value (search.rs:209)
      if (m.reseed_delta && rank + 1 == m.reseed_k) m.reseed_tau[qi] = reseed_key(key_score<HIB>(key), m.reseed_delta[qi]);
    }
  }
  for (uint32_t e = cnt + tid; e < k; e += 256) {
    m.out_ids[(size_t)qi * k + e] = ~0ull;
    m.out_scores[(size_t)qi * k + e] = __uint_as_float(0x7FC00000u);
  }
  if (tid == 0) {
    m.out_n[qi] = cnt;
    if (m.reseed_delta && (cnt < m.reseed_k || m.reseed_k == 0)) m.reseed_tau[qi] = kKeyInvalid;
  }
}

constexpr uint32_t kBitsFusedMaxK = 16;  // k of the extraction helpers below (4 k <= 64: their second level is one key per lane)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  auto a32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  v = min(a32[0], a32[1]);
  auto a16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = min(a16[0], a16[1]);
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));  // row_ror:8
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));  // row_ror:4
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false));  // row_ror:2
  v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}
// the smallest of the wave's 64 keys, in every lane (all invalid: kKeyInvalid): two 32-bit minimum butterflies on DPP / permlane
// swaps — the score word, then the row word among the lanes that tie
__device__ __forceinline__ uint64_t wave_min_key(uint64_t m) {
  const uint32_t hi = wave_min_u32((uint32_t)(m >> 32));
  const uint32_t lo = wave_min_u32((uint32_t)(m >> 32) == hi ? (uint32_t)m : 0xFFFFFFFFu);
  return ((uint64_t)hi << 32) | lo;
}
// the k smallest of a wave's 64 x R keys (R per lane, kKeyInvalid = none; keys distinct): k times { the lane's smallest, the wave's
// smallest of those, drop it }.  Lane e < k returns the e-th smallest; `mine` is consumed.
template <int R>
__device__ __forceinline__ uint64_t wave_k_smallest(uint64_t (&mine)[R], uint32_t k) {
  const uint32_t lane = (uint32_t)lane_id();
  uint64_t out = kKeyInvalid;
  for (uint32_t e = 0; e < k; e++) {
    uint64_t mloc = mine[0];
#pragma unroll
    for (int r = 1; r < R; r++) mloc = min(mloc, mine[r]);
    const uint64_t wm = wave_min_key(mloc);
    if (lane == e) out = wm;
    if (wm == kKeyInvalid) break;  // (wave-uniform) nothing left
#pragma unroll
    for (int r = 0; r < R; r++)
      if (mine[r] == wm) mine[r] = kKeyInvalid;
  }
  return out;
}
// the k smallest of a block's 256 x R keys: every wave's k smallest, then wave 0 the k smallest of those 4 k <= 64.  Returns, in wave
// 0, lane e < k: the block's e-th smallest; other waves / lanes: undefined.  wl: LDS scratch [4][kBitsFusedMaxK].  Block-uniform call
// (two barriers inside); k <= kBitsFusedMaxK.
template <int R>
__device__ __forceinline__ uint64_t block_k_smallest(uint64_t (&mine)[R], uint32_t k, uint64_t* wl) {
  const uint32_t lane = (uint32_t)lane_id(), wib = threadIdx.x >> 6;
  const uint64_t out = wave_k_smallest<R>(mine, k);
  __syncthreads();  // (wl may still be read from an earlier call)
  if (lane < k) wl[(size_t)wib * kBitsFusedMaxK + lane] = out;
  __syncthreads();
  uint64_t res = kKeyInvalid;
  if (wib == 0) {
    const uint32_t wsrc = lane / k, esrc = lane % k;
    uint64_t m2[1] = {wsrc < 4u ? wl[(size_t)wsrc * kBitsFusedMaxK + esrc] : kKeyInvalid};
    res = wave_k_smallest<1>(m2, k);
  }
  return res;
}
__device__ __forceinline__ uint64_t block_k_smallest(uint64_t mine, uint32_t k, uint64_t* wl) {
  uint64_t m1[1] = {mine};
  return block_k_smallest<1>(m1, k, wl);
}
// ------------------------------------------------------------------------------------------
// The same merge by EXTRACTION, for the small merges between the launches of the selection stage (<= 2 048 keys, k <= 16: the seed's
// 256 keys, 650 / 1 290 / 1 930 pool keys after the first three launches of a headline step).  merge_topk_select compacts into LDS
// and builds the k-th smallest key bit by bit — ~35 barriers; here a thread keeps its <= 8 keys in registers and the block extracts
// its k smallest (block_k_smallest: two barriers).  Equal keys keep their places (a seed row swept again may carry the same key):
// one instance is dropped per extraction.  Output identical to merge_topk_select, the next bound included (reseed_*).
// ------------------------------------------------------------------------------------------
constexpr uint32_t kMergeExtractMaxKeys = 8 * 256;
// wave_k_smallest for keys that may repeat: an extraction drops ONE instance (the lowest lane's first register)
template <int R>
__device__ __forceinline__ uint64_t wave_k_smallest_dup(uint64_t (&mine)[R], uint32_t k) {
  const uint32_t lane = (uint32_t)lane_id();
  uint64_t out = kKeyInvalid;
  for (uint32_t e = 0; e < k; e++) {
    uint64_t mloc = mine[0];
#pragma unroll
    for (int r = 1; r < R; r++) mloc = min(mloc, mine[r]);
    const uint64_t wm = wave_min_key(mloc);
    if (lane == e) out = wm;
    if (wm == kKeyInvalid) break;  // (wave-uniform) nothing left
    const uint64_t holders = __ballot(mloc == wm);
    if (lane == (uint32_t)__builtin_ctzll(holders)) {
      bool dropped = false;
#pragma unroll
      for (int r = 0; r < R; r++)
        if (!dropped && mine[r] == wm) {
          mine[r] = kKeyInvalid;
          dropped = true;
        }
    }
  }
  return out;
}
template <bool HIB>
__global__ __launch_bounds__(256) void merge_topk_extract(MergeArgs m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = (uint32_t)lane_id(), tid = threadIdx.x, wib = tid >> 6;
  const uint32_t qi = blockIdx.x;
  if (m.active && (qi >= *m.active || (m.active_max && *m.active > m.active_max))) return;  // (uniform per block)
  if (m.skip_cnt && *m.skip_cnt <= m.skip_le) return;
  if (m.gate && m.gate[qi] == 0u) return;
  const uint32_t kin = m.k;
  const uint32_t k = m.k_out ? m.k_out : m.k;
  const uint32_t total = m.n_lists * kin;
  uint64_t* wl = reinterpret_cast<uint64_t*>(smem);  // [4][kBitsFusedMaxK]
  const uint64_t* keys = m.part_keys + (size_t)qi * (m.list_stride ? m.list_stride : m.n_lists) * kin;
  uint64_t r8[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const uint32_t i = tid + 256u * (uint32_t)u;
    r8[u] = i < total ? keys[i] : kKeyInvalid;
  }
  const uint64_t mine = wave_k_smallest_dup<8>(r8, k);
  if (lane < k) wl[(size_t)wib * kBitsFusedMaxK + lane] = mine;
  __syncthreads();
  if (wib != 0) return;
  const uint32_t wsrc = lane / k, esrc = lane % k;
  uint64_t m2[1] = {wsrc < 4u ? wl[(size_t)wsrc * kBitsFusedMaxK + esrc] : kKeyInvalid};
  const uint64_t res = wave_k_smallest_dup<1>(m2, k);
  const uint32_t cnt = (uint32_t)__popcll(__ballot(lane < k && res != kKeyInvalid));
  if (lane < k) {
    if (lane < cnt) {
      const uint32_t row = key_row(res);
      m.out_ids[(size_t)qi * k + lane] = m.ext_ids ? m.ext_ids[row] : (uint64_t)row + m.row_base;
      m.out_scores[(size_t)qi * k + lane] = key_score<HIB>(res);  // raw compute_distance value (search.rs:209)
      if (m.reseed_delta && lane + 1 == m.reseed_k) m.reseed_tau[qi] = reseed_key(key_score<HIB>(res), m.reseed_delta[qi]);
    } else {
      m.out_ids[(size_t)qi * k + lane] = ~0ull;
      m.out_scores[(size_t)qi * k + lane] = __uint_as_float(0x7FC00000u);
    }
  }
  if (lane == 0) {
    m.out_n[qi] = cnt;
    if (m.reseed_delta && (cnt < m.reseed_k || m.reseed_k == 0)) m.reseed_tau[qi] = kKeyInvalid;
  }
}

// ------------------------------------------------------------------------------------------
// The merge for FEW queries over MANY partial lists (a one-query sweep leaves 1 024 lists x k keys: beyond merge_topk_select's LDS
// window, and merge_topk's serial insertions took 42.8 us of a 91-us one-query packed-bit call — as much as the sweep itself;
// profiles/r05m_*).  Heads first:
//   1. every list's smallest key (its "head"; the lists need not be sorted) goes to LDS — n_lists keys;
//   2. the k-th smallest HEAD bounds the answer: k lists hold a key <= it, so the k-th smallest key overall is <= it, and a list
//      whose head is larger holds nothing that matters.  Bit-by-bit selection as in merge_topk_select, over n_lists keys only;
//   3. the keys <= that bound — they all sit in the (at most k, keys are distinct) lists whose head passes: <= k x k_in keys — are
//      compacted and ranked among themselves, one key per thread.
// Output identical to merge_topk.  Used when (k_out or k) x k_in <= kMergeSelectMaxK and n_lists >= that count.
// ------------------------------------------------------------------------------------------
template <bool HIB>
__device__ __forceinline__ void merge_topk_heads_body(const MergeArgs& m, const uint32_t qi, unsigned char* smem) {
  const int lane = lane_id();
  const uint32_t tid = threadIdx.x;
  if (m.active && (qi >= *m.active || (m.active_max && *m.active > m.active_max))) return;  // (uniform per block)
  if (m.skip_cnt && *m.skip_cnt <= m.skip_le) return;
  if (m.gate && m.gate[qi] == 0u) return;
  const uint32_t kin = m.k;
  const uint32_t k = m.k_out ? m.k_out : m.k;
  const uint32_t nl = m.n_lists;
  uint64_t* heads = reinterpret_cast<uint64_t*>(smem);                     // [nl]
  uint64_t* sel = heads + nl;                                               // [kMergeSelectMaxK] the keys under the bound
  uint64_t* wl = sel + kMergeSelectMaxK;                                    // [4][kBitsFusedMaxK] block_k_smallest + [1] the bound
  uint32_t* cnt_s = reinterpret_cast<uint32_t*>(wl + 4 * kBitsFusedMaxK + 1);  // [64] one counter per bit, [64] valid heads, [65] selected, [66] lists under the bound
  uint32_t* plist = cnt_s + 68;                                             // [kMergeSelectMaxK] the lists under the bound
  const uint64_t* keys = m.part_keys + (size_t)qi * (m.list_stride ? m.list_stride : m.n_lists) * kin;
  if (tid < 68) cnt_s[tid] = 0;
  for (uint32_t l = tid; l < nl; l += 256) heads[l] = kKeyInvalid;
  __syncthreads();
  auto mbcnt = [](uint64_t mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); };
  // 1. heads: ONE coalesced pass over all keys, eight independent loads per thread in flight, the minimum per list by LDS atomics.
  //    (The first form gave a thread whole lists and walked each key by key — 40 dependent round trips at 1 024 lists of 10 — and the
  //    third step did the same for the lists under the bound: 36 us per merge, profiles/r05u_*.)
  {
    const uint32_t total = nl * kin;
    const uint32_t dl = 256u / kin, de = 256u % kin;  // a step of 256 keys in (list, entry) coordinates
    uint32_t l = tid / kin, e = tid % kin;
    for (uint32_t i0 = tid; i0 < total; i0 += 8u * 256u) {
      uint64_t kk8[8];
      uint32_t l8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const uint32_t i = i0 + (uint32_t)u * 256u;
        kk8[u] = i < total ? keys[i] : kKeyInvalid;
        l8[u] = l;
        l += dl;
        e += de;
        if (e >= kin) {
          e -= kin;
          l++;
        }
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (kk8[u] != kKeyInvalid) atomicMin(reinterpret_cast<unsigned long long*>(&heads[l8[u]]), (unsigned long long)kk8[u]);
    }
  }
  __syncthreads();
  uint32_t nvalid_mine = 0;
  for (uint32_t l = tid; l < nl; l += 256) nvalid_mine += heads[l] != kKeyInvalid ? 1u : 0u;
  {
    uint32_t v = nvalid_mine;
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) v += __shfl_xor(v, sft, 64);
    if (lane == 0 && v) atomicAdd(&cnt_s[64], v);
  }
  __syncthreads();
  const uint32_t nh = cnt_s[64];  // lists that hold anything
  uint64_t bound = kKeyInvalid;   // fewer than k non-empty lists: every key may matter
  if (nh > k && k <= kBitsFusedMaxK && nl <= 1024u) {
    // 2a. the k-th smallest head by extraction (a thread holds <= 4 heads): two barriers instead of one per bit
    uint64_t h4[4];
#pragma unroll
    for (int u = 0; u < 4; u++) h4[u] = tid + 256u * (uint32_t)u < nl ? heads[tid + 256u * (uint32_t)u] : kKeyInvalid;
    const uint64_t hk = block_k_smallest<4>(h4, k, wl);
    if (tid == k - 1) wl[4 * kBitsFusedMaxK] = hk;
    __syncthreads();
    bound = wl[4 * kBitsFusedMaxK];
  } else if (nh > k) {
    // 2b. bit by bit, as merge_topk_select
    uint64_t P = 0;
    for (int bit = 63; bit >= 0; bit--) {
      const uint64_t cand = P | ((1ull << bit) - 1ull);
      uint32_t wc = 0;
      for (uint32_t base = 0; base < nl; base += 256) {
        const uint32_t i = base + tid;
        wc += (uint32_t)__popcll(__ballot(i < nl && heads[i] <= cand));   // (an empty list's head is kKeyInvalid: above every bound)
      }
      if (lane == 0 && wc) atomicAdd(&cnt_s[bit], wc);
      __syncthreads();
      const uint32_t c = cnt_s[bit];
      if (c == k) {  // (block-uniform)
        bound = cand;
        break;
      }
      if (c < k) P |= 1ull << bit;
      bound = P;  // after the last bit: P is the k-th smallest head itself
    }
  }
  // 3. the lists under the bound (at most k: keys are distinct), then ONE load per key of those lists
  for (uint32_t base = 0; base < nl; base += 256) {
    const uint32_t l = base + tid;
    const bool mine = l < nl && heads[l] != kKeyInvalid && heads[l] <= bound;
    const uint64_t mask = __ballot(mine);
    if (!mask) continue;
    uint32_t off = 0;
    if (lane == 0) off = atomicAdd(&cnt_s[66], (uint32_t)__popcll(mask));
    off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
    if (mine) {
      const uint32_t slot = off + mbcnt(mask);
      if (slot < kMergeSelectMaxK) plist[slot] = l;
    }
  }
  __syncthreads();
  {
    const uint32_t np = min(cnt_s[66], kMergeSelectMaxK);
    const uint32_t nk = np * kin;  // <= k x k_in <= kMergeSelectMaxK (the launcher's condition)
    for (uint32_t base = 0; base < nk; base += 256) {
      const uint32_t t = base + tid;
      const uint64_t key = t < nk ? keys[(size_t)plist[t / kin] * kin + t % kin] : kKeyInvalid;
      const bool take = key != kKeyInvalid && key <= bound;
      const uint64_t mask = __ballot(take);
      if (!mask) continue;
      uint32_t off = 0;
      if (lane == 0) off = atomicAdd(&cnt_s[65], (uint32_t)__popcll(mask));
      off = (uint32_t)__builtin_amdgcn_readfirstlane((int)off);
      if (take) {
        const uint32_t slot = off + mbcnt(mask);
        if (slot < kMergeSelectMaxK) sel[slot] = key;
      }
    }
  }
  __syncthreads();
  const uint32_t ns = min(cnt_s[65], kMergeSelectMaxK);
  const uint32_t cnt = min(ns, k);
  if (tid < ns) {
    const uint64_t key = sel[tid];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < ns; j++) {
      const uint64_t kj = sel[j];
      rank += (kj < key || (kj == key && j < tid)) ? 1u : 0u;
    }
    if (rank < k) {
      const uint32_t row = key_row(key);
      m.out_ids[(size_t)qi * k + rank] = m.ext_ids ? m.ext_ids[row] : (uint64_t)row + m.row_base;
      m.out_scores[(size_t)qi * k + rank] = key_score<HIB>(key);  // raw compute_distance value (search.rs:209)
      if (m.reseed_delta && rank + 1 == m.reseed_k) m.reseed_tau[qi] = reseed_key(key_score<HIB>(key), m.reseed_delta[qi]);
    }
  }
  for (uint32_t e = cnt + tid; e < k; e += 256) {
    m.out_ids[(size_t)qi * k + e] = ~0ull;
    m.out_scores[(size_t)qi * k + e] = __uint_as_float(0x7FC00000u);
  }
  if (tid == 0) {
    m.out_n[qi] = cnt;
    if (m.reseed_delta && (cnt < m.reseed_k || m.reseed_k == 0)) m.reseed_tau[qi] = kKeyInvalid;
  }
}
template <bool HIB>
__global__ __launch_bounds__(256) void merge_topk_heads(MergeArgs m) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  merge_topk_heads_body<HIB>(m, blockIdx.x, smem);
}

// ------------------------------------------------------------------------------------------
// packed-bit sweep for Hamming / Jaccard (simd_explicit.rs:234-287,372-443 on the exact
// re-encoding bit = (x > 0.5)).  Lane per row; grid.y = query.  Rows are W words (16-B
// aligned), read as dwordx4.  Algorithmic bytes per launch: n_rows * W * 4 per query.
// ------------------------------------------------------------------------------------------
template <int METRIC>
__global__ __launch_bounds__(256) void sweep_topk_bits(BitsArgs a) {
  constexpr bool HIB = higher_is_better(METRIC);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = (int)(threadIdx.x >> 6);
  const uint32_t wave = blockIdx.x * 4 + wib;
  const uint32_t nwaves = gridDim.x * 4;
  const uint32_t qi = blockIdx.y;
  const uint32_t k = a.k;
  const uint32_t W = a.words;  // multiple of 4
  // LDS: list[k] u64 (block-shared, locked) | cnt, lock | query words
  lds_vu64* list = (lds_vu64*)(lds_void_p)(smem);
  lds_vu32* cnt = (lds_vu32*)(lds_void_p)(smem + (size_t)k * 8);
  uint32_t* lock = reinterpret_cast<uint32_t*>(smem + (size_t)k * 8 + 4);
  uint32_t* qw = reinterpret_cast<uint32_t*>(smem + (((size_t)k * 8 + 8 + 15) & ~(size_t)15));
  if (threadIdx.x == 0) {
    *cnt = 0;
    *lock = 0;
  }
  for (uint32_t i = threadIdx.x; i < W; i += 256) qw[i] = a.qbits[(size_t)qi * W + i];
  __syncthreads();
  for (uint64_t base = (uint64_t)wave * 64; base < a.n_rows; base += (uint64_t)nwaves * 64) {
    const uint32_t row = (uint32_t)base + lane;
    const bool valid = row < a.n_rows;
    uint32_t ham = 0, inter = 0, uni = 0;
    if (valid) {
      const uint4* p = reinterpret_cast<const uint4*>(a.bits + (size_t)row * W);
      for (uint32_t w = 0; w < W; w += 4) {
        const uint4 x = p[w / 4];
        const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const uint32_t qq = qw[w + e];
          if (METRIC == kHamming) {
            ham += __popc(xs[e] ^ qq);
          } else {
            inter += __popc(xs[e] & qq);
            uni += __popc(xs[e] | qq);
          }
        }
      }
    }
    float score;
    if (METRIC == kHamming) {
      score = (float)ham;
    } else {
      score = (uni == 0) ? 1.0f : (float)inter / (float)uni;  // simd_explicit.rs:431-442
    }
    uint64_t key = valid ? make_key<HIB>(score, row) : kKeyInvalid;
    const uint64_t tau = (*cnt == k) ? list[k - 1] : kKeyInvalid;
    uint64_t mask = __ballot(key < tau);
    if ((uint32_t)__popcll(mask) > k) {
      // Many lanes pass (always in a block's first steps, while its list is still filling): only the wave's own k best can
      // end up in the list, so rank the passing keys among themselves first — one readlane + compare per passing lane
      // instead of one locked list insertion (~20x the instructions) per passing lane.  Soft-deleted rows must not take
      // a rank: they are dropped here (a rare, divergent load).
      if (a.alive && (mask >> lane & 1ull) && a.alive[row] == 0) key = kKeyInvalid;
      mask = __ballot(key < tau);
      uint32_t rank = 0;
      uint64_t rest = mask;
      while (rest) {
        const int src = __ffsll((long long)rest) - 1;
        rest &= rest - 1;
        rank += readlane64(key, src) < key ? 1u : 0u;
      }
      mask = __ballot((mask >> lane & 1ull) && rank < k);
    }
    while (mask) {
      const int src = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const uint64_t kk = readlane64(key, src);
      if (a.alive && a.alive[key_row(kk)] == 0) continue;  // soft-deleted rows are filtered where it is rare
      shared_list_offer(list, cnt, lock, k, kk, lane);
    }
  }
  __syncthreads();
  if (wib != 0) return;
  const uint32_t c = *cnt;
  uint64_t* dst = a.part_keys + ((size_t)qi * gridDim.x + blockIdx.x) * k;
  for (uint32_t e = lane; e < k; e += 64) dst[e] = e < c ? list[e] : kKeyInvalid;
}

// ------------------------------------------------------------------------------------------
// Batched packed-bit sweep: B queries per corpus pass (grid.y = query tile).  The kernel above reads the whole corpus
// once per query — fine for one query (HBM / latency bound, 96 B per row), wasteful for a batch: 1 024 queries at
// 1 M x 768 cost 1 024 passes (28 ms).  Here a lane holds TWO rows' words in registers and walks the B queries of the
// tile out of LDS (one broadcast ds_read_b128 = 4 words of one query, used for both rows):
//   inter[r][b] += popc(x_r & q_b)   — one v_and + one v_bcnt (popcount-accumulate) per word,
// and both metrics finish from the same counts: |x ^ q| = |x| + |q| - 2 inter, |x | q| = |x| + |q| - inter
// (|x| once per row, |q| once per query).  Bound: the vector ALUs — 2 instructions per 32 bits and query.
// Top-k: per query a block-shared sorted list (shared_list_offer); the filter in front of it is a conservative
// integer / float threshold per query in LDS (Hamming: distance <= the k-th best's; Jaccard: inter >= tau * union with
// tau a hair below the k-th best's score), refreshed by whoever inserts; survivors (rare) go through the exact
// 64-bit key compare.  Integer work: results are bit-identical to the per-query kernel and to the oracle.
// ------------------------------------------------------------------------------------------
// the rare path of sweep_topk_bits_batch: lanes in `mask` passed the conservative filter of one query; exact keys,
// offers to the query's block-shared list, threshold refresh.  Not inlined: it runs for a few rows per thousand, and
// keeping it out of line keeps the (R x B)-fold unrolled filter loop small enough to stay fully unrolled.
template <int METRIC>
__device__ __noinline__ void bits_offer(lds_vu64* list, lds_vu32* cnt, uint32_t* lock, uint32_t* thr,
                                        uint32_t k, uint32_t pqb, uint32_t px, uint32_t inter, uint32_t row, uint64_t mask,
                                        const uint8_t* alive, int lane) {
  constexpr bool HIB = higher_is_better(METRIC);
  float score;
  if (METRIC == kHamming) {
    score = (float)(px + pqb - 2u * inter);
  } else {
    const uint32_t uni = px + pqb - inter;
    score = (uni == 0) ? 1.0f : (float)inter / (float)uni;  // simd_explicit.rs:431-442
  }
  const uint64_t key = make_key<HIB>(score, row);
  bool inserted = false;
  while (mask) {
    const int src = __ffsll((long long)mask) - 1;
    mask &= mask - 1;
    const uint64_t kk = readlane64(key, src);
    if (*cnt == k && kk >= list[k - 1]) continue;
    if (alive && alive[key_row(kk)] == 0) continue;  // soft-deleted rows are filtered where it is rare
    shared_list_offer(list, cnt, lock, k, kk, lane);
    inserted = true;
    if (*cnt == k) mask &= __ballot(key < list[k - 1]);  // drop every pending lane the (new) k-th best already beats
  }
  if (inserted && *cnt == k && lane == 0) {
    // publish a threshold from the current k-th best (a stale, looser value written by a racing wave is still
    // conservative: the exact compare above decides)
    const float sk = key_score<HIB>(list[k - 1]);
    if (METRIC == kHamming)
      *thr = (uint32_t)((int32_t)sk + 1 - (int32_t)pqb);
    else
      *thr = __float_as_uint(sk > 0.0f ? sk * 0.999999f : -1.0f);
  }
}

template <int METRIC, int B>
__global__ __launch_bounds__(256) void sweep_topk_bits_batch(BitsArgs a, uint32_t nq) {
  constexpr int R = 2;  // rows per lane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = (int)threadIdx.x;
  const int lane = lane_id();
  const uint32_t k = a.k, W = a.words, W4 = W / 4;
  const uint32_t q0 = blockIdx.y * B;
  const uint32_t nb = min((uint32_t)B, nq - q0);
  // LDS: qw [W4][B][4] u32 | pq[B] | thr[B] (Hamming: int32 "d_k + 1 - |q|", Jaccard: float tau) | cnt[B] | lock[B] | lists [B][k] u64
  uint32_t* qw = reinterpret_cast<uint32_t*>(smem);
  uint32_t* pq = qw + (size_t)W * B;
  uint32_t* thr = pq + B;
  lds_vu32* cnt = (lds_vu32*)(lds_void_p)(thr + B);
  uint32_t* lock = thr + 2 * B;
  lds_vu64* lists = (lds_vu64*)(lds_void_p)(lock + B);
  for (uint32_t i = tid; i < W * B; i += 256) {
    const uint32_t b = i / W, w = i % W;
    const uint32_t src = min(q0 + b, nq - 1);  // padded slots repeat the last query (their lists are never written out)
    qw[((size_t)(w >> 2) * B + b) * 4 + (w & 3)] = a.qbits[(size_t)src * W + w];
  }
  __syncthreads();
  if (tid < B) {
    uint32_t c = 0;
    for (uint32_t w = 0; w < W; w++) c += __popc(qw[((size_t)(w >> 2) * B + tid) * 4 + (w & 3)]);
    pq[tid] = c;
    // nothing selected yet: everything passes (Hamming: px - 2 inter < INT_MAX; Jaccard: inter >= -1 * union)
    thr[tid] = METRIC == kHamming ? 0x7FFFFFFFu : __float_as_uint(-1.0f);
    cnt[tid] = 0;
    lock[tid] = 0;
  }
  __syncthreads();
  const uint32_t* bits = a.bits;
  for (uint64_t base = (uint64_t)blockIdx.x * (256 * R); base < a.n_rows; base += (uint64_t)gridDim.x * (256 * R)) {
    uint32_t row[R];
    bool valid[R];
    const uint4* rp[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      row[r] = (uint32_t)base + r * 256 + tid;
      valid[r] = row[r] < a.n_rows;
      rp[r] = reinterpret_cast<const uint4*>(bits + (size_t)(valid[r] ? row[r] : a.n_rows - 1) * W);
    }
    uint32_t inter[R][B];
    uint32_t px[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      px[r] = 0;
#pragma unroll
      for (int b = 0; b < B; b++) inter[r][b] = 0;
    }
    for (uint32_t c = 0; c < W4; c++) {
      uint4 x[R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        x[r] = rp[r][c];
        px[r] += __popc(x[r].x) + __popc(x[r].y) + __popc(x[r].z) + __popc(x[r].w);
      }
      const uint4* qc = reinterpret_cast<const uint4*>(qw) + (size_t)c * B;
#pragma unroll
      for (int b = 0; b < B; b++) {
        const uint4 q = qc[b];
#pragma unroll
        for (int r = 0; r < R; r++)
          inter[r][b] += __popc(x[r].x & q.x) + __popc(x[r].y & q.y) + __popc(x[r].z & q.z) + __popc(x[r].w & q.w);
      }
    }
    // ---- filter + (rare) exact offers ----
#pragma unroll
    for (int b = 0; b < B; b++) {
      const uint32_t pqb = pq[b];
      const uint32_t tb = thr[b];
#pragma unroll
      for (int r = 0; r < R; r++) {
        bool pass;
        if (METRIC == kHamming) {
          // ham = px + pq - 2 inter <= d_k  <=>  px - 2 inter < thr (= d_k + 1 - pq, signed)
          pass = (int32_t)(px[r] - 2u * inter[r][b]) < (int32_t)tb;
        } else {
          const uint32_t uni = px[r] + pqb - inter[r][b];
          pass = !((float)inter[r][b] < __uint_as_float(tb) * (float)uni);  // union 0 (score 1.0): 0 >= 0 passes
        }
        const uint64_t mask = __ballot(pass && valid[r]);
        if (mask) bits_offer<METRIC>(lists + (size_t)b * k, &cnt[b], &lock[b], &thr[b], k, pqb, px[r], inter[r][b], row[r], mask, a.alive, lane);
      }
    }
  }
  __syncthreads();
  const int wib = tid >> 6;
  for (uint32_t b = wib; b < nb; b += 4) {
    const uint32_t c = cnt[b];
    uint64_t* dst = a.part_keys + ((size_t)(q0 + b) * gridDim.x + blockIdx.x) * k;
    for (uint32_t e = lane; e < k; e += 64) dst[e] = e < c ? lists[(size_t)b * k + e] : kKeyInvalid;
  }
}

// ------------------------------------------------------------------------------------------
// Batched packed-bit sweep, k <= 48: same popcount core as sweep_topk_bits_batch, lock-free selection.  Measured
// on the locked variant above (1 M x 768 bits, 32 queries): popcounts + filter 0.09 ms, the selection 1.6 ms — every
// block starts with empty lists, so its first rows all qualify and queue up one locked sorted insertion at a time.
// Here (the scheme of the GEMM sweep's epilogue, sweep_gemm.hip): per query a candidate buffer cand[64] + counter in
// LDS; a row that passes the cheap per-query filter APPENDS its exact 64-bit key (one LDS atomic add + one store, all
// lanes at once).  After the appends of a row step: barrier, wave w compacts the buffers of queries w, w+4, ... that
// are past the watermark (one key per lane, rank = number of smaller keys, the k best written back in order, k-th
// best and the filter threshold published), barrier.  Keys that found their buffer full stay pending in a per-lane
// bit mask (one bit per (query, row slot)) and the block repeats append / compact for them against the tightened
// k-th best — only the first row step of a block does.  The final lists hold the k smallest keys offered whatever
// the interleaving (keys are unique; a key is only ever dropped against a k-th best that never gets worse).
// ------------------------------------------------------------------------------------------
constexpr int kBitsCap = 64;   // candidate buffer entries per query (= one key per lane when compacted)
template <int METRIC, int B>
__global__ __launch_bounds__(256) void sweep_topk_bits_tile(BitsArgs a, uint32_t nq) {
  constexpr bool HIB = higher_is_better(METRIC);
  constexpr int R = 2, CAP = kBitsCap;
  static_assert(B * R <= 64, "one pending bit per (query, row slot)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = (int)threadIdx.x;
  const int lane = lane_id();
  const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t k = a.k, W = a.words, W4 = W / 4;
  const uint32_t q0 = blockIdx.y * B;
  const uint32_t nb = min((uint32_t)B, nq - q0);
  // LDS: qw [W4][B][4] u32 | pq[B] | thr[B] | cnts[B] | ovf (+pad) | tauk[B] u64 | cand[B][CAP] u64
  uint32_t* qw = reinterpret_cast<uint32_t*>(smem);
  uint32_t* pq = qw + (size_t)W * B;
  uint32_t* thr = pq + B;
  uint32_t* cnts = thr + B;
  lds_vu32* ovf = (lds_vu32*)(lds_void_p)(cnts + B);
  uint64_t* tauk = reinterpret_cast<uint64_t*>(cnts + B + 4);
  uint64_t* cand = tauk + B;
  for (uint32_t i = tid; i < W * B; i += 256) {
    const uint32_t b = i / W, w = i % W;
    const uint32_t src = min(q0 + b, nq - 1);  // padded slots repeat the last query (their lists are never written out)
    qw[((size_t)(w >> 2) * B + b) * 4 + (w & 3)] = a.qbits[(size_t)src * W + w];
  }
  __syncthreads();
  if (tid < B) {
    uint32_t c = 0;
    for (uint32_t w = 0; w < W; w++) c += __popc(qw[((size_t)(w >> 2) * B + tid) * 4 + (w & 3)]);
    pq[tid] = c;
    thr[tid] = METRIC == kHamming ? 0x7FFFFFFFu : __float_as_uint(-1.0f);  // nothing selected yet: everything passes
    cnts[tid] = 0;
    tauk[tid] = kKeyInvalid;
  }
  if (tid == 0) *ovf = 0u;
  __syncthreads();

  const uint32_t watermark = (k + CAP) / 2;
  auto compact = [&](bool force) __attribute__((always_inline)) {
    const uint32_t bq = (uint32_t)wib + 4u * (uint32_t)lane;  // lane l looks at query wib + 4 l
    const uint32_t cq = bq < (uint32_t)B ? cnts[bq] : 0u;
    uint64_t need = __ballot(cq > k && (force || cq >= watermark));
    while (need) {
      const int src = __ffsll((long long)need) - 1;
      need &= need - 1;
      const uint32_t b = (uint32_t)wib + 4u * (uint32_t)src;
      const uint32_t n = min(cnts[b], (uint32_t)CAP);
      uint64_t* cb = cand + (size_t)b * CAP;
      const bool mine = (uint32_t)lane < n;
      const uint64_t key = mine ? cb[lane] : kKeyInvalid;
      uint32_t rank = 0;
      for (uint32_t j = 0; j < (uint32_t)CAP; j += 4) {
        uint64_t kj[4];
#pragma unroll
        for (int u = 0; u < 4; u++) kj[u] = cb[j + u];
#pragma unroll
        for (int u = 0; u < 4; u++) rank += (j + u < n && kj[u] < key) ? 1u : 0u;
      }
      if (mine && rank < k) cb[rank] = key;
      if (mine && rank == k - 1) {
        tauk[b] = key;
        // the cheap filter in front of the exact compare: conservative (ties with the k-th best pass)
        const float sk = key_score<HIB>(key);
        if (METRIC == kHamming)
          thr[b] = (uint32_t)((int32_t)sk + 1 - (int32_t)pq[b]);
        else
          thr[b] = __float_as_uint(sk > 0.0f ? sk * 0.999999f : -1.0f);
      }
      if (lane == 0) cnts[b] = k;
    }
  };

  const uint32_t* bits = a.bits;
  uint32_t token = 0;
  for (uint64_t base = (uint64_t)blockIdx.x * (256 * R); base < a.n_rows; base += (uint64_t)gridDim.x * (256 * R)) {
    uint32_t row[R];
    bool valid[R];
    const uint4* rp[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      row[r] = (uint32_t)base + r * 256 + tid;
      valid[r] = row[r] < a.n_rows;
      rp[r] = reinterpret_cast<const uint4*>(bits + (size_t)(valid[r] ? row[r] : a.n_rows - 1) * W);
    }
    uint32_t inter[R][B];
    uint32_t px[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      px[r] = 0;
#pragma unroll
      for (int b = 0; b < B; b++) inter[r][b] = 0;
    }
    for (uint32_t c = 0; c < W4; c++) {
      uint4 x[R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        x[r] = rp[r][c];
        px[r] += __popc(x[r].x) + __popc(x[r].y) + __popc(x[r].z) + __popc(x[r].w);
      }
      const uint4* qc = reinterpret_cast<const uint4*>(qw) + (size_t)c * B;
#pragma unroll
      for (int b = 0; b < B; b++) {
        const uint4 q = qc[b];
#pragma unroll
        for (int r = 0; r < R; r++)
          inter[r][b] += __popc(x[r].x & q.x) + __popc(x[r].y & q.y) + __popc(x[r].z & q.z) + __popc(x[r].w & q.w);
      }
    }
    // ---- cheap filter: one pending bit per (query, row slot) ----
    uint32_t pend[2] = {0u, 0u};
#pragma unroll
    for (int b = 0; b < B; b++) {
      const uint32_t pqb = pq[b];
      const uint32_t tb = thr[b];
#pragma unroll
      for (int r = 0; r < R; r++) {
        bool pass;
        if (METRIC == kHamming) {
          pass = (int32_t)(px[r] - 2u * inter[r][b]) < (int32_t)tb;  // ham <= d_k  <=>  px - 2 inter < d_k + 1 - pq
        } else {
          const uint32_t uni = px[r] + pqb - inter[r][b];
          pass = !((float)inter[r][b] < __uint_as_float(tb) * (float)uni);  // union 0 (score 1.0): 0 >= 0 passes
        }
        const int bit = b * R + r;
        pend[bit >> 5] |= (pass && valid[r]) ? (1u << (bit & 31)) : 0u;
      }
    }
    // ---- append / compact rounds ----
    for (;;) {
      ++token;
      bool failed = false;
      if (__ballot((pend[0] | pend[1]) != 0u)) {
#pragma unroll
        for (int b = 0; b < B; b++) {
#pragma unroll
          for (int r = 0; r < R; r++) {
            const int bit = b * R + r;
            const uint32_t bm = 1u << (bit & 31);
            const bool want = (pend[bit >> 5] & bm) != 0u;
            if (__ballot(want) == 0) continue;
            const uint32_t pqb = pq[b];
            float score;
            if (METRIC == kHamming) {
              score = (float)(px[r] + pqb - 2u * inter[r][b]);
            } else {
              const uint32_t uni = px[r] + pqb - inter[r][b];
              score = (uni == 0) ? 1.0f : (float)inter[r][b] / (float)uni;  // simd_explicit.rs:431-442
            }
            const uint64_t key = make_key<HIB>(score, row[r]);
            bool ok = want && key < tauk[b];
            if (ok && a.alive) ok = a.alive[row[r]] != 0;  // soft-deleted rows are filtered where it is rare
            bool full = false;
            if (ok) {
              const uint32_t idx = atomicAdd(&cnts[b], 1u);
              if (idx < (uint32_t)CAP)
                cand[(size_t)b * CAP + idx] = key;
              else
                full = true;  // buffer full: stays pending for the round after the compaction
            }
            if (!full) pend[bit >> 5] &= ~bm;
            failed |= full;
          }
        }
      }
      if (failed) *ovf = token;
      __syncthreads();  // appends visible
      const bool again = *ovf == token;
      compact(again);
      __syncthreads();  // compacted lists, k-th bests and thresholds visible
      if (!again) break;
    }
  }
  compact(true);
  __syncthreads();
  for (uint32_t b = wib; b < nb; b += 4) {
    const uint32_t c = min(cnts[b], k);  // <= k entries; unsorted if never compacted (the merge kernel scans them all)
    uint64_t* dst = a.part_keys + ((size_t)(q0 + b) * gridDim.x + blockIdx.x) * k;
    for (uint32_t e = lane; e < k; e += 64) dst[e] = e < c ? cand[(size_t)b * CAP + e] : kKeyInvalid;
  }
}

// ------------------------------------------------------------------------------------------
// Euclidean batches on the matrix cores, second half (first half: sweep_topk_gemm_f32<kEuclidean>, which keeps the
// k' = k + slack rows with the smallest APPROXIMATE squared distance |v|^2 + |q|^2 - 2 q.v per query).  One block per
// query: (1) every candidate is re-scored with the canonical (q - v)^2 lane chain + butterfly + sqrt — the very
// arithmetic of the small-batch kernels, so a query's result does not depend on the batch it came in;
// (2) the candidates are ranked by (exact score, row); (3) the verdict: a row OUTSIDE the candidates has an
// approximate value >= A (the worst one kept), hence a canonical squared sum >= A - delta, where delta bounds
// |approx - canonical| for any row: both deviate from the true |q - v|^2 by at most ~2 n eps (|q|^2 + |v|^2), the
// bound used is 8 n eps (|q|^2 + max|v|^2).  If A - delta > the k-th best canonical sum, nothing outside can reach
// the top k and the answer is exact; otherwise (near-ties wider than the slack, NaN) the query is FLAGGED and the
// caller re-runs it through the exact vector-ALU sweep.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void max_norm_kernel(const float* norms, uint32_t n, uint32_t* out_bits) {
  float m = 0.0f;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float v = norms[i];
    m = (v > m || v != v) ? v : m;  // NaN propagates: the verdict then flags every query
  }
  // norms are >= 0 (or NaN, whose bit pattern is above every finite value): unsigned order = float order
  atomicMax(out_bits, __float_as_uint(m));
}

__global__ __launch_bounds__(256) void euclid_rerank_verify(EuclidRerankArgs a) {
  __shared__ uint64_t keys[64];
  __shared__ float sums[64];
  const int lane = lane_id();
  const int wib = (int)(threadIdx.x >> 6);
  const uint32_t qi = blockIdx.x;
  const uint32_t n = min(a.cand_n[qi], a.kp);
  const float* q = a.queries + (size_t)qi * a.q_stride;
  const int d4 = (int)((a.dim + 3) / 4);
  for (uint32_t c = wib; c < n; c += 4) {
    const uint32_t row = (uint32_t)a.cand_rows[(size_t)qi * a.kp + c];
    const float* p = a.rows + (size_t)row * a.row_stride;  // 16-B aligned, zero-padded to the stride
    float acc = 0.0f;
    for (int ch = lane; ch < d4; ch += 64) {
      const int nv = (int)a.dim - ch * 4;
      const float4 x = ld4(p + ch * 4);
      float4 qq;
      if (nv >= 4) {
        qq = make_float4(q[ch * 4], q[ch * 4 + 1], q[ch * 4 + 2], q[ch * 4 + 3]);
        acc = chain4<kOpL2>(acc, qq, x);
      } else {
        qq = make_float4(q[ch * 4], nv > 1 ? q[ch * 4 + 1] : 0.f, nv > 2 ? q[ch * 4 + 2] : 0.f, 0.f);
        acc = chain4_tail<kOpL2>(acc, qq, x, nv);
      }
    }
    const float sum = butterfly_all(acc);
    if (lane == 0) {
      sums[c] = sum;
      keys[c] = make_key<false>(finish_score<kEuclidean>(sum, 0.f, 0.f), row);
    }
  }
  __syncthreads();
  if (wib != 0) return;
  // rank by (exact score, row)
  const uint64_t key = (uint32_t)lane < n ? keys[lane] : kKeyInvalid;
  uint32_t rank = 0;
  for (uint32_t j = 0; j < n; j++) rank += keys[j] < key ? 1u : 0u;
  // lane e picks up the candidate of rank e (ranks are a permutation of 0..n-1: keys are unique)
  uint32_t mine = 0;
  for (uint32_t j = 0; j < n; j++) {
    const uint32_t rj = (uint32_t)__builtin_amdgcn_readlane((int)rank, (int)j);
    if (rj == (uint32_t)lane) mine = j;
  }
  // |q|^2 for the error bound (any accurate value will do)
  float qacc = 0.0f;
  for (uint32_t i = lane; i < a.dim; i += 64) qacc = __builtin_fmaf(q[i], q[i], qacc);
  const float qn2 = butterfly_all(qacc);
  const uint32_t kk = min(a.k, n);
  bool ok = true;
  if (n == a.kp && kk > 0) {  // the candidate buffer is full: rows were left out
    const float A = a.cand_approx[(size_t)qi * a.kp + a.kp - 1];
    const float nmax = __uint_as_float(*a.norm_max_bits);
    const float delta = 8.0f * (float)a.dim * 5.9604645e-8f * (qn2 + nmax * nmax) + fabsf(A) * 1e-6f;
    const float Ek = sums[__builtin_amdgcn_readlane((int)mine, (int)(kk - 1))];
    ok = (A - delta) > Ek;  // false for NaN anywhere
  }
  if (lane == 0) {
    a.flags[qi] = ok ? 0u : 1u;
    a.out_n[qi] = kk;
  }
  for (uint32_t e = lane; e < a.k; e += 64) {
    if (e < kk) {
      const uint64_t ke = keys[mine];
      const uint32_t row = key_row(ke);
      a.out_ids[(size_t)qi * a.k + e] = a.ext_ids ? a.ext_ids[row] : (uint64_t)row;
      a.out_scores[(size_t)qi * a.k + e] = key_score<false>(ke);
    } else {
      a.out_ids[(size_t)qi * a.k + e] = ~0ull;
      a.out_scores[(size_t)qi * a.k + e] = __uint_as_float(0x7FC00000u);
    }
  }
}

// ------------------------------------------------------------------------------------------
// Insert-time row preparation: canonical norms (cosine) and packed threshold bits.
// One wave per row.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_rows(PrepArgs a) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4;
  const int d4 = (int)((a.dim + 3) / 4);
  for (uint32_t r = wave; r < a.n_rows; r += nwaves) {
    const uint32_t row = a.row0 + r;
    const float* p = a.rows + (size_t)row * a.row_stride;
    if (a.norms) {
      float acc = 0.0f;
      for (int c = lane; c < d4; c += 64) {
        float4 x = ld4(p + c * 4);
        int nv = (int)a.dim - c * 4;
        acc = nv >= 4 ? chain4<kOpDot>(acc, x, x) : chain4_tail<kOpDot>(acc, x, x, nv);
      }
      float n = sqrtf(butterfly_all(acc));
      if (lane == 0) a.norms[row] = n;
    }
    if (a.bits) {
      uint32_t* dst = a.bits + (size_t)row * a.words;
      for (uint32_t e0 = 0; e0 < a.words * 32; e0 += 64) {
        const uint32_t e = e0 + lane;
        const bool bit = e < a.dim && p[e] > 0.5f;  // NaN > 0.5 is false, as on the CPU
        const uint64_t m = __ballot(bit);
        if (lane == 0) {
          dst[e0 / 32] = (uint32_t)m;
          if (e0 / 32 + 1 < a.words) dst[e0 / 32 + 1] = (uint32_t)(m >> 32);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// score-all kernel: DistanceEngine::batch_distance / GpuAccelerator::batch_* over an arbitrary
// row-major buffer (row stride = dim, any alignment >= 4 B handled by the scalar path).
// One wave per row; every metric computed from the raw f32 rows.
// ------------------------------------------------------------------------------------------
template <int METRIC>
__global__ __launch_bounds__(256) void score_rows(ScoreArgs a) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4;
  const int d4 = (int)((a.dim + 3) / 4);
  const bool vec_ok = a.aligned16 != 0;
  // query norm (cosine)
  float qn = 0.0f;
  if (METRIC == kCosine) {
    float acc = 0.0f;
    for (int c = lane; c < d4; c += 64)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        int i = c * 4 + e;
        if (i < (int)a.dim) acc = __builtin_fmaf(a.query[i], a.query[i], acc);
      }
    qn = sqrtf(butterfly_all(acc));
  }
  for (uint64_t r = wave; r < a.n_rows; r += nwaves) {
    const float* p = a.rows + (size_t)r * a.dim;
    float out;
    if (METRIC == kHamming || METRIC == kJaccard) {
      uint32_t ham = 0, inter = 0, uni = 0;
      for (uint32_t e0 = 0; e0 < a.dim; e0 += 64) {
        const uint32_t e = e0 + lane;
        const bool x = e < a.dim && p[e] > 0.5f;
        const bool y = e < a.dim && a.query[e] > 0.5f;
        ham += (uint32_t)__popcll(__ballot(x != y));
        inter += (uint32_t)__popcll(__ballot(x && y));
        uni += (uint32_t)__popcll(__ballot(x || y));
      }
      if (METRIC == kHamming) {
        out = (float)ham;
      } else {
        float sim = (uni == 0) ? 1.0f : (float)inter / (float)uni;
        out = (a.kind == 0) ? canon_nan(1.0f - sim) : sim;  // native/distance.rs:83
      }
    } else {
      constexpr int OP = (METRIC == kEuclidean) ? kOpL2 : kOpDot;
      float acc = 0.0f, nacc = 0.0f;
      for (int c = lane; c < d4; c += 64) {
        float4 x, qq;
        int nv = (int)a.dim - c * 4;
        if (vec_ok && nv >= 4) {
          x = ld4(p + c * 4);
          qq = ld4(a.query + c * 4);
        } else {
          float xa[4] = {0, 0, 0, 0}, qa[4] = {0, 0, 0, 0};
          for (int e = 0; e < 4 && e < nv; e++) {
            xa[e] = p[c * 4 + e];
            qa[e] = a.query[c * 4 + e];
          }
          x = make_float4(xa[0], xa[1], xa[2], xa[3]);
          qq = make_float4(qa[0], qa[1], qa[2], qa[3]);
        }
        if (nv >= 4) {
          acc = chain4<OP>(acc, qq, x);
          if (METRIC == kCosine) nacc = chain4<kOpDot>(nacc, x, x);
        } else {
          acc = chain4_tail<OP>(acc, qq, x, nv);
          if (METRIC == kCosine) nacc = chain4_tail<kOpDot>(nacc, x, x, nv);
        }
      }
      float sum = butterfly_all(acc);
      float vn = 1.0f;
      if (METRIC == kCosine) vn = sqrtf(butterfly_all(nacc));
      float s = finish_score<METRIC>(sum, qn, vn);
      if (METRIC == kEuclidean && a.kind == 2) s = canon_nan(sum);  // simd::squared_l2_distance (simd.rs:207-211): no sqrt
      if (a.kind == 0) {  // DistanceEngine::distance (native/distance.rs:78-80)
        if (METRIC == kCosine) s = canon_nan(1.0f - s);
        if (METRIC == kDot) s = -s;
      }
      out = s;
    }
    if (lane == 0) a.out[r] = out;
  }
}

// ---- host-callable launchers ------------------------------------------------------------
template <int METRIC, int B, int CPL>
static void launch_sweep_t(const SweepArgs& a, int blocks, size_t lds, hipStream_t st, int groups) {
  hipLaunchKernelGGL((sweep_topk_f32<METRIC, B, CPL>), dim3(blocks, groups), dim3(256), lds, st, a);
}
template <int METRIC, int B>
static void launch_sweep_cpl(const SweepArgs& a, int cpl, int blocks, size_t lds, hipStream_t st, int groups) {
  switch (cpl) {
    case 1: launch_sweep_t<METRIC, B, 1>(a, blocks, lds, st, groups); break;
    case 2: launch_sweep_t<METRIC, B, 2>(a, blocks, lds, st, groups); break;
    case 3: launch_sweep_t<METRIC, B, 3>(a, blocks, lds, st, groups); break;
    case 4: launch_sweep_t<METRIC, B, 4>(a, blocks, lds, st, groups); break;
    default: launch_sweep_t<METRIC, B, 0>(a, blocks, lds, st, groups); break;
  }
}
template <int METRIC>
static void launch_sweep_b(const SweepArgs& a, int B, int cpl, int blocks, size_t lds, hipStream_t st, int groups) {
  switch (B) {
    case 1: launch_sweep_cpl<METRIC, 1>(a, cpl, blocks, lds, st, groups); break;
    case 2: launch_sweep_cpl<METRIC, 2>(a, cpl, blocks, lds, st, groups); break;
    case 4: launch_sweep_cpl<METRIC, 4>(a, cpl, blocks, lds, st, groups); break;
    default: launch_sweep_cpl<METRIC, 8>(a, cpl, blocks, lds, st, groups); break;
  }
}

size_t sweep_lds_bytes(int B, uint32_t k, uint32_t dim, int cpl) {
  size_t s = (((size_t)B * k * 8 + (size_t)B * 8) + 15) & ~(size_t)15;
  if (cpl == 0) s += (size_t)B * ((dim + 3) / 4) * 16;
  return (s + 15) & ~(size_t)15;
}

int sweep_cpl_for_dim(uint32_t dim) {
  if (dim % 256 != 0) return 0;
  int c = (int)(dim / 256);
  return (c >= 1 && c <= 4) ? c : 0;
}

void launch_sweep_f32(int metric, int B, const SweepArgs& a, int blocks, hipStream_t st, int groups) {
  const int cpl = sweep_cpl_for_dim(a.dim);
  const size_t lds = sweep_lds_bytes(B, a.k, a.dim, cpl);
  switch (metric) {
    case kCosine: launch_sweep_b<kCosine>(a, B, cpl, blocks, lds, st, groups); break;
    case kEuclidean: launch_sweep_b<kEuclidean>(a, B, cpl, blocks, lds, st, groups); break;
    default: launch_sweep_b<kDot>(a, B, cpl, blocks, lds, st, groups); break;
  }
}

// ---- large-tile launcher -------------------------------------------------------------------
size_t sweep_qlds_lds_bytes(int B, uint32_t k, uint32_t dim, int waves) {
  (void)waves;
  return (((size_t)B * dim * 4 + (size_t)B * k * 8 + (size_t)B * 8) + 15) & ~(size_t)15;
}
template <int METRIC, int B, int CPL, int WAVES>
static hipError_t launch_qlds_t(const SweepArgs& a, int blocks, size_t lds, hipStream_t st) {
  static bool attr_done = false;  // per instantiation
  if (lds > 64 * 1024 && !attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_f32_qlds<METRIC, B, CPL, WAVES>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL((sweep_topk_f32_qlds<METRIC, B, CPL, WAVES>), dim3(blocks), dim3(WAVES * 64), lds, st, a);
  return hipGetLastError();
}
template <int METRIC, int B, int WAVES>
static hipError_t launch_qlds_cpl(const SweepArgs& a, int blocks, size_t lds, hipStream_t st) {
  switch (sweep_cpl_for_dim(a.dim)) {
    case 1: return launch_qlds_t<METRIC, B, 1, WAVES>(a, blocks, lds, st);
    case 2: return launch_qlds_t<METRIC, B, 2, WAVES>(a, blocks, lds, st);
    case 3: return launch_qlds_t<METRIC, B, 3, WAVES>(a, blocks, lds, st);
    case 4: return launch_qlds_t<METRIC, B, 4, WAVES>(a, blocks, lds, st);
    default: return hipErrorInvalidValue;
  }
}
template <int METRIC>
static hipError_t launch_qlds_m(int B, const SweepArgs& a, int blocks, size_t lds, hipStream_t st) {
  if (B == 32) return launch_qlds_cpl<METRIC, 32, kQldsWaves32>(a, blocks, lds, st);
  return launch_qlds_cpl<METRIC, 16, kQldsWaves16>(a, blocks, lds, st);
}
hipError_t launch_sweep_f32_qlds(int metric, int B, const SweepArgs& a, int blocks, hipStream_t st) {
  const int waves = B == 32 ? kQldsWaves32 : kQldsWaves16;
  const size_t lds = sweep_qlds_lds_bytes(B, a.k, a.dim, waves);
  switch (metric) {
    case kCosine: return launch_qlds_m<kCosine>(B, a, blocks, lds, st);
    case kEuclidean: return launch_qlds_m<kEuclidean>(B, a, blocks, lds, st);
    default: return launch_qlds_m<kDot>(B, a, blocks, lds, st);
  }
}

// ---- MFMA launcher ---------------------------------------------------------------------------
size_t sweep_mfma_lds_bytes(int nqt, uint32_t k, uint32_t dim) {
  const size_t KU = (dim + 127) / 128, B = (size_t)nqt * 16;
  return ((KU * nqt * 8192 + B * k * 8 + B * 12) + 15) & ~(size_t)15;
}
template <int METRIC, int NQT, int WAVES>
static hipError_t launch_mfma_t(const SweepArgs& a, int blocks, size_t lds, hipStream_t st, int groups) {
  const uint32_t KT = (a.dim + 127) / 128;
  const bool full32 = (a.dim % 128) == 0;
  if (full32) {
    static bool done = false;
    if (lds > 64 * 1024 && !done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_mfma_f32<METRIC, NQT, WAVES, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      done = true;
    }
    hipLaunchKernelGGL((sweep_topk_mfma_f32<METRIC, NQT, WAVES, true>), dim3(blocks, groups), dim3(WAVES * 64), lds, st, a, KT);
  } else {
    static bool done = false;
    if (lds > 64 * 1024 && !done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_mfma_f32<METRIC, NQT, WAVES, false>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      done = true;
    }
    hipLaunchKernelGGL((sweep_topk_mfma_f32<METRIC, NQT, WAVES, false>), dim3(blocks, groups), dim3(WAVES * 64), lds, st, a, KT);
  }
  return hipGetLastError();
}
hipError_t launch_sweep_mfma(int metric, int nqt, const SweepArgs& a, int blocks, hipStream_t st, int groups) {
  const size_t lds = sweep_mfma_lds_bytes(nqt, a.k, a.dim);
  if (metric == kCosine) {
    if (nqt == 3) return launch_mfma_t<kCosine, 3, kMfmaWaves2>(a, blocks, lds, st, groups);
    if (nqt == 2) return launch_mfma_t<kCosine, 2, kMfmaWaves2>(a, blocks, lds, st, groups);
    return launch_mfma_t<kCosine, 1, kMfmaWaves1>(a, blocks, lds, st, groups);
  }
  if (nqt == 3) return launch_mfma_t<kDot, 3, kMfmaWaves2>(a, blocks, lds, st, groups);
  if (nqt == 2) return launch_mfma_t<kDot, 2, kMfmaWaves2>(a, blocks, lds, st, groups);
  return launch_mfma_t<kDot, 1, kMfmaWaves1>(a, blocks, lds, st, groups);
}

// ---- bf16 launchers --------------------------------------------------------------------------
size_t sweep_bf16_lds_bytes(int nqt, uint32_t k, uint32_t dim) {
  const size_t KU = (dim + 255) / 256, B = (size_t)nqt * 16;
  return ((KU * nqt * 8192 + B * k * 8 + B * 12) + 15) & ~(size_t)15;
}
void launch_prep_bf16(const float* rows, uint64_t row_stride, uint16_t* out, uint64_t out_stride, float* norms,
                      uint32_t row0, uint32_t n_rows, uint32_t dim, hipStream_t st, uint32_t* rho_max_bits) {
  if (n_rows == 0) return;
  const int blocks = (int)std::min<uint64_t>(((uint64_t)n_rows + 3) / 4, 4096);
  hipLaunchKernelGGL(prep_bf16_rows, dim3(blocks), dim3(256), 0, st, rows, row_stride, out, out_stride, norms, row0,
                     n_rows, dim, rho_max_bits);
}
template <int METRIC, int NQT, int WAVES>
static hipError_t launch_bf16_t(const Bf16SweepArgs& a, int blocks, size_t lds, hipStream_t st) {
  static bool done = false;
  if (lds > 64 * 1024 && !done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_mfma_bf16<METRIC, NQT, WAVES>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL((sweep_topk_mfma_bf16<METRIC, NQT, WAVES>), dim3(blocks), dim3(WAVES * 64), lds, st, a);
  return hipGetLastError();
}
hipError_t launch_sweep_bf16(int metric, int nqt, const uint16_t* rows, uint64_t row_stride, const float* norms,
                             const uint8_t* alive, const float* queries, uint64_t q_stride, uint64_t* part_keys,
                             uint32_t n_rows, uint32_t dim, uint32_t nq, uint32_t k, int blocks, hipStream_t st) {
  Bf16SweepArgs a{rows, norms, alive, queries, part_keys, row_stride, q_stride, n_rows, dim, nq, k, (dim + 255) / 256};
  const size_t lds = sweep_bf16_lds_bytes(nqt, k, dim);
  if (metric == kCosine) {
    switch (nqt) {
      case 6: return launch_bf16_t<kCosine, 6, kBf16WavesBig>(a, blocks, lds, st);
      case 4: return launch_bf16_t<kCosine, 4, kBf16WavesBig>(a, blocks, lds, st);
      case 2: return launch_bf16_t<kCosine, 2, kBf16WavesSmall>(a, blocks, lds, st);
      default: return launch_bf16_t<kCosine, 1, kBf16WavesSmall>(a, blocks, lds, st);
    }
  }
  switch (nqt) {
    case 6: return launch_bf16_t<kDot, 6, kBf16WavesBig>(a, blocks, lds, st);
    case 4: return launch_bf16_t<kDot, 4, kBf16WavesBig>(a, blocks, lds, st);
    case 2: return launch_bf16_t<kDot, 2, kBf16WavesSmall>(a, blocks, lds, st);
    default: return launch_bf16_t<kDot, 1, kBf16WavesSmall>(a, blocks, lds, st);
  }
}

void launch_euclid_rerank(const EuclidRerankArgs& a, const float* norms, uint32_t n_rows, uint32_t nq, hipStream_t st) {
  (void)hipMemsetAsync(const_cast<uint32_t*>(a.norm_max_bits), 0, 4, st);
  hipLaunchKernelGGL(max_norm_kernel, dim3(std::min<uint32_t>((n_rows + 255) / 256, 1024)), dim3(256), 0, st, norms, n_rows,
                     const_cast<uint32_t*>(a.norm_max_bits));
  hipLaunchKernelGGL(euclid_rerank_verify, dim3(nq), dim3(256), 0, st, a);
}

void launch_max_norm(const float* norms, uint32_t n_rows, uint32_t* out_bits, hipStream_t st) {
  (void)hipMemsetAsync(out_bits, 0, 4, st);
  hipLaunchKernelGGL(max_norm_kernel, dim3(std::min<uint32_t>((n_rows + 255) / 256, 1024)), dim3(256), 0, st, norms, n_rows, out_bits);
}

void launch_merge(bool hib, const MergeArgs& m, uint32_t nq, hipStream_t st) {
  const uint64_t total = (uint64_t)m.n_lists * m.k;
  static const bool extract_on = [] {  // (probe builds: VELESDB_MERGE_EXTRACT=0 keeps these merges on merge_topk_select)
    const char* e = probe_env("VELESDB_MERGE_EXTRACT");
    return !(e && e[0] == '0');
  }();
  if (extract_on && total <= kMergeExtractMaxKeys && (m.k_out ? m.k_out : m.k) <= kBitsFusedMaxK) {  // small merges: registers + extraction
    const size_t lds_x = (size_t)4 * kBitsFusedMaxK * 8;
    if (hib)
      hipLaunchKernelGGL((merge_topk_extract<true>), dim3(nq), dim3(256), lds_x, st, m);
    else
      hipLaunchKernelGGL((merge_topk_extract<false>), dim3(nq), dim3(256), lds_x, st, m);
    return;
  }
  if (total <= kMergeSelectMaxKeys && (m.k_out ? m.k_out : m.k) <= kMergeSelectMaxK) {  // selection: the whole query in LDS
    const size_t lds_r = (size_t)total * 8 + (size_t)kMergeSelectMaxK * 8 + 66 * 4 + 8;
    if (hib)
      hipLaunchKernelGGL((merge_topk_select<true>), dim3(nq), dim3(256), lds_r, st, m);
    else
      hipLaunchKernelGGL((merge_topk_select<false>), dim3(nq), dim3(256), lds_r, st, m);
    return;
  }
  // few queries over many lists (one-query sweeps): heads first
  const uint32_t kk = m.k_out ? m.k_out : m.k;
  const size_t lds_h = (size_t)m.n_lists * 8 + (size_t)kMergeSelectMaxK * 8 + (size_t)(4 * kBitsFusedMaxK + 1) * 8 + 68 * 4 + (size_t)kMergeSelectMaxK * 4 + 8;
  if (nq <= kMergeHeadsMaxQueries && (uint64_t)kk * m.k <= kMergeSelectMaxK && m.n_lists > 4 * kk && lds_h <= 64 * 1024) {
    if (hib)
      hipLaunchKernelGGL((merge_topk_heads<true>), dim3(nq), dim3(256), lds_h, st, m);
    else
      hipLaunchKernelGGL((merge_topk_heads<false>), dim3(nq), dim3(256), lds_h, st, m);
    return;
  }
  const size_t lds = ((size_t)4 * (m.k_out ? m.k_out : m.k) * 8 + 16 + 15) & ~(size_t)15;
  if (hib)
    hipLaunchKernelGGL((merge_topk<true>), dim3(nq), dim3(256), lds, st, m);
  else
    hipLaunchKernelGGL((merge_topk<false>), dim3(nq), dim3(256), lds, st, m);
}

size_t sweep_bits_batch_lds_bytes(int B, uint32_t words, uint32_t k) {
  return (((size_t)words * B * 4 + (size_t)B * 16 + (size_t)B * k * 8) + 15) & ~(size_t)15;
}
template <int METRIC, int B>
static hipError_t launch_bits_batch_t(const BitsArgs& a, int blocks, uint32_t nq, size_t lds, hipStream_t st) {
  static bool done = false;
  if (lds > 64 * 1024 && !done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_bits_batch<METRIC, B>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL((sweep_topk_bits_batch<METRIC, B>), dim3(blocks, (nq + B - 1) / B), dim3(256), lds, st, a, nq);
  return hipGetLastError();
}
size_t sweep_bits_tile_lds_bytes(int B, uint32_t words) {
  return (((size_t)words * B * 4 + (size_t)B * 12 + 16 + (size_t)B * 8 + (size_t)B * kBitsCap * 8) + 15) & ~(size_t)15;
}
template <int METRIC, int B>
static hipError_t launch_bits_tile_t(const BitsArgs& a, int blocks, uint32_t nq, size_t lds, hipStream_t st) {
  static bool done = false;
  if (lds > 64 * 1024 && !done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_bits_tile<METRIC, B>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL((sweep_topk_bits_tile<METRIC, B>), dim3(blocks, (nq + B - 1) / B), dim3(256), lds, st, a, nq);
  return hipGetLastError();
}
hipError_t launch_sweep_bits_tile(int metric, int B, const BitsArgs& a, int blocks, uint32_t nq, hipStream_t st) {
  const size_t lds = sweep_bits_tile_lds_bytes(B, a.words);
  if (metric == kHamming)
    return B == 32 ? launch_bits_tile_t<kHamming, 32>(a, blocks, nq, lds, st) : launch_bits_tile_t<kHamming, 8>(a, blocks, nq, lds, st);
  return B == 32 ? launch_bits_tile_t<kJaccard, 32>(a, blocks, nq, lds, st) : launch_bits_tile_t<kJaccard, 8>(a, blocks, nq, lds, st);
}

hipError_t launch_sweep_bits_batch(int metric, int B, const BitsArgs& a, int blocks, uint32_t nq, hipStream_t st) {
  const size_t lds = sweep_bits_batch_lds_bytes(B, a.words, a.k);
  if (metric == kHamming)
    return B == 32 ? launch_bits_batch_t<kHamming, 32>(a, blocks, nq, lds, st) : launch_bits_batch_t<kHamming, 8>(a, blocks, nq, lds, st);
  return B == 32 ? launch_bits_batch_t<kJaccard, 32>(a, blocks, nq, lds, st) : launch_bits_batch_t<kJaccard, 8>(a, blocks, nq, lds, st);
}

// Which packed-bit kernel serves a batch, and with how many row blocks (= partial lists per query):
//   1-2 queries: per-query kernel; >= 3 queries: B queries per corpus pass, B = 8 or 32, whichever pads the batch
//   less; k <= 48: lock-free selection (sweep_topk_bits_tile); larger k: block-shared locked lists
//   (sweep_topk_bits_batch), worth it from ~96 queries.
BitsPlan plan_bits_sweep(uint64_t n_rows, int n_cus, uint32_t words, uint32_t nq, uint32_t k) {
  BitsPlan p{};
  const uint64_t nchunks = (n_rows + 63) / 64;
  p.blocks = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)(nchunks + 3) / 4, (int64_t)n_cus * 4));
  if (nq >= 3) {
    const uint32_t pad8 = (nq + 7) / 8 * 8, pad32 = (nq + 31) / 32 * 32;
    p.B = (nq >= 64 || pad32 <= pad8) ? 32 : 8;
    if (k <= kBitsTileMaxK && sweep_bits_tile_lds_bytes(p.B, words) <= 64 * 1024) {
      p.tile = true;
    } else {
      if (nq < 96) p.B = 0;
      if (p.B && sweep_bits_batch_lds_bytes(p.B, words, k) > 64 * 1024) p.B = 8;
      if (p.B && sweep_bits_batch_lds_bytes(p.B, words, k) > 64 * 1024) p.B = 0;
    }
  }
  if (p.B) {
    const int64_t nqt = (nq + p.B - 1) / p.B;
    p.blocks = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)n_rows + 511) / 512, (int64_t)n_cus * 4 / nqt));
  }
  return p;
}
hipError_t launch_bits_plan(int metric, const BitsPlan& p, const BitsArgs& a, uint32_t nq, hipStream_t st) {
  if (p.B == 0) {
    launch_sweep_bits(metric, a, p.blocks, nq, st);
    return hipGetLastError();
  }
  return p.tile ? launch_sweep_bits_tile(metric, p.B, a, p.blocks, nq, st) : launch_sweep_bits_batch(metric, p.B, a, p.blocks, nq, st);
}

// ------------------------------------------------------------------------------------------
// The same per-query sweep with COALESCED row loads (round 5).  sweep_topk_bits gives every lane its own row: a 96-byte row (768
// bits) is six 16-byte loads per lane at a 96-byte stride between lanes — every load instruction of a wave touches 48 cache lines
// for 1 KiB of payload, and the kernel ran at 0.30 of HBM (40 us for the 96 MB of 1 M x 768 bits).  Here a wave's 64 rows are one
// contiguous chunk of 64 P sixteen-byte pieces (P = words / 4): load j takes pieces 64 j + lane — 1 KiB per instruction, 8 whole
// lines —, every lane reduces its piece against the query piece it belongs to (piece index mod P, stepped without a division), the
// per-piece counts go through the wave's own LDS strip, and lane r sums the P counts of row r.  Integer work: the same counts, the
// same scores, the same keys as the lane-per-row kernel (tests/test_gpu_sweep.py compares both with the oracle).
// Per-piece counts: Hamming |x ^ q| <= 128; Jaccard |x & q| and |x | q| <= 128 each, packed as inter | uni << 16.
// ------------------------------------------------------------------------------------------
template <int METRIC, int P>
__global__ __launch_bounds__(256) void sweep_topk_bits_co(BitsArgs a) {
  constexpr bool HIB = higher_is_better(METRIC);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = (int)(threadIdx.x >> 6);
  const uint32_t wave = blockIdx.x * 4 + wib;
  const uint32_t nwaves = gridDim.x * 4;
  const uint32_t qi = blockIdx.y;
  const uint32_t k = a.k;
  constexpr uint32_t W = 4u * P;
  // LDS: list[k] u64 (block-shared, locked) | cnt, lock | query words [W] | per-wave strips [4][64 P] u32
  lds_vu64* list = (lds_vu64*)(lds_void_p)(smem);
  lds_vu32* cnt = (lds_vu32*)(lds_void_p)(smem + (size_t)k * 8);
  uint32_t* lock = reinterpret_cast<uint32_t*>(smem + (size_t)k * 8 + 4);
  uint32_t* qw = reinterpret_cast<uint32_t*>(smem + (((size_t)k * 8 + 8 + 15) & ~(size_t)15));
  uint32_t* strip = qw + W + (size_t)wib * 64 * P;
  if (threadIdx.x == 0) {
    *cnt = 0;
    *lock = 0;
  }
  for (uint32_t i = threadIdx.x; i < W; i += 256) qw[i] = a.qbits[(size_t)qi * W + i];
  __syncthreads();
  // the query piece of (load j, lane): (64 j + lane) mod P — the lane's piece of load 0, then + (64 mod P) per load
  uint4 qp[P];
  {
    uint32_t pc = (uint32_t)lane % P;
#pragma unroll
    for (int j = 0; j < P; j++) {
      qp[j] = *reinterpret_cast<const uint4*>(qw + pc * 4);
      pc += 64u % P;
      if (pc >= (uint32_t)P) pc -= P;
    }
  }
  for (uint64_t base = (uint64_t)wave * 64; base < a.n_rows; base += (uint64_t)nwaves * 64) {
    const uint32_t row = (uint32_t)base + lane;
    const bool valid = row < a.n_rows;
    const uint4* chunk = reinterpret_cast<const uint4*>(a.bits + base * W);  // 64 rows = 64 P pieces, contiguous
    const uint64_t pieces_left = (a.n_rows - base) * P;                       // pieces that exist behind `chunk`
    uint4 x[P];
#pragma unroll
    for (int j = 0; j < P; j++) {
      const uint32_t pj = (uint32_t)j * 64u + (uint32_t)lane;
      x[j] = pj < pieces_left ? chunk[pj] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int j = 0; j < P; j++) {
      uint32_t c;
      if (METRIC == kHamming) {
        c = __popc(x[j].x ^ qp[j].x) + __popc(x[j].y ^ qp[j].y) + __popc(x[j].z ^ qp[j].z) + __popc(x[j].w ^ qp[j].w);
      } else {
        const uint32_t in = __popc(x[j].x & qp[j].x) + __popc(x[j].y & qp[j].y) + __popc(x[j].z & qp[j].z) + __popc(x[j].w & qp[j].w);
        const uint32_t un = __popc(x[j].x | qp[j].x) + __popc(x[j].y | qp[j].y) + __popc(x[j].z | qp[j].z) + __popc(x[j].w | qp[j].w);
        c = in | (un << 16);
      }
      strip[j * 64 + lane] = c;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < P; i++) sum += strip[lane * P + i];  // the P pieces of row `lane` (packed halves cannot carry: P * 128 < 2^16)
    __builtin_amdgcn_wave_barrier();  // strip reads done before the next chunk overwrites it
    float score;
    if (METRIC == kHamming) {
      score = (float)sum;
    } else {
      const uint32_t inter = sum & 0xFFFFu, uni = sum >> 16;
      score = (uni == 0) ? 1.0f : (float)inter / (float)uni;  // simd_explicit.rs:431-442
    }
    uint64_t key = valid ? make_key<HIB>(score, row) : kKeyInvalid;
    const uint64_t tau = (*cnt == k) ? list[k - 1] : kKeyInvalid;
    uint64_t mask = __ballot(key < tau);
    if ((uint32_t)__popcll(mask) > k) {  // (as sweep_topk_bits: rank the passing keys among themselves first)
      if (a.alive && (mask >> lane & 1ull) && a.alive[row] == 0) key = kKeyInvalid;
      mask = __ballot(key < tau);
      uint32_t rank = 0;
      uint64_t rest = mask;
      while (rest) {
        const int src = __ffsll((long long)rest) - 1;
        rest &= rest - 1;
        rank += readlane64(key, src) < key ? 1u : 0u;
      }
      mask = __ballot((mask >> lane & 1ull) && rank < k);
    }
    while (mask) {
      const int src = __ffsll((long long)mask) - 1;
      mask &= mask - 1;
      const uint64_t kk = readlane64(key, src);
      if (a.alive && a.alive[key_row(kk)] == 0) continue;
      shared_list_offer(list, cnt, lock, k, kk, lane);
    }
  }
  __syncthreads();
  if (wib != 0) return;
  const uint32_t c = *cnt;
  uint64_t* dst = a.part_keys + ((size_t)qi * gridDim.x + blockIdx.x) * k;
  for (uint32_t e = lane; e < k; e += 64) dst[e] = e < c ? list[e] : kKeyInvalid;
}
template <int P>
static void launch_sweep_bits_co(int metric, const BitsArgs& a, int blocks, uint32_t nq, size_t lds, hipStream_t st) {
  if (metric == kHamming)
    hipLaunchKernelGGL((sweep_topk_bits_co<kHamming, P>), dim3(blocks, nq), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((sweep_topk_bits_co<kJaccard, P>), dim3(blocks, nq), dim3(256), lds, st, a);
}

// ------------------------------------------------------------------------------------------
// ONE launch for a call of one or two packed-bit queries (round 5).  Such a call was three launches — prep_rows packing the query,
// the sweep, the merge of its partial lists — 68 us for the 96 MB of 1 M x 768 bits, of which the sweep was 40 (profiles/r05n_*).
// Here a block packs the query itself (the rule of prep_rows: x > 0.5), sweeps with the coalesced layout of sweep_topk_bits_co (a
// wave's 64 rows = one contiguous chunk) through range-checked raw buffer loads, three chunks ahead, one block per CU, and
// keeps NO list while it sweeps: a lane's keys stay in registers (kBitsFusedR chunks per batch), and the wave's k best are
// extracted afterwards — k times { the lane's smallest key, the wave's smallest of those (two 32-bit minimum butterflies on DPP /
// permlane swaps: score word, then row word among the lanes that tie), drop it }; lane e ends with the e-th best, which is also
// how a further batch (more than kBitsFusedR chunks per wave) carries the list on.  The block's four lists are merged the same
// way by wave 0 (4 k <= 64 keys, one per lane), the block writes ONE list and takes a ticket; the block that draws the last
// ticket merges all lists heads-first (fused_tail_merge) and writes the result.  Same keys as every other path: the same
// integer counts, make_key, the canonical (score, row) order.  k <= kBitsFusedMaxK.  Measured: 91.5 -> 37.6 us per one-query call
// at 1 M x 768 (an empty kernel of this shape 16.3, loads + 9, extraction + 4.5, last-block merge + 8-9;
// profiles/r05t_v_one_launch_packed_bit_query.txt).
// ------------------------------------------------------------------------------------------
constexpr int kBitsFusedR = 16;
// The merge of the one-launch search, run by the block that drew the last ticket: n_lists <= 256 sorted lists of k keys (one per
// thread).  Heads first, as merge_topk_heads — the k-th smallest first key bounds the answer, only the <= k lists under it matter —
// but with the extraction above instead of a bit-by-bit selection over LDS (35 barriers): ~5 us instead of ~12 at 256 lists.
template <bool HIB>
__device__ __forceinline__ void fused_tail_merge(const MergeArgs& m, uint32_t qi, unsigned char* smem) {
  const uint32_t tid = threadIdx.x, lane = (uint32_t)lane_id();
  const uint32_t k = m.k;
  uint64_t* wl = reinterpret_cast<uint64_t*>(smem);                 // [4][kBitsFusedMaxK]
  uint64_t* sel = wl + 4 * kBitsFusedMaxK;                          // [256] the keys under the bound
  uint64_t* bnd = sel + 256;                                        // [1]
  uint32_t* cnt_s = reinterpret_cast<uint32_t*>(bnd + 1);           // [1]
  const uint64_t* keys = m.part_keys + (size_t)qi * m.n_lists * k;
  const uint64_t head = tid < m.n_lists ? keys[(size_t)tid * k] : kKeyInvalid;  // (a list is sorted: its first key is its smallest)
  if (tid == 0) *cnt_s = 0u;
  const uint64_t hk = block_k_smallest(head, k, wl);
  if (tid == k - 1) *bnd = hk;  // the k-th smallest head (kKeyInvalid with fewer than k non-empty lists: every key may matter)
  __syncthreads();
  const uint64_t bound = *bnd;
  if (head != kKeyInvalid && head <= bound) {  // at most k lists (keys are distinct); <= k x k <= 256 keys
    uint64_t lk[kBitsFusedMaxK];  // the whole list first: independent loads, one round trip (a loop that stops at the bound makes k of them)
#pragma unroll
    for (uint32_t e = 0; e < kBitsFusedMaxK; e++) lk[e] = e < k ? keys[(size_t)tid * k + e] : kKeyInvalid;
#pragma unroll
    for (uint32_t e = 0; e < kBitsFusedMaxK; e++) {
      if (lk[e] != kKeyInvalid && lk[e] <= bound) {
        const uint32_t slot = atomicAdd(cnt_s, 1u);
        if (slot < 256u) sel[slot] = lk[e];
      }
    }
  }
  __syncthreads();
  const uint32_t ns = min(*cnt_s, 256u);
  const uint64_t fin = block_k_smallest(tid < ns ? sel[tid] : kKeyInvalid, k, wl);
  if (tid < 64u) {
    const uint32_t cnt = min(ns, k);
    if (lane < k) {
      if (lane < cnt) {
        const uint32_t row = key_row(fin);
        m.out_ids[(size_t)qi * k + lane] = m.ext_ids ? m.ext_ids[row] : (uint64_t)row + m.row_base;
        m.out_scores[(size_t)qi * k + lane] = key_score<HIB>(fin);  // raw compute_distance value (search.rs:209)
      } else {
        m.out_ids[(size_t)qi * k + lane] = ~0ull;
        m.out_scores[(size_t)qi * k + lane] = __uint_as_float(0x7FC00000u);
      }
    }
    if (lane == 0) m.out_n[qi] = cnt;
  }
}
typedef unsigned int bits_u32x4 __attribute__((ext_vector_type(4)));
template <int METRIC, int P>
__global__ __launch_bounds__(256) void sweep_bits_fused(BitsFusedArgs a) {
  constexpr bool HIB = higher_is_better(METRIC);
  constexpr int R = kBitsFusedR;
  constexpr uint32_t W = 4u * P;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform, and the compiler must know: the chunk descriptors live in scalar registers)
  const uint32_t wave = blockIdx.x * 4 + wib;
  const uint32_t nwaves = gridDim.x * 4;
  const uint32_t qi = blockIdx.y;
  const uint32_t k = a.k;
  // LDS while sweeping: query words [W] | per-wave strips [4][64 P] u32 | the four wave lists [4][k] u64 | last-block flag
  uint32_t* qw = reinterpret_cast<uint32_t*>(smem);
  uint32_t* strip = qw + W + (size_t)wib * 64 * P;
  uint64_t* wl = reinterpret_cast<uint64_t*>(smem + (((size_t)W * 4 + (size_t)4 * 64 * P * 4 + 15) & ~(size_t)15));
  uint32_t* lastf = reinterpret_cast<uint32_t*>(wl + 4 * kBitsFusedMaxK);
  {  // the query's bits (prep_rows: x > 0.5; Binary storage mode, sign_bits_rows: x >= 0)
    const float* qp = a.q + (size_t)qi * a.q_stride;
    for (uint32_t e0 = (uint32_t)wib * 64u; e0 < W * 32u; e0 += 256u) {
      const uint32_t e = e0 + (uint32_t)lane;
      const float xq = e < a.dim ? qp[e] : -1.0f;
      const uint64_t mb = __ballot(e < a.dim && (a.sign_rule ? xq >= 0.0f : xq > 0.5f));  // (NaN: false under both rules, as on the CPU)
      if (lane == 0) {
        qw[e0 / 32] = (uint32_t)mb;
        if (e0 / 32 + 1 < W) qw[e0 / 32 + 1] = (uint32_t)(mb >> 32);
      }
    }
  }
  __syncthreads();
  uint4 qpc[P];  // the query piece of (load j, lane): (64 j + lane) mod P
  {
    uint32_t pc = (uint32_t)lane % P;
#pragma unroll
    for (int j = 0; j < P; j++) {
      qpc[j] = *reinterpret_cast<const uint4*>(qw + pc * 4);
      pc += 64u % P;
      if (pc >= (uint32_t)P) pc -= P;
    }
  }
  uint64_t best = kKeyInvalid;  // lane e < k: the wave's e-th best so far
  for (uint64_t b0 = (uint64_t)wave * 64; b0 < a.n_rows; b0 += (uint64_t)R * nwaves * 64) {
    uint64_t keys[R];
    // chunks r + 1 .. r + 3 are requested before chunk r is counted (a wave's chunks are a dependent chain otherwise, and one block per
    // CU is one wave per SIMD: nothing else hides the latency; the registers are there)
    constexpr int AHEAD = 3;
    uint4 x[AHEAD + 1][P];
    // raw buffer loads: a chunk's descriptor says how many bytes exist behind its first row, lanes past them get zeros from the
    // hardware's range check — no branch around a load, so the compiler counts the loads in flight (vmcnt(n)) instead of draining
    // them all at every chunk, which is what the predicated global loads of the first version came to: 16 dependent round trips
#define VDB_BITS_FUSED_LOAD(DST, RR) do { \
      const uint64_t base_ = b0 + (uint64_t)(RR) * nwaves * 64; \
      const uint64_t left_ = (base_ < a.n_rows && !(a.probe_skip & 4u)) ? (a.n_rows - base_) * (uint64_t)W * 4u : 0ull;  /* bytes behind the chunk's first row */ \
      const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc( \
          const_cast<uint32_t*>(a.bits) + (base_ < a.n_rows ? base_ : 0ull) * W, 0, (int)(uint32_t)(left_ < 0x7FFFFFFFull ? left_ : 0x7FFFFFFFull), 0x00020000); \
_Pragma("unroll") \
      for (int j = 0; j < P; j++) { \
        const bits_u32x4 v_ = __builtin_amdgcn_raw_buffer_load_b128(rs_, (int)(((uint32_t)j * 64u + (uint32_t)lane) * 16u), 0, 0); \
        DST[j] = make_uint4(v_[0], v_[1], v_[2], v_[3]); \
      } \
    } while (0)
#pragma unroll
    for (int r = 0; r < AHEAD; r++) VDB_BITS_FUSED_LOAD(x[r], r);
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint64_t base = b0 + (uint64_t)r * nwaves * 64;
      keys[r] = kKeyInvalid;
      if (r + AHEAD < R) VDB_BITS_FUSED_LOAD(x[(r + AHEAD) % (AHEAD + 1)], r + AHEAD);
      if (base >= a.n_rows) continue;  // (wave-uniform)
      const uint32_t row = (uint32_t)base + lane;
      const uint4(&xc)[P] = x[r % (AHEAD + 1)];
#pragma unroll
      for (int j = 0; j < P; j++) {
        uint32_t c;
        if (METRIC == kHamming) {
          c = __popc(xc[j].x ^ qpc[j].x) + __popc(xc[j].y ^ qpc[j].y) + __popc(xc[j].z ^ qpc[j].z) + __popc(xc[j].w ^ qpc[j].w);
        } else {
          const uint32_t in = __popc(xc[j].x & qpc[j].x) + __popc(xc[j].y & qpc[j].y) + __popc(xc[j].z & qpc[j].z) + __popc(xc[j].w & qpc[j].w);
          const uint32_t un = __popc(xc[j].x | qpc[j].x) + __popc(xc[j].y | qpc[j].y) + __popc(xc[j].z | qpc[j].z) + __popc(xc[j].w | qpc[j].w);
          c = in | (un << 16);
        }
        strip[j * 64 + lane] = c;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      uint32_t sum = 0;
#pragma unroll
      for (int i = 0; i < P; i++) sum += strip[lane * P + i];  // the P pieces of row `lane` (packed halves cannot carry: P * 128 < 2^16)
      __builtin_amdgcn_wave_barrier();  // strip reads done before the next chunk overwrites it
      float score;
      if (METRIC == kHamming) {
        score = (float)sum;
      } else {
        const uint32_t inter = sum & 0xFFFFu, uni = sum >> 16;
        score = (uni == 0) ? 1.0f : (float)inter / (float)uni;  // simd_explicit.rs:431-442
      }
      bool valid = row < a.n_rows;
      if (valid && a.alive) valid = a.alive[row] != 0;
      if (valid) keys[r] = make_key<HIB>(score, row);
    }
#undef VDB_BITS_FUSED_LOAD
    // the wave's k best of { best, keys[] }
    uint64_t carry = best;
    uint64_t out = kKeyInvalid;
    for (uint32_t e = 0; e < ((a.probe_skip & 2u) ? 0u : k); e++) {
      uint64_t mloc = carry;
#pragma unroll
      for (int r = 0; r < R; r++) mloc = min(mloc, keys[r]);
      const uint64_t wm = wave_min_key(mloc);
      if ((uint32_t)lane == e) out = wm;
      if (wm == kKeyInvalid) break;  // (wave-uniform) nothing left
      if (carry == wm) carry = kKeyInvalid;
#pragma unroll
      for (int r = 0; r < R; r++)
        if (keys[r] == wm) keys[r] = kKeyInvalid;
    }
    best = out;
  }
  if ((uint32_t)lane < k) wl[(size_t)wib * kBitsFusedMaxK + lane] = best;
  __syncthreads();
  if (wib == 0) {  // the block's list: the k best of the four wave lists (4 k <= 64 keys, one per lane)
    const uint32_t wsrc = (uint32_t)lane / k, esrc = (uint32_t)lane % k;
    uint64_t mine = wsrc < 4u ? wl[(size_t)wsrc * kBitsFusedMaxK + esrc] : kKeyInvalid;
    uint64_t out = kKeyInvalid;
    for (uint32_t e = 0; e < k; e++) {
      const uint64_t wm = wave_min_key(mine);
      if ((uint32_t)lane == e) out = wm;
      if (wm == kKeyInvalid) break;  // (wave-uniform)
      if (mine == wm) mine = kKeyInvalid;
    }
    uint64_t* dst = a.part_keys + ((size_t)qi * gridDim.x + blockIdx.x) * k;
    if ((uint32_t)lane < k) dst[lane] = out;
    __threadfence();  // the list is visible device-wide before the ticket is
    if (lane == 0) {
      const uint32_t t = atomicAdd(&a.tickets[qi], 1u);
      *lastf = (t + 1u == gridDim.x) ? 1u : 0u;
    }
  }
  __syncthreads();
  if (*lastf == 0u) return;
  __threadfence();  // (acquire side: every other block's list was written before its ticket)
  if (threadIdx.x == 0) a.tickets[qi] = 0u;  // the next call finds it zero
  __syncthreads();  // (the merge re-uses the LDS from its start)
  if (a.probe_skip & 1u) return;
  fused_tail_merge<HIB>(a.m, qi, smem);
}
size_t sweep_bits_fused_lds_bytes(uint32_t words, uint32_t n_lists) {
  const size_t sweep = (((size_t)words * 4 + (size_t)4 * 64 * (words / 4) * 4 + 15) & ~(size_t)15) + (size_t)4 * kBitsFusedMaxK * 8 + 16;
  (void)n_lists;
  const size_t merge = (size_t)4 * kBitsFusedMaxK * 8 + 256 * 8 + 8 + 16;  // fused_tail_merge
  return std::max(sweep, merge);
}
template <int P>
static void launch_sweep_bits_fused_p(int metric, const BitsFusedArgs& a, int blocks, uint32_t nq, size_t lds, hipStream_t st) {
  if (metric == kHamming)
    hipLaunchKernelGGL((sweep_bits_fused<kHamming, P>), dim3(blocks, nq), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((sweep_bits_fused<kJaccard, P>), dim3(blocks, nq), dim3(256), lds, st, a);
}
// blocks for the one-launch path: every wave one batch of kBitsFusedR chunks when the corpus allows, at most one block per CU
// (measured, 1 M x 768 bits: 256 / 512 / 768 / 1 024 blocks = 44.5 / 48.3 / 47.9 / 48.0 us per one-query call, 58 / 89 / 90 / 90 us for two
// queries: fewer lists for the last block to merge, fewer extractions; profiles/r05u_*)
int sweep_bits_fused_blocks(uint64_t n_rows, int n_cus) {
  const uint64_t nchunks = (n_rows + 63) / 64;
  static const int forced = [] {  // (probe builds: VELESDB_BITS_FUSED_BLOCKS)
    const char* e = probe_env("VELESDB_BITS_FUSED_BLOCKS");
    return e ? std::max(1, atoi(e)) : 0;
  }();
  const int64_t cap = std::min<int64_t>(forced ? forced : (int64_t)n_cus, 256);  // (fused_tail_merge: one list per thread of the last block)
  return (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)(nchunks + 4 * kBitsFusedR - 1) / (4 * kBitsFusedR), cap));
}
bool sweep_bits_fused_supported(uint32_t words, uint32_t nq, uint32_t k) {
  if (nq == 0 || nq > 2 || k == 0 || k > kBitsFusedMaxK || words % 4 != 0) return false;
  switch (words / 4) {
    case 1: case 2: case 3: case 4: case 6: case 8: case 12: return true;
    default: return false;
  }
}
hipError_t launch_sweep_bits_fused(int metric, BitsFusedArgs a, int blocks, uint32_t nq, hipStream_t st) {
  static const uint32_t skip = [] {
    const char* e = probe_env("VELESDB_BITS_FUSED_SKIP");
    return e ? (uint32_t)atoi(e) : 0u;
  }();
  a.probe_skip = skip;
  a.m.part_keys = a.part_keys;
  a.m.n_lists = (uint32_t)blocks;
  a.m.k = a.k;
  const size_t lds = sweep_bits_fused_lds_bytes(a.words, (uint32_t)blocks);
  switch (a.words / 4) {
    case 1: launch_sweep_bits_fused_p<1>(metric, a, blocks, nq, lds, st); break;
    case 2: launch_sweep_bits_fused_p<2>(metric, a, blocks, nq, lds, st); break;
    case 3: launch_sweep_bits_fused_p<3>(metric, a, blocks, nq, lds, st); break;
    case 4: launch_sweep_bits_fused_p<4>(metric, a, blocks, nq, lds, st); break;
    case 6: launch_sweep_bits_fused_p<6>(metric, a, blocks, nq, lds, st); break;
    case 8: launch_sweep_bits_fused_p<8>(metric, a, blocks, nq, lds, st); break;
    case 12: launch_sweep_bits_fused_p<12>(metric, a, blocks, nq, lds, st); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

void launch_sweep_bits(int metric, const BitsArgs& a, int blocks, uint32_t nq, hipStream_t st) {
  const size_t lds = ((((size_t)a.k * 8 + 8 + 15) & ~(size_t)15) + (size_t)a.words * 4 + 15) & ~(size_t)15;
  // rows of 1, 2, 3, 4, 6, 8 or 12 sixteen-byte pieces (128 .. 1 536 bits): the coalesced kernel (+ 4 x 64 P counts of LDS)
  const uint32_t pcs = a.words / 4;
  const size_t lds_co = lds + (size_t)4 * 64 * pcs * 4;
  if (a.words % 4 == 0 && lds_co <= 64 * 1024) {
    switch (pcs) {
      case 1: return launch_sweep_bits_co<1>(metric, a, blocks, nq, lds_co, st);
      case 2: return launch_sweep_bits_co<2>(metric, a, blocks, nq, lds_co, st);
      case 3: return launch_sweep_bits_co<3>(metric, a, blocks, nq, lds_co, st);
      case 4: return launch_sweep_bits_co<4>(metric, a, blocks, nq, lds_co, st);
      case 6: return launch_sweep_bits_co<6>(metric, a, blocks, nq, lds_co, st);
      case 8: return launch_sweep_bits_co<8>(metric, a, blocks, nq, lds_co, st);
      case 12: return launch_sweep_bits_co<12>(metric, a, blocks, nq, lds_co, st);
      default: break;
    }
  }
  if (metric == kHamming)
    hipLaunchKernelGGL((sweep_topk_bits<kHamming>), dim3(blocks, nq), dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((sweep_topk_bits<kJaccard>), dim3(blocks, nq), dim3(256), lds, st, a);
}

void launch_prep_rows(const PrepArgs& a, hipStream_t st) {
  if (a.n_rows == 0) return;
  int blocks = (int)((a.n_rows + 3) / 4);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(prep_rows, dim3(blocks), dim3(256), 0, st, a);
}

void launch_score_rows(int metric, const ScoreArgs& a, hipStream_t st) {
  if (a.n_rows == 0) return;
  uint64_t want = (a.n_rows + 3) / 4;
  int blocks = (int)(want > 4096 ? 4096 : want);
  switch (metric) {
    case kCosine: hipLaunchKernelGGL((score_rows<kCosine>), dim3(blocks), dim3(256), 0, st, a); break;
    case kEuclidean: hipLaunchKernelGGL((score_rows<kEuclidean>), dim3(blocks), dim3(256), 0, st, a); break;
    case kDot: hipLaunchKernelGGL((score_rows<kDot>), dim3(blocks), dim3(256), 0, st, a); break;
    case kHamming: hipLaunchKernelGGL((score_rows<kHamming>), dim3(blocks), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((score_rows<kJaccard>), dim3(blocks), dim3(256), 0, st, a); break;
  }
}

}  // namespace vdb
