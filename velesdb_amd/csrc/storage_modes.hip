// storage_modes.hip — the storage modes of core/quantization.rs on the GPU: SQ8 (QuantizedVector: one byte per
// dimension + per-vector min/max, :204-316) and Binary (BinaryQuantizedVector: one sign bit per dimension, :48-202),
// the quantisers, and exact top-k searches of f32 queries over the quantised corpus with the reference's asymmetric
// distances dot_product_quantized_simd / euclidean_squared_quantized_simd / cosine_similarity_quantized_simd
// (:410-554) and BinaryQuantizedVector::hamming_distance (:123-135).  crud.rs:66-82 builds exactly these codes on
// upsert when a collection's StorageMode is SQ8 / Binary.
//
// Arithmetic: the reference's functions are plain scalar Rust — every sum is one left-to-right chain over the
// dimensions, nothing is fused (Rust never contracts a*b+c).  That order is reproduced exactly: ONE LANE OWNS ONE
// ROW and walks its dimensions in order with separately rounded multiplies and adds (__fmul_rn / __fadd_rn = plain
// * and + under -ffp-contract=off; sqrtf and / are correctly rounded; NOT __fsqrt_rn, which is the native
// approximation in this toolchain), so
// scores are bit-identical to the oracle's restatement (oracle/vdb_oracle.cpp vo_sq8_*), and — the reference's
// functions being deterministic scalar code — to the reference itself.
//
// sweep_topk_sq8<METRIC,B>: a wave takes 64 rows; their codes arrive through an LDS tile (16 rows x 64 B per load
// instruction, coalesced; each lane then reads its own row's 64 bytes with 4 conflict-free ds_read_b128), the B
// queries sit in LDS dimension-major so one broadcast ds_read_b128 feeds 4 queries; per element 3 VALU ops of
// dequantisation shared by the B queries + 2 per query.  Bound: HBM for B <= 4 (N*(dim+16) bytes per pass: 784 MB
// at 1 M x 768), the vector ALUs beyond.  Top-k: block-shared locked lists (vdb_device.hpp) + merge_topk.
#include <algorithm>
#include <cstring>
#include <memory>

#include "vdb_device.hpp"
#include "vdb_index.hpp"
#include "vdb_kernels.hpp"

namespace vdb {

constexpr float kF32Eps = 1.1920929e-07f;  // f32::EPSILON

// ---- quantisers ---------------------------------------------------------------------------------------------
// QuantizedVector::from_f32 (:229-255), one wave per row: min / max are order-independent, the codes elementwise.
// Also keeps quantized_norm_sq of cosine_similarity_quantized_simd (:531-546) — a left-to-right chain, so it is
// computed by sq8_norms below, one lane per row.
__global__ __launch_bounds__(256) void sq8_quantize_rows(const float* rows, uint64_t row_stride, uint8_t* codes,
                                                         uint64_t code_stride, float* vmin, float* vmax, uint32_t row0,
                                                         uint32_t n_rows, uint32_t dim) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t r = wave; r < n_rows; r += nwaves) {
    const uint32_t row = row0 + r;
    const float* p = rows + (size_t)row * row_stride;
    float mn = __uint_as_float(0x7F800000u), mx = __uint_as_float(0xFF800000u);
    for (uint32_t i = lane; i < dim; i += 64) {
      mn = fminf(mn, p[i]);  // f32::min / f32::max semantics: a NaN operand yields the other one
      mx = fmaxf(mx, p[i]);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, s, 64));
      mx = fmaxf(mx, __shfl_xor(mx, s, 64));
    }
    const float range = __fsub_rn(mx, mn);
    uint8_t* c = codes + (size_t)row * code_stride;
    if (range < kF32Eps) {
      for (uint32_t i = lane; i < code_stride; i += 64) c[i] = i < dim ? 128 : 0;
    } else {
      const float scale = __fdiv_rn(255.0f, range);
      for (uint32_t i = lane; i < code_stride; i += 64) {
        uint8_t q = 0;
        if (i < dim) {
          const float normalized = __fmul_rn(__fsub_rn(p[i], mn), scale);
          float rr = roundf(normalized);  // half away from zero, like f32::round
          rr = rr < 0.0f ? 0.0f : (rr > 255.0f ? 255.0f : rr);
          q = (rr != rr) ? (uint8_t)0 : (uint8_t)rr;  // NaN `as u8` saturates to 0
        }
        c[i] = q;
      }
    }
    if (lane == 0) {
      vmin[row] = mn;
      vmax[row] = mx;
    }
  }
}

__global__ __launch_bounds__(256) void sq8_norms(const uint8_t* codes, uint64_t code_stride, const float* vmin,
                                                 const float* vmax, float* nsq, uint32_t row0, uint32_t n_rows,
                                                 uint32_t dim) {
  const uint32_t r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  const uint32_t row = row0 + r;
  const float mn = vmin[row], range = __fsub_rn(vmax[row], mn);
  const float scale = range < kF32Eps ? 0.0f : __fdiv_rn(range, 255.0f);
  const uint32_t* c = reinterpret_cast<const uint32_t*>(codes + (size_t)row * code_stride);
  float s = 0.0f;
  for (uint32_t i = 0; i < dim; i += 4) {
    const uint32_t w = c[i / 4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      if (i + e < dim) {
        const float dq = __fadd_rn(__fmul_rn((float)((w >> (8 * e)) & 0xFFu), scale), mn);
        s = __fadd_rn(s, __fmul_rn(dq, dq));
      }
    }
  }
  nsq[row] = s;
}

// BinaryQuantizedVector::from_f32 (:68-86): bit i = (v[i] >= 0.0), LSB-first in bytes = bit i%32 of word i/32.
// One wave per row: a ballot over 64 consecutive dimensions IS eight of the reference's bytes.
__global__ __launch_bounds__(256) void sign_bits_rows(const float* rows, uint64_t row_stride, uint32_t* bits, uint32_t words,
                                                      uint32_t row0, uint32_t n_rows, uint32_t dim) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t r = wave; r < n_rows; r += nwaves) {
    const uint32_t row = row0 + r;
    const float* p = rows + (size_t)row * row_stride;
    uint32_t* dst = bits + (size_t)row * words;
    for (uint32_t e0 = 0; e0 < words * 32; e0 += 64) {
      const uint32_t e = e0 + lane;
      const bool bit = e < dim && p[e] >= 0.0f;  // NaN >= 0 is false, -0.0 >= 0 is true, as on the CPU
      const uint64_t m = __ballot(bit);
      if (lane == 0) {
        dst[e0 / 32] = (uint32_t)m;
        if (e0 / 32 + 1 < words) dst[e0 / 32 + 1] = (uint32_t)(m >> 32);
      }
    }
  }
}

// ---- SQ8 sweep ------------------------------------------------------------------------------------------------
struct Sq8Args {
  const uint8_t* codes;   // [n_rows][code_stride], code_stride % 16 == 0, padding bytes 0
  const float* vmin;
  const float* vmax;
  const float* nsq;       // quantized_norm_sq per row (cosine)
  const uint8_t* alive;
  const float* queries;   // [nq][q_stride] f32
  uint64_t* part_keys;    // [nq][n_blocks][k]
  uint64_t code_stride, q_stride;
  uint32_t n_rows, dim, nq, k;
  // gathered variant (the exact pass over the queries a selection stage could not prove): block row blockIdx.y serves the
  // queries qmap[B y .. B y + B - 1] of the *qcount listed ones and exits when there are none; part_keys slots follow the list
  const uint32_t* qmap;
  const uint32_t* qcount;
};

constexpr int kSq8TileStride = 80;  // bytes per row of the staging tile: 64 + 16 padding (conflict-free b128 reads)

typedef float sq8_f32x4 __attribute__((ext_vector_type(4)));
// One broadcast read of four queries' element (16 bytes at a wave-uniform LDS address), WRITTEN AS AN INSTRUCTION.  Left to the
// compiler the unrolled 64-code chunk came out as [one ds_read, s_waitcnt lgkmcnt(0), four vector instructions] x 128: every LDS
// round trip exposed, half of every wave's time in s_waitcnt (round 5 counters, profiles/r05k_*: vector ALU busy 0.54, LDS array
// 0.43, SQ_WAIT_ANY 0.51 of the wave cycles).  Here the reads of a whole code word (4 codes x B queries) are issued together, ONE
// WORD AHEAD of the arithmetic that uses them (two register sets), and one wait per word follows the arithmetic of the word before.
// (Measured and dropped in the same round: the queries' elements through scalar loads from a transposed global copy — no gain, the
// wait was the cost, not the broadcast's 64-fold write; plain instead of packed multiplies / adds — 18 % slower.)
template <int OFF>
__device__ __forceinline__ void sq8_lds_read4(sq8_f32x4& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// (the wait names the values it makes valid: a use cannot be scheduled in front of it)
__device__ __forceinline__ void sq8_lds_wait(sq8_f32x4 (&q)[4][1]) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0][0]), "+v"(q[1][0]), "+v"(q[2][0]), "+v"(q[3][0])::"memory");
}
__device__ __forceinline__ void sq8_lds_wait(sq8_f32x4 (&q)[4][2]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[2][0]), "+v"(q[2][1]), "+v"(q[3][0]), "+v"(q[3][1])::"memory");
}
template <int B, int WI>
__device__ __forceinline__ void sq8_read_word(sq8_f32x4 (&q)[4][B / 4], uint32_t qaddr) {
  // element e of the word, queries 4 j .. 4 j + 3: byte offset ((4 WI + e) B + 4 j) 4 behind the chunk's first element
  sq8_lds_read4<((4 * WI + 0) * B + 0) * 4>(q[0][0], qaddr);
  sq8_lds_read4<((4 * WI + 1) * B + 0) * 4>(q[1][0], qaddr);
  sq8_lds_read4<((4 * WI + 2) * B + 0) * 4>(q[2][0], qaddr);
  sq8_lds_read4<((4 * WI + 3) * B + 0) * 4>(q[3][0], qaddr);
  if constexpr (B == 8) {
    sq8_lds_read4<((4 * WI + 0) * B + 4) * 4>(q[0][1], qaddr);
    sq8_lds_read4<((4 * WI + 1) * B + 4) * 4>(q[1][1], qaddr);
    sq8_lds_read4<((4 * WI + 2) * B + 4) * 4>(q[2][1], qaddr);
    sq8_lds_read4<((4 * WI + 3) * B + 4) * 4>(q[3][1], qaddr);
  }
}
// one code word (4 codes of the lane's row) against the B queries: the reference's chains, element by element in order
// (dot: quantization.rs:452-466; squared L2: :495-507 — sum += ((f0^2 + f1^2) + f2^2) + f3^2 per group of four)
template <int METRIC, int B>
__device__ __forceinline__ void sq8_word_math(float (&acc)[B], uint32_t w, float scale, float mn, const sq8_f32x4 (&q)[4][B / 4]) {
  float dq[4];
#pragma unroll
  for (int e = 0; e < 4; e++) dq[e] = __fadd_rn(__fmul_rn((float)((w >> (8 * e)) & 0xFFu), scale), mn);
  constexpr int P = B / 2;  // query pairs: v_pk_mul_f32 / v_pk_add_f32 — each half is the same IEEE operation as the scalar form
  if (METRIC == kEuclidean) {
#pragma unroll
    for (int p = 0; p < P; p++) {
      f32x2 t = {0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const f32x2 f = f32x2{q[e][p / 2][2 * (p & 1)], q[e][p / 2][2 * (p & 1) + 1]} - f32x2{dq[e], dq[e]};
        t = e == 0 ? f * f : t + f * f;
      }
      acc[2 * p] = __fadd_rn(acc[2 * p], t.x);
      acc[2 * p + 1] = __fadd_rn(acc[2 * p + 1], t.y);
    }
  } else {
    // the products of TWO codes first, then their additions in the reference's order (code e before code e + 1 on every chain): an
    // addition then stands four to eight instructions behind the multiply it needs, instead of directly behind it
#pragma unroll
    for (int e0 = 0; e0 < 4; e0 += 2) {
      f32x2 pr[2][P];
#pragma unroll
      for (int e = 0; e < 2; e++)
#pragma unroll
        for (int p = 0; p < P; p++) pr[e][p] = f32x2{q[e0 + e][p / 2][2 * (p & 1)], q[e0 + e][p / 2][2 * (p & 1) + 1]} * f32x2{dq[e0 + e], dq[e0 + e]};
#pragma unroll
      for (int e = 0; e < 2; e++)
#pragma unroll
        for (int p = 0; p < P; p++) {
          const f32x2 r = f32x2{acc[2 * p], acc[2 * p + 1]} + pr[e][p];
          acc[2 * p] = r.x;
          acc[2 * p + 1] = r.y;
        }
    }
  }
}
// words WI, WI + 1 of a chunk: q0 holds word WI's elements (valid), q1 receives word WI + 1's while word WI is multiplied, and so on
template <int METRIC, int B, int WI>
__device__ __forceinline__ void sq8_word_pair(float (&acc)[B], const uint32_t (&wds)[16], float scale, float mn, uint32_t qaddr,
                                              sq8_f32x4 (&q0)[4][B / 4], sq8_f32x4 (&q1)[4][B / 4]) {
  sq8_read_word<B, WI + 1>(q1, qaddr);
  sq8_word_math<METRIC, B>(acc, wds[WI], scale, mn, q0);
  sq8_lds_wait(q1);
  if constexpr (WI + 2 < 16) sq8_read_word<B, WI + 2>(q0, qaddr);
  sq8_word_math<METRIC, B>(acc, wds[WI + 1], scale, mn, q1);
  if constexpr (WI + 2 < 16) sq8_lds_wait(q0);
}

template <int METRIC, int B>
__global__ __launch_bounds__(256) void sweep_topk_sq8(Sq8Args a) {
  constexpr bool HIB = METRIC != kEuclidean;  // cosine / dot similarity: larger is better; squared L2: smaller
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int wib = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t k = a.k, dim = a.dim;
  const uint32_t dpad = (dim + 3u) & ~3u;
  // LDS: queries [dpad][B] f32 | per-wave tiles [4][64][80] | lists [B][k] | cnt[B] | lock[B] | qsum[B] | qnsq[B]
  float* qs = reinterpret_cast<float*>(smem);
  unsigned char* tiles = smem + (size_t)dpad * B * 4;
  unsigned char* tile = tiles + (size_t)wib * 64 * kSq8TileStride;
  unsigned char* tail = tiles + (size_t)4 * 64 * kSq8TileStride;
  lds_vu64* lists = (lds_vu64*)(lds_void_p)(tail);
  lds_vu32* cnts = (lds_vu32*)(lds_void_p)(tail + (size_t)B * k * 8);
  uint32_t* locks = reinterpret_cast<uint32_t*>(tail + (size_t)B * k * 8 + B * 4);
  float* qsum = reinterpret_cast<float*>(tail + (size_t)B * k * 8 + B * 8);
  float* qnsq = reinterpret_cast<float*>(tail + (size_t)B * k * 8 + B * 12);

  uint32_t nq_here = a.nq, slot0 = 0;
  if (a.qmap) {
    const uint32_t listed = *a.qcount;
    slot0 = blockIdx.y * (uint32_t)B;
    if (slot0 >= listed) return;  // (uniform per block)
    nq_here = min((uint32_t)B, listed - slot0);
  }
  for (uint32_t i = tid; i < dpad * B; i += 256) {
    const uint32_t d = i / B, b = i % B;
    const uint32_t qrow = (a.qmap && b < nq_here) ? a.qmap[slot0 + b] : b;
    qs[i] = (d < dim && b < nq_here) ? a.queries[(size_t)qrow * a.q_stride + d] : 0.0f;
  }
  if (tid < B) {
    cnts[tid] = 0;
    locks[tid] = 0;
  }
  __syncthreads();
  if (tid < B) {  // per query: sum(q) (constant-vector branch, :332-334) and sum(q*q) (:528), left to right
    float s = 0.0f, n2 = 0.0f;
    for (uint32_t d = 0; d < dim; d++) {
      const float x = qs[d * B + tid];
      s = __fadd_rn(s, x);
      n2 = __fadd_rn(n2, __fmul_rn(x, x));
    }
    qsum[tid] = s;
    qnsq[tid] = n2;
  }
  __syncthreads();

  const uint32_t ngroups = (a.n_rows + 63) / 64;
  for (uint32_t g = blockIdx.x * 4 + wib; g < ngroups; g += gridDim.x * 4) {
    const uint32_t row = g * 64 + lane;
    const bool valid = row < a.n_rows;
    const uint32_t rowc = valid ? row : a.n_rows - 1;
    const float mn = a.vmin[rowc];
    const float range = __fsub_rn(a.vmax[rowc], mn);
    const bool flat = range < kF32Eps;
    const float scale = __fdiv_rn(range, 255.0f);
    float acc[B];
#pragma unroll
    for (int b = 0; b < B; b++) acc[b] = 0.0f;
    // query elements: LDS, dimension-major [dim][B] — one broadcast ds_read_b128 brings element i of all 4 queries as two
    // adjacent pairs, the operands of the packed multiplies below
    // staging loads run one 64-byte column chunk ahead of the arithmetic (registers), so their latency hides
    // behind the previous chunk's ~700 VALU instructions
    // (the loads are RAW — clamped address, always in bounds — and the rows / columns past the end are zeroed where the staged value
    // is USED, one chunk later: with the select beside the load, as rounds 1-4 had it, the compiler waited for the four loads
    // (vmcnt) right where it had issued them, and every chunk paid the codes' trip from HBM in full — half of every wave's time,
    // profiles/r05k_*: SQ_WAIT_ANY 0.51)
    uint4 stg[4];
    auto stage_load = [&](uint32_t c0) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t rr = g * 64 + 16 * j + (lane >> 2);
        const uint32_t off = c0 + (lane & 3) * 16;
        const uint32_t rc = rr < a.n_rows ? rr : a.n_rows - 1;
        const uint32_t oc = off < a.code_stride ? off : 0u;
        stg[j] = *reinterpret_cast<const uint4*>(a.codes + (size_t)rc * a.code_stride + oc);
      }
    };
    stage_load(0);
    for (uint32_t c0 = 0; c0 < dim; c0 += 64) {
      // stage 64 rows x 64 B: lane l moves 16 B of row 16 j + l/4, part l%4
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const bool ok = g * 64 + 16 * j + (lane >> 2) < a.n_rows && c0 + (lane & 3) * 16 < a.code_stride;
        *reinterpret_cast<uint4*>(tile + (16 * j + (lane >> 2)) * kSq8TileStride + (lane & 3) * 16) =
            make_uint4(ok ? stg[j].x : 0u, ok ? stg[j].y : 0u, ok ? stg[j].z : 0u, ok ? stg[j].w : 0u);
      }
      if (c0 + 64 < dim) stage_load(c0 + 64);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      uint4 w4[4];
#pragma unroll
      for (int j = 0; j < 4; j++) w4[j] = *reinterpret_cast<const uint4*>(tile + lane * kSq8TileStride + j * 16);
      __builtin_amdgcn_wave_barrier();  // tile reads done before the next chunk overwrites it
      const uint32_t wds[16] = {w4[0].x, w4[0].y, w4[0].z, w4[0].w, w4[1].x, w4[1].y, w4[1].z, w4[1].w,
                                w4[2].x, w4[2].y, w4[2].z, w4[2].w, w4[3].x, w4[3].y, w4[3].z, w4[3].w};
      if ((B == 4 || B == 8) && c0 + 64 <= dim) {  // full chunk, 4 or 8 queries: grouped query reads, one word ahead
        if constexpr (B == 4 || B == 8) {
          const uint32_t qaddr = (uint32_t)(size_t)(lds_void_p)smem + c0 * (uint32_t)B * 4u;  // (qs starts the dynamic LDS; uniform)
          sq8_f32x4 qa[4][B / 4], qb[4][B / 4];
          sq8_read_word<B, 0>(qa, qaddr);
          sq8_lds_wait(qa);
          sq8_word_pair<METRIC, B, 0>(acc, wds, scale, mn, qaddr, qa, qb);
          sq8_word_pair<METRIC, B, 2>(acc, wds, scale, mn, qaddr, qa, qb);
          sq8_word_pair<METRIC, B, 4>(acc, wds, scale, mn, qaddr, qa, qb);
          sq8_word_pair<METRIC, B, 6>(acc, wds, scale, mn, qaddr, qa, qb);
          sq8_word_pair<METRIC, B, 8>(acc, wds, scale, mn, qaddr, qa, qb);
          sq8_word_pair<METRIC, B, 10>(acc, wds, scale, mn, qaddr, qa, qb);
          sq8_word_pair<METRIC, B, 12>(acc, wds, scale, mn, qaddr, qa, qb);
          sq8_word_pair<METRIC, B, 14>(acc, wds, scale, mn, qaddr, qa, qb);
        }
      } else if (c0 + 64 <= dim) {  // full chunk: no bounds tests => ONE basic block, the scheduler overlaps the LDS reads of
                             // later groups with the arithmetic of earlier ones (per-group branches pinned them)
  #pragma unroll
        for (int wi = 0; wi < 16; wi++) {
          const uint32_t i0 = c0 + wi * 4;
          const uint32_t w = wds[wi];
          float dq[4];
  #pragma unroll
          for (int e = 0; e < 4; e++) dq[e] = __fadd_rn(__fmul_rn((float)((w >> (8 * e)) & 0xFFu), scale), mn);
          // Two queries per instruction where B is even (v_pk_mul_f32 / v_pk_add_f32: each half is the same IEEE
          // operation as the scalar form, so the bits do not change; a plain f32 VALU op runs at half the packed rate).
          constexpr int P = B / 2;  // query pairs
          if (METRIC == kEuclidean) {
            if (true) {  // a full group of four: sum += ((f0^2 + f1^2) + f2^2) + f3^2 (:495-507)
  #pragma unroll
              for (int p = 0; p < P; p++) {
                f32x2 t = {0.f, 0.f};
  #pragma unroll
                for (int e = 0; e < 4; e++) {
                  const f32x2 f = f32x2{qs[(size_t)(i0 + e) * B + (2 * p)], qs[(size_t)(i0 + e) * B + (2 * p + 1)]} - f32x2{dq[e], dq[e]};
                  t = e == 0 ? f * f : t + f * f;
                }
                acc[2 * p] = __fadd_rn(acc[2 * p], t.x);
                acc[2 * p + 1] = __fadd_rn(acc[2 * p + 1], t.y);
              }
              if (B & 1) {
                const int b = B - 1;
                const float f0 = __fsub_rn(qs[(size_t)(i0 + 0) * B + (b)], dq[0]), f1 = __fsub_rn(qs[(size_t)(i0 + 1) * B + (b)], dq[1]);
                const float f2 = __fsub_rn(qs[(size_t)(i0 + 2) * B + (b)], dq[2]), f3 = __fsub_rn(qs[(size_t)(i0 + 3) * B + (b)], dq[3]);
                const float t = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(f0, f0), __fmul_rn(f1, f1)), __fmul_rn(f2, f2)),
                                          __fmul_rn(f3, f3));
                acc[b] = __fadd_rn(acc[b], t);
              }
            } else {  // remainder: one element at a time (:510-515)
  #pragma unroll
              for (int e = 0; e < 4; e++) {
                if (true) {
  #pragma unroll
                  for (int b = 0; b < B; b++) {
                    const float f = __fsub_rn(qs[(size_t)(i0 + e) * B + (b)], dq[e]);
                    acc[b] = __fadd_rn(acc[b], __fmul_rn(f, f));
                  }
                }
              }
            }
          } else {  // dot chain (:452-466), also the numerator of the cosine
  #pragma unroll
            for (int e = 0; e < 4; e++) {
              if (true) {
  #pragma unroll
                for (int p = 0; p < P; p++) {
                  const f32x2 r = f32x2{acc[2 * p], acc[2 * p + 1]} +
                                  f32x2{qs[(size_t)(i0 + e) * B + (2 * p)], qs[(size_t)(i0 + e) * B + (2 * p + 1)]} * f32x2{dq[e], dq[e]};
                  acc[2 * p] = r.x;
                  acc[2 * p + 1] = r.y;
                }
                if (B & 1) acc[B - 1] = __fadd_rn(acc[B - 1], __fmul_rn(qs[(size_t)(i0 + e) * B + (B - 1)], dq[e]));
              }
            }
          }
        }
      } else {
  #pragma unroll
        for (int wi = 0; wi < 16; wi++) {
          const uint32_t i0 = c0 + wi * 4;
          if (i0 >= dim) break;  // uniform
          const uint32_t w = wds[wi];
          float dq[4];
  #pragma unroll
          for (int e = 0; e < 4; e++) dq[e] = __fadd_rn(__fmul_rn((float)((w >> (8 * e)) & 0xFFu), scale), mn);
          // Two queries per instruction where B is even (v_pk_mul_f32 / v_pk_add_f32: each half is the same IEEE
          // operation as the scalar form, so the bits do not change; a plain f32 VALU op runs at half the packed rate).
          constexpr int P = B / 2;  // query pairs
          if (METRIC == kEuclidean) {
            if (i0 + 3 < dim) {  // a full group of four: sum += ((f0^2 + f1^2) + f2^2) + f3^2 (:495-507)
  #pragma unroll
              for (int p = 0; p < P; p++) {
                f32x2 t = {0.f, 0.f};
  #pragma unroll
                for (int e = 0; e < 4; e++) {
                  const f32x2 f = f32x2{qs[(size_t)(i0 + e) * B + (2 * p)], qs[(size_t)(i0 + e) * B + (2 * p + 1)]} - f32x2{dq[e], dq[e]};
                  t = e == 0 ? f * f : t + f * f;
                }
                acc[2 * p] = __fadd_rn(acc[2 * p], t.x);
                acc[2 * p + 1] = __fadd_rn(acc[2 * p + 1], t.y);
              }
              if (B & 1) {
                const int b = B - 1;
                const float f0 = __fsub_rn(qs[(size_t)(i0 + 0) * B + (b)], dq[0]), f1 = __fsub_rn(qs[(size_t)(i0 + 1) * B + (b)], dq[1]);
                const float f2 = __fsub_rn(qs[(size_t)(i0 + 2) * B + (b)], dq[2]), f3 = __fsub_rn(qs[(size_t)(i0 + 3) * B + (b)], dq[3]);
                const float t = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(f0, f0), __fmul_rn(f1, f1)), __fmul_rn(f2, f2)),
                                          __fmul_rn(f3, f3));
                acc[b] = __fadd_rn(acc[b], t);
              }
            } else {  // remainder: one element at a time (:510-515)
  #pragma unroll
              for (int e = 0; e < 4; e++) {
                if (i0 + e < dim) {
  #pragma unroll
                  for (int b = 0; b < B; b++) {
                    const float f = __fsub_rn(qs[(size_t)(i0 + e) * B + (b)], dq[e]);
                    acc[b] = __fadd_rn(acc[b], __fmul_rn(f, f));
                  }
                }
              }
            }
          } else {  // dot chain (:452-466), also the numerator of the cosine
  #pragma unroll
            for (int e = 0; e < 4; e++) {
              if (i0 + e < dim) {
  #pragma unroll
                for (int p = 0; p < P; p++) {
                  const f32x2 r = f32x2{acc[2 * p], acc[2 * p + 1]} +
                                  f32x2{qs[(size_t)(i0 + e) * B + (2 * p)], qs[(size_t)(i0 + e) * B + (2 * p + 1)]} * f32x2{dq[e], dq[e]};
                  acc[2 * p] = r.x;
                  acc[2 * p + 1] = r.y;
                }
                if (B & 1) acc[B - 1] = __fadd_rn(acc[B - 1], __fmul_rn(qs[(size_t)(i0 + e) * B + (B - 1)], dq[e]));
              }
            }
          }
        }
      }
    }
    // constant-vector rows (range < EPSILON) take the reference's other branch; rare, so a divergent redo
    if (__ballot(valid && flat)) {
      if (valid && flat) {
#pragma unroll
        for (int b = 0; b < B; b++) {
          if (METRIC == kEuclidean) {  // sum((q - value)^2), left to right (:357-361)
            float s = 0.0f;
            for (uint32_t d = 0; d < dim; d++) {
              const float f = __fsub_rn(qs[d * B + b], mn);
              s = __fadd_rn(s, __fmul_rn(f, f));
            }
            acc[b] = s;
          } else {
            acc[b] = __fmul_rn(qsum[b], mn);  // sum(q) * value (:330-334)
          }
        }
      }
    }
    const float vn2 = (METRIC == kCosine) ? a.nsq[rowc] : 0.0f;
    uint64_t taus[B];  // the lists' bounds, read together (vdb_device.hpp list_tau_relaxed)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int b = 0; b < B; b++) taus[b] = list_tau_relaxed(lists, cnts, (uint32_t)b, k);
#pragma unroll
    for (int b = 0; b < B; b++) {
      float score = acc[b];
      if (METRIC == kCosine) {  // :548-553
        const float denom = sqrtf(__fmul_rn(qnsq[b], vn2));
        score = denom < kF32Eps ? 0.0f : __fdiv_rn(acc[b], denom);
      }
      const bool ok = valid && (uint32_t)b < nq_here;
      const uint64_t key = ok ? make_key<HIB>(score, row) : kKeyInvalid;
      uint64_t mask = __ballot(key < taus[b]);
      while (mask) {
        const int src = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const uint64_t kk = readlane64(key, src);
        if (a.alive && a.alive[key_row(kk)] == 0) continue;
        shared_list_offer(lists + (size_t)b * k, cnts + b, locks + b, k, kk, lane);
      }
    }
  }
  __syncthreads();
  for (uint32_t b = wib; b < nq_here && b < (uint32_t)B; b += 4) {
    const uint32_t c = cnts[b];
    uint64_t* out = a.part_keys + ((size_t)(slot0 + b) * gridDim.x + blockIdx.x) * k;
    for (uint32_t e = lane; e < k; e += 64) out[e] = e < c ? lists[(size_t)b * k + e] : kKeyInvalid;
  }
}

static size_t sq8_lds_bytes(int B, uint32_t k, uint32_t dim) {
  const size_t dpad = (dim + 3u) & ~3u;
  return (dpad * B * 4 + (size_t)4 * 64 * kSq8TileStride + (size_t)B * k * 8 + (size_t)B * 16 + 15) & ~(size_t)15;
}

template <int METRIC, int B>
static hipError_t launch_sq8_t(const Sq8Args& a, int blocks, size_t lds, hipStream_t st, int groups = 1) {
  static bool done = false;
  if (lds > 64 * 1024 && !done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sweep_topk_sq8<METRIC, B>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    done = true;
  }
  hipLaunchKernelGGL((sweep_topk_sq8<METRIC, B>), dim3(blocks, groups), dim3(256), lds, st, a);
  return hipGetLastError();
}
template <int B>
static hipError_t launch_sq8_m(int metric, const Sq8Args& a, int blocks, size_t lds, hipStream_t st, int groups = 1) {
  if (metric == kCosine) return launch_sq8_t<kCosine, B>(a, blocks, lds, st, groups);
  if (metric == kEuclidean) return launch_sq8_t<kEuclidean, B>(a, blocks, lds, st, groups);
  return launch_sq8_t<kDot, B>(a, blocks, lds, st, groups);
}

// ---- selection stage over the SQ8 storage mode (select_stage.hip brute_split_dev, level 3) -------------------------------------
// The dequantised rows d = code * scale + min as a bf16 image (what the matrix cores select on), their norms sqrt(nsq), and
// the first kSplitSeedRows dequantised rows in f32 (what the exact matrix-core kernel seeds the thresholds from).
// Euclidean (aug): the augmented form of sweep_split.hip — image rows of dim + 64 (slots dim, dim + 1 = hi / lo of -nsq / 2), seed
// rows of dim + 4 (slot dim = -nsq / 2)
__global__ __launch_bounds__(256) void sq8_dequant_rows(const uint8_t* codes, uint64_t code_stride, const float* vmin, const float* vmax,
                                                        const float* nsq, uint16_t* img, float* nrm, float* seed, uint32_t seed_rows,
                                                        uint32_t row0, uint32_t n_rows, uint32_t dim, uint32_t dim_a, uint32_t dim_s,
                                                        uint32_t* rho_max_bits) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t r = wave; r < n_rows; r += nwaves) {
    const uint32_t row = row0 + r;
    double se = 0.0, sx = 0.0;  // the dequantised row's bf16 rounding residual ratio (sweep_split.hip select_eps_q)
    const float mn = vmin[row], range = __fsub_rn(vmax[row], mn);
    const bool flat = range < kF32Eps;
    const float scale = __fdiv_rn(range, 255.0f);
    const uint8_t* c = codes + (size_t)row * code_stride;
    for (uint32_t i = lane * 4; i < dim; i += 256) {  // dim % 4 == 0 (selection needs dim % 64 == 0)
      const uint32_t w = *reinterpret_cast<const uint32_t*>(c + i);
      float d[4];
#pragma unroll
      for (int e = 0; e < 4; e++) d[e] = flat ? mn : __fadd_rn(__fmul_rn((float)((w >> (8 * e)) & 0xFFu), scale), mn);
      uint32_t h[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {  // round to nearest even; NaN stays NaN
        uint32_t u = __float_as_uint(d[e]);
        h[e] = (u & 0x7FFFFFFFu) > 0x7F800000u ? ((u >> 16) | 0x0040u) : ((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
        const float de = d[e] - __uint_as_float(h[e] << 16);  // exact in f32
        se += (double)de * (double)de;
        sx += (double)d[e] * (double)d[e];
      }
      *reinterpret_cast<uint2*>(img + (size_t)row * dim_a + i) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
      if (row < seed_rows) *reinterpret_cast<float4*>(seed + (size_t)row * dim_s + i) = make_float4(d[0], d[1], d[2], d[3]);
    }
    if (dim_a > dim) {
      const float hh = -0.5f * nsq[row];
      uint32_t u = __float_as_uint(hh);
      const uint32_t hi = (u & 0x7FFFFFFFu) > 0x7F800000u ? ((u >> 16) | 0x0040u) : ((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
      u = __float_as_uint(hh - __uint_as_float(hi << 16));
      const uint32_t lo = (u & 0x7FFFFFFFu) > 0x7F800000u ? ((u >> 16) | 0x0040u) : ((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
      if (dim + lane < dim_a) img[(size_t)row * dim_a + dim + lane] = lane == 0 ? (uint16_t)hi : (lane == 1 ? (uint16_t)lo : (uint16_t)0);
      if (row < seed_rows && lane < 4 && dim + lane < dim_s) seed[(size_t)row * dim_s + dim + lane] = lane == 0 ? hh : 0.0f;
    }
    if (lane == 0) nrm[row] = sqrtf(nsq[row]);
    if (rho_max_bits) {
#pragma unroll
      for (int s2 = 32; s2 > 0; s2 >>= 1) {
        se += __shfl_xor(se, s2, 64);
        sx += __shfl_xor(sx, s2, 64);
      }
      if (lane == 0 && sx > 0.0) {
        const float rho = (float)(sqrt(se / sx) * 1.0000002);
        if (rho == rho && rho < __uint_as_float(0x7F800000u)) atomicMax(rho_max_bits, __float_as_uint(rho));
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
static void quantize_range(vdb_hip_index* ix, uint64_t first, uint64_t n) {
  if (n == 0) return;
  const int blocks = (int)std::min<uint64_t>((n + 3) / 4, 4096);
  if (ix->storage_mode == VDB_STORAGE_SQ8) {
    hipLaunchKernelGGL(sq8_quantize_rows, dim3(blocks), dim3(256), 0, ix->stream, ix->rows.as<float>(), ix->row_stride,
                       ix->sq8_codes.as<uint8_t>(), ix->sq8_stride, ix->sq8_min.as<float>(), ix->sq8_max.as<float>(),
                       (uint32_t)first, (uint32_t)n, ix->dim);
    hipLaunchKernelGGL(sq8_norms, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ix->stream, ix->sq8_codes.as<uint8_t>(),
                       ix->sq8_stride, ix->sq8_min.as<float>(), ix->sq8_max.as<float>(), ix->sq8_nsq.as<float>(),
                       (uint32_t)first, (uint32_t)n, ix->dim);
  } else if (ix->storage_mode == VDB_STORAGE_BINARY) {
    hipLaunchKernelGGL(sign_bits_rows, dim3(blocks), dim3(256), 0, ix->stream, ix->rows.as<float>(), ix->row_stride,
                       ix->sign_bits.as<uint32_t>(), ix->words, (uint32_t)first, (uint32_t)n, ix->dim);
  }
}

// grows the quantised arrays with the index and encodes rows [first, first + n)
int32_t storage_mode_append(vdb_hip_index* ix, uint64_t first, uint64_t n) {
  if (ix->storage_mode == VDB_STORAGE_FULL) return VDB_OK;
  const uint64_t cap = std::max<uint64_t>(ix->capacity, 1);
  hipError_t e = hipSuccess;
  if (ix->storage_mode == VDB_STORAGE_SQ8) {
    if ((e = ix->sq8_codes.reserve(cap * ix->sq8_stride, true, ix->stream)) != hipSuccess ||
        (e = ix->sq8_min.reserve(cap * 4, true, ix->stream)) != hipSuccess ||
        (e = ix->sq8_max.reserve(cap * 4, true, ix->stream)) != hipSuccess ||
        (e = ix->sq8_nsq.reserve(cap * 4, true, ix->stream)) != hipSuccess)
      return fail(VDB_ERR_OOM, std::string("SQ8 codes: ") + hipGetErrorString(e));
  } else {
    if ((e = ix->sign_bits.reserve(cap * ix->words * 4, true, ix->stream)) != hipSuccess)
      return fail(VDB_ERR_OOM, std::string("sign bits: ") + hipGetErrorString(e));
  }
  quantize_range(ix, first, n);
  if (first < ix->sq8_img_rows) ix->sq8_img_rows = first;  // codes rebuilt from `first`: the selection image follows at next use
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}

// selection stage (level 3): bf16 image of the dequantised rows, their norms, the f32 seed prefix — built at first use,
// kept current by storage_mode_append
static int32_t ensure_sq8_select_impl(vdb_hip_index* ix, hipStream_t st);
int32_t ensure_sq8_select(vdb_hip_index* cx, hipStream_t st) {  // (inside a search: see select_stage.hip build_image_on_primary)
  vdb_hip_index* p = primary_of(cx);
  {
    std::lock_guard<std::mutex> il(p->img_mu);
    const bool stale = p->sq8_img.cap == 0 || p->sq8_img_rows < p->n_rows;
    const int32_t rc = ensure_sq8_select_impl(p, st);
    if (rc != VDB_OK) return rc;
    if (stale) VDB_HIP(hipStreamSynchronize(st));  // complete before another context may take the image over (index.hip)
    if (cx != p) copy_image_fields(cx, p);
  }
  if (!cx->sel_stats) {
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocDefault) != hipSuccess) return fail(VDB_ERR_OOM, "pinned selection statistics");
    memset(h, 0, 64);
    cx->sel_stats = static_cast<volatile uint32_t*>(h);
  }
  return VDB_OK;
}
static int32_t ensure_sq8_select_impl(vdb_hip_index* ix, hipStream_t st) {
  const uint64_t cap = std::max<uint64_t>(ix->capacity, 1) + kRowSlack;
  const bool aug = ix->metric == VDB_EUCLIDEAN;
  const uint32_t dim_a = aug ? ix->dim + 64 : ix->dim, dim_s = aug ? ix->dim + 4 : ix->dim;
  hipError_t e;
  if ((e = ix->sq8_img.reserve(cap * (size_t)dim_a * 2, true, st)) != hipSuccess ||
      (e = ix->sq8_nrm.reserve(cap * 4, true, st)) != hipSuccess ||
      (e = ix->sq8_seed.reserve((size_t)kSplitSeedRows * dim_s * 4, true, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("SQ8 selection image: ") + hipGetErrorString(e));
  if (!ix->sq8_rho.p) {  // (the image is built from row 0 behind this: every row contributes)
    if ((e = ix->sq8_rho.reserve(256, false, st)) != hipSuccess || (e = hipMemsetAsync(ix->sq8_rho.p, 0, 256, st)) != hipSuccess)
      return fail(VDB_ERR_OOM, std::string("SQ8 residual bound: ") + hipGetErrorString(e));
    ix->sq8_img_rows = 0;
  }
  if (ix->sq8_img_rows < ix->n_rows) {
    const uint64_t first = ix->sq8_img_rows, n = ix->n_rows - first;
    const int blocks = (int)std::min<uint64_t>((n + 3) / 4, 4096);
    hipLaunchKernelGGL(sq8_dequant_rows, dim3(blocks), dim3(256), 0, st, ix->sq8_codes.as<uint8_t>(), ix->sq8_stride,
                       ix->sq8_min.as<float>(), ix->sq8_max.as<float>(), ix->sq8_nsq.as<float>(), ix->sq8_img.as<uint16_t>(),
                       ix->sq8_nrm.as<float>(), ix->sq8_seed.as<float>(), kSplitSeedRows, (uint32_t)first, (uint32_t)n, ix->dim, dim_a, dim_s,
                       ix->sq8_rho.as<uint32_t>());
    ix->sq8_img_rows = ix->n_rows;
    VDB_HIP(hipGetLastError());
  }
  return VDB_OK;
}

// the reference chain for the queries flagged by a selection batch, decided on the device: list them, one gathered launch of
// the exact SQ8 sweep (block rows without a listed query exit at once), merge per listed query, scatter.  qmap / fin.qcount / fin.qslot:
// the list the re-scoring launch made (sweep_split.hip list_unproven); fin: the batch's last launch, completed and issued here
int32_t sq8_fallback_flagged(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nqg, uint32_t k, const uint32_t* qmap,
                             const SelectFinishArgs& fin_in, hipStream_t st) {
  SelectFinishArgs fin = fin_in;
  const uint32_t* qcount = fin.qcount;
  const uint8_t* alive = ix->any_dead ? ix->alive.as<uint8_t>() : nullptr;
  int B = 8;
  if (sq8_lds_bytes(8, k, ix->dim) > 160 * 1024) B = 4;
  if (B == 4 && sq8_lds_bytes(4, k, ix->dim) > 160 * 1024) B = 1;
  const size_t lds = sq8_lds_bytes(B, k, ix->dim);
  if (lds > 160 * 1024) return fail(VDB_ERR_UNSUPPORTED, "SQ8 search: dim / k too large for the LDS query tile");
  const uint32_t ngroups = (uint32_t)((ix->n_rows + 63) / 64);
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds, 4));
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ngroups + 3) / 4, (int64_t)ix->n_cus * per_cu));
  const int qgroups = (int)((nqg + (uint32_t)B - 1) / (uint32_t)B);
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 15) & ~(size_t)15;
    return o;
  };
  const size_t o_keys = take((size_t)nqg * blocks * k * 8), o_ids = take((size_t)nqg * k * 8), o_sc = take((size_t)nqg * k * 4),
               o_n = take((size_t)nqg * 4);
  hipError_t e;
  if ((e = ix->s_fb_keys.reserve(off, false, st)) != hipSuccess) return fail(VDB_ERR_OOM, "SQ8 fallback scratch");
  unsigned char* sd = ix->s_fb_keys.as<unsigned char>();
  Sq8Args a{};
  a.codes = ix->sq8_codes.as<uint8_t>();
  a.vmin = ix->sq8_min.as<float>();
  a.vmax = ix->sq8_max.as<float>();
  a.nsq = ix->sq8_nsq.as<float>();
  a.alive = alive;
  a.queries = d_q;
  a.part_keys = reinterpret_cast<uint64_t*>(sd + o_keys);
  a.code_stride = ix->sq8_stride;
  a.q_stride = q_stride;
  a.n_rows = (uint32_t)ix->n_rows;
  a.dim = ix->dim;
  a.nq = (uint32_t)B;
  a.k = k;
  a.qmap = qmap;
  a.qcount = qcount;
  e = B == 8 ? launch_sq8_m<8>(ix->metric, a, blocks, lds, st, qgroups)
             : (B == 4 ? launch_sq8_m<4>(ix->metric, a, blocks, lds, st, qgroups) : launch_sq8_m<1>(ix->metric, a, blocks, lds, st, qgroups));
  if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("SQ8 fallback sweep launch: ") + hipGetErrorString(e));
  MergeArgs m{};
  m.part_keys = a.part_keys;
  m.ext_ids = ix->ext_ids.as<uint64_t>();
  m.out_ids = reinterpret_cast<uint64_t*>(sd + o_ids);
  m.out_scores = reinterpret_cast<float*>(sd + o_sc);
  m.out_n = reinterpret_cast<uint32_t*>(sd + o_n);
  m.n_lists = (uint32_t)blocks;
  m.k = k;
  m.active = qcount;
  launch_merge(ix->metric != VDB_EUCLIDEAN, m, nqg, st);
  fin.g_ids = m.out_ids;
  fin.g_scores = m.out_scores;
  fin.g_n = m.out_n;
  launch_select_finish(fin, st);
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}

int32_t brute_sq8_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k, uint64_t* d_ids,
                      float* d_scores, uint32_t* d_n, hipStream_t st) {
  if (ix->storage_mode != VDB_STORAGE_SQ8) return fail(VDB_ERR_STATE, "SQ8 search: set the storage mode to SQ8 first");
  if (ix->metric != VDB_COSINE && ix->metric != VDB_EUCLIDEAN && ix->metric != VDB_DOT)
    return fail(VDB_ERR_UNSUPPORTED, "SQ8 search: Cosine, Euclidean and DotProduct only");
  if (nq == 0) return VDB_OK;
  if (k == 0 || ix->n_rows == 0) {
    VDB_HIP(hipMemsetAsync(d_n, 0, (size_t)nq * 4, st));
    return VDB_OK;
  }
  const uint8_t* alive = ix->any_dead ? ix->alive.as<uint8_t>() : nullptr;
  for (uint32_t q0 = 0; q0 < nq;) {
    int B = nq - q0 >= 8 ? 8 : (nq - q0 >= 4 ? 4 : 1);
    if (B == 8 && sq8_lds_bytes(8, k, ix->dim) > 160 * 1024) B = 4;
    if (B == 4 && sq8_lds_bytes(4, k, ix->dim) > 160 * 1024) B = 1;
    const size_t lds = sq8_lds_bytes(B, k, ix->dim);
    if (lds > 160 * 1024) return fail(VDB_ERR_UNSUPPORTED, "SQ8 search: dim / k too large for the LDS query tile");
    const uint32_t tile = std::min<uint32_t>((uint32_t)B, nq - q0);
    const uint32_t ngroups = (uint32_t)((ix->n_rows + 63) / 64);
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds, 4));
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ngroups + 3) / 4, (int64_t)ix->n_cus * per_cu));
    hipError_t e;
    if ((e = ix->s_part_keys.reserve((size_t)B * blocks * k * 8, false, st)) != hipSuccess)
      return fail(VDB_ERR_OOM, "top-k scratch");
    Sq8Args a{};
    a.codes = ix->sq8_codes.as<uint8_t>();
    a.vmin = ix->sq8_min.as<float>();
    a.vmax = ix->sq8_max.as<float>();
    a.nsq = ix->sq8_nsq.as<float>();
    a.alive = alive;
    a.queries = d_q + (size_t)q0 * q_stride;
    a.part_keys = ix->s_part_keys.as<uint64_t>();
    a.code_stride = ix->sq8_stride;
    a.q_stride = q_stride;
    a.n_rows = (uint32_t)ix->n_rows;
    a.dim = ix->dim;
    a.nq = tile;
    a.k = k;
    EventPair* ev = next_events(ix);
    if (ev) (void)hipEventRecord(ev->a, st);
    e = B == 8 ? launch_sq8_m<8>(ix->metric, a, blocks, lds, st)
               : (B == 4 ? launch_sq8_m<4>(ix->metric, a, blocks, lds, st) : launch_sq8_m<1>(ix->metric, a, blocks, lds, st));
    if (ev) (void)hipEventRecord(ev->b, st);
    if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("SQ8 sweep launch: ") + hipGetErrorString(e));
    MergeArgs m{};
    m.part_keys = a.part_keys;
    m.ext_ids = ix->ext_ids.as<uint64_t>();
    m.out_ids = d_ids + (size_t)q0 * k;
    m.out_scores = d_scores + (size_t)q0 * k;
    m.out_n = d_n + q0;
    m.n_lists = (uint32_t)blocks;
    m.k = k;
    launch_merge(ix->metric != VDB_EUCLIDEAN, m, tile, st);
    q0 += tile;
  }
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}

// BinaryQuantizedVector::hamming_distance between the sign bits of the queries and of every row: the packed-bit
// sweep of sweep.hip over the sign-bit array (scores = distance as f32, smallest first)
// the four-bit image of the sign-bit codes (bits_gemm.hip: batches of >= 32 queries run on the matrix cores), in the storage mode's
// image buffers (vdb_index.hpp); built at first use, extended lazily behind inserts — inside a search: see index.hip
// build_image_on_primary for why a build is complete before the image is handed on
static int32_t ensure_sign_image(vdb_hip_index* cx, hipStream_t st) {
  vdb_hip_index* p = primary_of(cx);
  std::lock_guard<std::mutex> il(p->img_mu);
  const bool stale = p->sq8_img.cap == 0 || p->sq8_img_rows < p->n_rows;
  const uint64_t cap = std::max<uint64_t>(p->capacity, 1) + kRowSlack;
  const uint32_t stride = bits_image_stride(p->dim);
  hipError_t e;
  if ((e = p->sq8_img.reserve(cap * (size_t)stride, true, st)) != hipSuccess || (e = p->sq8_nrm.reserve(cap * 4, true, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("sign-bit image: ") + hipGetErrorString(e));
  if (p->sq8_img_rows < p->n_rows) {
    launch_bits_expand(VDB_HAMMING, p->sign_bits.as<uint32_t>(), p->words, p->sq8_img.as<uint8_t>(), stride, p->sq8_nrm.as<float>(), (uint32_t)p->sq8_img_rows,
                       (uint32_t)(p->n_rows - p->sq8_img_rows), p->dim, 0.0f, st);
    p->sq8_img_rows = p->n_rows;
    VDB_HIP(hipGetLastError());
  }
  if (stale) VDB_HIP(hipStreamSynchronize(st));
  if (cx != p) copy_image_fields(cx, p);
  return VDB_OK;
}

int32_t brute_binary_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k, uint64_t* d_ids,
                         float* d_scores, uint32_t* d_n, hipStream_t st) {
  if (ix->storage_mode != VDB_STORAGE_BINARY) return fail(VDB_ERR_STATE, "Binary search: set the storage mode to Binary first");
  if (nq == 0) return VDB_OK;
  if (k == 0 || ix->n_rows == 0) {
    VDB_HIP(hipMemsetAsync(d_n, 0, (size_t)nq * 4, st));
    return VDB_OK;
  }
  if ((size_t)4 * k * 8 + (size_t)ix->words * 4 + 32 > 60 * 1024)
    return fail(VDB_ERR_UNSUPPORTED, "k too large for the fused top-k path");
  hipError_t e;
  // one or two queries: ONE launch — the blocks take the query's sign bits themselves, the one that finishes last merges (sweep.hip)
  if (opt_bits_fused(ix) && sweep_bits_fused_supported(ix->words, nq, k)) {
    const int fblocks = sweep_bits_fused_blocks(ix->n_rows, ix->n_cus);
    if (ix->s_tickets.cap == 0) {
      if ((e = ix->s_tickets.reserve(16, false, st)) != hipSuccess) return fail(VDB_ERR_OOM, "ticket scratch");
      VDB_HIP(hipMemsetAsync(ix->s_tickets.p, 0, 16, st));
    }
    if ((e = ix->s_part_keys.reserve((size_t)nq * fblocks * k * 8, false, st)) != hipSuccess) return fail(VDB_ERR_OOM, "top-k scratch");
    BitsFusedArgs fa{};
    fa.bits = ix->sign_bits.as<uint32_t>();
    fa.q = d_q;
    fa.q_stride = q_stride;
    fa.alive = ix->any_dead ? ix->alive.as<uint8_t>() : nullptr;
    fa.part_keys = ix->s_part_keys.as<uint64_t>();
    fa.tickets = ix->s_tickets.as<uint32_t>();
    fa.n_rows = (uint32_t)ix->n_rows;
    fa.words = ix->words;
    fa.dim = ix->dim;
    fa.k = k;
    fa.sign_rule = 1u;
    fa.m.ext_ids = ix->ext_ids.as<uint64_t>();
    fa.m.out_ids = d_ids;
    fa.m.out_scores = d_scores;
    fa.m.out_n = d_n;
    EventPair* evf = next_events(ix);
    if (evf) (void)hipEventRecord(evf->a, st);
    if ((e = launch_sweep_bits_fused(VDB_HAMMING, fa, fblocks, nq, st)) != hipSuccess)
      return fail(VDB_ERR_HIP, std::string("one-launch sign-bit search: ") + hipGetErrorString(e));
    if (evf) (void)hipEventRecord(evf->b, st);
    return VDB_OK;
  }
  e = ix->s_qbits.reserve((size_t)nq * ix->words * 4, false, st);
  if (e != hipSuccess) return fail(VDB_ERR_OOM, "qbits scratch");
  hipLaunchKernelGGL(sign_bits_rows, dim3((unsigned)std::min<uint32_t>((nq + 3) / 4, 4096)), dim3(256), 0, st, d_q, q_stride,
                     ix->s_qbits.as<uint32_t>(), ix->words, 0u, nq, ix->dim);
  // large batches: Hamming between sign-bit codes as a four-bit GEMM distance on the matrix cores (bits_gemm.hip), exact; the rest of
  // the batch and every other shape on the vector ALUs — the same keys either way
  uint32_t qdone = 0;
  while (opt_value(ix, VDB_OPT_SWEEP_ENGINE) == 1 && opt_value(ix, VDB_OPT_MAX_QUERY_TILE) >= 128) {
    const uint32_t nqg = bits_gemm_chunk(ix, nq - qdone, k);
    if (!nqg) break;
    int32_t rg = ensure_sign_image(ix, st);
    if (rg == VDB_OK)
      rg = brute_bits_gemm_dev(ix, VDB_HAMMING, ix->sq8_img.as<uint8_t>(), ix->sq8_nrm.as<float>(), ix->s_qbits.as<uint32_t>() + (size_t)qdone * ix->words, nqg,
                               k, d_ids + (size_t)qdone * k, d_scores + (size_t)qdone * k, d_n + qdone, st);
    if (rg != VDB_OK) return rg;
    qdone += nqg;
  }
  if (qdone == nq) return VDB_OK;
  d_ids += (size_t)qdone * k;
  d_scores += (size_t)qdone * k;
  d_n += qdone;
  const uint32_t* qbits_rest = ix->s_qbits.as<uint32_t>() + (size_t)qdone * ix->words;
  nq -= qdone;
  const BitsPlan bp = plan_bits_sweep(ix->n_rows, ix->n_cus, ix->words, nq, k);  // batches: 8 / 32 queries per corpus pass
  const int blocks = bp.blocks;
  if ((e = ix->s_part_keys.reserve((size_t)nq * blocks * k * 8, false, st)) != hipSuccess ||
      (e = ix->s_part_cnt.reserve((size_t)nq * blocks * 4, false, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, "top-k scratch");
  BitsArgs ba{};
  ba.bits = ix->sign_bits.as<uint32_t>();
  ba.qbits = qbits_rest;
  ba.alive = ix->any_dead ? ix->alive.as<uint8_t>() : nullptr;
  ba.part_keys = ix->s_part_keys.as<uint64_t>();
  ba.part_cnt = ix->s_part_cnt.as<uint32_t>();
  ba.n_rows = (uint32_t)ix->n_rows;
  ba.words = ix->words;
  ba.k = k;
  EventPair* ev = next_events(ix);
  if (ev) (void)hipEventRecord(ev->a, st);
  if ((e = launch_bits_plan(VDB_HAMMING, bp, ba, nq, st)) != hipSuccess)
    return fail(VDB_ERR_HIP, std::string("sign-bit sweep launch: ") + hipGetErrorString(e));
  if (ev) (void)hipEventRecord(ev->b, st);
  MergeArgs m{};
  m.part_keys = ba.part_keys;
  m.part_cnt = ba.part_cnt;
  m.ext_ids = ix->ext_ids.as<uint64_t>();
  m.out_ids = d_ids;
  m.out_scores = d_scores;
  m.out_n = d_n;
  m.n_lists = (uint32_t)blocks;
  m.k = k;
  launch_merge(false, m, nq, st);
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}

}  // namespace vdb

using namespace vdb;

extern "C" {

// StorageMode of a collection (quantization.rs:17-29; crud.rs:66-82 encodes every upserted vector accordingly)
int32_t vdb_hip_index_set_storage_mode(vdb_hip_index* ix, int32_t mode) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (mode != VDB_STORAGE_FULL && mode != VDB_STORAGE_SQ8 && mode != VDB_STORAGE_BINARY)
    return fail(VDB_ERR_INVALID_ARG, "bad storage mode");
  if (ix->group) return group_for_all(ix, 2, (uint32_t)mode);
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  if (ix->storage_mode == mode) return VDB_OK;
  VDB_ENTER(ix);
  for (DevBuf* b : {&ix->sq8_codes, &ix->sq8_min, &ix->sq8_max, &ix->sq8_nsq, &ix->sign_bits, &ix->sq8_img, &ix->sq8_nrm, &ix->sq8_seed})
    b->release();
  ix->sq8_img_rows = 0;
  ix->storage_mode = mode;
  ix->sq8_stride = ((uint64_t)ix->dim + 15) / 16 * 16;
  int32_t rc = storage_mode_append(ix, 0, ix->n_rows);
  if (rc != VDB_OK) {
    ix->storage_mode = VDB_STORAGE_FULL;
    return rc;
  }
  VDB_HIP(hipStreamSynchronize(ix->stream));
  return VDB_OK;
  });
}

// The stored code of one vector in the reference's byte format: QuantizedVector::to_bytes (min f32, max f32, dim
// bytes; quantization.rs:289-295) or BinaryQuantizedVector::to_bytes (dimension u32, ceil(dim/8) bytes; :155-169).
int32_t vdb_hip_index_get_quantized(vdb_hip_index* ix, uint64_t id, uint8_t* out, size_t cap, size_t* len) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !len) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) {  // every shard that holds the id holds the same code
    for (size_t s = 0; s < group_size(ix); s++) {
      int32_t rc = vdb_hip_index_get_quantized(group_shard(ix, s), id, out, cap, len);
      if (rc == VDB_OK || s + 1 == group_size(ix)) return rc;
    }
  }
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  if (ix->storage_mode == VDB_STORAGE_FULL) return fail(VDB_ERR_STATE, "storage mode is Full: nothing is quantised");
  auto it = ix->id_to_idx.find(id);
  if (it == ix->id_to_idx.end()) return fail(VDB_ERR_INVALID_ARG, "unknown id");
  const uint64_t row = it->second;
  const size_t need = ix->storage_mode == VDB_STORAGE_SQ8 ? 8 + (size_t)ix->dim : 4 + ((size_t)ix->dim + 7) / 8;
  *len = need;
  if (!out || cap < need) return fail(VDB_ERR_INVALID_ARG, "buffer too small");
  VDB_ENTER(ix);
  if (ix->storage_mode == VDB_STORAGE_SQ8) {
    VDB_HIP(hipMemcpyAsync(out, ix->sq8_min.as<float>() + row, 4, hipMemcpyDeviceToHost, ix->stream));
    VDB_HIP(hipMemcpyAsync(out + 4, ix->sq8_max.as<float>() + row, 4, hipMemcpyDeviceToHost, ix->stream));
    VDB_HIP(hipMemcpyAsync(out + 8, ix->sq8_codes.as<uint8_t>() + row * ix->sq8_stride, ix->dim, hipMemcpyDeviceToHost, ix->stream));
  } else {
    const uint32_t d = ix->dim;
    std::memcpy(out, &d, 4);
    VDB_HIP(hipMemcpyAsync(out + 4, ix->sign_bits.as<uint32_t>() + row * ix->words, need - 4, hipMemcpyDeviceToHost, ix->stream));
  }
  VDB_HIP(hipStreamSynchronize(ix->stream));
  return VDB_OK;
  });
}

}  // extern "C"
