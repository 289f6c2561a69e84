// bits_gemm.hip — Hamming / Jaccard batches on the matrix cores, bit-exact.
//
// The reference's Hamming and Jaccard on f32 vectors are functions of bit counts (simd_explicit.rs:234-287,372-443 on the exact
// re-encoding bit = x > 0.5, SURVEY 8a note 10):  jaccard = |q & v| / (|q| + |v| - |q & v|) (1.0 for an empty union), and
// hamming = |q ^ v| = (dim - q' . v') / 2 for the +-1 re-encoding q' = 2 q - 1 (the XNOR form).  Both are dot products of the
// values 0 and +-1 — which the matrix cores' four-bit format holds exactly (E2M1: +1.0 = 0b0010, -1.0 = 0b1010) and multiplies
// at four times the bf16 rate: v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales, f32 accumulators = exact integers.  A
// batch therefore runs as a four-bit GEMM distance: rows and queries as nibble images (dim / 2 bytes per row, padded with zeros
// to whole 128-byte k-tiles of 256 values; Hamming: +-1, Jaccard: {0,1} with |v| and |q| where the cosine kernel keeps its norms)
// through the four-bit instance of the ping-pong selection kernel (sweep_gemm_bf16.hip, sweep_topk_gemm_bf16_pp<METRIC, true>)
// with the metric's own per-element bound and finish in its epilogue.  Scores are exact integers (ratios of exact integers), so the
// kernel's keys ARE the result: no re-scoring, no proof, the same keys the vector-ALU kernels of sweep.hip produce
// ((score total order, row) — ties by internal row).
// Bound: the four-bit matrix pipe (~10 PFLOP/s dense); algorithmic operations = 2 * rows * dim * queries.  The vector-ALU batch
// kernel (sweep_topk_bits_tile<B = 32>, AND + popcount) takes 2.8 / 3.2 ms per 1 024 queries at 1 M x 768 (vector-ALU issue-bound);
// this path 0.72 / 0.82 ms (the first version, on v_mfma_i32_16x16x64_i8 over byte images: 0.96 / 1.02 ms).
#include <algorithm>
#include <cstdlib>
#include <string>

#include "vdb_probe_env.hpp"
#include "vdb_device.hpp"
#include "vdb_index.hpp"
#include "vdb_kernels.hpp"

namespace vdb {

// one thread = one 32-bit word of a packed row -> 32 nibbles = 16 bytes; the thread of a row's first word also leaves the row's bit
// count (or `fill`).  pm = false: values {0, +1}; pm = true: {-1, +1} for the dim real columns, 0 for the padding (it must not
// contribute).  E2M1: +1.0 = 0b0010, -1.0 = 0b1010.  Value i of a word sits in nibble i of the 16 bytes — any order would do: rows
// and queries are packed alike and the kernels read both with the same fragment mapping.
__global__ __launch_bounds__(256) void bits_expand_kernel(const uint32_t* __restrict__ bits, uint32_t words, uint8_t* __restrict__ img,
                                                          uint32_t img_stride, float* __restrict__ cnt, uint32_t row0, uint32_t n_rows,
                                                          uint32_t dim, bool pm, float fill) {
  const uint32_t chunks = img_stride / 16u;
  const uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  if (gid >= (uint64_t)n_rows * chunks) return;
  const uint32_t r = (uint32_t)(gid / chunks), c = (uint32_t)(gid % chunks);
  const uint32_t* w = bits + (size_t)(row0 + r) * words;
  const uint32_t h = c < words ? w[c] : 0u;
  uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int i = 0; i < 32; i++) {
    const uint32_t col = c * 32u + (uint32_t)i;
    const uint32_t nib = ((h >> i) & 1u) ? 0x2u : ((pm && col < dim) ? 0xAu : 0u);
    o[i >> 3] |= nib << (4 * (i & 7));
  }
  *reinterpret_cast<uint4*>(img + (size_t)(row0 + r) * img_stride + (size_t)c * 16u) = make_uint4(o[0], o[1], o[2], o[3]);
  if (c == 0 && cnt) {
    uint32_t s = 0;
    for (uint32_t i = 0; i < words; i++) s += __popc(w[i]);
    cnt[row0 + r] = pm ? fill : (float)s;
  }
}
// metric VDB_HAMMING: the +-1 image (q . v = dim - 2 |q ^ v|: no counts needed, cnt[] = fill); VDB_JACCARD: {0,1} + bit counts
void launch_bits_expand(int metric, const uint32_t* bits, uint32_t words, uint8_t* img, uint32_t img_stride, float* cnt, uint32_t row0,
                        uint32_t n_rows, uint32_t dim, float fill, hipStream_t st) {
  if (!n_rows) return;
  const uint64_t threads = (uint64_t)n_rows * (img_stride / 16u);
  hipLaunchKernelGGL(bits_expand_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, bits, words, img, img_stride, cnt, row0, n_rows,
                     dim, metric == VDB_HAMMING, fill);
}

// ---- the seed: a plain four-bit GEMM of the first rows x the batch, straight from L2 (as seed_scores_bf16, sweep_split.hip) ----
// The selection kernel needs a bound per query before its first row tile, or that tile's 65 536 elements all queue up.  The seed
// is a SAMPLE: the k-th best score over ANY rows bounds the k-th best over all of them, so the kernel keeps the best key of
// every 16 rows ([nq][seed_rows / 16] keys), merge_topk_select picks the k best, their k-th + 1 is the bound — and the launches
// sweep the seed rows again (0.4 % of a 1 M corpus).  (The vector-ALU tile kernel over 16 384 rows did this first: 190 us of a
// 1.0 ms batch, nearly all of it per-block start-up.)  One wave = 64 rows x 64 queries, 16-byte fragments from global memory.
typedef int i32x4_s __attribute__((ext_vector_type(4)));
typedef int i32x8_s __attribute__((ext_vector_type(8)));
typedef float f32x4_q __attribute__((ext_vector_type(4)));
template <int METRIC>
__global__ __launch_bounds__(256) void seed_scores_fp4(const uint8_t* rows8, uint32_t stride, const float* cnt, const uint8_t* alive,
                                                      const uint8_t* q8, const float* qcnt, uint64_t* keys, uint32_t seed_rows, uint32_t nq) {
  constexpr bool HIB = METRIC != kHamming;
  const uint32_t lane = threadIdx.x & 63u, wib = threadIdx.x >> 6;
  const uint32_t row0 = (blockIdx.x * 4u + wib) * 64u;
  const uint32_t qb = blockIdx.y * 64u;
  if (row0 >= seed_rows) return;
  const uint32_t i = lane & 15u, kk = lane >> 4;
  const uint8_t* ap[4];
#pragma unroll
  for (int rb = 0; rb < 4; rb++) ap[rb] = rows8 + (size_t)min(row0 + (uint32_t)rb * 16u + i, seed_rows - 1u) * stride + kk * 16u;
  const uint8_t* bp[4];
#pragma unroll
  for (int t = 0; t < 4; t++) bp[t] = q8 + (size_t)min(qb + (uint32_t)t * 16u + i, nq - 1u) * stride + kk * 16u;
  f32x4_q acc[4][4];
#pragma unroll
  for (int rb = 0; rb < 4; rb++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[rb][t] = f32x4_q{0.f, 0.f, 0.f, 0.f};
  for (uint32_t k0 = 0; k0 < stride; k0 += 128) {  // stride % 128 == 0; two steps of 128 values (64 bytes per row) in flight
    i32x4_s av[4][2], bv[2][4];
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
      for (int rb = 0; rb < 4; rb++) av[rb][s] = *reinterpret_cast<const i32x4_s*>(ap[rb] + k0 + s * 64);
#pragma unroll
      for (int t = 0; t < 4; t++) bv[s][t] = *reinterpret_cast<const i32x4_s*>(bp[t] + k0 + s * 64);
    }
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int rb = 0; rb < 4; rb++)
#pragma unroll
        for (int t = 0; t < 4; t++) {  // (four-bit operands use the first four of the builtin's eight registers; scales 2^0)
          const i32x8_s a8 = {av[rb][s][0], av[rb][s][1], av[rb][s][2], av[rb][s][3], 0, 0, 0, 0};
          const i32x8_s b8 = {bv[s][t][0], bv[s][t][1], bv[s][t][2], bv[s][t][3], 0, 0, 0, 0};
          acc[rb][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[rb][t], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
  }
  const uint32_t ngrp = (seed_rows + 15u) / 16u;
#pragma unroll
  for (int t = 0; t < 4; t++) {
    const uint32_t q = qb + (uint32_t)t * 16u + i;
    if (q >= nq) continue;
    const float qn = qcnt[q];
    uint64_t best = kKeyInvalid;
#pragma unroll
    for (int rb = 0; rb < 4; rb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint32_t row = row0 + (uint32_t)rb * 16u + 4u * kk + (uint32_t)r;
        if (row >= seed_rows) continue;
        const float x = acc[rb][t][r];
        float sc;
        if (METRIC == kHamming) {
          sc = 0.5f * (qn - x);  // (the finish of the selection kernel: g16_protocol.inc)
        } else {
          const float uni = qn + cnt[row] - x;
          sc = uni == 0.0f ? 1.0f : x / uni;
        }
        const bool live = !alive || alive[row] != 0;
        const uint64_t key = live ? make_key<HIB>(sc, row) : kKeyInvalid;
        best = key < best ? key : best;
      }
    keys[(size_t)q * ngrp + (row0 / 64u) * 4u + kk] = best;
  }
}
static void launch_seed_scores_fp4(int metric, const uint8_t* rows8, uint32_t stride, const float* cnt, const uint8_t* alive, const uint8_t* q8,
                                  const float* qcnt, uint64_t* keys, uint32_t seed_rows, uint32_t nq, hipStream_t st) {
  const dim3 grid((seed_rows + 255) / 256, (nq + 63) / 64);
  if (metric == VDB_HAMMING)
    hipLaunchKernelGGL((seed_scores_fp4<kHamming>), grid, dim3(256), 0, st, rows8, stride, cnt, alive, q8, qcnt, keys, seed_rows, nq);
  else
    hipLaunchKernelGGL((seed_scores_fp4<kJaccard>), grid, dim3(256), 0, st, rows8, stride, cnt, alive, q8, qcnt, keys, seed_rows, nq);
}
constexpr uint32_t kBitsSeedRows = 4096;
// smallest batch that takes the matrix cores (VELESDB_BITS_GEMM_MIN_QUERIES overrides: probes).  1 M x 768 (profiles/
// r04r_bits_gemm_min_queries.log): 16 queries 0.21 / 0.17 ms on the vector ALUs against 0.28 / 0.29 ms here (one partly filled
// query tile costs what a full one costs), 32 queries 0.32 / 0.37 against 0.29 / 0.30, 64 queries 0.52 / 0.56 against 0.30 / 0.32
constexpr uint32_t kBitsGemmMinQueries = 32;

uint32_t bits_image_stride(uint32_t dim) { return std::max<uint32_t>(256u, (dim + 255u) / 256u * 128u); }  // bytes (two values each); >= 2 k-tiles

// does a chunk of the batch take the matrix cores?
uint32_t bits_gemm_chunk(const vdb_hip_index* ix, uint32_t nq_left, uint32_t k) {
  if (k == 0 || k > kGemmBf16MaxK || ix->n_rows < kGemmBf16MinRows || ix->n_rows >= 0xFFFFFF00ull) return 0;
  // up to 1 024 queries, whatever that leaves of the last 256-query tile: a partly filled tile costs what a full one costs, a second
  // pass costs the whole fixed part again (the selection stage's rule, select_stage.hip select_chunk)
  static const uint32_t min_q = [] {
    const char* e = probe_env("VELESDB_BITS_GEMM_MIN_QUERIES");
    return e ? (uint32_t)atoi(e) : kBitsGemmMinQueries;
  }();
  const uint32_t nqg = std::min<uint32_t>(nq_left, kGemmMaxQueries);
  return nqg >= min_q ? nqg : 0;
}

// nqg packed queries (qbits [nqg][words]) against the index's four-bit image: exact top-k per query into d_ids / d_scores / d_n
// (metric, img, cnt: the index's own bit metric over bits_img / bits_cnt, or VDB_HAMMING over the sign-bit codes' image — the Binary
// storage mode, storage_modes.hip)
int32_t brute_bits_gemm_dev(vdb_hip_index* ix, int metric, const uint8_t* img, const float* cnt, const uint32_t* qbits, uint32_t nqg, uint32_t k,
                            uint64_t* d_ids, float* d_scores, uint32_t* d_n, hipStream_t st) {
  const bool hib = metric == VDB_JACCARD;
  const uint8_t* alive = ix->any_dead ? ix->alive.as<uint8_t>() : nullptr;
  const uint32_t n = (uint32_t)ix->n_rows, stride = bits_image_stride(ix->dim), dim2 = stride / 2;  // the kernel's unit: two bytes
  // Launch schedule: the sample seed over the first rows (bounds only), then the four-bit GEMM in launches of growing size (gemm_schedule,
  // vdb_kernels.hpp); every launch starts from the k-th best key over all rows before it
  const uint32_t R0 = kBitsSeedRows;
  GemmSchedule sch;
  {
    const uint32_t head[3] = {1u, 4u, 16u};  // (the bound of a 4 096-row sample is weak: the first launches stay small)
    gemm_schedule(nqg, 0, n, ix->n_cus, head, 1u << 21, &sch);
  }
  const uint32_t lists = sch.lists;
  const uint32_t seed_lists = R0 / 16u;  // one key per 16 seed rows
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 15) & ~(size_t)15;
    return o;
  };
  const size_t o_seedp = take((size_t)nqg * seed_lists * 8), o_ids = take((size_t)nqg * k * 8), o_sc = take((size_t)nqg * k * 4),
               o_n = take((size_t)nqg * 4), o_tau = take((size_t)nqg * 8), o_qc = take((size_t)nqg * 4);
  hipError_t e;
  if ((e = ix->s_seed.reserve(off, false, st)) != hipSuccess ||
      (e = ix->s_part_keys.reserve((size_t)nqg * lists * k * 8, false, st)) != hipSuccess ||
      (e = ix->s_misc.reserve(((size_t)nqg + 256) * stride, false, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("bit-metric GEMM scratch: ") + hipGetErrorString(e));
  unsigned char* sd = ix->s_seed.as<unsigned char>();
  uint64_t* parts = ix->s_part_keys.as<uint64_t>();
  uint64_t* tau0 = reinterpret_cast<uint64_t*>(sd + o_tau);
  float* qcnt = reinterpret_cast<float*>(sd + o_qc);
  uint8_t* qimg = ix->s_misc.as<uint8_t>();
  launch_bits_expand(metric, qbits, ix->words, qimg, stride, qcnt, 0, nqg, ix->dim, (float)ix->dim, st);  // (Hamming: qn[b] = dim)
  // the kernel stages whole 256-query tiles: zero rows behind the batch
  if (nqg % 256u) VDB_HIP(hipMemsetAsync(qimg + (size_t)nqg * stride, 0, (size_t)256 * stride, st));
  // (no fill of the partial lists: every block writes all k slots of every query of its tile, and a merge reads only lists written)
  ix->last_kernels |= VDB_KERNEL_BITS | VDB_KERNEL_BITS_GEMM;
  EventPair* evg = next_events(ix);
  if (evg) (void)hipEventRecord(evg->a, st);
  MergeArgs ms{};
  ms.out_ids = reinterpret_cast<uint64_t*>(sd + o_ids);
  ms.out_scores = reinterpret_cast<float*>(sd + o_sc);
  ms.out_n = reinterpret_cast<uint32_t*>(sd + o_n);
  ms.ext_ids = nullptr;  // internal rows
  {
    const uint32_t seed_rows = std::min(R0, n);
    uint64_t* skeys = reinterpret_cast<uint64_t*>(sd + o_seedp);
    launch_seed_scores_fp4(metric, img, stride, cnt, alive, qimg, qcnt, skeys, seed_rows, nqg, st);
    ms.part_keys = skeys;
    ms.n_lists = (seed_rows + 15u) / 16u;
    ms.k = 1;
    ms.k_out = k;
    launch_merge(hib, ms, nqg, st);
    launch_seed_tau(ms.out_ids, ms.out_scores, ms.out_n, tau0, nullptr, 0, nqg, k, st, hib);
    ms.k_out = 0;
  }
  ms.k = k;
  e = run_gemm_schedule(
      sch, metric, reinterpret_cast<const uint16_t*>(img), dim2, cnt, alive, reinterpret_cast<const uint16_t*>(qimg), dim2,
      tau0, parts, lists, /*list_first=*/0, dim2, nqg, k, st, /*split=*/false, nullptr, nullptr, qcnt, [](int) {},
      [&](int, uint32_t list_off, bool last) {
        if (last) return;  // bound of the next launch: k-th best key over everything swept so far
        ms.part_keys = parts;
        ms.n_lists = list_off;   // the lists written so far ...
        ms.list_stride = lists;  // ... of `lists` per query
        launch_merge(hib, ms, nqg, st);
        launch_seed_tau(ms.out_ids, ms.out_scores, ms.out_n, tau0, nullptr, 0, nqg, k, st, hib);
      });
  if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("bit-metric GEMM launch: ") + hipGetErrorString(e));
  if (evg) (void)hipEventRecord(evg->b, st);
  MergeArgs mg{};
  mg.part_keys = parts;
  mg.ext_ids = ix->ext_ids.as<uint64_t>();
  mg.out_ids = d_ids;
  mg.out_scores = d_scores;
  mg.out_n = d_n;
  mg.n_lists = lists;
  mg.k = k;
  launch_merge(hib, mg, nqg, st);
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}

}  // namespace vdb
