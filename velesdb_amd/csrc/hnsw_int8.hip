// hnsw_int8.hip — dual-precision search: DualPrecisionHnsw::search_with_config(use_int8_traversal)
// (native/dual_precision.rs:223-441) with the per-dimension scalar quantiser of native/quantization.rs:191-260.
//
//   * ScalarQuantizer::train  — per-dimension min / max over the first min(1000, n) rows, scale = 255 / range
//     (1.0 for a constant dimension), kernel sq_train.
//   * quantize               — code = clamp(round((x - min) * scale), 0, 255), f32::round (half away from zero),
//     kernel sq_quantize_rows; it also stores sum(code^2) per row.
//   * traversal              — the graph walk of hnsw_kernels.hip, but every distance is the INTEGER L2^2 between
//     the u8 codes (quantization.rs:42-91): a row costs 768 B instead of 3 072 B, the path is bound by exactly these
//     gathers.  d = qsq + rsq[row] - 2 * dot(q, row) with v_dot4_u32_u8: exact integers, so ids, ranks and counters
//     are bit-identical to the oracle's restatement of the reference.
//   * re-rank                — the k * oversampling best candidates are re-scored with the exact f32 engine distance
//     (canonical arithmetic), stable-sorted (total_cmp), cut to k (dual_precision.rs:267-281).
// Algorithmic HBM bytes per query: n_dist * (dim + 4) + n_expand * M0 * 4 + k * oversampling * dim * 4.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "vdb_probe_env.hpp"
#include "vdb_hnsw_device.hpp"
#include "vdb_index.hpp"

namespace vdb {

// VELESDB_I8_WAVES2=0 keeps four-wave blocks for every batch (A / B measurements)
static std::atomic<bool> g_i8_waves2{[] {
  const char* e = probe_env("VELESDB_I8_WAVES2");
  return !(e && e[0] == '0');
}()};

namespace {

enum QPhase : int { Q_START = 0, Q_G_ENTRY, Q_G_LOAD, Q_G_SCAN, Q_G_DONE, Q_Z_ENTRY, Q_Z_POP, Q_Z_ADMIT, Q_FINISH, Q_R_DONE };

// one thread per dimension: min / max over the first n_train rows (quantization.rs:199-213)
__global__ void sq_train(const float* rows, uint64_t row_stride, uint32_t n_train, uint32_t dim, float* min_vals,
                         float* scales) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dim) return;
  float mn = 3.402823466e+38f, mx = -3.402823466e+38f;  // f32::MAX / f32::MIN
  for (uint32_t r = 0; r < n_train; r++) {
    const float v = rows[(size_t)r * row_stride + i];
    mn = fminf(mn, v);  // f32::min / max: the non-NaN operand wins
    mx = fmaxf(mx, v);
  }
  const float range = mx - mn;
  min_vals[i] = mn;
  scales[i] = fabsf(range) < 1e-10f ? 1.0f : 255.0f / range;  // quantization.rs:215-226
}

// one wave per row: codes (quantization.rs:236-252) + sum of squared codes
__global__ __launch_bounds__(256) void sq_quantize_rows(const float* rows, uint64_t row_stride, const float* min_vals,
                                                        const float* scales, uint32_t* codes, uint32_t code_words,
                                                        uint32_t* rsq, uint32_t row0, uint32_t n_rows, uint32_t dim) {
  const int lane = lane_id();
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t r = wave; r < n_rows; r += nwaves) {
    const uint32_t row = row0 + r;
    const float* p = rows + (size_t)row * row_stride;
    uint32_t sq = 0;
    for (uint32_t w = lane; w < code_words; w += 64) {
      uint32_t word = 0;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint32_t i = w * 4 + e;
        if (i < dim) {
          float q = roundf((p[i] - min_vals[i]) * scales[i]);
          q = q < 0.0f ? 0.0f : (q > 255.0f ? 255.0f : q);
          const uint32_t c = (q != q) ? 0u : (uint32_t)q;  // NaN `as u8` = 0
          word |= c << (8 * e);
          sq += c * c;
        }
      }
      codes[(size_t)row * code_words + w] = word;
    }
    sq = wave_sum_u32(sq);
    if (lane == 0 && rsq) rsq[row] = sq;
  }
}

}  // namespace

struct HnswInt8Args {
  HnswSearchArgs s;        // graph, f32 rows (re-rank), outputs, scratch; s.ef = max(ef_search, k * oversampling)
  CodeCtx codes;
  const float* min_vals;   // [dim]
  const float* scales;     // [dim]
  uint32_t cand_k;         // k * oversampling
};

// LDS: keys[cap] u64 | nb_id[nbmax] | nb_d[nbmax] | ctl[4] | flags[cap] (pad 16) | f32 query scratch (generic dims)
//      | qcode words [code_words] | re-rank output (node, dist) [nbmax] u64
// WAVES: 4 (a 256-thread block per query in flight) or 2: the traversal reads 772 bytes per visited node instead of 3 KB, so
// a step's distance phase is short and the leader's serial part (pop, visited test-and-set, admission) dominates — at 142
// registers a CU holds 12 waves: 3 queries in flight with four-wave blocks, 6 with two-wave blocks.
// Round 4: __launch_bounds__(.., 4) = at most 128 registers (12 dwords of scratch): 16 waves per CU = 8 two-wave walks.  The walk is
// bound by its chain of dependent memory round trips with too few of them in flight, not by bytes or by the vector ALUs:
// 1 M x 768, ef 128, 8 192 queries on one box 25.1 ms at 3 waves per SIMD, 19.7 at 4, 19.2 at 5 (96 registers; needs scratch for
// 10 walks per CU), 24.2 at 6 (80 registers: the spills cost more than the walks gain) — profiles/r04l_int8_occupancy_ab.log.
template <int METRIC, int CPL, int NS, int WAVES, bool VIS = false>
__global__ __launch_bounds__(WAVES * 64, 4) void hnsw_search_int8_kernel(HnswInt8Args A) {
  const HnswSearchArgs& a = A.s;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = (int)rfl(threadIdx.x >> 6);
  const uint32_t cap = a.cap, nbmax = a.nbmax, ef = a.ef;
  lds_vu64* keys = (lds_vu64*)(lds_void_p)(smem);
  lds_vu32* nb_id = (lds_vu32*)(lds_void_p)(smem + (size_t)cap * 8);
  lds_vf32* nb_d = (lds_vf32*)(lds_void_p)(smem + (size_t)cap * 8 + (size_t)nbmax * 4);
  lds_vu32* ctl = (lds_vu32*)(lds_void_p)(smem + (size_t)cap * 8 + (size_t)nbmax * 8);
  lds_vu8* flags = (lds_vu8*)(lds_void_p)(smem + (size_t)cap * 8 + (size_t)nbmax * 8 + 16);
  const size_t qoff = (size_t)cap * 8 + (size_t)nbmax * 8 + 16 + (((size_t)cap + 15) & ~(size_t)15);
  const int d4 = (int)((a.dim + 3) / 4);
  float* qgen = reinterpret_cast<float*>(smem + qoff);
  uint32_t* qcw = reinterpret_cast<uint32_t*>(smem + qoff + (CPL > 0 ? 0 : (size_t)d4 * 16));
  lds_vu64* outp = (lds_vu64*)(lds_void_p)(
      smem + ((qoff + (CPL > 0 ? 0 : (size_t)d4 * 16) + (size_t)A.codes.code_words * 4 + 15) & ~(size_t)15));

  uint32_t* vis = a.visited + (size_t)blockIdx.x * a.vis_words;
  uint32_t* vlog = a.vlog + (size_t)blockIdx.x * a.vlog_cap;
  const DistCtx dc{a.rows, a.norms, a.bits, a.row_stride, a.dim, a.words};
  const uint32_t CW = A.codes.code_words;
  // VIS: the exact visited set in LDS (VisSet, vdb_hnsw_device.hpp) instead of the HBM bitmap
  const VisSet vs{reinterpret_cast<uint32_t*>(smem + a.vis_off), (1u << a.vis_log2) - 1u, 32u - a.vis_log2};
  const uint32_t vis_limit = VIS ? (3u << a.vis_log2) / 4u : 0xFFFFFFFFu;
  if (VIS) {
    vs.clear(threadIdx.x, WAVES * 64);
    __syncthreads();
  }

  for (uint32_t qi = blockIdx.x; qi < a.nq; qi += gridDim.x) {
    const float* qp = a.queries + (size_t)qi * a.q_stride;
    // f32 query (re-rank) exactly as in hnsw_search_kernel
    float4 q[CPL > 0 ? CPL : 1];
    float qnorm = 0.0f;
    if (CPL > 0) {
      float nacc = 0.0f;
#pragma unroll
      for (int c = 0; c < CPL; c++) {
        q[c] = ld4(qp + (size_t)(c * 64 + lane) * 4);
        nacc = chain4<kOpDot>(nacc, q[c], q[c]);
      }
      if (METRIC == kCosine) qnorm = sqrtf(butterfly_all(nacc));
    } else {
      for (int i = threadIdx.x; i < d4 * 4; i += WAVES * 64) qgen[i] = i < (int)a.dim ? qp[i] : 0.0f;
      __syncthreads();
      if (METRIC == kCosine) {
        float nacc = 0.0f;
        for (int c = lane; c < d4; c += 64) {
          const float4 x = ld4(qgen + c * 4);
          const int nv = (int)a.dim - c * 4;
          nacc = nv >= 4 ? chain4<kOpDot>(nacc, x, x) : chain4_tail<kOpDot>(nacc, x, x, nv);
        }
        qnorm = sqrtf(butterfly_all(nacc));
      }
    }
    // quantised query (quantizer.quantize(query), dual_precision.rs:235)
    for (uint32_t w = threadIdx.x; w < CW; w += WAVES * 64) {
      uint32_t word = 0;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const uint32_t i = w * 4 + e;
        if (i < a.dim) {
          float v = roundf((qp[i] - A.min_vals[i]) * A.scales[i]);
          v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
          word |= ((v != v) ? 0u : (uint32_t)v) << (8 * e);
        }
      }
      qcw[w] = word;
    }
    __syncthreads();
    constexpr int QW = 4;  // words per lane: dims up to 1024
    uint32_t qw[QW];
    uint32_t qsq = 0;
#pragma unroll
    for (int t = 0; t < QW; t++) {
      qw[t] = ((uint32_t)lane + 64u * t < CW) ? qcw[lane + 64 * t] : 0u;
      qsq = __builtin_amdgcn_udot4(qw[t], qw[t], qsq, false);
    }
    qsq = wave_sum_u32(qsq);

    CandList<NS, true> list;
    list.init(keys, flags, cap);
    uint32_t n_dist = 0, n_expand = 0, logn = 0, overflow = 0, m_prev = 0, rr_m = 0;
    int phase = Q_START;
    int layer = (int)a.max_layer;
    uint32_t cur = a.entry_point;
    uint32_t best_d = 0;

    for (;;) {
      if (wib == 0) {
        bool ready = false;
        uint32_t m = 0, done = 0, exact = 0;
        while (!ready) {
          if (phase == Q_START) {
            if (lane == 0) nb_id[0] = cur;
            m = 1;
            ready = true;
            phase = layer > 0 ? Q_G_ENTRY : Q_Z_ENTRY;
          } else if (phase == Q_G_ENTRY) {
            best_d = rfl(__float_as_uint(nb_d[0]));
            n_dist += 1;
            phase = Q_G_LOAD;
          } else if (phase == Q_G_LOAD) {
            const HnswLayerRef L = a.layers[layer];
            uint32_t nc = rfl(L.cnt[cur]);
            nc = min(nc, min(L.stride, nbmax));
            for (uint32_t base = 0; base < nc; base += 64) {
              const uint32_t t = base + lane;
              if (t < nc) nb_id[t] = L.nbr[(size_t)cur * L.stride + t];
            }
            if (nc == 0) {
              phase = Q_G_DONE;
            } else {
              m = nc;
              ready = true;
              phase = Q_G_SCAN;
            }
          } else if (phase == Q_G_SCAN) {  // dual_precision.rs:422-433: strict <, first minimum
            n_dist += m_prev;
            uint32_t mn = 0, besti = kNoIndex;
            for (uint32_t base = 0; base < m_prev; base += 64) {
              const uint32_t t = base + lane;
              const uint32_t d = t < m_prev ? __float_as_uint(nb_d[t]) : 0xFFFFFFFFu;
              const bool ok = t < m_prev && d < best_d;
              const uint64_t okm = __ballot(ok);
              if (okm) {
                uint32_t v = ok ? d : 0xFFFFFFFFu;
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, s, 64));
                v = rfl(v);
                if (besti == kNoIndex || v < mn) {
                  const uint64_t eq = __ballot(ok && d == v);
                  besti = base + (uint32_t)__ffsll((long long)eq) - 1;
                  mn = v;
                }
              }
            }
            if (besti != kNoIndex) {
              cur = rfl(nb_id[besti]);
              best_d = mn;
              phase = Q_G_LOAD;
            } else {
              phase = Q_G_DONE;
            }
          } else if (phase == Q_G_DONE) {
            layer -= 1;
            phase = Q_START;
          } else if (phase == Q_Z_ENTRY) {
            const uint32_t d = rfl(__float_as_uint(nb_d[0]));
            n_dist += 1;
            list.insert(((uint64_t)d << 32) | cur, lane, overflow);
            if (lane == 0) {
              if (VIS) {
                (void)vs.test_and_set(cur);
              } else {
                atomicOr(&vis[cur >> 5], 1u << (cur & 31));
                if (a.vlog_cap) vlog[0] = cur;
              }
            }
            logn = 1;
            phase = Q_Z_POP;
          } else if (phase == Q_Z_POP) {
            const uint32_t idx = list.first_unexpanded(lane);
            if (idx == kNoIndex) {
              phase = Q_FINISH;
            } else {
              const uint64_t ckey = list.key_at(idx, lane);
              bool stop = false;
              if (list.size() >= ef) stop = (uint32_t)(ckey >> 32) > (uint32_t)(list.key_at(ef - 1, lane) >> 32);  // :341
              if (!stop && VIS && logn + nbmax > vis_limit) {  // the LDS set could pass 3/4: the caller re-runs on the bitmap
                overflow = 1;
                stop = true;
              }
              if (stop) {
                phase = Q_FINISH;
              } else {
                list.mark_expanded(idx, lane);
                n_expand += 1;
                const uint32_t cnode = (uint32_t)ckey;
                const HnswLayerRef L = a.layers[0];
                const uint32_t lim = min(L.stride, nbmax);
                uint32_t nb0 = 0;
                if ((uint32_t)lane < lim) nb0 = L.nbr[(size_t)cnode * L.stride + lane];
                uint32_t nc = rfl(L.cnt[cnode]);
                nc = min(nc, lim);
                for (uint32_t base = 0; base < nc; base += 64) {
                  const uint32_t t = base + lane;
                  const bool valid = t < nc;
                  uint32_t nb = nb0;
                  bool newly = false;
                  if (valid) {
                    if (base != 0) nb = L.nbr[(size_t)cnode * L.stride + t];
                    const uint32_t bit = 1u << (nb & 31);
                    newly = VIS ? vs.test_and_set(nb) : (atomicOr(&vis[nb >> 5], bit) & bit) == 0;
                  }
                  const uint64_t mask = __ballot(newly);
                  const uint32_t before = (uint32_t)__popcll(mask & lt_mask(lane));
                  if (newly) {
                    nb_id[m + before] = nb;
                    if (!VIS && logn + before < a.vlog_cap) vlog[logn + before] = nb;
                  }
                  m += (uint32_t)__popcll(mask);
                  logn += (uint32_t)__popcll(mask);
                }
                if (m != 0) {
                  ready = true;
                  phase = Q_Z_ADMIT;
                }
              }
            }
          } else if (phase == Q_Z_ADMIT) {  // dual_precision.rs:356-369
            n_dist += m_prev;
            for (uint32_t base = 0; base < m_prev; base += 64) {
              const uint32_t t = base + lane;
              const uint32_t d = t < m_prev ? __float_as_uint(nb_d[t]) : 0xFFFFFFFFu;
              uint32_t size = list.size() < ef ? list.size() : ef;
              uint32_t far = (uint32_t)(list.key_at(size - 1, lane) >> 32);
              uint64_t mask = __ballot(t < m_prev && (d < far || size < ef));
              while (mask) {
                const int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const uint32_t dj = (uint32_t)__builtin_amdgcn_readlane((int)d, src);
                size = list.size() < ef ? list.size() : ef;
                far = (uint32_t)(list.key_at(size - 1, lane) >> 32);
                if (dj < far || size < ef) {
                  const uint32_t nbj = nb_id[base + src];
                  list.insert(((uint64_t)dj << 32) | nbj, lane, overflow);
                  list.truncate(ef, lane);
                }
              }
            }
            phase = Q_Z_POP;
          } else if (phase == Q_FINISH) {
            // coarse candidates = the cand_k best by (int distance, node) (dual_precision.rs:377-383), then exact re-scoring
            const uint32_t size = list.size() < ef ? list.size() : ef;
            const uint32_t kk = A.cand_k < size ? A.cand_k : size;
            for (uint32_t base = 0; base < kk; base += 64) {
              const uint32_t e = base + lane;
              if (e < kk) nb_id[e] = (uint32_t)list.chunk_key(base, lane);
            }
            m = kk;
            rr_m = kk;
            exact = 1;
            phase = Q_R_DONE;
            if (m == 0) done = 1;
            ready = true;
          } else {  // Q_R_DONE
            done = 1;
            ready = true;
          }
        }
        if (lane == 0) {
          ctl[0] = m;
          ctl[1] = done;
          ctl[2] = logn;
          ctl[3] = exact;
        }
        m_prev = m;
      }
      __syncthreads();
      const uint32_t m = ctl[0];
      if (ctl[1]) break;
      if (ctl[3])
        dist_phase_f32<METRIC, CPL, WAVES>(dc, q, qnorm, qgen, m, nb_id, nb_d, lane, wib, false);  // exact engine distance
      else
        dist_phase_int8<QW, WAVES>(A.codes, qw, qsq, m, nb_id, nb_d, lane, wib);
      __syncthreads();
    }

    // ---- re-rank: stable sort of the exact distances ascending (total_cmp; equal distances keep the coarse
    // order), cut to k; soft-deleted rows are dropped after the cut like in the f32 path (search.rs:86-91);
    // scores through transform_score.  sort key = (total-order(dist) << 32 | coarse position), unique. ----
    if (wib == 0) {
      const uint32_t m = rr_m;
      for (uint32_t i = lane; i < m; i += 64) keys[i] = make_key<false>(nb_d[i], i);
      for (uint32_t i = lane; i < m; i += 64) {
        const uint64_t mine = keys[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m; j++) rank += keys[j] < mine ? 1u : 0u;
        outp[rank] = ((uint64_t)nb_id[i] << 32) | (uint64_t)__float_as_uint(nb_d[i]);
      }
      const uint32_t kk = m < a.k ? m : a.k;
      uint32_t outn = 0;
      for (uint32_t base = 0; base < kk; base += 64) {
        const uint32_t e = base + lane;
        const bool v = e < kk;
        const uint64_t pk = v ? outp[e] : 0;
        const uint32_t node = (uint32_t)(pk >> 32);
        bool al = v;
        if (v && a.alive) al = a.alive[node] != 0;
        const uint64_t mask = __ballot(al);
        const uint32_t p = outn + (uint32_t)__popcll(mask & lt_mask(lane));
        if (al) {
          a.out_ids[(size_t)qi * a.k + p] = a.ext_ids ? a.ext_ids[node] : (uint64_t)node;
          a.out_scores[(size_t)qi * a.k + p] = transform_score_dev(METRIC, __uint_as_float((uint32_t)pk));
        }
        outn += (uint32_t)__popcll(mask);
      }
      for (uint32_t e = outn + lane; e < a.k; e += 64) {
        a.out_ids[(size_t)qi * a.k + e] = ~0ull;
        a.out_scores[(size_t)qi * a.k + e] = __uint_as_float(0x7FC00000u);
      }
      if (lane == 0) {
        a.out_n[qi] = overflow ? 0xFFFFFFFFu : outn;
        if (a.stats) {
          atomicAdd(&a.stats[0], (unsigned long long)n_dist);
          atomicAdd(&a.stats[1], (unsigned long long)n_expand);
        }
      }
    }
    const uint32_t nlog = ctl[2];
    if (VIS) {
      __syncthreads();
      vs.clear(threadIdx.x, WAVES * 64);
    } else if (nlog <= a.vlog_cap) {
      for (uint32_t i = threadIdx.x; i < nlog; i += WAVES * 64) vis[vlog[i] >> 5] = 0;
    } else {
      for (uint64_t i = threadIdx.x; i < a.vis_words; i += WAVES * 64) vis[i] = 0;
    }
    __syncthreads();
  }
}

// ---- host side -----------------------------------------------------------------------------
template <int METRIC, int CPL, int NS, int WAVES, bool VIS>
static hipError_t launch_i8_wv(const HnswInt8Args& A, int slots, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hnsw_search_int8_kernel<METRIC, CPL, NS, WAVES, VIS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  int occ = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, hnsw_search_int8_kernel<METRIC, CPL, NS, WAVES, VIS>, WAVES * 64, lds);
  if (e != hipSuccess) return e;
  occ = std::max(1, std::min(occ, kTraversalSlotsPerCu));
  const int grid = (int)std::min<int64_t>((int64_t)slots, (int64_t)A.s.n_cus * occ);
  hipLaunchKernelGGL((hnsw_search_int8_kernel<METRIC, CPL, NS, WAVES, VIS>), dim3(grid), dim3(WAVES * 64), lds, st, A);
  return hipGetLastError();
}
// VELESDB_INT8_VIS_LDS: 0 = HBM bitmaps, 1 = the exact LDS visited set (2^14 entries = 64 KiB: two queries in flight per CU);
// unset = the measured default
static const int g_i8_vis = [] {
  const char* e = probe_env("VELESDB_INT8_VIS_LDS");
  return e ? atoi(e) : -1;
}();
template <int METRIC, int CPL, int NS, int WAVES>
static hipError_t launch_i8_w(const HnswInt8Args& A0, int slots, size_t lds, hipStream_t st) {
  HnswInt8Args A = A0;
  const uint32_t lg = 14;
  const bool use = A0.s.vis_log2 != 0 && g_i8_vis == 1 && (uint64_t)A0.s.ef * 90 + A0.s.nbmax <= (3ull << lg) / 4 &&
                   lds + ((size_t)4 << lg) <= 160 * 1024;
  A.s.vis_log2 = use ? lg : 0u;
  A.s.vis_off = (uint32_t)lds;
  if (use) return launch_i8_wv<METRIC, CPL, NS, WAVES, true>(A, slots, lds + ((size_t)4 << lg), st);
  return launch_i8_wv<METRIC, CPL, NS, WAVES, false>(A, slots, lds, st);
}
template <int METRIC, int CPL, int NS>
static hipError_t launch_i8_ns(const HnswInt8Args& A, int slots, size_t lds, hipStream_t st) {
  // batches that can fill more than four-wave blocks' worth of slots run two-wave blocks (twice the queries in flight)
  if (g_i8_waves2.load(std::memory_order_relaxed) && (int64_t)slots > (int64_t)A.s.n_cus * 3)
    return launch_i8_w<METRIC, CPL, NS, 2>(A, slots, lds, st);
  return launch_i8_w<METRIC, CPL, NS, 4>(A, slots, lds, st);
}
template <int METRIC, int CPL>
static hipError_t launch_i8_t(const HnswInt8Args& A, int slots, size_t lds, hipStream_t st) {
  if (A.s.list_slots == kSearchRegSlots) return launch_i8_ns<METRIC, CPL, kSearchRegSlots>(A, slots, lds, st);
  return launch_i8_ns<METRIC, CPL, 0>(A, slots, lds, st);
}
template <int METRIC>
static hipError_t launch_i8_cpl(const HnswInt8Args& A, int slots, size_t lds, hipStream_t st) {
  switch (sweep_cpl_for_dim(A.s.dim)) {
    case 1: return launch_i8_t<METRIC, 1>(A, slots, lds, st);
    case 2: return launch_i8_t<METRIC, 2>(A, slots, lds, st);
    case 3: return launch_i8_t<METRIC, 3>(A, slots, lds, st);
    case 4: return launch_i8_t<METRIC, 4>(A, slots, lds, st);
    default: return launch_i8_t<METRIC, 0>(A, slots, lds, st);
  }
}

// ScalarQuantizer::train on the first sample_rows rows (0 = min(1000, n_rows), dual_precision.rs:95,134-157) and
// quantisation of every row present; later rows are quantised as they arrive (index.hip finish_append)
int32_t quantizer_train(vdb_hip_index* ix, uint32_t sample_rows) {
  if (ix->n_rows == 0) return fail(VDB_ERR_STATE, "Cannot train on empty vectors");  // quantization.rs:193
  if (ix->dim > 1024) return fail(VDB_ERR_UNSUPPORTED, "int8 traversal: dim > 1024");
  hipStream_t st = ix->stream;
  const uint32_t n_train = (uint32_t)std::min<uint64_t>(sample_rows ? sample_rows : 1000, ix->n_rows);
  ix->code_words = ((ix->dim + 3) / 4 + 3) / 4 * 4;
  hipError_t e;
  if ((e = ix->sq_min.reserve((size_t)ix->dim * 4, false, st)) != hipSuccess ||
      (e = ix->sq_scale.reserve((size_t)ix->dim * 4, false, st)) != hipSuccess ||
      (e = ix->codes.reserve(std::max<uint64_t>(ix->capacity, 1) * ix->code_words * 4, false, st)) != hipSuccess ||
      (e = ix->codes_sq.reserve(std::max<uint64_t>(ix->capacity, 1) * 4, false, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("quantiser: ") + hipGetErrorString(e));
  hipLaunchKernelGGL(sq_train, dim3((ix->dim + 255) / 256), dim3(256), 0, st, ix->rows.as<float>(), ix->row_stride, n_train,
                     ix->dim, ix->sq_min.as<float>(), ix->sq_scale.as<float>());
  VDB_HIP(hipGetLastError());
  ix->quantizer_trained = true;
  return quantize_rows(ix, 0, ix->n_rows);
}

int32_t quantize_rows(vdb_hip_index* ix, uint64_t first, uint64_t n) {
  if (!ix->quantizer_trained || n == 0) return VDB_OK;
  const int blocks = (int)std::min<uint64_t>((n + 3) / 4, 4096);
  hipLaunchKernelGGL(sq_quantize_rows, dim3(blocks), dim3(256), 0, ix->stream, ix->rows.as<float>(), ix->row_stride,
                     ix->sq_min.as<float>(), ix->sq_scale.as<float>(), ix->codes.as<uint32_t>(), ix->code_words,
                     ix->codes_sq.as<uint32_t>(), (uint32_t)first, (uint32_t)n, ix->dim);
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}

// DualPrecisionHnsw::search_with_config(use_int8_traversal = true) for nq device-resident queries
int32_t hnsw_search_int8_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k,
                             uint32_t ef_search, uint32_t oversampling, uint32_t cap_mult, uint64_t* d_ids,
                             float* d_scores, uint32_t* d_n, hipStream_t st) {
  if (!ix->graph_valid) return fail(VDB_ERR_STATE, "HNSW graph not built for all rows (use mode BRUTE or build it)");
  if (!ix->quantizer_trained) return fail(VDB_ERR_STATE, "int8 traversal: call vdb_hip_index_train_quantizer first");
  if (ix->metric == VDB_HAMMING || ix->metric == VDB_JACCARD)
    return fail(VDB_ERR_UNSUPPORTED, "int8 traversal: Cosine / Euclidean / DotProduct indexes only");
  if (nq == 0) return VDB_OK;
  if (ix->entry_point < 0 || ix->graph_nodes == 0 || k == 0) {
    VDB_HIP(hipMemsetAsync(d_n, 0, (size_t)nq * 4, st));
    return VDB_OK;
  }
  if (oversampling == 0) oversampling = 4;  // DualPrecisionConfig::default (dual_precision.rs:57)
  const uint64_t cand_k = (uint64_t)k * oversampling;
  const uint64_t ef = std::max<uint64_t>(ef_search, cand_k);  // dual_precision.rs:334
  HnswInt8Args A{};
  HnswSearchArgs& a = A.s;
  uint32_t nbmax = 0;
  for (size_t l = 0; l < ix->layers.size() && l < (size_t)kMaxLayers; l++) {
    a.layers[l].nbr = ix->layers[l].nbr.as<uint32_t>();
    a.layers[l].cnt = ix->layers[l].cnt.as<uint32_t>();
    a.layers[l].stride = ix->layers[l].stride;
    nbmax = std::max(nbmax, ix->layers[l].stride);
  }
  nbmax = (uint32_t)((std::max<uint64_t>(nbmax, cand_k) + 63) / 64 * 64);
  uint64_t cap = ef + std::max<uint64_t>(64, ef * cap_mult / 2);
  cap = (cap + 63) / 64 * 64;
  // integer distances tie often: the register list (256 entries) is used for small ef only on the first attempt
  const bool reg_list = cap_mult == 1 && ef + 64 <= (uint64_t)kSearchRegSlots * 64 && ix->n_rows < (1ull << 31);
  if (reg_list) cap = (uint64_t)kSearchRegSlots * 64;
  cap = std::max<uint64_t>(cap, cand_k);
  const int cpl = sweep_cpl_for_dim(ix->dim);
  size_t lds = (size_t)cap * 8 + (size_t)nbmax * 8 + 16 + (((size_t)cap + 15) & ~(size_t)15);
  if (cpl == 0) lds += (size_t)((ix->dim + 3) / 4) * 16;
  lds += (size_t)ix->code_words * 4;
  lds = (lds + 15) & ~(size_t)15;
  lds += (size_t)nbmax * 8;
  if (cap > 0xFFFFFFFFull || lds > 160 * 1024)
    return fail(VDB_ERR_UNSUPPORTED, "ef too large for the LDS-resident candidate list");
  int32_t rcs = ensure_traversal_scratch(ix, st);
  if (rcs != VDB_OK) return rcs;
  VDB_HIP(hipMemsetAsync(ix->s_stats.p, 0, 24, st));
  a.rows = ix->rows.as<float>();
  a.norms = ix->norms.as<float>();
  a.bits = nullptr;
  a.alive = ix->any_dead ? ix->alive.as<uint8_t>() : nullptr;
  a.ext_ids = ix->ext_ids.as<uint64_t>();
  a.queries = d_q;
  a.row_stride = ix->row_stride;
  a.q_stride = q_stride;
  a.visited = ix->s_visited.as<uint32_t>();
  a.vlog = ix->s_vlog.as<uint32_t>();
  a.vis_words = ix->vis_words;
  a.out_ids = d_ids;
  a.out_scores = d_scores;
  a.out_n = d_n;
  a.stats = ix->s_stats.as<unsigned long long>();
  a.dim = ix->dim;
  a.words = ix->words;
  a.n_rows = (uint32_t)ix->n_rows;
  a.nq = nq;
  a.k = k;
  a.ef = (uint32_t)ef;
  a.cap = (uint32_t)cap;
  a.nbmax = nbmax;
  a.vlog_cap = kVlogCap;
  a.max_layer = ix->max_layer;
  a.entry_point = (uint32_t)ix->entry_point;
  a.metric = ix->metric;
  a.n_cus = (uint32_t)ix->n_cus;
  a.list_slots = reg_list ? kSearchRegSlots : 0;
  a.vis_log2 = cap_mult == 1 ? 1u : 0u;  // "the LDS visited set is allowed" (a re-run after an overflow takes the bitmap)
  A.codes = CodeCtx{ix->codes.as<uint32_t>(), ix->codes_sq.as<uint32_t>(), ix->code_words};
  A.min_vals = ix->sq_min.as<float>();
  A.scales = ix->sq_scale.as<float>();
  A.cand_k = (uint32_t)cand_k;
  const int slots = (int)std::min<int64_t>((int64_t)nq, (int64_t)ix->n_cus * kTraversalSlotsPerCu);
  EventPair* ev = next_events(ix);
  if (ev) (void)hipEventRecord(ev->a, st);
  hipError_t e;
  switch (ix->metric) {
    case kCosine: e = launch_i8_cpl<kCosine>(A, slots, lds, st); break;
    case kEuclidean: e = launch_i8_cpl<kEuclidean>(A, slots, lds, st); break;
    default: e = launch_i8_cpl<kDot>(A, slots, lds, st); break;
  }
  if (ev) (void)hipEventRecord(ev->b, st);
  if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("int8 search launch: ") + hipGetErrorString(e));
  ix->stats_pending = true;
  return VDB_OK;
}

}  // namespace vdb
