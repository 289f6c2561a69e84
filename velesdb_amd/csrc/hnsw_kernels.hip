// hnsw_kernels.hip — graph traversal on the GPU: NativeHnsw::search (native/graph.rs:251-270) =
// greedy descent search_layer_single (graph.rs:405-428) on layers max_layer..1, then the layer-0
// best-first beam search_layer (graph.rs:438-520); result mapping of HnswIndex::search_with_quality
// (index/hnsw/index/search.rs:79-93, transform_score backend_adapter.rs:160-168).
//
// Design (MI355X-first; the path is bound by random 3 KB row gathers from HBM):
//   * one 256-thread block per query in flight ("slot"); slots loop over the batch.  With 4 blocks
//     per CU that is 1024 queries x 4 waves x 8 rows x 3 KB = several hundred KB of loads in flight
//     per CU, far more than the latency-bandwidth product needs.
//   * a step = (leader wave decides what to evaluate) -> barrier -> (all 4 waves evaluate up to
//     `nbmax` distances: wave w takes groups of 8 neighbours, every row is read as float4 per lane,
//     1 KiB per load instruction, canonical per-lane fmaf chains + xor-butterfly, vdb_device.hpp)
//     -> barrier.  The decision logic is the reference's, statement for statement, executed by one
//     wave with wave-uniform control flow; the 64 lanes are used for the list operations.
//   * candidates + results live in ONE sorted list in LDS keyed by (total-order(dist), node) with
//     an "expanded" flag per entry:  results  = the first min(len, ef) entries (the reference's
//     max-heap bounded to ef holds exactly the ef smallest keys pushed so far), candidates = the
//     entries not yet expanded (the reference pushes every admitted node to both heaps).  Entries
//     past position ef were evicted from results; they stay only while the reference's
//     termination test `c_dist > furthest` (graph.rs:474) could still let them be expanded
//     (exact ties with the current furthest distance: common for Hamming, never for f32).
//   * visited set = one bit per node in HBM per slot (atomicOr test-and-set, L2-resident), undone
//     after each query from a log of the ids it set.
// Algorithmic HBM bytes per query: n_dist * dim * 4 + n_expand * M0 * 4, with n_dist / n_expand
// counted by the kernel (stats), SURVEY.md §8(d).
#include <algorithm>
#include <cstdlib>

#include "vdb_probe_env.hpp"
#include "vdb_hnsw_device.hpp"
#include "vdb_index.hpp"

namespace vdb {

namespace {

enum Phase : int {
  P_START = 0,   // evaluate dist(q, cur) on `layer`
  P_G_ENTRY,     // consume it as best_dist (graph.rs:407)
  P_G_LOAD,      // load neighbours of best (graph.rs:410)
  P_G_SCAN,      // scan them with strict < (graph.rs:413-421)
  P_G_DONE,      // no improvement: next layer down
  P_Z_ENTRY,     // layer 0: push the entry point (graph.rs:464-469)
  P_Z_POP,       // pop nearest candidate, termination test, gather unvisited neighbours (:471-499)
  P_Z_ADMIT,     // admission of the evaluated neighbours in list order (:500-511)
  P_FINISH,
  P_R_DONE       // rerank: raw scores of the candidates are in nb_d
};

}  // namespace

// ------------------------------------------------------------------------------------------
// LDS: keys[cap] u64 | nb_id[nbmax] u32 | nb_d[nbmax] f32 | ctl[4] u32 | flags[cap] u8 (padded to 16)
//      | query scratch: generic f32 dims: d4*4 floats; bit metrics: `words` u32
// ------------------------------------------------------------------------------------------
// LAT (latency mode: calls of at most one query per CU, f32 metrics; the speculative step below: layer-0 lists of <= 64 neighbours,
// longer lists take the test-first form — launch_hnsw_search): a 1 024-thread block per query and a speculative layer-0 step.  A walk is a chain of dependent memory round trips — neighbour ids, visited
// test-and-set, rows — and a single query cannot hide them behind other queries: here all (<= 64) neighbours' rows are fetched
// at once by 16 waves (4 rows each) WITHOUT waiting for the visited test, which the leader wave issues alongside; the
// verdicts select, in list order, which of the evaluated distances are admitted.  Same ids, scores and counters (n_dist counts
// the unvisited neighbours, as the reference's loop evaluates them); the rows of visited neighbours are wasted bandwidth that a
// single query has to spare.
// __launch_bounds__(.., 4): four waves per SIMD = four 256-thread walks per CU.  Left alone the register-list instance took 143
// registers = three walks per CU; the walk is a chain of dependent memory round trips, and the rows in flight per CU are what
// the chip's random-gather bandwidth follows: 126 registers (no scratch) took the 8 192-query launch at 1 M x 768, ef 128 from
// 49.1 to 42.1 ms on one box (0.61 -> 0.72 of HBM; profiles/r04m_f32_walk_occupancy_ab.log), same ids / scores / counters.
// RAWEF: NativeHnsw-level calls whose ef_search may be smaller than the number of layer-0 entry points (search_multi_entry with
// ef_search < 4 and several probes, or ef_search = 0): ef is then raised per query at the layer-0 start (see P_START).  Only the
// generic instance (LDS list, any dimension) exists in this form: in every other instance ef stays the launch constant it was —
// as a loop-carried value it put 48 bytes of the register-list instances into scratch memory.
template <int METRIC, int CPL, int NS, bool LAT = false, bool VIS = false, bool RAWEF = false>
__global__ __launch_bounds__(LAT ? 1024 : 256, 4) void hnsw_search_kernel(HnswSearchArgs a) {
  constexpr bool BITS = (METRIC == kHamming || METRIC == kJaccard);
  constexpr int WAVES = LAT ? 16 : 4, TPB = WAVES * 64, RR = LAT ? 4 : 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = (int)rfl(threadIdx.x >> 6);
  const uint32_t cap = a.cap, nbmax = a.nbmax;
  const uint32_t ef_launch = a.ef;
  uint32_t ef_query = a.ef;  // RAWEF only (wave-uniform)
#define ef (RAWEF ? ef_query : ef_launch)
  lds_vu64* keys = (lds_vu64*)(lds_void_p)(smem);
  lds_vu32* nb_id = (lds_vu32*)(lds_void_p)(smem + (size_t)cap * 8);
  lds_vf32* nb_d = (lds_vf32*)(lds_void_p)(smem + (size_t)cap * 8 + (size_t)nbmax * 4);
  lds_vu32* ctl = (lds_vu32*)(lds_void_p)(smem + (size_t)cap * 8 + (size_t)nbmax * 8);
  lds_vu8* flags = (lds_vu8*)(lds_void_p)(smem + (size_t)cap * 8 + (size_t)nbmax * 8 + 16);
  const size_t qoff = (size_t)cap * 8 + (size_t)nbmax * 8 + 16 + (((size_t)cap + 15) & ~(size_t)15);
  float* qgen = reinterpret_cast<float*>(smem + qoff);
  uint32_t* qbits = reinterpret_cast<uint32_t*>(smem + qoff);

  uint32_t* vis = a.visited + (size_t)blockIdx.x * a.vis_words;
  uint32_t* vlog = a.vlog + (size_t)blockIdx.x * a.vlog_cap;
  // VIS: the exact visited set in LDS (zero between queries); a query stops with the overflow flag before it is 3/4 full
  const VisSet vs{reinterpret_cast<uint32_t*>(smem + a.vis_off), (1u << a.vis_log2) - 1u, 32u - a.vis_log2};
  const uint32_t vis_limit = VIS ? (3u << a.vis_log2) / 4u : 0xFFFFFFFFu;
  if (VIS) {
    vs.clear(threadIdx.x, TPB);
    __syncthreads();
  }
  const int d4 = (int)((a.dim + 3) / 4);
  const DistCtx dc{a.rows, a.norms, a.bits, a.row_stride, a.dim, a.words};

  for (uint32_t qi = blockIdx.x; qi < a.nq; qi += gridDim.x) {
    const float* qp = a.queries + (size_t)qi * a.q_stride;
    float4 q[CPL > 0 ? CPL : 1];
    float qnorm = 0.0f;
    if (BITS) {
      for (uint32_t w = threadIdx.x; w < a.words; w += TPB) {
        uint32_t bitsw = 0;
        for (uint32_t e = 0; e < 32; e++) {
          const uint32_t i = w * 32 + e;
          if (i < a.dim && qp[i] > 0.5f) bitsw |= 1u << e;
        }
        qbits[w] = bitsw;
      }
    } else if (CPL > 0) {
      float nacc = 0.0f;
#pragma unroll
      for (int c = 0; c < CPL; c++) {
        q[c] = ld4(qp + (size_t)(c * 64 + lane) * 4);
        nacc = chain4<kOpDot>(nacc, q[c], q[c]);
      }
      if (METRIC == kCosine) qnorm = sqrtf(butterfly_all(nacc));
    } else {
      const int qlen = d4 * 4;
      for (int i = threadIdx.x; i < qlen; i += TPB) qgen[i] = i < (int)a.dim ? qp[i] : 0.0f;
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
    __syncthreads();

    // ---- leader state (meaningful in wave 0 only; every value is wave-uniform) ----
    CandList<NS> list;
    list.init(keys, flags, cap);
    uint32_t n_dist = 0, n_expand = 0, logn = 0, overflow = 0, m_prev = 0, rr_m = 0;
    bool spec = false;        // LAT: the distance phase in flight evaluates ALL neighbours of the expanded node ...
    uint64_t spec_mask = 0;   // ... and these lanes' neighbours were unvisited (known behind that phase)
    // the neighbour list of the PREDICTED next pop — the nearest unexpanded candidate left behind by this pop — requested with this
    // pop's own list: when the admission puts nothing in front of it (most expansions once the beam has settled), the next pop finds
    // its ids in registers and the walk is one dependent memory round trip shorter.  A wrong guess costs 260 bytes.  Not counted:
    // n_dist / n_expand are what the reference's loop counts (graph.rs:471-511).
    uint32_t pf_node = 0xFFFFFFFFu, pf_nb = 0, pf_cnt = 0, pf_hits = 0;
    // (latency-mode instances only: the throughput instances sit at the 128-register line that lets four walks share a CU, and the
    // three registers this needs across the distance phase put 20 of theirs into scratch)
    const bool PF = LAT && a.pf_ids != 0;
    int phase = P_START;
    int layer = (int)a.max_layer;
    uint32_t cur = a.entry_point;
    float best_d = 0.0f;

    for (;;) {
      if (wib == 0) {
        bool ready = false;
        uint32_t m = 0, done = 0, raw = 0;
        while (!ready) {
          if (phase == P_START) {
            if (lane == 0) nb_id[0] = cur;
            m = 1;
            if (layer == 0 && a.extra_eps) {  // search_multi_entry: the drawn ids that are not yet entry points (graph.rs:335-338)
              uint32_t e0 = cur, e1 = 0xFFFFFFFFu, e2 = 0xFFFFFFFFu;
              for (uint32_t j = 0; j < 3; j++) {
                const uint32_t id = a.extra_eps[(size_t)qi * 3 + j];
                if (id == 0xFFFFFFFFu || id == e0 || id == e1 || id == e2) continue;
                if (lane == 0) nb_id[m] = id;
                if (m == 1) e1 = id; else e2 = id;  // (a third new id has nothing behind it to be compared with)
                m++;
              }
            }
            // graph.rs:463-468 pushes EVERY entry point into `results` with no cut, and graph.rs:503-509 pops at most one entry per
            // push: with more entry points than ef_search the result set simply stays at that size — which is what a search with
            // ef = the number of entry points does from its first step (results full from the start: same admission test
            // `dist < furthest`, same single pop, same stop rule `len >= ef`).  Only NativeHnsw-level calls get here with ef < m:
            // search_multi_entry with ef_search < 4 and several probes, or ef_search = 0 (which therefore acts as 1).
            if (RAWEF && layer == 0) ef_query = max(ef_launch, m);
            ready = true;
            phase = layer > 0 ? P_G_ENTRY : P_Z_ENTRY;
          } else if (phase == P_G_ENTRY) {
            best_d = rflf(nb_d[0]);
            n_dist += 1;
            phase = P_G_LOAD;
          } else if (phase == P_G_LOAD) {
            const HnswLayerRef L = a.layers[layer];
            uint32_t nc = rfl(L.cnt[cur]);
            nc = min(nc, min(L.stride, nbmax));
            for (uint32_t base = 0; base < nc; base += 64) {
              const uint32_t t = base + lane;
              if (t < nc) nb_id[t] = L.nbr[(size_t)cur * L.stride + t];
            }
            if (nc == 0) {
              phase = P_G_DONE;
            } else {
              m = nc;
              ready = true;
              phase = P_G_SCAN;
            }
          } else if (phase == P_G_SCAN) {
            n_dist += m_prev;
            // sequential scan with strict `<` == first index attaining the minimum, if below best
            float mn = 0.0f;
            uint32_t besti = 0xFFFFFFFFu;
            for (uint32_t base = 0; base < m_prev; base += 64) {
              const uint32_t t = base + lane;
              const float d = t < m_prev ? nb_d[t] : 0.0f;
              const bool ok = t < m_prev && d < best_d;  // raw compare: NaN never improves
              const uint64_t okm = __ballot(ok);
              if (okm) {
                float v = ok ? d : __uint_as_float(0x7F800000u);
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) v = fminf(v, shx(v, s));
                v = rflf(v);
                if (besti == 0xFFFFFFFFu || v < mn) {
                  const uint64_t eq = __ballot(ok && d == v);
                  besti = base + (uint32_t)__ffsll((long long)eq) - 1;
                  mn = v;
                }
              }
            }
            if (besti != 0xFFFFFFFFu) {
              cur = rfl(nb_id[besti]);
              best_d = rflf(nb_d[besti]);
              phase = P_G_LOAD;
            } else {
              phase = P_G_DONE;
            }
          } else if (phase == P_G_DONE) {
            layer -= 1;
            phase = P_START;
          } else if (phase == P_Z_ENTRY) {
            // graph.rs:463-468: every entry point is evaluated, pushed to both heaps and marked visited (one unless search_multi_entry)
            for (uint32_t t = 0; t < m_prev; t++) {
              const float d = rflf(nb_d[t]);
              const uint32_t ep = rfl(nb_id[t]);
              n_dist += 1;
              list.insert(make_key<false>(d, ep), lane, overflow);
              if (lane == 0) {
                if (VIS) {
                  (void)vs.test_and_set(ep);
                } else {
                  atomicOr(&vis[ep >> 5], 1u << (ep & 31));
                  if (a.vlog_cap) vlog[t] = ep;
                }
              }
            }
            logn = m_prev;
            phase = P_Z_POP;
          } else if (phase == P_Z_POP) {
            const uint32_t idx = list.first_unexpanded(lane);
            if (idx == kNoIndex) {
              phase = P_FINISH;  // candidates empty (graph.rs:471)
            } else {
              const uint64_t ckey = list.key_at(idx, lane);
              bool stop = false;
              if (list.size() >= ef) stop = key_dist(ckey) > key_dist(list.key_at(ef - 1, lane));  // graph.rs:474
              if (!stop && VIS && logn + nbmax > vis_limit) {  // the LDS set could pass 3/4: the caller re-runs on the bitmap
                overflow = 1;
                stop = true;
              }
              if (stop) {
                phase = P_FINISH;
              } else {
                list.mark_expanded(idx, lane);
                n_expand += 1;
                const uint32_t cnode = (uint32_t)ckey;
                const HnswLayerRef L = a.layers[0];
                // the neighbour ids are requested together with the count (one memory round trip instead of two)
                const uint32_t lim = min(L.stride, nbmax);
                uint32_t nb0 = 0, ncv = 0;
                if (PF && pf_node == cnode) {
                  nb0 = pf_nb;
                  ncv = pf_cnt;
                  pf_hits += 1;
                } else {
                  if ((uint32_t)lane < lim) nb0 = L.nbr[(size_t)cnode * L.stride + lane];
                  ncv = L.cnt[cnode];
                }
                uint32_t nc = rfl(ncv);
                nc = min(nc, lim);
                if (PF) {
                  // (behind the wait for this pop's own list: where the two paths above join, the compiler waits for EVERY load in
                  // flight — requested in front of that, the prediction's round trip was paid by every pop: 2 % slower than none)
                  const uint32_t idx2 = list.first_unexpanded(lane);
                  pf_node = 0xFFFFFFFFu;
                  if (idx2 != kNoIndex) {
                    pf_node = (uint32_t)list.key_at(idx2, lane);
                    if ((uint32_t)lane < lim) pf_nb = L.nbr[(size_t)pf_node * L.stride + lane];
                    pf_cnt = L.cnt[pf_node];
                  }
                }
                if (LAT && a.lat_spec) {  // (nc <= 64, host) every neighbour is evaluated; the visited verdicts follow with the distances
                  if ((uint32_t)lane < nc) nb_id[lane] = nb0;
                  m = nc;
                  spec = nc != 0;
                } else {
                for (uint32_t base = 0; base < nc; base += 64) {
                    const uint32_t t = base + lane;
                    const bool valid = t < nc;
                    uint32_t nb = nb0;
                    bool newly = false;
                    if (valid) {
                      if (base != 0) nb = L.nbr[(size_t)cnode * L.stride + t];
                      const uint32_t bit = 1u << (nb & 31);
                      newly = VIS ? vs.test_and_set(nb) : (atomicOr(&vis[nb >> 5], bit) & bit) == 0;  // visited.insert (graph.rs:499)
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
                }
                if (m != 0) {
                  ready = true;
                  phase = P_Z_ADMIT;
                }
              }
            }
          } else if (phase == P_Z_ADMIT) {
            if (LAT && a.lat_spec) {  // the unvisited ones of the evaluated neighbours, in list order: counters and the undo log
              n_dist += (uint32_t)__popcll(spec_mask);
              if (!VIS && (spec_mask >> lane & 1ull)) {
                const uint32_t pos = logn + (uint32_t)__popcll(spec_mask & lt_mask(lane));
                if (pos < a.vlog_cap) vlog[pos] = nb_id[lane];
              }
              logn += (uint32_t)__popcll(spec_mask);
            } else {
              n_dist += m_prev;
            }
            for (uint32_t base = 0; base < m_prev; base += 64) {
              const uint32_t t = base + lane;
              const float d = t < m_prev ? nb_d[t] : 0.0f;
              uint32_t size = list.size() < ef ? list.size() : ef;
              float far = key_dist(list.key_at(size - 1, lane));
              // pre-filter against the furthest distance at chunk start: it only decreases while the
              // result set is full, so a neighbour rejected now would be rejected at its turn too
              uint64_t mask = __ballot(t < m_prev && (d < far || size < ef));
              if (LAT && a.lat_spec) mask &= spec_mask;
              // two or more to admit: all at once (vdb_hnsw_device.hpp admit_batch) unless distances tie exactly
              if (NS > 0 && (mask & (mask - 1)) != 0ull &&
                  list.admit_batch(mask, d, t < m_prev ? nb_id[t] : 0u, lane, ef, keys, flags))
                mask = 0ull;
              while (mask) {
                const int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const float dj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d), src));
                size = list.size() < ef ? list.size() : ef;
                far = key_dist(list.key_at(size - 1, lane));
                if (dj < far || size < ef) {  // graph.rs:503
                  const uint32_t nbj = nb_id[base + src];
                  list.insert(make_key<false>(dj, nbj), lane, overflow);
                  list.truncate(ef, lane);
                }
              }
            }
            phase = P_Z_POP;
          } else if (phase == P_FINISH) {
            if (a.rerank_k == 0) {
              done = 1;
              ready = true;
            } else {
              // search_with_rerank (search.rs:118-160): candidates = the search result for k = rerank_k (soft-deleted
              // rows dropped, search.rs:86-91); their rows are re-scored with the raw compute_distance
              const uint32_t size = list.size() < ef ? list.size() : ef;
              const uint32_t kk = a.rerank_k < size ? a.rerank_k : size;
              for (uint32_t base = 0; base < kk; base += 64) {
                const uint32_t e = base + lane;
                const bool v = e < kk;
                const uint32_t node = v ? (uint32_t)list.chunk_key(base, lane) : 0;
                bool al = v;
                if (v && a.alive) al = a.alive[node] != 0;
                const uint64_t mask = __ballot(al);
                if (al) nb_id[m + (uint32_t)__popcll(mask & lt_mask(lane))] = node;
                m += (uint32_t)__popcll(mask);
              }
              raw = 1;
              rr_m = m;
              phase = P_R_DONE;
              if (m == 0) done = 1;
              ready = true;
            }
          } else {  // P_R_DONE
            done = 1;
            ready = true;
          }
        }
        if (lane == 0) {
          ctl[0] = m;
          ctl[1] = done;
          ctl[2] = logn;
          ctl[3] = raw;
        }
        m_prev = m;
      }
      __syncthreads();
      const uint32_t m = ctl[0];
      if (ctl[1]) break;
      const bool raw = ctl[3] != 0;
      uint32_t spec_old = 0, spec_bit = 0;
      const bool spec_lane = LAT && wib == 0 && spec && (uint32_t)lane < m;
      if (spec_lane) {  // visited.insert (graph.rs:499) for every neighbour, in flight beside the row fetches below
        const uint32_t nb = nb_id[lane];
        spec_bit = 1u << (nb & 31);
        spec_old = VIS ? (vs.test_and_set(nb) ? 0u : spec_bit) : atomicOr(&vis[nb >> 5], spec_bit);
      }
      if (BITS)
        dist_phase_bits<METRIC>(dc, qbits, m, nb_id, nb_d, raw);
      else
        dist_phase_f32<METRIC, CPL, WAVES, RR>(dc, q, qnorm, qgen, m, nb_id, nb_d, lane, wib, raw);
      if (LAT && wib == 0) {
        spec_mask = spec ? __ballot(spec_lane && (spec_old & spec_bit) == 0) : 0ull;
        spec = false;
      }
      __syncthreads();
    }

    // ---- rerank results: stable sort of the re-scored candidates in the metric's order (distance.rs:95-103),
    // cut to k.  sort key = (order-key(score) << 32 | candidate position): unique, rank = #smaller keys ----
    if (a.rerank_k != 0) {
      if (wib == 0) {
        const uint32_t m = rr_m;  // candidates re-scored in the last distance phase (0 if none)
        constexpr bool HIB = higher_is_better(METRIC);
        const uint32_t outn = m < a.k ? m : a.k;
        for (uint32_t i = lane; i < m; i += 64) keys[i] = make_key<HIB>(nb_d[i], i);
        for (uint32_t i = lane; i < m; i += 64) {
          const uint64_t mine = keys[i];
          uint32_t rank = 0;
          for (uint32_t j = 0; j < m; j++) rank += keys[j] < mine ? 1u : 0u;
          if (rank < a.k) {
            const uint32_t node = nb_id[i];
            a.out_ids[(size_t)qi * a.k + rank] = a.ext_ids ? a.ext_ids[node] : (uint64_t)node;
            a.out_scores[(size_t)qi * a.k + rank] = nb_d[i];
          }
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
            if (pf_hits) atomicAdd(&a.stats[2], (unsigned long long)pf_hits);
          }
        }
      }
    } else
    // ---- results: first k of the sorted result set, soft-deleted rows dropped after the cut
    // (search.rs:86-91), scores through transform_score ----
    if (wib == 0) {
      const uint32_t size = list.size() < ef ? list.size() : ef;
      const uint32_t kk = a.k < size ? a.k : size;
      uint32_t outn = 0;
      for (uint32_t base = 0; base < kk; base += 64) {
        const uint32_t e = base + lane;
        const bool v = e < kk;
        const uint64_t key = v ? list.chunk_key(base, lane) : 0;
        const uint32_t node = (uint32_t)key;
        bool al = v;
        if (v && a.alive) al = a.alive[node] != 0;
        const uint64_t mask = __ballot(al);
        const uint32_t p = outn + (uint32_t)__popcll(mask & lt_mask(lane));
        if (al) {
          a.out_ids[(size_t)qi * a.k + p] = a.ext_ids ? a.ext_ids[node] : (uint64_t)node;
          a.out_scores[(size_t)qi * a.k + p] = transform_score_dev(METRIC, key_dist(key));
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
          if (pf_hits) atomicAdd(&a.stats[2], (unsigned long long)pf_hits);
        }
      }
    }
    // ---- undo the visited bits of this query ----
    const uint32_t nlog = ctl[2];
    if (VIS) {
      __syncthreads();  // (ctl[2] read by everybody before the next query's leader rewrites it)
      vs.clear(threadIdx.x, TPB);
    } else if (nlog <= a.vlog_cap) {
      for (uint32_t i = threadIdx.x; i < nlog; i += TPB) vis[vlog[i] >> 5] = 0;
    } else {
      for (uint64_t i = threadIdx.x; i < a.vis_words; i += TPB) vis[i] = 0;
    }
    __syncthreads();
  }
}

// ---- host side -----------------------------------------------------------------------------
size_t hnsw_lds_bytes(uint32_t cap, uint32_t nbmax, uint32_t dim, uint32_t words, int metric) {
  size_t s = (size_t)cap * 8 + (size_t)nbmax * 8 + 16 + (((size_t)cap + 15) & ~(size_t)15);
  if (metric == kHamming || metric == kJaccard)
    s += (size_t)words * 4;
  else if (sweep_cpl_for_dim(dim) == 0)
    s += (size_t)((dim + 3) / 4) * 16;
  return (s + 15) & ~(size_t)15;
}

// latency mode: one 1 024-thread block per query (at most one query per CU per call)
#undef ef
template <int METRIC, int CPL, bool VIS>
static hipError_t launch_lat_v(const HnswSearchArgs& a, int slots, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hnsw_search_kernel<METRIC, CPL, kSearchRegSlots, true, VIS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((hnsw_search_kernel<METRIC, CPL, kSearchRegSlots, true, VIS>), dim3(slots), dim3(1024), lds, st, a);
  return hipGetLastError();
}
template <int METRIC, int CPL>
static hipError_t launch_lat(const HnswSearchArgs& a, int slots, size_t lds, hipStream_t st) {
  return a.vis_log2 ? launch_lat_v<METRIC, CPL, true>(a, slots, lds, st) : launch_lat_v<METRIC, CPL, false>(a, slots, lds, st);
}
template <int METRIC, int CPL, int NS, bool VIS>
static hipError_t launch_ns_v(const HnswSearchArgs& a, int slots, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hnsw_search_kernel<METRIC, CPL, NS, false, VIS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  // resident blocks per CU of THIS instantiation (registers / LDS): a grid larger than what is resident would
  // queue whole blocks behind the persistent ones
  int occ = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, hnsw_search_kernel<METRIC, CPL, NS, false, VIS>, 256, lds);
  if (e != hipSuccess) return e;
  occ = std::max(1, std::min(occ, 4));
  const int grid = (int)std::min<int64_t>((int64_t)slots, (int64_t)a.n_cus * occ);
  hipLaunchKernelGGL((hnsw_search_kernel<METRIC, CPL, NS, false, VIS>), dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}
template <int METRIC, int CPL, int NS>
static hipError_t launch_ns(const HnswSearchArgs& a, int slots, size_t lds, hipStream_t st) {
  return a.vis_log2 ? launch_ns_v<METRIC, CPL, NS, true>(a, slots, lds, st) : launch_ns_v<METRIC, CPL, NS, false>(a, slots, lds, st);
}
// list_slots: 0 = LDS list (any ef), kSearchRegSlots = register list (ef + 64 <= kSearchRegSlots * 64)
template <int METRIC, int CPL>
static hipError_t launch_t(const HnswSearchArgs& a, int slots, size_t lds, hipStream_t st) {
  if (a.list_slots == kSearchRegSlots) return launch_ns<METRIC, CPL, kSearchRegSlots>(a, slots, lds, st);
  return launch_ns<METRIC, CPL, 0>(a, slots, lds, st);
}
template <int METRIC>
static hipError_t launch_cpl(const HnswSearchArgs& a, int slots, size_t lds, hipStream_t st) {
  switch (sweep_cpl_for_dim(a.dim)) {
    case 1: return launch_t<METRIC, 1>(a, slots, lds, st);
    case 2: return launch_t<METRIC, 2>(a, slots, lds, st);
    case 3: return launch_t<METRIC, 3>(a, slots, lds, st);
    case 4: return launch_t<METRIC, 4>(a, slots, lds, st);
    default: return launch_t<METRIC, 0>(a, slots, lds, st);
  }
}

template <int METRIC>
static hipError_t launch_lat_cpl(const HnswSearchArgs& a, int slots, size_t lds, hipStream_t st) {
  switch (sweep_cpl_for_dim(a.dim)) {
    case 1: return launch_lat<METRIC, 1>(a, slots, lds, st);
    case 2: return launch_lat<METRIC, 2>(a, slots, lds, st);
    case 3: return launch_lat<METRIC, 3>(a, slots, lds, st);
    case 4: return launch_lat<METRIC, 4>(a, slots, lds, st);
    default: return launch_lat<METRIC, 0>(a, slots, lds, st);
  }
}
// VELESDB_HNSW_LATENCY_MODE=0 keeps the throughput kernel for every call (A / B measurements), =2 takes the latency-mode kernel
// whatever the corpus size (fuzzing it over small graphs: tools/fuzz_hnsw.py)
static const int g_hnsw_lat = [] {
  const char* e = probe_env("VELESDB_HNSW_LATENCY_MODE");
  return e ? atoi(e) : 1;
}();
static const uint32_t g_hnsw_lat_max = [] {  // 0: one query per CU (the default); VELESDB_HNSW_LATENCY_MAX_QUERIES overrides
  const char* e = probe_env("VELESDB_HNSW_LATENCY_MAX_QUERIES");
  return e ? (uint32_t)atoi(e) : 0u;
}();

// VELESDB_HNSW_PREFETCH_IDS: 0 = every pop requests its own neighbour list, 1 = the latency-mode walk always asks for the predicted
// next pop's list too, unset = the measured default: only over a corpus beyond the Infinity Cache (1 M x 768: 1 102 -> 1 091 us per
// one-query call at a 48 % hit rate; 10 K x 768, cache-resident: 659 -> 671 us at 55 % — profiles/r04q2_*)
static const int g_hnsw_pf = [] {
  const char* e = probe_env("VELESDB_HNSW_PREFETCH_IDS");
  return e ? atoi(e) : -1;
}();
// VELESDB_HNSW_VIS_LDS: 0 = HBM bitmaps everywhere, 1 = the exact LDS set in the throughput kernel too (two blocks per CU
// instead of four), unset = the measured default (see pick_vis)
static const int g_hnsw_vis = [] {
  const char* e = probe_env("VELESDB_HNSW_VIS_LDS");
  return e ? atoi(e) : -1;
}();
// the LDS visited set of a launch: 2^log2 entries behind the kernel's other LDS, or none.  A walk at ef visits ~75 ef nodes
// (1 M x 768, M0 = 64: 9 570 at ef 128) and the kernel stops a query at 3/4 of the table (the caller re-runs it on the bitmap):
// offered where that is unlikely.
static uint32_t pick_vis(const HnswSearchArgs& a, size_t lds, bool lat) {
  if (!a.vis_log2 || g_hnsw_vis == 0) return 0;  // (a.vis_log2 != 0 on entry: the caller allows it — not a re-run)
  if (!lat && g_hnsw_vis != 1) return 0;
  for (uint32_t lg = lat ? 15u : 14u; lg >= 14u; lg--) {
    const uint64_t limit = (3ull << lg) / 4;
    if ((uint64_t)a.ef * 90 + a.nbmax <= limit && lds + ((size_t)4 << lg) <= 160 * 1024) return lg;
  }
  return 0;
}

template <int METRIC>
static hipError_t launch_raw_ef(const HnswSearchArgs& a, int slots, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hnsw_search_kernel<METRIC, 0, 0, false, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  int occ = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, hnsw_search_kernel<METRIC, 0, 0, false, false, true>, 256, lds);
  if (e != hipSuccess) return e;
  occ = std::max(1, std::min(occ, 4));
  const int grid = (int)std::min<int64_t>((int64_t)slots, (int64_t)a.n_cus * occ);
  hipLaunchKernelGGL((hnsw_search_kernel<METRIC, 0, 0, false, false, true>), dim3(grid), dim3(256), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_hnsw_search(const HnswSearchArgs& a0, int slots, hipStream_t st) {
  HnswSearchArgs a = a0;
  size_t lds = hnsw_lds_bytes(a.cap, a.nbmax, a.dim, a.words, a.metric);
  if (a.raw_small_ef) {  // NativeHnsw-level call with ef_search < 4: the generic instance that raises ef to the entry points per query
    a.vis_log2 = 0;      // (HBM bitmaps)
    a.pf_ids = 0;
    switch (a.metric) {
      case kCosine: return launch_raw_ef<kCosine>(a, slots, lds, st);
      case kEuclidean: return launch_raw_ef<kEuclidean>(a, slots, lds, st);
      case kDot: return launch_raw_ef<kDot>(a, slots, lds, st);
      case kHamming: return launch_raw_ef<kHamming>(a, slots, lds, st);
      default: return launch_raw_ef<kJaccard>(a, slots, lds, st);
    }
  }
  // at most one query per CU (measured at 1 M x 768, ef 128: 64 queries 1.57 ms against 2.68 ms on the throughput kernel, 256
  // queries 2.12 against 3.21 ms — a 1 024-thread block per CU is all the chip holds of this kernel) over a corpus that does not
  // sit in the 256 MB Infinity Cache: the latency-mode kernel (f32 metrics,
  // register list, layer-0 lists of <= 64 neighbours).  Over a cache-resident corpus the walk is not latency-bound the same way
  // (10 K x 768: 408 us per query on the throughput kernel, 599 us in latency mode, whose speculation fetches visited rows too)
  // (round 3: with the visited set in LDS the latency-mode kernel also serves cache-resident corpora — WITHOUT the speculation:
  // test first, fetch the unvisited; VELESDB_HNSW_LATENCY_MODE=2 forces the speculative form, =3 the other one)
  const bool beyond_cache = (uint64_t)a.n_rows * a.row_stride * 4 >= (256ull << 20);
  a.pf_ids = g_hnsw_pf >= 0 ? (g_hnsw_pf ? 1u : 0u) : (beyond_cache ? 1u : 0u);
  const uint32_t lat_vis = pick_vis(a0, lds, true);
  // Layer-0 lists of more than 64 neighbours (HnswParams::for_dataset_size, params.rs:72-147: M 128 => M0 256 from 10 001 vectors of
  // more than 256 dimensions up): the speculative step holds one neighbour per lane of the leader wave, so such graphs take the
  // latency-mode kernel in its test-first form — the LDS set answers in a few cycles, the 16 waves then fetch only the unvisited
  // neighbours' rows (at M0 256 most of a list is already visited: fetching every row would be 768 KB per expansion) — and only
  // when that set fits (otherwise the throughput kernel: same walk, four waves).
  const bool wide_lists = a.layers[0].stride > 64;
  if (g_hnsw_lat && a.nq <= (g_hnsw_lat_max ? g_hnsw_lat_max : a.n_cus) && (g_hnsw_lat >= 2 || beyond_cache || lat_vis) && a.list_slots == kSearchRegSlots && a.rerank_k == 0 &&
      (!wide_lists || lat_vis) && a.nbmax >= 64 && (a.metric == kCosine || a.metric == kEuclidean || a.metric == kDot)) {
    a.lat_spec = wide_lists ? 0u : (g_hnsw_lat == 2 ? 1u : (g_hnsw_lat == 3 ? 0u : (beyond_cache ? 1u : 0u)));
    a.vis_log2 = pick_vis(a0, lds, true);
    a.vis_off = (uint32_t)lds;
    if (a.vis_log2) lds += (size_t)4 << a.vis_log2;
    switch (a.metric) {
      case kCosine: return launch_lat_cpl<kCosine>(a, (int)a.nq, lds, st);
      case kEuclidean: return launch_lat_cpl<kEuclidean>(a, (int)a.nq, lds, st);
      default: return launch_lat_cpl<kDot>(a, (int)a.nq, lds, st);
    }
  }
  a.vis_log2 = pick_vis(a0, lds, false);
  a.vis_off = (uint32_t)lds;
  if (a.vis_log2) lds += (size_t)4 << a.vis_log2;
  switch (a.metric) {
    case kCosine: return launch_cpl<kCosine>(a, slots, lds, st);
    case kEuclidean: return launch_cpl<kEuclidean>(a, slots, lds, st);
    case kDot: return launch_cpl<kDot>(a, slots, lds, st);
    case kHamming: return launch_t<kHamming, 0>(a, slots, lds, st);
    default: return launch_t<kJaccard, 0>(a, slots, lds, st);
  }
}

// one visited bitmap (capacity bits) + one id log per resident block, zero between launches.  want_slots = the blocks the
// coming launch runs (0: as many as the chip holds of any walk kernel): a search context that only ever serves small calls — the
// combining front's batches, search_front.hip — keeps 125 KB + 64 KB per query in flight at 1 M rows instead of 0.4 GB
int32_t ensure_traversal_scratch(vdb_hip_index* ix, hipStream_t st, int want_slots) {
  const uint64_t vis_words = (ix->capacity + 31) / 32;
  const int chip_slots = ix->n_cus * kTraversalSlotsPerCu;
  int max_slots = want_slots > 0 ? std::min(want_slots, chip_slots) : chip_slots;
  if (max_slots < chip_slots) max_slots = std::min(chip_slots, (max_slots + 63) / 64 * 64);
  if (ix->vis_words == vis_words) max_slots = std::max<int>(max_slots, (int)std::min<size_t>((size_t)chip_slots, ix->s_visited.cap / std::max<uint64_t>(vis_words * 4, 1)));  // (never shrinks)
  if (ix->vis_words != vis_words || ix->s_visited.cap < (size_t)max_slots * vis_words * 4) {
    hipError_t e = ix->s_visited.reserve((size_t)max_slots * vis_words * 4, false, st);
    if (e == hipSuccess) e = ix->s_vlog.reserve((size_t)max_slots * kVlogCap * 4, false, st);
    if (e == hipSuccess) e = ix->s_stats.reserve(64, false, st);
    if (e == hipSuccess) e = hipMemsetAsync(ix->s_visited.p, 0, ix->s_visited.cap, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return fail(VDB_ERR_OOM, std::string("visited scratch: ") + hipGetErrorString(e));
    ix->vis_words = vis_words;
  }
  return VDB_OK;
}

// NativeHnsw::search for nq device-resident queries (graph.rs:251-270) + result mapping
// (search.rs:79-93).  Enqueues on `st`; no host synchronisation.
int32_t hnsw_search_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k, uint32_t ef,
                        uint32_t cap_mult, uint64_t* d_ids, float* d_scores, uint32_t* d_n, hipStream_t st,
                        uint32_t rerank_k, const uint32_t* d_extra_eps) {
  if (!ix->graph_valid) return fail(VDB_ERR_STATE, "HNSW graph not built for all rows (use mode BRUTE or build it)");
  if (nq == 0) return VDB_OK;
  if (ix->entry_point < 0 || ix->graph_nodes == 0 || k == 0) {  // graph.rs:252-255: no entry point => empty
    VDB_HIP(hipMemsetAsync(d_n, 0, (size_t)nq * 4, st));
    return VDB_OK;
  }
  if (ix->layers.size() > (size_t)kMaxLayers) return fail(VDB_ERR_UNSUPPORTED, "more than 16 graph layers");
  HnswSearchArgs a{};
  uint32_t nbmax = 0;
  for (size_t l = 0; l < ix->layers.size(); l++) {
    a.layers[l].nbr = ix->layers[l].nbr.as<uint32_t>();
    a.layers[l].cnt = ix->layers[l].cnt.as<uint32_t>();
    a.layers[l].stride = ix->layers[l].stride;
    nbmax = std::max(nbmax, ix->layers[l].stride);
  }
  nbmax = (std::max(nbmax, rerank_k) + 63) / 64 * 64;
  // list capacity: ef results + room for evicted candidates that tie with the furthest result
  uint64_t cap = (uint64_t)ef + std::max<uint64_t>(64, (uint64_t)ef * cap_mult / 2);
  cap = (cap + 63) / 64 * 64;
  // small ef: the list lives in registers (first attempt only; an overflow re-run uses the larger LDS list)
  const bool raw_small = ix->raw_ef && ef < 4;  // (search_multi_entry: more entry points than ef_search are possible)
  const bool reg_list = !raw_small && cap_mult == 1 && rerank_k == 0 && (uint64_t)ef + 64 <= (uint64_t)kSearchRegSlots * 64 &&
                        ix->n_rows < (1ull << 31);
  if (reg_list) cap = (uint64_t)kSearchRegSlots * 64;
  const size_t lds = hnsw_lds_bytes((uint32_t)cap, nbmax, ix->dim, ix->words, ix->metric);
  if (cap > 0xFFFFFFFFull || lds > 160 * 1024)
    return fail(VDB_ERR_UNSUPPORTED, "ef too large for the LDS-resident candidate list (" + std::to_string(lds) + " B)");
  // slots: resident blocks; 4 per CU unless LDS limits it
  int per_cu = (int)std::min<size_t>(4, std::max<size_t>(1, (160 * 1024) / lds));
  int slots = (int)std::min<int64_t>((int64_t)nq, (int64_t)ix->n_cus * per_cu);
  int32_t rcs = ensure_traversal_scratch(ix, st, slots);
  if (rcs != VDB_OK) return rcs;
  const uint64_t vis_words = ix->vis_words;
  const uint32_t vlog_cap = kVlogCap;
  VDB_HIP(hipMemsetAsync(ix->s_stats.p, 0, 24, st));
  a.rows = ix->rows.as<float>();
  a.norms = ix->norms.as<float>();
  a.bits = ix->bits.as<uint32_t>();
  a.alive = ix->any_dead ? ix->alive.as<uint8_t>() : nullptr;
  a.ext_ids = ix->ext_ids.as<uint64_t>();
  a.queries = d_q;
  a.row_stride = ix->row_stride;
  a.q_stride = q_stride;
  a.visited = ix->s_visited.as<uint32_t>();
  a.vlog = ix->s_vlog.as<uint32_t>();
  a.vis_words = vis_words;
  a.out_ids = d_ids;
  a.out_scores = d_scores;
  a.out_n = d_n;
  a.stats = ix->s_stats.as<unsigned long long>();
  a.dim = ix->dim;
  a.words = ix->words;
  a.n_rows = (uint32_t)ix->n_rows;
  a.nq = nq;
  a.k = k;
  a.ef = ef;
  a.cap = (uint32_t)cap;
  a.nbmax = nbmax;
  a.vlog_cap = vlog_cap;
  a.max_layer = ix->max_layer;
  a.entry_point = (uint32_t)ix->entry_point;
  a.metric = ix->metric;
  a.rerank_k = rerank_k;
  a.extra_eps = d_extra_eps;
  a.raw_small_ef = raw_small ? 1u : 0u;
  a.list_slots = reg_list ? kSearchRegSlots : 0;
  a.vis_log2 = cap_mult == 1 ? 1u : 0u;  // "the LDS visited set is allowed" (a re-run after an overflow takes the bitmap)
  a.n_cus = (uint32_t)ix->n_cus;
  EventPair* ev = next_events(ix);
  if (ev) (void)hipEventRecord(ev->a, st);
  hipError_t e = launch_hnsw_search(a, slots, st);
  if (ev) (void)hipEventRecord(ev->b, st);
  if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("hnsw_search launch: ") + hipGetErrorString(e));
  ix->stats_pending = true;
  return VDB_OK;
}

}  // namespace vdb
