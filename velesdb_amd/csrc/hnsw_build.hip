// hnsw_build.hip — graph construction on the GPU: NativeHnsw::insert (native/graph.rs:158-237) with
// select_neighbors (graph.rs:526-581) and add_bidirectional_connection (graph.rs:592-639).
//
// One algorithm, two uses:
//   * batch of ONE node  = the reference's sequential insert, link for link (VectorIndex::insert,
//     insert_batch_sequential, batch.rs:128-149);
//   * batch of B nodes   = batch-synchronous insertion, the deterministic stand-in for the reference's
//     rayon parallel_insert (backend_adapter.rs:110-123, batch.rs:83-108): all B nodes search the graph
//     as it was before the batch, then links are applied target by target with sources in ascending
//     node order.  oracle/vdb_oracle.cpp hnsw_insert_batch_sync restates exactly this, so the result
//     is checked link for link as well.
//
// Kernel A (hnsw_insert_kernel): one 256-thread block per new node.  Greedy descent to the node's level,
//   then per layer: search_layer(ef_construction) with the traversal kernel's machinery
//   (vdb_hnsw_device.hpp), select_neighbors, write the node's own list and emit one link request per
//   selected neighbour.  select_neighbors is evaluated "by selected node": when a candidate is selected,
//   ONE distance phase evaluates it against every still-undecided later candidate; a candidate c falls
//   as soon as alpha*d(q,c) <= d(c,s) fails for a selected s.  Same decisions and the same number of
//   distance evaluations as the reference's per-candidate loop with its short-circuit `all()`, but
//   batched into at most max_conn phases of up to ef_construction rows instead of ef_construction
//   phases of up to max_conn rows.
// Kernel B (hnsw_link_kernel): requests sorted by (layer, target, source); one wave per target applies
//   its requests in order: append while the list has room, otherwise stable-sort by distance and keep
//   the max_conn closest (graph.rs:604-637).  The distances to a node's current neighbours are CACHED
//   next to the lists (every one of them was produced by a search: d(a,b) and d(b,a) are bit-identical
//   in the canonical arithmetic), so pruning needs no vector reads at all: the reference recomputes
//   max_conn+1 distances per full back-link, 4 160 row reads per inserted node at M0 = 64.
#include <algorithm>
#include <cmath>

#include <cstring>
#include <limits>


#include "vdb_hnsw_device.hpp"
#include "vdb_index.hpp"

namespace vdb {

struct HnswLayerMut {
  uint32_t* nbr;
  uint32_t* cnt;
  float* ndist;  // [capacity][stride] distance node <-> neighbour, same slots as nbr
  uint32_t stride;
  uint32_t pad;
};

struct HnswInsertArgs {
  DistCtx dc;
  HnswLayerMut layers[kMaxLayers];
  const uint8_t* levels;  // [B] level of node first + b
  uint32_t* visited;
  uint32_t* vlog;
  uint64_t vis_words;
  uint64_t* req_keys;  // link requests: (layer << 52 | target << 20 | batch index)
  uint64_t* req_vals;  //                (source << 32 | f32 bits of the distance)
  uint32_t* req_n;
  uint32_t* overflow;  // [1] set when a candidate list or the request buffer overflowed
  unsigned long long* stats;  // nullable; [0] += rows evaluated (search_layer + select_neighbors distance phases), [1] += distance
                              // phases (dependent memory round trips), [2] += nodes inserted, [3] += the rows of [0] that were
                              // select_neighbors evaluations (re-reads of <= ef_construction candidate rows: cache hits): the
                              // construction roofline's counters
  uint32_t first, B, ef, cap, nbmax, vlog_cap, max_layer, entry_point, req_cap;
  float alpha;
};

struct HnswLinkArgs {
  HnswLayerMut layers[kMaxLayers];
  const uint64_t* keys;
  const uint64_t* vals;
  uint32_t n;           // entries to look at (unused ones hold ~0)
  uint32_t singletons;  // 1: every request is its own group (batch of one node, unsorted input)
};

namespace {

enum BPhase : int {
  B_START = 0, B_G_ENTRY, B_G_LOAD, B_G_SCAN, B_G_DONE, B_Z_ENTRY, B_Z_POP, B_Z_ADMIT,
  B_S_BEGIN, B_S_NEXT, B_S_MARK, B_S_FILL, B_S_WRITE, B_L_NEXT, B_FINISH
};
enum Cmd : uint32_t { CMD_DIST = 0, CMD_DONE = 1, CMD_CLEAN = 2 };
constexpr uint32_t kNone = 0xFFFFFFFFu;

}  // namespace

// LDS: keys[cap] u64 | nb_id, nb_d, nbx, sel [nbmax] u32 each | ctl[8] u32 | flags[cap] u8 (pad 16) | query scratch
template <int METRIC, int CPL>
__global__ __launch_bounds__(256) void hnsw_insert_kernel(HnswInsertArgs a) {
  constexpr bool BITS = (METRIC == kHamming || METRIC == kJaccard);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = (int)rfl(threadIdx.x >> 6);
  const uint32_t cap = a.cap, nbmax = a.nbmax, ef = a.ef;
  lds_vu64* keys = (lds_vu64*)(lds_void_p)(smem);
  lds_vu32* nb_id = (lds_vu32*)(lds_void_p)(smem + (size_t)cap * 8);
  lds_vf32* nb_d = (lds_vf32*)(nb_id + nbmax);
  lds_vu32* nbx = nb_id + 2 * (size_t)nbmax;
  lds_vu32* sel = nb_id + 3 * (size_t)nbmax;
  lds_vu32* ctl = nb_id + 4 * (size_t)nbmax;
  lds_vu8* flags = (lds_vu8*)(ctl + 8);
  const size_t qoff = (size_t)cap * 8 + (size_t)nbmax * 16 + 32 + (((size_t)cap + 15) & ~(size_t)15);
  float* qgen = reinterpret_cast<float*>(smem + qoff);
  uint32_t* qbits = reinterpret_cast<uint32_t*>(smem + qoff);

  uint32_t* vis = a.visited + (size_t)blockIdx.x * a.vis_words;
  uint32_t* vlog = a.vlog + (size_t)blockIdx.x * a.vlog_cap;
  const DistCtx& dc = a.dc;
  const int d4 = (int)((dc.dim + 3) / 4);

  for (uint32_t bi = blockIdx.x; bi < a.B; bi += gridDim.x) {
    const uint32_t x = a.first + bi;
    const int lx = (int)a.levels[bi];
    float4 q[CPL > 0 ? CPL : 1];
    float qnorm = 0.0f;
    uint32_t qrow_loaded = kNone;

    uint32_t cnt = 0, logn = 0, m_prev = 0, selc = 0, spos = 0, ssize = 0;
    if (threadIdx.x == 0) {
      ctl[4] = 0;
      ctl[5] = 0;
      ctl[6] = 0;
    }
    int phase = B_START;
    int layer = max((int)a.max_layer, lx);
    uint32_t cur = a.entry_point;
    float best_d = 0.0f;

    for (;;) {
      if (wib == 0) {
        bool ready = false;
        uint32_t m = 0, cmd = CMD_DIST, qrow = x;
        while (!ready) {
          if (phase == B_START) {
            if (lane == 0) nb_id[0] = cur;
            m = 1;
            ready = true;
            phase = layer > lx ? B_G_ENTRY : B_Z_ENTRY;
          } else if (phase == B_G_ENTRY) {
            best_d = rflf(nb_d[0]);
            phase = B_G_LOAD;
          } else if (phase == B_G_LOAD) {
            const HnswLayerMut L = a.layers[layer];
            uint32_t nc = rfl(L.cnt[cur]);
            nc = min(nc, min(L.stride, nbmax));
            for (uint32_t base = 0; base < nc; base += 64) {
              const uint32_t t = base + lane;
              if (t < nc) nb_id[t] = L.nbr[(size_t)cur * L.stride + t];
            }
            if (nc == 0) {
              phase = B_G_DONE;
            } else {
              m = nc;
              ready = true;
              phase = B_G_SCAN;
            }
          } else if (phase == B_G_SCAN) {
            float mn = 0.0f;
            uint32_t besti = kNone;
            for (uint32_t base = 0; base < m_prev; base += 64) {
              const uint32_t t = base + lane;
              const float d = t < m_prev ? nb_d[t] : 0.0f;
              const bool ok = t < m_prev && d < best_d;
              const uint64_t okm = __ballot(ok);
              if (okm) {
                float v = ok ? d : __uint_as_float(0x7F800000u);
#pragma unroll
                for (int s = 32; s >= 1; s >>= 1) v = fminf(v, shx(v, s));
                v = rflf(v);
                if (besti == kNone || v < mn) {
                  const uint64_t eq = __ballot(ok && d == v);
                  besti = base + (uint32_t)__ffsll((long long)eq) - 1;
                  mn = v;
                }
              }
            }
            if (besti != kNone) {
              cur = rfl(nb_id[besti]);
              best_d = rflf(nb_d[besti]);
              phase = B_G_LOAD;
            } else {
              phase = B_G_DONE;
            }
          } else if (phase == B_G_DONE) {
            layer -= 1;
            phase = B_START;
          } else if (phase == B_Z_ENTRY) {
            const float d = rflf(nb_d[0]);
            uint64_t dr;
            uint32_t df;
            cnt = 0;
            list_insert(keys, flags, cnt, cap, make_key<false>(d, cur), lane, dr, df);
            if (lane == 0) {
              atomicOr(&vis[cur >> 5], 1u << (cur & 31));
              if (a.vlog_cap) vlog[0] = cur;
            }
            logn = 1;
            phase = B_Z_POP;
          } else if (phase == B_Z_POP) {
            uint32_t idx = kNone;
            for (uint32_t c = 0; c < cnt; c += 64) {
              const uint32_t e = c + lane;
              const uint64_t un = __ballot(e < cnt && flags[e] == 0);
              if (un) {
                idx = c + (uint32_t)__ffsll((long long)un) - 1;
                break;
              }
            }
            bool stop = idx == kNone;
            uint64_t ckey = 0;
            if (!stop) {
              ckey = keys[idx];
              if (cnt >= ef) stop = key_dist(ckey) > key_dist(keys[ef - 1]);  // graph.rs:474
            }
            if (stop) {
              phase = B_S_BEGIN;
            } else {
              if (lane == 0) flags[idx] = 1;
              const uint32_t cnode = (uint32_t)ckey;
              const HnswLayerMut L = a.layers[layer];
              // the neighbour ids are requested together with the count (one memory round trip instead of two)
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
                  newly = (atomicOr(&vis[nb >> 5], bit) & bit) == 0;  // visited.insert (graph.rs:499)
                }
                const uint64_t mask = __ballot(newly);
                const uint32_t before = (uint32_t)__popcll(mask & lt_mask(lane));
                if (newly) {
                  nb_id[m + before] = nb;
                  if (logn + before < a.vlog_cap) vlog[logn + before] = nb;
                }
                m += (uint32_t)__popcll(mask);
                logn += (uint32_t)__popcll(mask);
              }
              if (m != 0) {
                ready = true;
                phase = B_Z_ADMIT;
              }
            }
          } else if (phase == B_Z_ADMIT) {
            for (uint32_t base = 0; base < m_prev; base += 64) {
              const uint32_t t = base + lane;
              const float d = t < m_prev ? nb_d[t] : 0.0f;
              uint32_t size = cnt < ef ? cnt : ef;
              float far = key_dist(keys[size - 1]);
              uint64_t mask = __ballot(t < m_prev && (d < far || size < ef));
              while (mask) {
                const int src = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const float dj = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d), src));
                size = cnt < ef ? cnt : ef;
                far = key_dist(keys[size - 1]);
                if (dj < far || size < ef) {  // graph.rs:503
                  const uint32_t nbj = nb_id[base + src];
                  uint64_t dr;
                  uint32_t df;
                  list_insert(keys, flags, cnt, cap, make_key<false>(dj, nbj), lane, dr, df);
                  if (dr != kKeyInvalid && df == 0 && lane == 0) atomicOr(a.overflow, 1u);
                  list_truncate(keys, cnt, ef, lane);
                }
              }
            }
            phase = B_Z_POP;
          } else if (phase == B_S_BEGIN) {
            // ---- select_neighbors (graph.rs:526-581) over the sorted result set keys[0..ssize) ----
            // flags: 0 undecided, 1 rejected by the diversity test, 2 selected
            ssize = cnt < ef ? cnt : ef;
            const uint32_t maxc = a.layers[layer].stride;
            selc = 0;
            spos = 0;
            if (ssize <= maxc) {  // graph.rs:536-538: few candidates, take all
              for (uint32_t e = lane; e < ssize; e += 64) sel[e] = e;
              selc = ssize;
              phase = B_S_WRITE;
            } else {
              for (uint32_t e = lane; e < ssize; e += 64) flags[e] = 0;
              phase = B_S_NEXT;
            }
          } else if (phase == B_S_NEXT) {
            const uint32_t maxc = a.layers[layer].stride;
            uint32_t e0 = kNone;
            for (uint32_t c = spos & ~63u; c < ssize; c += 64) {
              const uint32_t e = c + lane;
              const uint64_t un = __ballot(e >= spos && e < ssize && flags[e] == 0);
              if (un) {
                e0 = c + (uint32_t)__ffsll((long long)un) - 1;
                break;
              }
            }
            if (e0 == kNone) {
              phase = B_S_FILL;
            } else {
              if (lane == 0) {
                flags[e0] = 2;
                sel[selc] = e0;
              }
              selc += 1;
              spos = e0 + 1;
              if (selc == maxc) {
                phase = B_S_WRITE;
              } else {
                // the undecided later candidates are evaluated against the newly selected one
                for (uint32_t c = spos & ~63u; c < ssize; c += 64) {
                  const uint32_t e = c + lane;
                  const bool un = e >= spos && e < ssize && flags[e] == 0;
                  const uint64_t mask = __ballot(un);
                  const uint32_t p = m + (uint32_t)__popcll(mask & lt_mask(lane));
                  if (un) {
                    nb_id[p] = (uint32_t)keys[e];
                    nbx[p] = e;
                  }
                  m += (uint32_t)__popcll(mask);
                }
                if (m == 0) {
                  phase = B_S_FILL;
                } else {
                  qrow = (uint32_t)keys[e0];
                  ready = true;
                  phase = B_S_MARK;
                }
              }
            }
          } else if (phase == B_S_MARK) {
            for (uint32_t j = lane; j < m_prev; j += 64) {
              const uint32_t e = nbx[j];
              const float cd = key_dist(keys[e]);
              if (!(a.alpha * cd <= nb_d[j])) flags[e] = 1;  // graph.rs:553: fails the diversity test
            }
            phase = B_S_NEXT;
          } else if (phase == B_S_FILL) {
            // graph.rs:569-578: under quota -> back-fill with the closest unselected candidates
            const uint32_t maxc = a.layers[layer].stride;
            for (uint32_t c = 0; c < ssize && selc < maxc; c += 64) {
              const uint32_t e = c + lane;
              const bool un = e < ssize && flags[e] != 2;
              const uint64_t mask = __ballot(un);
              const uint32_t p = selc + (uint32_t)__popcll(mask & lt_mask(lane));
              if (un && p < maxc) sel[p] = e;
              selc = min(maxc, selc + (uint32_t)__popcll(mask));
            }
            phase = B_S_WRITE;
          } else if (phase == B_S_WRITE) {
            const HnswLayerMut L = a.layers[layer];
            uint32_t base = 0;
            if (lane == 0 && selc) base = atomicAdd(a.req_n, selc);
            base = rfl(base);
            const bool fits = base + selc <= a.req_cap;
            if (!fits && lane == 0) atomicOr(a.overflow, 2u);
            for (uint32_t j = lane; j < selc; j += 64) {
              const uint64_t key = keys[sel[j]];
              const uint32_t node = (uint32_t)key;
              const float d = key_dist(key);
              L.nbr[(size_t)x * L.stride + j] = node;      // set_neighbors (graph.rs:213)
              L.ndist[(size_t)x * L.stride + j] = d;
              if (fits) {
                a.req_keys[base + j] = ((uint64_t)layer << 52) | ((uint64_t)node << 20) | (uint64_t)bi;
                a.req_vals[base + j] = ((uint64_t)x << 32) | (uint64_t)__float_as_uint(d);
              }
            }
            if (lane == 0) L.cnt[x] = selc;
            if (ssize) cur = (uint32_t)keys[0];  // graph.rs:220-222: closest found feeds the next layer
            cmd = CMD_CLEAN;
            ready = true;
            phase = B_L_NEXT;
          } else if (phase == B_L_NEXT) {
            cnt = 0;
            logn = 0;
            if (layer == 0) {
              phase = B_FINISH;
            } else {
              layer -= 1;
              phase = B_START;
            }
          } else {  // B_FINISH
            cmd = CMD_DONE;
            ready = true;
          }
        }
        if (lane == 0) {
          ctl[0] = m;
          ctl[1] = cmd;
          ctl[2] = logn;
          ctl[3] = qrow;
        }
        m_prev = m;
      }
      __syncthreads();
      const uint32_t m = ctl[0];
      const uint32_t cmd = ctl[1];
      if (cmd == CMD_DONE) break;
      if (cmd == CMD_DIST && threadIdx.x == 0 && a.stats) {  // the roofline's counters live in LDS (two registers more cost the kernel a wave per SIMD)
        ctl[4] += m;
        ctl[5] += 1;
        if (ctl[3] != x) ctl[6] += m;  // a select_neighbors phase: the rows are evaluated against a SELECTED neighbour, not the new node
      }
      if (cmd == CMD_CLEAN) {
        const uint32_t nlog = ctl[2];
        if (nlog <= a.vlog_cap) {
          for (uint32_t i = threadIdx.x; i < nlog; i += 256) vis[vlog[i] >> 5] = 0;
        } else {
          for (uint64_t i = threadIdx.x; i < a.vis_words; i += 256) vis[i] = 0;
        }
        __syncthreads();
        continue;
      }
      const uint32_t qrow = ctl[3];
      if (qrow != qrow_loaded) {  // block-uniform
        if (BITS) {
          for (uint32_t w = threadIdx.x; w < dc.words; w += 256) qbits[w] = dc.bits[(size_t)qrow * dc.words + w];
          __syncthreads();
        } else if (CPL > 0) {
          const float* qp = dc.rows + (size_t)qrow * dc.row_stride;
#pragma unroll
          for (int c = 0; c < CPL; c++) q[c] = ld4(qp + (size_t)(c * 64 + lane) * 4);
          if (METRIC == kCosine) qnorm = dc.norms[qrow];
        } else {
          const float* qp = dc.rows + (size_t)qrow * dc.row_stride;
          for (int i = threadIdx.x; i < d4 * 4; i += 256) qgen[i] = qp[i];  // row padding is zero
          if (METRIC == kCosine) qnorm = dc.norms[qrow];
          __syncthreads();
        }
        qrow_loaded = qrow;
      }
      if (BITS)
        dist_phase_bits<METRIC>(dc, qbits, m, nb_id, nb_d);
      else
        dist_phase_f32<METRIC, CPL>(dc, q, qnorm, qgen, m, nb_id, nb_d, lane, wib);
      __syncthreads();
    }
    if (a.stats && threadIdx.x == 0) {
      atomicAdd(&a.stats[0], (unsigned long long)ctl[4]);
      atomicAdd(&a.stats[1], (unsigned long long)ctl[5]);
      atomicAdd(&a.stats[2], 1ull);
      atomicAdd(&a.stats[3], (unsigned long long)ctl[6]);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// Kernel B: add_bidirectional_connection (graph.rs:592-639) for every request, one wave per target.
// LDS per wave: (stride + 1) u64 sort keys.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hnsw_link_kernel(HnswLinkArgs a, uint32_t lds_per_wave) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = (int)rfl(threadIdx.x >> 6);
  lds_vu64* sk = (lds_vu64*)(lds_void_p)(smem + (size_t)wib * lds_per_wave);
  const uint32_t nwaves = gridDim.x * 4;
  for (uint32_t p = blockIdx.x * 4 + wib; p < a.n; p += nwaves) {
    const uint64_t key = a.keys[p];
    if (key == ~0ull) continue;
    if (!a.singletons && p > 0 && (a.keys[p - 1] >> 20) == (key >> 20)) continue;  // not the head of its group
    const uint32_t layer = (uint32_t)(key >> 52);
    const uint32_t t = (uint32_t)(key >> 20);
    const HnswLayerMut L = a.layers[layer];
    const uint32_t maxc = L.stride;
    uint32_t* ids = L.nbr + (size_t)t * maxc;
    float* ds = L.ndist + (size_t)t * maxc;
    uint32_t c = rfl(L.cnt[t]);
    for (uint32_t g = p; g < a.n; g++) {
      const uint64_t kg = a.keys[g];
      if (kg == ~0ull || (kg >> 20) != (key >> 20)) break;
      const uint64_t v = a.vals[g];
      const uint32_t src = (uint32_t)(v >> 32);
      const float d = __uint_as_float((uint32_t)v);
      if (c < maxc) {  // graph.rs:604-607
        if (lane == 0) {
          ids[c] = src;
          ds[c] = d;
        }
        c += 1;
      } else {
        // graph.rs:608-637: all = current + new, stable sort by distance (total order), keep max_conn.
        // sort key = (total-order(dist) << 32 | position): unique, so rank = number of smaller keys
        const uint32_t n = c + 1;
        for (uint32_t i = lane; i < n; i += 64) {
          const float di = i < c ? ds[i] : d;
          sk[i] = ((uint64_t)asc_key(di) << 32) | i;
        }
        // every lane owns elements lane, lane+64, ...; reads of ids/ds happen before any write below
        uint32_t my_id[5], my_rank[5];
        float my_d[5];
#pragma unroll
        for (int r = 0; r < 5; r++) {
          const uint32_t i = lane + 64 * r;
          my_rank[r] = kNone;
          if (i < n) {
            my_id[r] = i < c ? ids[i] : src;
            my_d[r] = i < c ? ds[i] : d;
            const uint64_t mine = sk[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < n; j++) rank += sk[j] < mine ? 1u : 0u;
            my_rank[r] = rank;
          }
        }
#pragma unroll
        for (int r = 0; r < 5; r++) {
          if (my_rank[r] < maxc) {
            ids[my_rank[r]] = my_id[r];
            ds[my_rank[r]] = my_d[r];
          }
        }
        c = maxc;
      }
      if (a.singletons) break;
    }
    if (lane == 0) L.cnt[t] = c;
  }
}

// ------------------------------------------------------------------------------------------
// distance cache for a graph that arrived without one (load_reference_files): ndist[node][j] =
// distance(node, nbr[node][j]).  One block per node, same distance phase as everything else.
// ------------------------------------------------------------------------------------------
struct HnswNdistArgs {
  DistCtx dc;
  HnswLayerMut layer;
  uint32_t n_nodes, nbmax;
};
template <int METRIC, int CPL>
__global__ __launch_bounds__(256) void hnsw_ndist_kernel(HnswNdistArgs a) {
  constexpr bool BITS = (METRIC == kHamming || METRIC == kJaccard);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id();
  const int wib = (int)rfl(threadIdx.x >> 6);
  lds_vu32* nb_id = (lds_vu32*)(lds_void_p)(smem);
  lds_vf32* nb_d = (lds_vf32*)(nb_id + a.nbmax);
  float* qgen = reinterpret_cast<float*>(smem + (size_t)a.nbmax * 8);
  uint32_t* qbits = reinterpret_cast<uint32_t*>(smem + (size_t)a.nbmax * 8);
  const DistCtx& dc = a.dc;
  const int d4 = (int)((dc.dim + 3) / 4);
  for (uint32_t node = blockIdx.x; node < a.n_nodes; node += gridDim.x) {
    const uint32_t c = min(a.layer.cnt[node], a.layer.stride);
    if (c == 0) continue;  // block-uniform
    for (uint32_t t = threadIdx.x; t < c; t += 256) nb_id[t] = a.layer.nbr[(size_t)node * a.layer.stride + t];
    float4 q[CPL > 0 ? CPL : 1];
    float qnorm = 0.0f;
    if (BITS) {
      for (uint32_t w = threadIdx.x; w < dc.words; w += 256) qbits[w] = dc.bits[(size_t)node * dc.words + w];
    } else if (CPL > 0) {
      const float* qp = dc.rows + (size_t)node * dc.row_stride;
#pragma unroll
      for (int cc = 0; cc < CPL; cc++) q[cc] = ld4(qp + (size_t)(cc * 64 + lane) * 4);
      if (METRIC == kCosine) qnorm = dc.norms[node];
    } else {
      const float* qp = dc.rows + (size_t)node * dc.row_stride;
      for (int i = threadIdx.x; i < d4 * 4; i += 256) qgen[i] = qp[i];
      if (METRIC == kCosine) qnorm = dc.norms[node];
    }
    __syncthreads();
    if (BITS)
      dist_phase_bits<METRIC>(dc, qbits, c, nb_id, nb_d);
    else
      dist_phase_f32<METRIC, CPL>(dc, q, qnorm, qgen, c, nb_id, nb_d, lane, wib);
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < c; t += 256) a.layer.ndist[(size_t)node * a.layer.stride + t] = nb_d[t];
    __syncthreads();
  }
}

// ---- host side -----------------------------------------------------------------------------
static size_t insert_lds_bytes(uint32_t cap, uint32_t nbmax, uint32_t dim, uint32_t words, int metric) {
  size_t s = (size_t)cap * 8 + (size_t)nbmax * 16 + 32 + (((size_t)cap + 15) & ~(size_t)15);
  if (metric == kHamming || metric == kJaccard)
    s += (size_t)words * 4;
  else if (sweep_cpl_for_dim(dim) == 0)
    s += (size_t)((dim + 3) / 4) * 16;
  return (s + 15) & ~(size_t)15;
}

template <int METRIC, int CPL>
static hipError_t launch_insert_t(const HnswInsertArgs& a, int slots, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hnsw_insert_kernel<METRIC, CPL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((hnsw_insert_kernel<METRIC, CPL>), dim3(slots), dim3(256), lds, st, a);
  return hipGetLastError();
}
template <int METRIC, int CPL>
static hipError_t launch_ndist_t(const HnswNdistArgs& a, int blocks, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((hnsw_ndist_kernel<METRIC, CPL>), dim3(blocks), dim3(256), lds, st, a);
  return hipGetLastError();
}

#define VDB_DISPATCH_METRIC_CPL(FN, metric, dim, ...)                                   \
  [&]() -> hipError_t {                                                                 \
    const int cpl_ = sweep_cpl_for_dim(dim);                                            \
    switch (metric) {                                                                   \
      case kCosine:                                                                     \
        switch (cpl_) {                                                                 \
          case 1: return FN<kCosine, 1>(__VA_ARGS__);                                   \
          case 2: return FN<kCosine, 2>(__VA_ARGS__);                                   \
          case 3: return FN<kCosine, 3>(__VA_ARGS__);                                   \
          case 4: return FN<kCosine, 4>(__VA_ARGS__);                                   \
          default: return FN<kCosine, 0>(__VA_ARGS__);                                  \
        }                                                                               \
      case kEuclidean:                                                                  \
        switch (cpl_) {                                                                 \
          case 1: return FN<kEuclidean, 1>(__VA_ARGS__);                                \
          case 2: return FN<kEuclidean, 2>(__VA_ARGS__);                                \
          case 3: return FN<kEuclidean, 3>(__VA_ARGS__);                                \
          case 4: return FN<kEuclidean, 4>(__VA_ARGS__);                                \
          default: return FN<kEuclidean, 0>(__VA_ARGS__);                               \
        }                                                                               \
      case kDot:                                                                        \
        switch (cpl_) {                                                                 \
          case 1: return FN<kDot, 1>(__VA_ARGS__);                                      \
          case 2: return FN<kDot, 2>(__VA_ARGS__);                                      \
          case 3: return FN<kDot, 3>(__VA_ARGS__);                                      \
          case 4: return FN<kDot, 4>(__VA_ARGS__);                                      \
          default: return FN<kDot, 0>(__VA_ARGS__);                                     \
        }                                                                               \
      case kHamming: return FN<kHamming, 0>(__VA_ARGS__);                               \
      default: return FN<kJaccard, 0>(__VA_ARGS__);                                     \
    }                                                                                   \
  }()

static void fill_layers(vdb_hip_index* ix, HnswLayerMut* out, uint32_t* max_stride) {
  uint32_t ms = 0;
  for (size_t l = 0; l < ix->layers.size() && l < (size_t)kMaxLayers; l++) {
    out[l].nbr = ix->layers[l].nbr.as<uint32_t>();
    out[l].cnt = ix->layers[l].cnt.as<uint32_t>();
    out[l].ndist = ix->layers[l].ndist.as<float>();
    out[l].stride = ix->layers[l].stride;
    ms = std::max(ms, ix->layers[l].stride);
  }
  if (max_stride) *max_stride = ms;
}

// native/graph.rs:368-403: xorshift64 + exponential level, capped at 15
static uint32_t next_level(uint64_t& state, double level_mult) {
  uint64_t s = state;
  if (s == 0) s = 0x853c49e6748fea9bULL;
  s ^= s << 13;
  s ^= s >> 7;
  s ^= s << 17;
  state = s;
  const double uniform = (double)s / (double)UINT64_MAX;
  const double safe = std::max(uniform, std::numeric_limits<double>::min());
  const double lv = std::floor(-std::log(safe) * level_mult);
  const uint64_t level = lv >= 18446744073709551615.0 ? UINT64_MAX : (uint64_t)lv;
  return (uint32_t)std::min<uint64_t>(level, 15);
}

uint32_t build_batch_size(uint64_t linked, uint32_t max_batch) {
  uint64_t b = linked / 16;
  if (b < 1) b = 1;
  if (b > max_batch) b = max_batch;
  return (uint32_t)b;
}

// distance cache of a loaded graph (one-off, before the first insert into it)
int32_t graph_fill_ndist(vdb_hip_index* ix) {
  if (ix->ndist_valid) return VDB_OK;
  hipStream_t st = ix->stream;
  for (size_t l = 0; l < ix->layers.size(); l++) {
    GraphLayer& L = ix->layers[l];
    hipError_t e = L.ndist.reserve(std::max<size_t>(ix->capacity * L.stride * 4, 4), false, st);
    if (e != hipSuccess) return fail(VDB_ERR_OOM, std::string("neighbour-distance cache: ") + hipGetErrorString(e));
  }
  if (ix->graph_nodes) {
    for (size_t l = 0; l < ix->layers.size(); l++) {
      HnswNdistArgs a{};
      a.dc = DistCtx{ix->rows.as<float>(), ix->norms.as<float>(), ix->bits.as<uint32_t>(), ix->row_stride, ix->dim,
                     ix->words};
      HnswLayerMut lm[kMaxLayers] = {};
      fill_layers(ix, lm, nullptr);
      a.layer = lm[l];
      a.n_nodes = (uint32_t)ix->graph_nodes;
      a.nbmax = (a.layer.stride + 63) / 64 * 64;
      size_t lds = (size_t)a.nbmax * 8;
      if (ix->metric == kHamming || ix->metric == kJaccard)
        lds += (size_t)ix->words * 4;
      else if (sweep_cpl_for_dim(ix->dim) == 0)
        lds += (size_t)((ix->dim + 3) / 4) * 16;
      const int blocks = (int)std::min<uint64_t>(ix->graph_nodes, (uint64_t)ix->n_cus * 8);
      hipError_t e = VDB_DISPATCH_METRIC_CPL(launch_ndist_t, ix->metric, ix->dim, a, blocks, lds, st);
      if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("ndist launch: ") + hipGetErrorString(e));
    }
  }
  ix->ndist_valid = true;
  return VDB_OK;
}

// Links rows [first, first+n) into the graph.  max_batch == 1: the reference's sequential insert;
// otherwise batch-synchronous insertion with the schedule build_batch_size().
static int32_t graph_insert_rows_impl(vdb_hip_index* ix, uint64_t first, uint64_t n, uint32_t max_batch) {
  if (n == 0) return VDB_OK;
  if (first != ix->graph_nodes) return fail(VDB_ERR_STATE, "graph_insert_rows: rows must be linked in order");
  if (max_batch == 0) max_batch = 2048;
  for (auto& L : ix->layers)
    if (L.stride > 256) return fail(VDB_ERR_UNSUPPORTED, "max_connections > 128 is not supported by the link kernel");
  if (ix->M > 128) return fail(VDB_ERR_UNSUPPORTED, "max_connections > 128 is not supported by the link kernel");
  max_batch = std::min<uint32_t>(max_batch, 1u << 20);
  hipStream_t st = ix->stream;
  int32_t rc = graph_fill_ndist(ix);
  if (rc != VDB_OK) return rc;
  const double level_mult = 1.0 / std::log((double)ix->M);  // graph.rs:63
  std::vector<uint8_t> levels(n);
  uint32_t top = 0;
  for (uint64_t i = 0; i < n; i++) {
    levels[i] = (uint8_t)next_level(ix->rng_state, level_mult);
    top = std::max<uint32_t>(top, levels[i]);
  }
  rc = ensure_layers(ix, top + 1);  // graph.rs:171-179
  if (rc != VDB_OK) return rc;
  hipError_t e = ix->s_levels.reserve(n, false, st);
  if (e == hipSuccess) e = hipMemcpyAsync(ix->s_levels.p, levels.data(), n, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);  // `levels` is host stack/heap memory
  if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("levels upload: ") + hipGetErrorString(e));

  HnswInsertArgs a{};
  uint32_t max_stride = 0;
  fill_layers(ix, a.layers, &max_stride);
  a.dc = DistCtx{ix->rows.as<float>(), ix->norms.as<float>(), ix->bits.as<uint32_t>(), ix->row_stride, ix->dim,
                 ix->words};
  const uint32_t ef = std::max<uint32_t>(ix->efc, 1);
  a.ef = ef;
  uint64_t cap = (uint64_t)ef + std::max<uint64_t>(64, ef / 2);
  cap = (cap + 63) / 64 * 64;
  a.cap = (uint32_t)cap;
  a.nbmax = (std::max(max_stride, ef) + 63) / 64 * 64;
  a.alpha = 1.0f;  // graph.rs:77
  size_t lds = insert_lds_bytes(a.cap, a.nbmax, ix->dim, ix->words, ix->metric);
  if (lds > 160 * 1024) return fail(VDB_ERR_UNSUPPORTED, "ef_construction too large for the LDS-resident candidate list");
  int per_cu = (int)std::min<size_t>(4, std::max<size_t>(1, (160 * 1024) / lds));
  rc = ensure_traversal_scratch(ix, st);
  if (rc != VDB_OK) return rc;
  const uint64_t vis_words = ix->vis_words;
  const uint32_t vlog_cap = kVlogCap;
  a.visited = ix->s_visited.as<uint32_t>();
  a.vlog = ix->s_vlog.as<uint32_t>();
  a.vis_words = vis_words;
  a.vlog_cap = vlog_cap;
  // link-request buffers: every node emits at most sum over its layers of max_conn requests
  uint64_t per_node = 0;
  for (auto& L : ix->layers) per_node += L.stride;
  const uint32_t bmax = (uint32_t)std::min<uint64_t>(max_batch, n);
  const uint64_t req_cap = (uint64_t)bmax * per_node;
  const size_t sort_tmp = bmax > 1 ? radix_sort_scratch_bytes((uint32_t)std::min<uint64_t>(req_cap, 0xFFFFFFFFull)) : 0;
  if (req_cap > 0xFFFFFFFFull) return fail(VDB_ERR_UNSUPPORTED, "graph construction: batch too large for the link-request sort");
  if ((e = ix->s_req_keys.reserve(req_cap * 8 * 2, false, st)) != hipSuccess ||
      (e = ix->s_req_vals.reserve(req_cap * 8 * 2, false, st)) != hipSuccess ||
      (e = ix->s_sort_tmp.reserve(std::max<size_t>(sort_tmp, 16), false, st)) != hipSuccess ||
      (e = ix->s_misc.reserve(64, false, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("link-request scratch: ") + hipGetErrorString(e));
  uint64_t* keys_in = ix->s_req_keys.as<uint64_t>();
  uint64_t* keys_out = keys_in + req_cap;
  uint64_t* vals_in = ix->s_req_vals.as<uint64_t>();
  uint64_t* vals_out = vals_in + req_cap;
  uint32_t* d_req_n = ix->s_misc.as<uint32_t>();
  uint32_t* d_overflow = d_req_n + 1;
  VDB_HIP(hipMemsetAsync(d_req_n, 0, 8, st));
  a.req_keys = keys_in;
  a.req_vals = vals_in;
  a.req_n = d_req_n;
  a.overflow = d_overflow;
  a.levels = ix->s_levels.as<uint8_t>();
  if (!ix->s_build_stats.p) {  // first construction on this handle: the cumulative counters start at zero
    if ((e = ix->s_build_stats.reserve(64, false, st)) != hipSuccess || (e = hipMemsetAsync(ix->s_build_stats.p, 0, 64, st)) != hipSuccess)
      return fail(VDB_ERR_OOM, std::string("construction counters: ") + hipGetErrorString(e));
  }
  a.stats = ix->s_build_stats.as<unsigned long long>();

  uint64_t pos = 0;
  while (pos < n) {
    uint64_t b = build_batch_size(ix->graph_nodes, max_batch);
    if (ix->entry_point < 0) b = 1;
    b = std::min<uint64_t>(b, n - pos);
    const uint64_t node0 = first + pos;
    if (ix->entry_point >= 0) {
      a.first = (uint32_t)node0;
      a.B = (uint32_t)b;
      a.levels = ix->s_levels.as<uint8_t>() + pos;
      a.max_layer = ix->max_layer;
      a.entry_point = (uint32_t)ix->entry_point;
      const uint32_t nreq = (uint32_t)(b * per_node);
      a.req_cap = nreq;
      for (;;) {
        VDB_HIP(hipMemsetAsync(d_req_n, 0, 8, st));  // request counter + overflow flags
        VDB_HIP(hipMemsetAsync(keys_in, 0xFF, (size_t)nreq * 8, st));
        const int slots = (int)std::min<uint64_t>(b, (uint64_t)ix->n_cus * per_cu);
        e = VDB_DISPATCH_METRIC_CPL(launch_insert_t, ix->metric, ix->dim, a, slots, lds, st);
        if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("insert launch: ") + hipGetErrorString(e));
        // The insert kernel only writes the NEW nodes' own lists and the request buffer (back-links are applied by the
        // link kernel below), so a batch can be repeated.  Candidates tied with the worst result stay in the list
        // unexpanded (the reference's candidates heap is unbounded, graph.rs:449-510): with very many exact ties
        // (sparse Jaccard: most pairs at distance 1.0) the list outgrows ef + slack — repeat with twice the room.
        uint32_t h_batch_over = 0;
        VDB_HIP(hipMemcpyAsync(&h_batch_over, d_overflow, 4, hipMemcpyDeviceToHost, st));
        VDB_HIP(hipStreamSynchronize(st));
        if (!(h_batch_over & 1u)) {
          if (h_batch_over)
            return fail(VDB_ERR_UNSUPPORTED, "graph construction: link-request buffer overflow, code " + std::to_string(h_batch_over));
          break;
        }
        const uint64_t ncap = (uint64_t)a.cap * 2;
        const size_t nlds = insert_lds_bytes((uint32_t)ncap, a.nbmax, ix->dim, ix->words, ix->metric);
        if (nlds > 160 * 1024)
          return fail(VDB_ERR_UNSUPPORTED,
                      "graph construction: candidate list overflow (more exact distance ties than the LDS list can hold)");
        a.cap = (uint32_t)ncap;
        lds = nlds;
        per_cu = (int)std::min<size_t>(4, std::max<size_t>(1, (160 * 1024) / lds));
      }
      HnswLinkArgs la{};
      fill_layers(ix, la.layers, nullptr);
      la.n = nreq;
      if (b > 1) {
        // requests grouped by (layer, target), inside a group by batch index = the order of the reference's sequential insert:
        // stable LSD radix sort (radix_sort.hip) over the digits that can differ — batch index < b (key bits 0..), node id <
        // node0 + b (bits 20..), layer (bits 52..55); unused slots (all ones) sort behind everything
        RadixDigit dg[12];
        int nd = 0;
        auto add_field = [&](uint32_t shift, uint32_t bits) {
          for (uint32_t o = 0; o < bits; o += 8) dg[nd++] = RadixDigit{shift + o, std::min<uint32_t>(8, bits - o)};
        };
        auto bits_for = [](uint64_t v) {
          uint32_t nb = 1;
          while (nb < 32 && (v >> nb)) nb++;
          return nb;
        };
        add_field(0, std::min<uint32_t>(20, bits_for(b)));             // (all-ones slots need one bit more than b - 1: bits_for(b))
        add_field(20, std::min<uint32_t>(32, bits_for(node0 + b)));
        add_field(52, 4);
        bool in_b = false;
        e = radix_sort_pairs_u64(keys_in, vals_in, keys_out, vals_out, nreq, dg, nd, ix->s_sort_tmp.p, &in_b, st);
        if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("link-request sort: ") + hipGetErrorString(e));
        la.keys = in_b ? keys_out : keys_in;
        la.vals = in_b ? vals_out : vals_in;
        la.singletons = 0;
      } else {
        la.keys = keys_in;
        la.vals = vals_in;
        la.singletons = 1;
      }
      const uint32_t lds_per_wave = ((max_stride + 1) * 8 + 15) & ~15u;
      const int lblocks = (int)std::min<uint64_t>(((uint64_t)nreq + 3) / 4, (uint64_t)ix->n_cus * 8);
      hipLaunchKernelGGL(hnsw_link_kernel, dim3(lblocks), dim3(256), (size_t)lds_per_wave * 4, st, la, lds_per_wave);
      VDB_HIP(hipGetLastError());
    }
    // entry point / max layer bookkeeping in node order (graph.rs:225-233)
    for (uint64_t i = 0; i < b; i++) {
      const uint64_t node = node0 + i;
      const uint32_t lv = levels[pos + i];
      if (ix->entry_point < 0) ix->entry_point = (int64_t)node;
      if (lv > ix->max_layer) {
        ix->max_layer = lv;
        ix->entry_point = (int64_t)node;
      }
    }
    ix->graph_nodes += b;
    pos += b;
  }
  uint32_t h_over = 0;
  VDB_HIP(hipMemcpyAsync(&h_over, d_overflow, 4, hipMemcpyDeviceToHost, st));
  VDB_HIP(hipStreamSynchronize(st));
  if (h_over)
    return fail(VDB_ERR_UNSUPPORTED,
                "graph construction: candidate list overflow (too many exact distance ties for the LDS list), code " +
                    std::to_string(h_over));
  return VDB_OK;
}

// A failure (candidate-list overflow on exact ties, max_connections > 128, LDS limit, out of memory) can strike after
// some sub-batches are linked.  The rows stay registered (exact search keeps serving them) but the graph no longer
// covers them: graph_valid = false makes every HNSW mode answer VDB_ERR_STATE instead of silently omitting rows, later
// inserts only append, and vdb_hip_index_build_graph resumes from graph_nodes — with the level stream (graph.rs:368-403)
// rewound to the last linked node, so the finished graph is the one an undisturbed run builds.
int32_t graph_insert_rows(vdb_hip_index* ix, uint64_t first, uint64_t n, uint32_t max_batch) {
  const uint64_t rng0 = ix->rng_state, nodes0 = ix->graph_nodes;
  const int32_t rc = graph_insert_rows_impl(ix, first, n, max_batch);
  if (rc != VDB_OK) {
    ix->graph_valid = false;
    const double level_mult = 1.0 / std::log((double)ix->M);
    ix->rng_state = rng0;
    for (uint64_t i = nodes0; i < ix->graph_nodes; i++) (void)next_level(ix->rng_state, level_mult);
    (void)hipStreamSynchronize(ix->stream);
  }
  return rc;
}

}  // namespace vdb
