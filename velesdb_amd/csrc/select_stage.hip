// select_stage.hip — the batches of HnswIndex::search_brute_force (index/hnsw/index/search.rs:176-219) that SELECT on the matrix
// cores and SCORE exactly: the bf16 result path (VDB_SEARCH_BRUTE_BF16: half_precision.rs:199-255 semantics), the selection stage
// of exact Cosine / DotProduct / Euclidean / SQ8 batches (split-bf16 or plain bf16 selection + exact re-scoring + per-query proof,
// sweep_split.hip), and the derived images of the rows they read (built at first use inside a search, complete before another
// search context may take them over).  Host side only: the kernels live in sweep_gemm_bf16.hip / sweep_split.hip / sweep_gemm.hip;
// index.hip dispatches here (brute_dev / search_dev).
#include <algorithm>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vdb_probe_env.hpp"
#include "vdb_select_stage.hpp"
#include "vdb_wide.hpp"

namespace vdb {

static inline uint32_t opt_max_tile(const vdb_hip_index* ix) { return (uint32_t)opt_value(ix, VDB_OPT_MAX_QUERY_TILE); }
static inline int opt_engine(const vdb_hip_index* ix) { return (int)opt_value(ix, VDB_OPT_SWEEP_ENGINE); }
static inline int opt_selector(const vdb_hip_index* ix) { return (int)opt_value(ix, VDB_OPT_SELECTOR_LEVEL); }
// VELESDB_BF16_SEED=0: level 2 keeps the exact f32 seed sweep (A / B probes)
static const bool g_bf16_seed = [] {
  const char* e = probe_env("VELESDB_BF16_SEED");
  return !(e && e[0] == '0');
}();
// VELESDB_POOL_SELECT=0: the bounds between the launches of a selection batch and its final pool come from merge_topk_select again (A / B probes)
static const bool g_pool_select = [] {
  const char* e = probe_env("VELESDB_POOL_SELECT");
  return !(e && e[0] == '0');
}();
// VELESDB_BF16_GLDS=0: big bf16 batches stay on the register-staged kernel of sweep_gemm.hip (A/B probes)
static bool gemm_bf16_glds_enabled() {
  static const bool on = [] {
    const char* e = probe_env("VELESDB_BF16_GLDS");
    return !(e && e[0] == '0');
  }();
  return on;
}


// exact sweep over the bf16 copy of the rows: half_precision::dot_product / cosine_similarity semantics
// (half_precision.rs:199-255) for nq device-resident f32 queries (rounded to bf16 by the kernel)
int32_t brute_bf16_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k,
                              uint64_t* d_ids, float* d_scores, uint32_t* d_n, hipStream_t st) {
  if (!ix->bf16_enabled) return fail(VDB_ERR_STATE, "bf16 sweep: call vdb_hip_index_enable_bf16 first");
  if (ix->metric != VDB_COSINE && ix->metric != VDB_DOT)
    return fail(VDB_ERR_UNSUPPORTED, "bf16 sweep: Cosine and DotProduct only");
  if (nq == 0) return VDB_OK;
  if (k == 0 || ix->n_rows == 0) {
    VDB_HIP(hipMemsetAsync(d_n, 0, (size_t)nq * 4, st));
    return VDB_OK;
  }
  const uint8_t* alive = ix->any_dead ? ix->alive.as<uint8_t>() : nullptr;
  for (uint32_t q0 = 0; q0 < nq;) {
    const uint32_t rem = nq - q0;
    // large batches over a large corpus: the 256 x 256 LDS-DMA kernel (sweep_gemm_bf16.hip).  Its thresholds are seeded:
    // the 128 x 128 kernel first sweeps the first kGemmBf16SeedRows rows, the k-th best key found there (+ 1) is every
    // block's starting bound — without it the first row tile of every block floods the 12-key candidate buffers.
    if (opt_max_tile(ix) >= 128 && rem >= kGemmBigMinQueries && k <= kGemmBf16MaxK && ix->dim % 64 == 0 && ix->dim >= 128 &&
        ix->n_rows >= kGemmBf16MinRows && ix->n_rows < 0xFFFFFF00ull && gemm_bf16_glds_enabled()) {
      // the chunk this kernel takes: up to 1 024 queries that fill their 256-query tiles to >= 7/8; when the last tile would
      // be emptier than that, the whole tiles in front of it (the rest is the next chunk: the 128 x 128 / streaming kernels)
      uint32_t nqg = std::min<uint32_t>(rem, kGemmMaxQueries);
      if ((uint64_t)nqg * 8 < (uint64_t)((nqg + 255) / 256) * 256 * 7) nqg = nqg / 256 * 256;
      if (nqg >= kGemmBigMinQueries) {
        // Launch schedule: rows [0, R0) by the 128 x 128 kernel (seed), then the LDS-DMA kernel over [R0, R1) and [R1, n).
        // Every launch starts from the k-th best key over ALL rows swept before it: the number of candidates a wave has
        // to look at per row tile falls as k / rows seen (5 per wave tile behind 16 K rows, 0.1 behind 640 K).  All
        // launches write their partial lists into one [nq][lists][k] array that the final merge scans once.
        const uint32_t n = (uint32_t)ix->n_rows;
        const uint32_t R0 = kGemmBf16SeedRows;
        // first launch: ~max(2^18, n / 16) rows, then launches of <= 2 M rows (gemm_schedule, vdb_kernels.hpp)
        GemmSchedule sch;
        {
          const uint32_t G2 = (uint32_t)std::max(8, ix->n_cus / (int)((nqg + 255) / 256) / 8 * 8);
          const uint32_t head[3] = {(uint32_t)((std::max<uint64_t>(1u << 18, n / 16) + (uint64_t)G2 * 256 - 1) / ((uint64_t)G2 * 256)), 0u, 0u};
          gemm_schedule(nqg, R0, n, ix->n_cus, head, 1u << 21, &sch);
        }
        const uint32_t lists = 1 + sch.lists;
        GemmPlan sp;  // seeding sweep over the first rows
        sweep_gemm_plan(nqg, R0, ix->n_cus, k, &sp, /*allow_big=*/false);
        const size_t off_ids = ((size_t)nqg * sp.G * k * 8 + 15) & ~(size_t)15, off_sc = off_ids + (size_t)nqg * k * 8,
                     off_n = off_sc + (size_t)nqg * k * 4, off_tau = (off_n + (size_t)nqg * 4 + 15) & ~(size_t)15,
                     off_qn = off_tau + (size_t)nqg * 8;
        hipError_t e3;
        if ((e3 = ix->s_part_keys.reserve((size_t)nqg * lists * k * 8, false, st)) != hipSuccess ||
            (e3 = ix->s_seed.reserve(off_qn + (size_t)nqg * 4, false, st)) != hipSuccess ||
            (e3 = ix->s_misc.reserve(((size_t)nqg + 256) * ix->bf16_stride * 2, false, st)) != hipSuccess)
          return fail(VDB_ERR_OOM, "bf16 GEMM scratch");
        unsigned char* sd = ix->s_seed.as<unsigned char>();
        uint64_t* parts = ix->s_part_keys.as<uint64_t>();
        uint64_t* tau0 = reinterpret_cast<uint64_t*>(sd + off_tau);
        const uint16_t* q16 = ix->s_misc.as<uint16_t>();
        launch_round_queries_bf16(d_q + (size_t)q0 * q_stride, q_stride, ix->s_misc.as<uint16_t>(), ix->bf16_stride, nqg,
                                  ix->dim, st);
        float* qn_half = reinterpret_cast<float*>(sd + off_qn);  // norms of the rounded queries, once per batch
        launch_query_norms_bf16(q16, ix->bf16_stride, qn_half, nqg, ix->dim, st);
        // the 256 x 256 kernel stages whole 256-query tiles: zero rows behind the batch
        VDB_HIP(hipMemsetAsync(ix->s_misc.as<uint16_t>() + (size_t)nqg * ix->bf16_stride, 0, (size_t)256 * ix->bf16_stride * 2, st));
        VDB_HIP(hipMemsetAsync(parts, 0xFF, (size_t)nqg * lists * k * 8, st));  // every slot: kKeyInvalid
        ix->last_kernels |= VDB_KERNEL_GEMM_BF16_GLDS | VDB_KERNEL_GEMM_BF16;  // (the seed prefix: the 128 x 128 kernel)
        EventPair* evg = next_events(ix);
        if (evg) (void)hipEventRecord(evg->a, st);
        e3 = launch_sweep_gemm_bf16(ix->metric, sp, ix->rows_bf16.as<uint16_t>(), ix->bf16_stride, ix->norms_bf16.as<float>(),
                                    alive, q16, ix->bf16_stride, reinterpret_cast<uint64_t*>(sd), R0, ix->dim, nqg, k, st);
        if (e3 != hipSuccess) return fail(VDB_ERR_HIP, std::string("bf16 seed sweep launch: ") + hipGetErrorString(e3));
        MergeArgs ms{};
        ms.part_keys = reinterpret_cast<const uint64_t*>(sd);
        ms.ext_ids = nullptr;  // internal rows
        ms.out_ids = reinterpret_cast<uint64_t*>(sd + off_ids);
        ms.out_scores = reinterpret_cast<float*>(sd + off_sc);
        ms.out_n = reinterpret_cast<uint32_t*>(sd + off_n);
        ms.n_lists = sp.G;
        ms.k = k;
        launch_merge(true, ms, nqg, st);
        launch_seed_tau(ms.out_ids, ms.out_scores, ms.out_n, tau0, parts, lists, nqg, k, st);  // list 0 = the seed's top-k
        e3 = run_gemm_schedule(
            sch, ix->metric, ix->rows_bf16.as<uint16_t>(), ix->bf16_stride, ix->norms_bf16.as<float>(), alive, q16, ix->bf16_stride, tau0, parts, lists,
            /*list_first=*/1, ix->dim, nqg, k, st, /*split=*/false, nullptr, nullptr, qn_half, [](int) {},
            [&](int, uint32_t, bool last) {
              if (last) return;  // bound for the next launch: k-th best key over everything swept so far
              ms.part_keys = parts;
              ms.n_lists = lists;
              launch_merge(true, ms, nqg, st);
              launch_seed_tau(ms.out_ids, ms.out_scores, ms.out_n, tau0, nullptr, 0, nqg, k, st);
            });
        if (e3 != hipSuccess) return fail(VDB_ERR_HIP, std::string("bf16 gemm sweep launch: ") + hipGetErrorString(e3));
        if (evg) (void)hipEventRecord(evg->b, st);
        MergeArgs mg{};
        mg.part_keys = parts;
        mg.ext_ids = ix->ext_ids.as<uint64_t>();
        mg.out_ids = d_ids + (size_t)q0 * k;
        mg.out_scores = d_scores + (size_t)q0 * k;
        mg.out_n = d_n + q0;
        mg.n_lists = lists;
        mg.k = k;
        launch_merge(true, mg, nqg, st);
        q0 += nqg;
        continue;
      }
    }
    // large batches: the GEMM-structured kernel over the bf16 rows (sweep_gemm.hip, BF16 variant): the corpus is read
    // once per <= 128 queries instead of once per 96, both operands through LDS, lock-free top-k epilogue
    if (opt_max_tile(ix) >= 128 && rem >= kGemmMinQueries && k <= kGemmMaxK && ix->dim % 64 == 0) {
      const uint32_t nqg = std::min<uint32_t>(rem, kGemmMaxQueries);
      GemmPlan gp;
      sweep_gemm_plan(nqg, (uint32_t)ix->n_rows, ix->n_cus, k, &gp, /*allow_big=*/true);
      if (gp.lds <= 160 * 1024) {
        hipError_t e3;
        if ((e3 = ix->s_part_keys.reserve((size_t)nqg * gp.G * k * 8, false, st)) != hipSuccess ||
            (e3 = ix->s_misc.reserve((size_t)nqg * ix->bf16_stride * 2, false, st)) != hipSuccess)
          return fail(VDB_ERR_OOM, "bf16 GEMM scratch");
        launch_round_queries_bf16(d_q + (size_t)q0 * q_stride, q_stride, ix->s_misc.as<uint16_t>(), ix->bf16_stride, nqg,
                                  ix->dim, st);
        ix->last_kernels |= VDB_KERNEL_GEMM_BF16;
        EventPair* evg = next_events(ix);
        if (evg) (void)hipEventRecord(evg->a, st);
        e3 = launch_sweep_gemm_bf16(ix->metric, gp, ix->rows_bf16.as<uint16_t>(), ix->bf16_stride, ix->norms_bf16.as<float>(),
                                    alive, ix->s_misc.as<uint16_t>(), ix->bf16_stride, ix->s_part_keys.as<uint64_t>(),
                                    (uint32_t)ix->n_rows, ix->dim, nqg, k, st);
        if (evg) (void)hipEventRecord(evg->b, st);
        if (e3 != hipSuccess) return fail(VDB_ERR_HIP, std::string("bf16 gemm sweep launch: ") + hipGetErrorString(e3));
        MergeArgs mg{};
        mg.part_keys = ix->s_part_keys.as<uint64_t>();
        mg.ext_ids = ix->ext_ids.as<uint64_t>();
        mg.out_ids = d_ids + (size_t)q0 * k;
        mg.out_scores = d_scores + (size_t)q0 * k;
        mg.out_n = d_n + q0;
        mg.n_lists = gp.G;
        mg.k = k;
        launch_merge(true, mg, nqg, st);
        q0 += nqg;
        continue;
      }
    }
    int nqt = rem > 64 ? 6 : (rem > 32 ? 4 : (rem > 16 ? 2 : 1));
    while (nqt > 1 && sweep_bf16_lds_bytes(nqt, k, ix->dim) > 160 * 1024) nqt = nqt == 6 ? 4 : nqt / 2;
    if (sweep_bf16_lds_bytes(nqt, k, ix->dim) > 160 * 1024)
      return fail(VDB_ERR_UNSUPPORTED, "bf16 sweep: dim / k too large for the LDS query tile");
    const uint32_t Bq = (uint32_t)nqt * 16;
    const uint32_t tile = std::min<uint32_t>(Bq, rem);
    const int waves = nqt >= 4 ? kBf16WavesBig : kBf16WavesSmall;
    const size_t lds = sweep_bf16_lds_bytes(nqt, k, ix->dim);
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds, (size_t)(16 / waves)));
    const uint32_t ntiles = (uint32_t)((ix->n_rows + 15) / 16);
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ntiles + waves - 1) / waves,
                                                                 (int64_t)ix->n_cus * per_cu));
    hipError_t e;
    if ((e = ix->s_part_keys.reserve((size_t)Bq * blocks * k * 8, false, st)) != hipSuccess)
      return fail(VDB_ERR_OOM, "top-k scratch");
    EventPair* ev = next_events(ix);
    if (ev) (void)hipEventRecord(ev->a, st);
    ix->last_kernels |= VDB_KERNEL_SWEEP_MFMA_BF16;
    e = launch_sweep_bf16(ix->metric, nqt, ix->rows_bf16.as<uint16_t>(), ix->bf16_stride, ix->norms_bf16.as<float>(),
                          alive, d_q + (size_t)q0 * q_stride, q_stride, ix->s_part_keys.as<uint64_t>(),
                          (uint32_t)ix->n_rows, ix->dim, tile, k, blocks, st);
    if (ev) (void)hipEventRecord(ev->b, st);
    if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("bf16 sweep launch: ") + hipGetErrorString(e));
    MergeArgs m{};
    m.part_keys = ix->s_part_keys.as<uint64_t>();
    m.ext_ids = ix->ext_ids.as<uint64_t>();
    m.out_ids = d_ids + (size_t)q0 * k;
    m.out_scores = d_scores + (size_t)q0 * k;
    m.out_n = d_n + q0;
    m.n_lists = (uint32_t)blocks;
    m.k = k;
    launch_merge(true, m, tile, st);
    q0 += tile;
  }
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}

// ---- exact Cosine / DotProduct batches: split-bf16 selection + exact re-scoring + proof (sweep_split.hip) ----------------
// first use: build the split image of every row (and the canonical norms a DotProduct index did not need so far)
// (The three builders below run inside searches, i.e. under the SHARED lock: serialised by the primary's img_mu, always on the
// primary's fields — a search context then takes over the views — and complete on the building search's stream before
// another context may read the image.)
static int32_t pinned_select_stats(vdb_hip_index* ix) {  // per context: the adaptive level state is the context's own
  if (!ix->sel_stats) {
    void* h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocDefault) != hipSuccess) return fail(VDB_ERR_OOM, "pinned selection statistics");
    memset(h, 0, 64);
    ix->sel_stats = static_cast<volatile uint32_t*>(h);
  }
  return VDB_OK;
}
static int32_t ensure_split_impl(vdb_hip_index* ix, hipStream_t st);
static int32_t ensure_sel16_impl(vdb_hip_index* ix, hipStream_t st);
static int32_t ensure_l2_select_impl(vdb_hip_index* ix, hipStream_t st);
// `stale(p)` is evaluated under img_mu.  A build is enqueued on the building search's stream `st` — a caller's stream for the
// device-resident entry point — and another context may take over the image (its fields say "complete") the moment img_mu is
// released: whatever was built is therefore COMPLETE on the device before the lock is dropped.  (First use, and the first
// search behind inserts for the images that are extended lazily: a synchronisation there is noise next to the build.)
template <class S, class F>
static int32_t build_image_on_primary(vdb_hip_index* ix, hipStream_t st, S&& stale, F&& impl) {
  vdb_hip_index* p = primary_of(ix);
  std::lock_guard<std::mutex> il(p->img_mu);
  const bool was_stale = stale(p);
  const int32_t rc = impl(p, st);
  if (rc != VDB_OK) return rc;
  if (was_stale) VDB_HIP(hipStreamSynchronize(st));
  if (ix != p) copy_image_fields(ix, p);
  return VDB_OK;
}
static int32_t ensure_split(vdb_hip_index* ix, hipStream_t st) {
  return build_image_on_primary(ix, st, [](const vdb_hip_index* p) { return !p->split_enabled || p->split_rows < p->n_rows; }, ensure_split_impl);
}
static int32_t ensure_split_impl(vdb_hip_index* ix, hipStream_t st) {
  if (!ix->split_enabled) {
    hipError_t e = ix->rows_split.reserve((std::max<uint64_t>(ix->capacity, 1) + kRowSlack) * (size_t)ix->dim * 4, false, st);
    if (e == hipSuccess) e = ix->norms.reserve((std::max<uint64_t>(ix->capacity, 1) + kRowSlack) * 4, ix->metric != VDB_DOT, st);
    if (e != hipSuccess) return fail(VDB_ERR_OOM, std::string("split rows: ") + hipGetErrorString(e));
    ix->split_enabled = true;
    ix->split_rows = 0;
  }
  if (ix->split_rows < ix->n_rows) {
    launch_split_vectors(ix->rows.as<float>(), ix->row_stride, ix->rows_split.as<uint16_t>(),
                         ix->metric == VDB_DOT ? ix->norms.as<float>() : nullptr, (uint32_t)ix->split_rows,
                         (uint32_t)(ix->n_rows - ix->split_rows), ix->dim, st);
    ix->split_rows = ix->n_rows;
    VDB_HIP(hipGetLastError());
  }
  return VDB_OK;
}

// level 2: the bf16 copy of the rows (what vdb_hip_index_enable_bf16 keeps) + canonical f32 norms for every metric
static int32_t ensure_sel16(vdb_hip_index* ix, hipStream_t st) {
  const int32_t rc = build_image_on_primary(ix, st, [](const vdb_hip_index* p) { return !p->bf16_enabled || p->bf16_rows < p->n_rows || !p->sel_norms; },
                                            ensure_sel16_impl);
  return rc != VDB_OK ? rc : pinned_select_stats(ix);
}
// the residual-ratio scalar of a bf16 copy that is about to be (re)built from row 0
int32_t reset_bf16_rho(vdb_hip_index* ix, hipStream_t st) {
  hipError_t e = ix->bf16_rho.reserve(256, false, st);
  if (e == hipSuccess) e = hipMemsetAsync(ix->bf16_rho.p, 0, 256, st);
  return e == hipSuccess ? VDB_OK : fail(VDB_ERR_OOM, std::string("bf16 residual bound: ") + hipGetErrorString(e));
}
static int32_t ensure_sel16_impl(vdb_hip_index* ix, hipStream_t st) {
  hipError_t e;
  if (!ix->bf16_enabled) {
    ix->bf16_stride = ((uint64_t)ix->dim + 7) / 8 * 8;
    if ((e = ix->rows_bf16.reserve((std::max<uint64_t>(ix->capacity, 1) + kRowSlack) * ix->bf16_stride * 2, false, st)) != hipSuccess ||
        (e = ix->norms_bf16.reserve((std::max<uint64_t>(ix->capacity, 1) + kRowSlack) * 4, false, st)) != hipSuccess)
      return fail(VDB_ERR_OOM, std::string("bf16 rows: ") + hipGetErrorString(e));
    const int32_t rr = reset_bf16_rho(ix, st);
    if (rr != VDB_OK) return rr;
    ix->bf16_enabled = true;
    ix->bf16_rows = 0;
  }
  if (ix->bf16_rows < ix->n_rows) {
    launch_prep_bf16(ix->rows.as<float>(), ix->row_stride, ix->rows_bf16.as<uint16_t>(), ix->bf16_stride, ix->norms_bf16.as<float>(),
                     (uint32_t)ix->bf16_rows, (uint32_t)(ix->n_rows - ix->bf16_rows), ix->dim, st, ix->bf16_rho.as<uint32_t>());
    ix->bf16_rows = ix->n_rows;
  }
  if (!ix->sel_norms) {
    if (ix->metric == VDB_DOT && !ix->split_enabled) {  // a DotProduct index did not need norms so far
      if ((e = ix->norms.reserve((std::max<uint64_t>(ix->capacity, 1) + kRowSlack) * 4, false, st)) != hipSuccess)
        return fail(VDB_ERR_OOM, std::string("norms: ") + hipGetErrorString(e));
      PrepArgs pa{};
      pa.rows = ix->rows.as<float>();
      pa.norms = ix->norms.as<float>();
      pa.row_stride = ix->row_stride;
      pa.n_rows = (uint32_t)ix->n_rows;
      pa.dim = ix->dim;
      pa.words = ix->words;
      if (pa.n_rows) launch_prep_rows(pa, st);
    }
    ix->sel_norms = true;
  }
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}

// Hamming / Jaccard batches on the matrix cores (bits_gemm.hip): the four-bit image of the packed bit rows + their bit counts
static int32_t ensure_bits_image_impl(vdb_hip_index* ix, hipStream_t st) {
  const uint64_t cap = std::max<uint64_t>(ix->capacity, 1) + kRowSlack;
  const uint32_t stride = bits_image_stride(ix->dim);
  hipError_t e;
  if ((e = ix->bits_img.reserve(cap * (size_t)stride, true, st)) != hipSuccess || (e = ix->bits_cnt.reserve(cap * 4, true, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("four-bit row image: ") + hipGetErrorString(e));
  if (ix->bits_img_rows < ix->n_rows) {
    launch_bits_expand(ix->metric, ix->bits.as<uint32_t>(), ix->words, ix->bits_img.as<uint8_t>(), stride, ix->bits_cnt.as<float>(), (uint32_t)ix->bits_img_rows,
                       (uint32_t)(ix->n_rows - ix->bits_img_rows), ix->dim, 0.0f, st);
    ix->bits_img_rows = ix->n_rows;
    VDB_HIP(hipGetLastError());
  }
  return VDB_OK;
}
int32_t ensure_bits_image(vdb_hip_index* ix, hipStream_t st) {
  return build_image_on_primary(ix, st, [](const vdb_hip_index* p) { return p->bits_img.cap == 0 || p->bits_img_rows < p->n_rows; }, ensure_bits_image_impl);
}

// does a chunk of nqg queries take the selection stage?  Whole 256-query tiles filled to >= 7/8 — or ONE partly filled tile
// from kSelectMinQueries up: a single-tile launch spreads its row groups over the whole chip, and a half-empty tile on the
// bf16 pipe still beats the f32 pipe's exact kernels (measured: see DESIGN 4.1b)
static uint32_t select_min_queries() {
  static const uint32_t v = [] {
    const char* e = probe_env("VELESDB_SELECT_MIN_QUERIES");
    return e ? (uint32_t)atoi(e) : kSelectMinQueries;
  }();
  return v;
}
// the chunk of the remaining nq_left queries the selection stage takes next (0: none): up to 1 024, whatever that leaves of the
// last 256-query tile — a partly filled tile costs what a full one costs, a second pass costs the whole fixed part again
// (384 queries as 256 + 128: 1.34 ms; as one pass of two tiles: what 512 cost, 1.01 ms)
uint32_t select_chunk(uint32_t nq_left, uint32_t min_queries) {
  const uint32_t nqg = std::min<uint32_t>(nq_left, kGemmMaxQueries);
  return nqg >= (min_queries ? min_queries : select_min_queries()) ? nqg : 0;
}
// (kSelectMinQueriesSq8 = 6, vdb_select_stage.hpp: the SQ8 storage mode's exact sweep keeps the reference's left-to-right chain — one
// lane per row, 0.39 ms per 4-query pass and 0.73 ms per 8-query pass at 1 M x 768 — so the selection stage, 0.55 ms whatever the
// fill of its one query tile, is ahead from 6 queries up, where the f32 kernels hold out until 16)

// 0: no selection stage (exact kernel); 1: split-bf16 selection; 2: plain bf16 selection
int select_level(vdb_hip_index* ix, uint32_t nq_left, uint32_t k) {
  const int want = opt_selector(ix);
  if (!want || opt_engine(ix) != 1 || opt_max_tile(ix) < 128) return 0;
  if (ix->metric != VDB_COSINE && ix->metric != VDB_DOT) return 0;
  if (ix->dim % 32 != 0 || ix->dim < 64 || ix->row_stride != ix->dim) return 0;
  if (k == 0 || k > kGemmBf16MaxK || ix->n_rows < kGemmBf16MinRows || ix->n_rows >= 0xFFFFFF00ull) return 0;
  if (!select_chunk(nq_left)) return 0;
  if (want < 2 || ix->dim % 64 != 0 || ix->dim < 128) return 1;
  // what did the finished level-2 batches of this handle look like?  (pinned host memory, written by the re-scoring launch, sweep_split.hip selection_batch_tail)
  if (ix->sel_stats && ix->sel_stats[2] != ix->sel_seq_seen) {
    ix->sel_seq_seen = ix->sel_stats[2];
    if (ix->sel_stats[3] == 2u && (uint64_t)ix->sel_stats[0] * 16 > ix->sel_stats[1]) ix->sel16_hold = 64;  // > 1/16 unproven
  }
  if (ix->sel16_hold) {
    ix->sel16_hold--;
    return 1;
  }
  return 2;
}

// Euclidean batches: augmented bf16 image + augmented f32 seed prefix (sweep_split.hip), built at first use, extended lazily
static int32_t ensure_l2_select(vdb_hip_index* ix, hipStream_t st) {
  const int32_t rc = build_image_on_primary(ix, st, [](const vdb_hip_index* p) { return p->l2_img.cap == 0 || p->l2_rows < p->n_rows; }, ensure_l2_select_impl);
  return rc != VDB_OK ? rc : pinned_select_stats(ix);
}
static int32_t ensure_l2_select_impl(vdb_hip_index* ix, hipStream_t st) {
  const uint64_t cap = std::max<uint64_t>(ix->capacity, 1) + kRowSlack;
  const uint32_t dim_a = ix->dim + 64, dim_s = ix->dim + 4;
  hipError_t e;
  if ((e = ix->l2_img.reserve(cap * (size_t)dim_a * 2, true, st)) != hipSuccess ||
      (e = ix->l2_seed.reserve((size_t)kSplitSeedRows * dim_s * 4, true, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("Euclidean selection image: ") + hipGetErrorString(e));
  if (!ix->l2_rho.p) {  // (the image is built from row 0 behind this: every row contributes)
    if ((e = ix->l2_rho.reserve(256, false, st)) != hipSuccess || (e = hipMemsetAsync(ix->l2_rho.p, 0, 256, st)) != hipSuccess)
      return fail(VDB_ERR_OOM, std::string("Euclidean residual bound: ") + hipGetErrorString(e));
    ix->l2_rows = 0;
  }
  if (ix->l2_rows < ix->n_rows) {
    launch_l2_augment_rows(ix->rows.as<float>(), ix->row_stride, ix->norms.as<float>(), ix->l2_img.as<uint16_t>(), dim_a,
                           ix->l2_seed.as<float>(), dim_s, kSplitSeedRows, (uint32_t)ix->l2_rows, (uint32_t)(ix->n_rows - ix->l2_rows),
                           ix->dim, st, ix->l2_rho.as<uint32_t>());
    ix->l2_rows = ix->n_rows;
    VDB_HIP(hipGetLastError());
  }
  return VDB_OK;
}
// Cosine batches at level 2 (and the WIDE selection): the bf16 image of the NORMALISED rows (sweep_split.hip seln_rows_kernel), built at
// first use, extended lazily, grown by ensure_capacity.  VELESDB_COSINE_NORMALISED=0 (probe builds): the round-5 path over the plain
// bf16 copy + per-row norms in the kernel (A / B runs)
static const bool g_cosn = [] {
  const char* e = probe_env("VELESDB_COSINE_NORMALISED");
  return !(e && e[0] == '0');
}();
static int32_t ensure_cosn_impl(vdb_hip_index* ix, hipStream_t st) {
  const uint64_t cap = std::max<uint64_t>(ix->capacity, 1) + kRowSlack;
  hipError_t e;
  if ((e = ix->cosn_img.reserve(cap * (size_t)ix->dim * 2, true, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("normalised selection image: ") + hipGetErrorString(e));
  if (!ix->cosn_rho.p) {  // (the image is built from row 0 behind this: every row contributes)
    if ((e = ix->cosn_rho.reserve(256, false, st)) != hipSuccess || (e = hipMemsetAsync(ix->cosn_rho.p, 0, 256, st)) != hipSuccess)
      return fail(VDB_ERR_OOM, std::string("normalised residual bound: ") + hipGetErrorString(e));
    ix->cosn_rows = 0;
  }
  if (ix->cosn_rows < ix->n_rows) {
    launch_seln_rows(ix->rows.as<float>(), ix->row_stride, ix->norms.as<float>(), ix->cosn_img.as<uint16_t>(), ix->dim, (uint32_t)ix->cosn_rows,
                     (uint32_t)(ix->n_rows - ix->cosn_rows), ix->dim, ix->cosn_rho.as<uint32_t>(), st);
    ix->cosn_rows = ix->n_rows;
    VDB_HIP(hipGetLastError());
  }
  return VDB_OK;
}
static int32_t ensure_cosn(vdb_hip_index* ix, hipStream_t st) {
  const int32_t rc = build_image_on_primary(ix, st, [](const vdb_hip_index* p) { return p->cosn_img.cap == 0 || p->cosn_rows < p->n_rows; }, ensure_cosn_impl);
  return rc != VDB_OK ? rc : pinned_select_stats(ix);
}
static inline bool cosine_normalised(const vdb_hip_index* ix) { return g_cosn && ix->metric == VDB_COSINE && ix->dim % 64 == 0; }

int select_level_l2(vdb_hip_index* ix, uint32_t nq_left, uint32_t k) {
  if (!opt_selector(ix) || opt_engine(ix) != 1 || opt_max_tile(ix) < 128) return 0;
  if (ix->metric != VDB_EUCLIDEAN) return 0;
  if (ix->dim % 64 != 0 || ix->dim < 128 || ix->row_stride != ix->dim) return 0;
  if (k == 0 || k > kGemmBf16MaxK || ix->n_rows < kGemmBf16MinRows || ix->n_rows >= 0xFFFFFF00ull) return 0;
  if (!select_chunk(nq_left)) return 0;
  if (ix->sel_stats && ix->sel_stats[2] != ix->sel_seq_seen) {
    ix->sel_seq_seen = ix->sel_stats[2];
    if (ix->sel_stats[3] == 5u && (uint64_t)ix->sel_stats[0] * 16 > ix->sel_stats[1]) ix->l2_hold = 64;
  }
  if (ix->l2_hold) {
    ix->l2_hold--;
    return 0;
  }
  return 2;
}

// the SQ8 storage mode's batches (VDB_SEARCH_BRUTE_SQ8): the same eligibility on the shapes the bf16 selection kernel takes
int select_level_sq8(vdb_hip_index* ix, uint32_t nq_left, uint32_t k) {
  if (!opt_selector(ix) || opt_max_tile(ix) < 128) return 0;
  if (ix->metric != VDB_COSINE && ix->metric != VDB_DOT && ix->metric != VDB_EUCLIDEAN) return 0;
  if (ix->dim % 64 != 0 || ix->dim < 128 || ix->row_stride != ix->dim) return 0;
  if (k == 0 || k > kGemmBf16MaxK || ix->n_rows < kGemmBf16MinRows || ix->n_rows >= 0xFFFFFF00ull) return 0;
  if (!select_chunk(nq_left, kSelectMinQueriesSq8)) return 0;
  if (ix->sel_stats && ix->sel_stats[2] != ix->sel_seq_seen) {
    ix->sel_seq_seen = ix->sel_stats[2];
    if (ix->sel_stats[3] == 3u && (uint64_t)ix->sel_stats[0] * 16 > ix->sel_stats[1]) ix->sq8_hold = 64;
  }
  if (ix->sq8_hold) {
    ix->sq8_hold--;
    return 0;
  }
  return 3;
}

int32_t brute_split_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nqg, uint32_t k, uint64_t* d_ids,
                        float* d_scores, uint32_t* d_n, hipStream_t st, int level) {
  const bool sq8 = level >= 3;
  const bool l2 = ix->metric == VDB_EUCLIDEAN;  // (level 2) the augmented DotProduct form of |q - v|^2, sweep_split.hip
  // (level 2, Cosine) both sides normalised before the rounding: the selection is a DotProduct of unit vectors (sweep_split.hip seln_rows_kernel)
  const bool cosn = level == 2 && !sq8 && !l2 && cosine_normalised(ix);
  const int sel_metric = (l2 || cosn) ? VDB_DOT : ix->metric;
  int32_t rc = sq8 ? ensure_sq8_select(ix, st)
                   : (l2 ? ensure_l2_select(ix, st) : (cosn ? ensure_cosn(ix, st) : (level >= 2 ? ensure_sel16(ix, st) : ensure_split(ix, st))));
  if (rc != VDB_OK) return rc;
  ix->last_select_level = level;
  ix->last_kernels |= (level < 2 ? VDB_KERNEL_SELECT_SPLIT : VDB_KERNEL_SELECT_BF16) | VDB_KERNEL_GEMM_F32;  // (exact seed sweep)
  const uint8_t* alive = ix->any_dead ? ix->alive.as<uint8_t>() : nullptr;
  const uint32_t n = (uint32_t)ix->n_rows, dim = ix->dim, K2 = level >= 2 ? kSelect16Pool : kSplitPool;
  const uint32_t dim_a = l2 ? dim + 64 : dim, dim_s = dim + 4;  // augmented image / f32 seed widths (Euclidean)
  const uint32_t ks = std::min<uint32_t>(kGemmBf16MaxK, k + 3);  // rows a selection block keeps per query (sweep_split.hip)
  // Launch schedule: exact seed sweep over [0, R0) (at 1/16 of the selection's rate: kept short), then selection launches
  // of GROWING size — G, 4 G, 16 G row tiles (G = row groups the chip holds at once: one tile per block first), then the
  // rest in whole tiles per row group — with the thresholds re-seeded from the merged pool between launches.  A block's
  // epilogue costs ~0.2 us per candidate it has to finish, and a threshold seeded from few rows lets hundreds of
  // candidates per 256 x 256 tile through: the rows swept under a weak threshold are kept few (measured at 1 M x 1 024
  // queries, level 2: two launches 2 x 1.32 ms).
  const uint32_t R0 = kSplitSeedRows;
  // level >= 2: the seed runs on the bf16 pipe over the selection's own image (sweep_split.hip seed_scores_bf16) and is a SAMPLE:
  // it only supplies the starting bounds — the selection launches sweep its rows again (0.4 % of the corpus), so it may keep
  // one key per 16 rows instead of all of them (33 MB of keys and a 55-us merge per batch otherwise)
  const bool bf16_seed = level >= 2 && g_bf16_seed;
  const uint32_t row_first = bf16_seed ? 0u : R0;
  GemmSchedule sch;
  {
    // (VELESDB_SEL_STEPS="a,b,c": tiles per row group of the first launches — schedule probes)
    static const std::array<uint32_t, 3> mult = [] {
      std::array<uint32_t, 3> m{1, 4, 16};
      if (const char* e = probe_env("VELESDB_SEL_STEPS")) {
        unsigned a = 0, b = 0, c = 0;
        const int got = sscanf(e, "%u,%u,%u", &a, &b, &c);
        m = {got >= 1 ? a : 0u, got >= 2 ? b : 0u, got >= 3 ? c : 0u};
      }
      return m;
    }();
    const uint32_t head[3] = {mult[0], mult[1], mult[2]};
    gemm_schedule(nqg, row_first, n, ix->n_cus, head, 0, &sch);
  }
  const uint32_t lists = 1 + sch.lists;
  GemmPlan sp, fp;  // exact kernel: seed sweep over the first rows; fallback over everything
  sweep_gemm_plan(nqg, R0, ix->n_cus, k, &sp);
  sweep_gemm_plan(nqg, n, ix->n_cus, k, &fp);
  if (sp.lds > 160 * 1024 || fp.lds > 160 * 1024) return fail(VDB_ERR_UNSUPPORTED, "k too large for the fused top-k path");
  // Unproven queries of a Cosine / DotProduct batch take the exact streaming matrix-core kernel in GATHERED mode (one corpus pass per
  // g_B listed queries, none when nothing is listed).  Round 6: for ANY number of them — rounds 2-5 sent batches with more than 96 to a
  // launch of the GEMM-structured kernel, which (with its merge) had to be enqueued for every batch to stand aside on the device: two
  // launches less per batch; a batch with hundreds of unproven queries costs a few more corpus passes, and parks the handle anyway.
  // (VELESDB_GATHER_ALL=0, probe builds: the old pair)
  int g_nqt = 3;
  while (g_nqt > 1 && sweep_mfma_lds_bytes(g_nqt, k, dim) > 160 * 1024) g_nqt--;
  const bool gather_ok = sweep_mfma_lds_bytes(g_nqt, k, dim) <= 160 * 1024;
  static const bool g_gather_all = [] {
    const char* e = probe_env("VELESDB_GATHER_ALL");
    return !(e && e[0] == '0');
  }();
  const bool gather_all = gather_ok && g_gather_all && !sq8 && !l2;
  // scratch map (s_seed)
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 15) & ~(size_t)15;
    return o;
  };
  // level 2 over the f32 rows: the seed runs on the bf16 pipe (sweep_split.hip seed_scores_bf16: every seed score as a key)
  const size_t o_seedp = take(std::max((size_t)nqg * sp.G * k * 8, bf16_seed ? (size_t)nqg * (R0 / 16) * 8 : (size_t)0)), o_ids = take((size_t)nqg * K2 * 8), o_sc = take((size_t)nqg * K2 * 4),
               o_n = take((size_t)nqg * 4), o_tau = take((size_t)nqg * 8), o_delta = take((size_t)nqg * 4),
               o_qn = take((size_t)nqg * 4), o_rho = take((size_t)nqg * 4), o_flags = take((size_t)nqg * 4 + 64 * 4 + 16), o_btau = take((size_t)nqg * lists * 8),
               o_fid = take((size_t)nqg * k * 8), o_fsc = take((size_t)nqg * k * 4), o_fn = take((size_t)nqg * 4),
               o_qmap = take((size_t)nqg * 8), o_gid = take((size_t)nqg * k * 8), o_gsc = take((size_t)nqg * k * 4),
               o_gn = take((size_t)nqg * 4);
  hipError_t e;
  if ((e = ix->s_seed.reserve(off, false, st)) != hipSuccess ||
      (e = ix->s_part_keys.reserve((size_t)nqg * lists * ks * 8, false, st)) != hipSuccess ||
      (!sq8 && !l2 && !gather_all && (e = ix->s_fb_keys.reserve((size_t)nqg * fp.G * k * 8, false, st)) != hipSuccess) ||
      (e = ix->s_misc.reserve(((size_t)nqg + 256) * (dim + 64) * 4 + (size_t)nqg * dim_s * 4, false, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, "split sweep scratch");
  unsigned char* sd = ix->s_seed.as<unsigned char>();
  uint64_t* pool = ix->s_part_keys.as<uint64_t>();
  uint64_t* m_ids = reinterpret_cast<uint64_t*>(sd + o_ids);
  float* m_sc = reinterpret_cast<float*>(sd + o_sc);
  uint32_t* m_n = reinterpret_cast<uint32_t*>(sd + o_n);
  uint64_t* tau0 = reinterpret_cast<uint64_t*>(sd + o_tau);
  float* delta = reinterpret_cast<float*>(sd + o_delta);
  float* qnorms = reinterpret_cast<float*>(sd + o_qn);
  // level 2 over the f32 rows: the error bound from MEASURED rounding residuals (sweep_split.hip select_eps_q)
  const DevBuf& rho_buf = sq8 ? ix->sq8_rho : (l2 ? ix->l2_rho : (cosn ? ix->cosn_rho : ix->bf16_rho));
  float* rho_q = (level >= 2 && rho_buf.p) ? reinterpret_cast<float*>(sd + o_rho) : nullptr;
  const uint32_t* rho_max = rho_q ? rho_buf.as<uint32_t>() : nullptr;
  uint32_t* flags = reinterpret_cast<uint32_t*>(sd + o_flags);
  uint32_t* tile_needed = flags + nqg;          // [<= 64]
  uint32_t* norm_max = tile_needed + 64;
  uint32_t* qcount = norm_max + 1;              // (cleared with the flags) the unproven queries list themselves: sweep_split.hip list_unproven
  uint32_t* qmap = reinterpret_cast<uint32_t*>(sd + o_qmap);  // [nqg] slot -> query
  uint32_t* qslot = qmap + nqg;                                // [nqg] query -> slot
  uint64_t* blk_tau = reinterpret_cast<uint64_t*>(sd + o_btau);
  uint16_t* q16 = ix->s_misc.as<uint16_t>();

  EventPair* ev = next_events(ix);
  if (ev) (void)hipEventRecord(ev->a, st);
  // queries: split image / bf16 image (+ canonical norms), zero rows behind the batch (the kernel stages whole 256-query tiles)
  const uint64_t img_stride = l2 ? (uint64_t)dim_a : ((sq8 || cosn) ? (uint64_t)dim : (level >= 2 ? ix->bf16_stride : (uint64_t)dim * 2));  // elements per image row
  const uint16_t* img_rows = sq8 ? ix->sq8_img.as<uint16_t>()
                                 : (l2 ? ix->l2_img.as<uint16_t>()
                                       : (cosn ? ix->cosn_img.as<uint16_t>() : (level >= 2 ? ix->rows_bf16.as<uint16_t>() : ix->rows_split.as<uint16_t>())));
  const float* sel_norms = sq8 ? ix->sq8_nrm.as<float>() : ix->norms.as<float>();
  float* qaug = reinterpret_cast<float*>(ix->s_misc.as<unsigned char>() + ((size_t)nqg + 256) * (dim + 64) * 4);  // Euclidean: (q, 1, 0, 0, 0) f32
  bool flags_cleared = false;
  if (l2) {
    launch_l2_augment_queries(d_q, q_stride, q16, dim_a, qaug, dim_s, nqg, dim, st);
    if (rho_q) launch_query_round_error(d_q, q_stride, rho_q, nqg, dim, st);
    PrepArgs pq{};
    pq.rows = d_q;
    pq.norms = qnorms;
    pq.row_stride = q_stride;
    pq.n_rows = nqg;
    pq.dim = dim;
    pq.words = ix->words;
    launch_prep_rows(pq, st);
  } else if (cosn) {
    // one launch: image rows of q / |q|, canonical norms, residual ratios of the normalised queries, the cleared flag words
    launch_seln_prep_queries(d_q, q_stride, q16, img_stride, qnorms, rho_q, flags, nqg + 64 + 4, nqg, dim, st);
    flags_cleared = true;
  } else if (level >= 2) {
    // one launch: image rows, canonical norms, rounding residual ratios, the cleared flag words
    launch_sel16_prep_queries(d_q, q_stride, q16, img_stride, qnorms, rho_q, flags, nqg + 64 + 4, nqg, dim, st);
    flags_cleared = true;
  } else {
    launch_split_vectors(d_q, q_stride, q16, qnorms, 0, nqg, dim, st);
  }
  if (nqg % 256u) VDB_HIP(hipMemsetAsync(q16 + (size_t)nqg * img_stride, 0, (size_t)256 * img_stride * 2, st));  // (whole tiles read nothing behind the batch)
  // (pool and blk_tau need no fill: the seed kernel writes slot 0 of every query, every selection block writes its slot of
  // every query of its tile — all ks keys, padded with invalid ones, and its bound — and every merge reads only the slots
  // written so far)
  if (!flags_cleared) VDB_HIP(hipMemsetAsync(flags, 0, (size_t)nqg * 4 + 64 * 4 + 16, st));
  // (s_fb_keys needs no fill: the whole-tile fallback writes every slot of every query of the tiles it runs for, and its merge
  // only looks at the unproven queries — MergeArgs::gate)
  if (sel_metric == VDB_DOT && !cosn) launch_max_norm(sel_norms, n, norm_max, st);  // (unit vectors: the Cosine bound needs no norm)
  // exact seed sweep over the first rows
  SweepArgs ag{};
  ag.rows = sq8 ? ix->sq8_seed.as<float>() : (l2 ? ix->l2_seed.as<float>() : ix->rows.as<float>());  // SQ8: the dequantised prefix (f32)
  ag.norms = sel_norms;
  ag.alive = alive;
  ag.queries = l2 ? qaug : d_q;
  ag.part_keys = reinterpret_cast<uint64_t*>(sd + o_seedp);
  ag.row_stride = l2 ? (uint64_t)dim_s : ix->row_stride;
  ag.q_stride = l2 ? (uint64_t)dim_s : q_stride;
  ag.n_rows = R0;
  ag.dim = l2 ? dim_s : dim;
  ag.nq = nqg;
  ag.k = k;
  MergeArgs ms{};
  ms.part_keys = ag.part_keys;
  ms.ext_ids = nullptr;  // internal rows
  ms.out_ids = m_ids;
  ms.out_scores = m_sc;
  ms.out_n = m_n;
  if (bf16_seed) {
    launch_seed_scores_bf16(sel_metric, img_rows, img_stride, sel_norms, alive, q16, img_stride, qnorms, ag.part_keys, R0, nqg,
                            l2 ? dim_a : dim, st);
    if (!l2 && g_pool_select && R0 / 16 <= 256) {  // the bound straight from the sample keys: one launch (sweep_split.hip split_seed_sample_kernel)
      launch_split_seed_sample(ix->metric, ag.part_keys, R0 / 16, qnorms, norm_max, tau0, delta, pool, blk_tau, lists, nqg, k, ks, dim, level, st, rho_q, rho_max);
    } else {
    ms.n_lists = R0 / 16;  // one "list" of one key per 16 seed rows (their best): the selection merge picks the ks best
    ms.k = 1;
    ms.k_out = ks;
    launch_merge(true, ms, nqg, st);
    if (l2)
      launch_l2_seed(m_ids, m_sc, m_n, qnorms, norm_max, tau0, delta, pool, blk_tau, lists, nqg, k, ks, dim_a, sq8 ? 1.5e-4f : 0.0f, st, rho_q, rho_max,
                     kSeedIsSample);
    else
      launch_split_seed_approx(ix->metric, m_ids, m_sc, m_n, qnorms, norm_max, tau0, delta, pool, blk_tau, lists, nqg, k, ks, kSeedIsSample, dim, level, st,
                               rho_q, rho_max);
    }
  } else {
  // (the exact seed scores the index' OWN metric over the f32 rows — Cosine stays Cosine when the selection runs as a DotProduct of
  // normalised images; Euclidean: the DotProduct of the augmented prefix)
  e = launch_sweep_gemm(l2 ? VDB_DOT : ix->metric, sp, ag, st);
  if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("split seed sweep launch: ") + hipGetErrorString(e));
  ms.n_lists = sp.G;
  ms.k = k;
  launch_merge(true, ms, nqg, st);
  }
  if (bf16_seed) {
  } else if (l2)
    launch_l2_seed(m_ids, m_sc, m_n, qnorms, norm_max, tau0, delta, pool, blk_tau, lists, nqg, k, ks, dim_a, sq8 ? 1.5e-4f : 0.0f, st, rho_q, rho_max);
  else
    launch_split_seed(ix->metric, m_ids, m_sc, m_n, qnorms, norm_max, tau0, delta, pool, blk_tau, lists, nqg, k, ks, dim, level, st, rho_q, rho_max);
  // selection launches over the split images
  {
    EventPair* evs = nullptr;
    e = run_gemm_schedule(
        sch, sel_metric, img_rows, img_stride, sel_norms, alive, q16, img_stride, tau0, pool, lists, /*list_first=*/1, l2 ? dim_a : dim, nqg, ks, st,
        /*split=*/level < 2, qnorms, blk_tau, nullptr,
        [&](int) {
          evs = next_sel_events(ix);
          if (evs) (void)hipEventRecord(evs->a, st);
        },
        [&](int, uint32_t list_off, bool last) {
          if (evs) (void)hipEventRecord(evs->b, st);
          if (last) return;  // bound of the next launch: k-th best pool score so far (the lists written so far)
          if (g_pool_select && pool_select_supported(list_off, ks, K2)) {  // ... by selection, not by a merge (pool_select.hip)
            launch_pool_kth_reseed(pool, list_off, lists, ks, k, delta, tau0, nqg, st);
            return;
          }
          ms.part_keys = pool;
          ms.n_lists = list_off;
          ms.list_stride = lists;
          ms.k = ks;
          ms.k_out = k;
          ms.reseed_delta = delta;  // ... and the bound itself, in the merge's own pass (sweep_split.hip split_reseed_kernel's rule)
          ms.reseed_tau = tau0;
          ms.reseed_k = k;
          launch_merge(true, ms, nqg, st);
          ms.reseed_delta = nullptr;
        });
    if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("split selection launch: ") + hipGetErrorString(e));
  }
  // pool -> K2 best by pool score -> exact re-scoring, ranking, proof
  if (g_pool_select && pool_select_supported(lists, ks, K2)) {
    launch_pool_topk(pool, lists, lists, ks, K2, m_ids, m_sc, m_n, nqg, st);
  } else {
    ms.part_keys = pool;
    ms.n_lists = lists;
    ms.list_stride = 0;
    ms.k = ks;
    ms.k_out = K2;
    launch_merge(true, ms, nqg, st);
  }
  SplitRerankArgs ra{};
  ra.rows = ix->rows.as<float>();
  ra.norms = ix->norms.as<float>();
  ra.queries = d_q;
  ra.qnorms = qnorms;
  ra.cand_rows = m_ids;
  ra.cand_scores = m_sc;
  ra.cand_n = m_n;
  ra.blk_tau = blk_tau;
  ra.delta = delta;
  ra.ext_ids = ix->ext_ids.as<uint64_t>();
  ra.out_ids = d_ids;
  ra.out_scores = d_scores;
  ra.out_n = d_n;
  ra.flags = flags;
  ra.tile_needed = sq8 ? nullptr : tile_needed;
  if (sq8) {
    ra.sq8_codes = ix->sq8_codes.as<uint8_t>();
    ra.sq8_min = ix->sq8_min.as<float>();
    ra.sq8_max = ix->sq8_max.as<float>();
    ra.sq8_nsq = ix->sq8_nsq.as<float>();
    ra.sq8_stride = ix->sq8_stride;
  }
  ra.row_stride = ix->row_stride;
  ra.q_stride = q_stride;
  ra.dim = dim;
  ra.dim_pad = (dim + 127) / 128 * 128;
  ra.k = k;
  ra.k2 = K2;
  ra.lists = lists;
  ra.fb_qper = fp.qper;
  ra.norm_max_bits = norm_max;
  ra.qcount = qcount;
  ra.qmap = qmap;
  ra.qslot = qslot;
  SelectFinishArgs fin{};  // the batch's last launch: exact results for the unproven queries, the counts to pinned host memory
  fin.flags = flags;
  fin.qcount = qcount;
  fin.qslot = qslot;
  fin.out_ids = d_ids;
  fin.out_scores = d_scores;
  fin.out_n = d_n;
  fin.nq = nqg;
  fin.k = k;
  if (ix->sel_stats) {
    fin.stats_host = ix->sel_stats;
    fin.stats_seq = ++ix->sel_seq;
    fin.stats_level = sq8 ? 3u : (l2 ? 5u : (uint32_t)level);
  }
  if (l2) launch_l2_rerank(ra, nqg, st);
  else launch_split_rerank(ix->metric, ra, nqg, st);
  if (sq8) {  // the reference chain for the unproven queries only, decided on the device
    const int32_t rf = sq8_fallback_flagged(ix, d_q, q_stride, nqg, k, qmap, fin, st);
    if (rf != VDB_OK) return rf;
  } else if (l2) {  // the canonical vector-ALU sweep for the unproven queries only, listed and gathered on the device
    const uint32_t ngroups8 = (n + 7) / 8;
    const int f_blocks = blocks_for(ix, 8, ngroups8);
    if ((e = ix->s_part_cnt.reserve((size_t)nqg * f_blocks * k * 8, false, st)) != hipSuccess) return fail(VDB_ERR_OOM, "gathered fallback scratch");
    SweepArgs af{};
    af.rows = ix->rows.as<float>();
    af.norms = ix->norms.as<float>();
    af.alive = alive;
    af.queries = d_q;
    af.part_keys = ix->s_part_cnt.as<uint64_t>();
    af.row_stride = ix->row_stride;
    af.q_stride = q_stride;
    af.n_rows = n;
    af.dim = dim;
    af.nq = 8;
    af.k = k;
    af.qmap = qmap;
    af.qcount = qcount;
    launch_sweep_f32(VDB_EUCLIDEAN, 8, af, f_blocks, st, (int)((nqg + 7) / 8));
    MergeArgs mg{};
    mg.part_keys = af.part_keys;
    mg.ext_ids = ix->ext_ids.as<uint64_t>();
    mg.out_ids = reinterpret_cast<uint64_t*>(sd + o_fid);
    mg.out_scores = reinterpret_cast<float*>(sd + o_fsc);
    mg.out_n = reinterpret_cast<uint32_t*>(sd + o_fn);
    mg.n_lists = (uint32_t)f_blocks;
    mg.k = k;
    mg.active = qcount;
    launch_merge(false, mg, nqg, st);
    fin.g_ids = mg.out_ids;
    fin.g_scores = mg.out_scores;
    fin.g_n = mg.out_n;
    launch_select_finish(fin, st);
  } else {
    // Unproven queries, decided on the device.  A few (<= kFallbackGatherMax): listed, and the streaming matrix-core kernel
    // makes ONE gathered corpus pass per 48 of them (0.9 ms; same mode-M bits).  More: the GEMM-structured kernel for the
    // query tiles that hold one (a tile costs the whole launch's duration: its row groups are all it parallelises over).
    // Both are launched; the one whose turn it is not exits at once.
    const uint32_t kFallbackGatherMax = gather_all ? nqg : 96u;
    const int g_waves = g_nqt >= 2 ? kMfmaWaves2 : kMfmaWaves1;
    const uint32_t g_B = (uint32_t)g_nqt * 16;
    MergeArgs mg{};
    if (gather_ok) {
      const uint32_t ntiles16 = (n + 15) / 16;
      const int g_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ntiles16 + g_waves - 1) / g_waves, (int64_t)ix->n_cus));
      const size_t gk = (size_t)kFallbackGatherMax * g_blocks * k * 8;
      if ((e = ix->s_part_cnt.reserve(gk, false, st)) != hipSuccess) return fail(VDB_ERR_OOM, "gathered fallback scratch");
      SweepArgs am{};
      am.rows = ix->rows.as<float>();
      am.norms = ix->norms.as<float>();
      am.alive = alive;
      am.queries = d_q;
      am.part_keys = ix->s_part_cnt.as<uint64_t>();
      am.row_stride = ix->row_stride;
      am.q_stride = q_stride;
      am.n_rows = n;
      am.dim = dim;
      am.nq = g_B;
      am.k = k;
      am.qmap = qmap;
      am.qcount = qcount;
      am.qcount_max = kFallbackGatherMax;
      e = launch_sweep_mfma(ix->metric, g_nqt, am, g_blocks, st, (int)((kFallbackGatherMax + g_B - 1) / g_B));
      if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("gathered fallback launch: ") + hipGetErrorString(e));
      mg.part_keys = am.part_keys;
      mg.ext_ids = ix->ext_ids.as<uint64_t>();
      mg.out_ids = reinterpret_cast<uint64_t*>(sd + o_gid);
      mg.out_scores = reinterpret_cast<float*>(sd + o_gsc);
      mg.out_n = reinterpret_cast<uint32_t*>(sd + o_gn);
      mg.n_lists = (uint32_t)g_blocks;
      mg.k = k;
      mg.active = qcount;
      mg.active_max = gather_all ? 0u : kFallbackGatherMax;
      launch_merge(true, mg, kFallbackGatherMax, st);
      ag.qcount = qcount;
      ag.qcount_max = kFallbackGatherMax;
    }
    if (!gather_all) {
      ag.part_keys = ix->s_fb_keys.as<uint64_t>();
      ag.n_rows = n;
      ag.rows = ix->rows.as<float>();
      e = launch_sweep_gemm(ix->metric, fp, ag, st, tile_needed);
      if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("exact fallback launch: ") + hipGetErrorString(e));
      MergeArgs mf{};
      mf.part_keys = ag.part_keys;
      mf.ext_ids = ix->ext_ids.as<uint64_t>();
      mf.out_ids = reinterpret_cast<uint64_t*>(sd + o_fid);
      mf.out_scores = reinterpret_cast<float*>(sd + o_fsc);
      mf.out_n = reinterpret_cast<uint32_t*>(sd + o_fn);
      mf.n_lists = fp.G;
      mf.k = k;
      mf.gate = flags;
      if (gather_ok) {  // (the GEMM pass did not run for a batch the gathered pass answered: nothing to merge)
        mf.skip_cnt = qcount;
        mf.skip_le = kFallbackGatherMax;
      }
      launch_merge(true, mf, nqg, st);
      fin.fb_ids = mf.out_ids;
      fin.fb_scores = mf.out_scores;
      fin.fb_n = mf.out_n;
    }
    // one launch: an unproven query takes the gathered pass's slot or, when that pass stood aside, the whole-tile fallback's
    if (gather_ok) {
      fin.max_listed = gather_all ? 0u : kFallbackGatherMax;
      fin.g_ids = mg.out_ids;
      fin.g_scores = mg.out_scores;
      fin.g_n = mg.out_n;
    }
    launch_select_finish(fin, st);
  }
  ix->split_flags_off = o_flags;
  ix->split_flags_n = nqg;
  ix->split_flags_stream = st;
  if (ev) (void)hipEventRecord(ev->b, st);
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}


// ---- exact Cosine / DotProduct batches with 10 < k <= kWideMaxK: the WIDE selection (sweep_wide.hip) -------------------------------
// the same eligibility as level 2 (the bf16 image of the rows, whole 64-element k-tiles), its own park-after-failure state
// VELESDB_WIDE_SMALL_K=0 (probe builds): k <= 10 stays on the block-local lists whatever the selector level (A / B probes)
static const bool g_wide_small_k = [] {
  const char* e = probe_env("VELESDB_WIDE_SMALL_K");
  return !(e && e[0] == '0');
}();
// VELESDB_WIDE_FUSE=0 (probe builds): the final bound and the pool are taken by a wide_reseed launch of their own, as between launches
static const bool g_wide_fuse = [] {
  const char* e = probe_env("VELESDB_WIDE_FUSE");
  return !(e && e[0] == '0');
}();
int select_level_wide(vdb_hip_index* ix, uint32_t nq_left, uint32_t k, bool sq8) {
  if (opt_selector(ix) < 2 || (!sq8 && opt_engine(ix) != 1) || opt_max_tile(ix) < 128) return 0;
  if (ix->metric != VDB_COSINE && ix->metric != VDB_DOT && (sq8 || ix->metric != VDB_EUCLIDEAN)) return 0;  // (SQ8: Cosine / DotProduct)
  if (ix->dim % 64 != 0 || ix->dim < 128 || ix->row_stride != ix->dim) return 0;
  // k <= 10: the block-local lists of levels 1 / 2 exist for it; selector level 3 (the default since round 6) sends Cosine / DotProduct
  // / Euclidean batches over f32 rows and the SQ8 mode's Cosine / DotProduct batches through the WIDE selection all the same — measured faster at every k (no candidate buffers, no compaction,
  // ~25 instead of 64 rows to re-score: DESIGN 4.1f) — and a handle the data defeats falls back to those levels, not to the exact kernels
  if (k == 0 || k > kWideMaxK || ix->n_rows < kGemmBf16MinRows || ix->n_rows >= 0xFFFFFF00ull) return 0;
  if (k <= kGemmBf16MaxK && (opt_selector(ix) < 3 || !g_wide_small_k)) return 0;
  if (!sq8 && ix->metric != VDB_EUCLIDEAN && sweep_mfma_lds_bytes(1, k, ix->dim) > 160 * 1024) return 0;  // (the gathered exact pass of the unproven queries)
  if (!select_chunk(nq_left, sq8 ? kSelectMinQueriesSq8 : 0)) return 0;
  if (ix->sel_stats && ix->sel_stats[2] != ix->sel_seq_seen) {
    ix->sel_seq_seen = ix->sel_stats[2];
    if (ix->sel_stats[3] == 4u && (uint64_t)ix->sel_stats[0] * 16 > ix->sel_stats[1]) ix->wide_hold = 64;  // > 1/16 unproven
  }
  if (ix->wide_hold) {
    ix->wide_hold--;
    return 0;
  }
  return 4;
}

int32_t brute_wide_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nqg, uint32_t k, uint64_t* d_ids, float* d_scores,
                       uint32_t* d_n, hipStream_t st, bool sq8) {
  // (sq8: the SQ8 storage mode's batches, VDB_SEARCH_BRUTE_SQ8 — selection over the dequantised bf16 image with its own norms, the
  // candidates re-scored with the reference's chain over the codes, storage_modes.hip's gathered sweep for the unproven)
  const bool cosn = !sq8 && cosine_normalised(ix);  // Cosine: both sides normalised before the rounding, the DotProduct instance selects
  const bool l2 = !sq8 && ix->metric == VDB_EUCLIDEAN;  // the augmented DotProduct form s = q.v - |v|^2 / 2 (sweep_split.hip), images of dim + 64 columns
  const int sel_metric = (cosn || l2) ? VDB_DOT : ix->metric;
  int32_t rc = sq8 ? ensure_sq8_select(ix, st) : (l2 ? ensure_l2_select(ix, st) : (cosn ? ensure_cosn(ix, st) : ensure_sel16(ix, st)));
  if (rc != VDB_OK) return rc;
  const uint32_t dim_a = ix->dim + 64, dim_s = ix->dim + 4;
  const uint16_t* img_rows = sq8 ? ix->sq8_img.as<uint16_t>()
                                 : (l2 ? ix->l2_img.as<uint16_t>() : (cosn ? ix->cosn_img.as<uint16_t>() : ix->rows_bf16.as<uint16_t>()));
  const uint64_t img_stride = l2 ? (uint64_t)dim_a : ((cosn || sq8) ? (uint64_t)ix->dim : ix->bf16_stride);
  const DevBuf& rho_buf = sq8 ? ix->sq8_rho : (l2 ? ix->l2_rho : (cosn ? ix->cosn_rho : ix->bf16_rho));
  const float* sel_norms = sq8 ? ix->sq8_nrm.as<float>() : ix->norms.as<float>();  // what the kernel's Cosine bound and `force` rule read
  DevBuf& list_buf = sq8 ? ix->s_part_cnt : ix->s_fb_keys;  // (storage_modes.hip's gathered fallback keeps its lists in s_fb_keys)
  const uint32_t sel_dim = l2 ? dim_a : ix->dim;  // the k-extent the selection kernel and the seed contract over
  ix->last_select_level = 4;
  ix->last_kernels |= VDB_KERNEL_SELECT_BF16;
  const uint8_t* alive = ix->any_dead ? ix->alive.as<uint8_t>() : nullptr;
  const uint32_t n = (uint32_t)ix->n_rows, dim = ix->dim;
  // the seed sample: one key per 16 rows, its k-th best is the first bound — 4 096 rows (256 keys) bound a small k well enough that the
  // first launch passes ~100 rows per query; a k of 100 needs the 1 024 keys of 16 384 rows (seed_scores_bf16: 32 us against 100)
  // (VELESDB_WIDE_SEED_ROWS=n, probe builds: the sample's size for k <= kWideSmallSeedMaxK)
  static const uint32_t g_small_seed_rows = [] {
    const char* e = probe_env("VELESDB_WIDE_SEED_ROWS");
    const long v = e ? atol(e) : 0;
    return v >= 1024 && v <= (long)kWideSeedRows ? (uint32_t)v / 256u * 256u : kSplitSeedRows;
  }();
  const uint32_t R0 = std::min<uint32_t>(k <= kWideSmallSeedMaxK ? g_small_seed_rows : kWideSeedRows, n), ngrp = (R0 + 15) / 16;
  GemmSchedule sch;
  {
    // tiles per row group of the first launches (VELESDB_WIDE_STEPS="a,b,c": schedule probes)
    static const std::array<uint32_t, 3> mult = [] {
      std::array<uint32_t, 3> m{1, 4, 16};
      if (const char* e = probe_env("VELESDB_WIDE_STEPS")) {
        unsigned a = 0, b = 0, c = 0;
        const int got = sscanf(e, "%u,%u,%u", &a, &b, &c);
        m = {got >= 1 ? a : 0u, got >= 2 ? b : 0u, got >= 3 ? c : 0u};
      }
      return m;
    }();
    const uint32_t head[3] = {mult[0], mult[1], mult[2]};
    gemm_schedule(nqg, 0, n, ix->n_cus, head, 0, &sch);
  }
  // the gathered exact pass of the unproven queries: the streaming matrix-core kernel, as many 16-query tiles per pass as k leaves room
  // for (Euclidean: the canonical vector-ALU sweep, 8 queries per pass)
  int g_nqt = 3;
  while (g_nqt > 1 && sweep_mfma_lds_bytes(g_nqt, k, dim) > 160 * 1024) g_nqt--;
  const int g_waves = g_nqt >= 2 ? kMfmaWaves2 : kMfmaWaves1;
  const uint32_t g_B = l2 ? 8u : (uint32_t)g_nqt * 16;
  const uint32_t ntiles16 = (n + 15) / 16;
  const int g_blocks = l2 ? blocks_for(ix, 8, (n + 7) / 8)
                          : (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ntiles16 + g_waves - 1) / g_waves, (int64_t)ix->n_cus));
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 15) & ~(size_t)15;
    return o;
  };
  const size_t o_tau = take((size_t)nqg * 8), o_delta = take((size_t)nqg * 4), o_qn = take((size_t)nqg * 4), o_rho = take((size_t)nqg * 4),
               o_cnt = take((size_t)nqg * 4), o_state = take((size_t)nqg * 4), o_flags = take((size_t)nqg * 4 + 16), o_qmap = take((size_t)nqg * 8),
               o_gid = take((size_t)nqg * k * 8), o_gsc = take((size_t)nqg * k * 4), o_gn = take((size_t)nqg * 4), o_nmax = take(16),
               o_extra = take((size_t)nqg * 4);
  hipError_t e;
  if ((e = ix->s_seed.reserve(off, false, st)) != hipSuccess || (e = list_buf.reserve((size_t)nqg * kWideCap * 8, false, st)) != hipSuccess ||
      (e = ix->s_part_keys.reserve((size_t)nqg * ngrp * 8, false, st)) != hipSuccess ||
      (!sq8 && (e = ix->s_part_cnt.reserve((size_t)nqg * g_blocks * k * 8, false, st)) != hipSuccess) ||
      (e = ix->s_misc.reserve(((size_t)nqg + 256) * img_stride * 2 + (l2 ? (size_t)nqg * dim_s * 4 + 16 : 0), false, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, "wide selection scratch");
  unsigned char* sd = ix->s_seed.as<unsigned char>();
  uint16_t* q16 = ix->s_misc.as<uint16_t>();
  float* qnorms = reinterpret_cast<float*>(sd + o_qn);
  float* rho_q = rho_buf.p ? reinterpret_cast<float*>(sd + o_rho) : nullptr;
  uint32_t* flags = reinterpret_cast<uint32_t*>(sd + o_flags);
  uint32_t* qcount = flags + nqg;  // (cleared with the flags)
  uint32_t* qmap = reinterpret_cast<uint32_t*>(sd + o_qmap);
  uint32_t* qslot = qmap + nqg;
  uint32_t* norm_max = reinterpret_cast<uint32_t*>(sd + o_nmax);
  EventPair* ev = next_events(ix);
  if (ev) (void)hipEventRecord(ev->a, st);
  // queries: bf16 image rows, canonical norms, rounding residuals, cleared flag words — one launch; whole 256-query tiles are staged
  if (l2) {
    float* qaug = reinterpret_cast<float*>(ix->s_misc.as<unsigned char>() + ((((size_t)nqg + 256) * img_stride * 2 + 15) & ~(size_t)15));  // (written, unused here: the exact seed's operand)
    launch_l2_augment_queries(d_q, q_stride, q16, dim_a, qaug, dim_s, nqg, dim, st);
    if (rho_q) launch_query_round_error(d_q, q_stride, rho_q, nqg, dim, st);
    PrepArgs pq{};
    pq.rows = d_q;
    pq.norms = qnorms;
    pq.row_stride = q_stride;
    pq.n_rows = nqg;
    pq.dim = dim;
    pq.words = ix->words;
    launch_prep_rows(pq, st);
    VDB_HIP(hipMemsetAsync(flags, 0, (size_t)nqg * 4 + 16, st));
  } else if (cosn) launch_seln_prep_queries(d_q, q_stride, q16, img_stride, qnorms, rho_q, flags, nqg + 4, nqg, dim, st);
  else launch_sel16_prep_queries(d_q, q_stride, q16, img_stride, qnorms, rho_q, flags, nqg + 4, nqg, dim, st);
  if (nqg % 256u) VDB_HIP(hipMemsetAsync(q16 + (size_t)nqg * img_stride, 0, (size_t)256 * img_stride * 2, st));
  if (ix->metric == VDB_DOT || l2) launch_max_norm(sel_norms, n, norm_max, st);
  WideArgs wa{};
  wa.keys = list_buf.as<uint64_t>();
  wa.eps_extra = sq8 ? 1.5e-4f : 0.0f;  // (select_eps level 3: the matrix-core score against the reference's left-to-right chain)
  wa.cnt = reinterpret_cast<uint32_t*>(sd + o_cnt);
  wa.state = reinterpret_cast<uint32_t*>(sd + o_state);
  wa.tau = reinterpret_cast<uint64_t*>(sd + o_tau);
  wa.delta = reinterpret_cast<float*>(sd + o_delta);
  wa.qnorms = qnorms;
  wa.rho_q = rho_q;
  wa.rho_max_bits = rho_q ? rho_buf.as<uint32_t>() : nullptr;
  wa.norm_max_bits = norm_max;
  wa.extra = l2 ? reinterpret_cast<float*>(sd + o_extra) : nullptr;
  wa.cap = kWideCap;
  wa.k = k;
  wa.dim = dim;
  // seed: a sample of the first rows on the bf16 pipe (one key per 16 rows), its k-th best -> the first bound
  launch_seed_scores_bf16(sel_metric, img_rows, img_stride, sel_norms, alive, q16, img_stride, qnorms, ix->s_part_keys.as<uint64_t>(), R0,
                          nqg, sel_dim, st);
  if (l2) launch_wide_seed_l2(wa, ix->s_part_keys.as<uint64_t>(), ngrp, dim_a, nqg, st);
  else launch_wide_seed(ix->metric, wa, ix->s_part_keys.as<uint64_t>(), ngrp, nqg, st);
  for (int j = 0; j < sch.n_launch; j++) {
    EventPair* evs = next_sel_events(ix);
    if (evs) (void)hipEventRecord(evs->a, st);
    e = launch_sweep_gemm_bf16_wide(sel_metric, sch.bp[j], img_rows, img_stride, sel_norms, alive, q16, img_stride, wa.tau, wa.keys, wa.cnt,
                                    wa.cap, sel_dim, nqg, st, qnorms);
    if (evs) (void)hipEventRecord(evs->b, st);
    if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("wide selection launch: ") + hipGetErrorString(e));
    // (behind the last launch: the final bound and the pool — wide_rerank_verify takes that step itself for f32 rows)
    if (j + 1 < sch.n_launch || l2 || sq8 || !g_wide_fuse) launch_wide_reseed(wa, nqg, st);
  }
  WideOutArgs wo{};
  wo.rows = ix->rows.as<float>();
  wo.norms = ix->norms.as<float>();
  wo.queries = d_q;
  wo.ext_ids = ix->ext_ids.as<uint64_t>();
  wo.out_ids = d_ids;
  wo.out_scores = d_scores;
  wo.out_n = d_n;
  wo.flags = flags;
  wo.qcount = qcount;
  wo.qmap = qmap;
  wo.qslot = qslot;
  wo.row_stride = ix->row_stride;
  wo.q_stride = q_stride;
  wo.dim_pad = (dim + 127) / 128 * 128;
  if (sq8) {
    wo.sq8_codes = ix->sq8_codes.as<uint8_t>();
    wo.sq8_min = ix->sq8_min.as<float>();
    wo.sq8_max = ix->sq8_max.as<float>();
    wo.sq8_nsq = ix->sq8_nsq.as<float>();
    wo.sq8_stride = ix->sq8_stride;
    launch_wide_rerank_sq8(ix->metric, wa, wo, nqg, st);
    SelectFinishArgs fq{};  // the reference chain for the unproven queries only, listed and gathered on the device (storage_modes.hip)
    fq.flags = flags;
    fq.qcount = qcount;
    fq.qslot = qslot;
    fq.out_ids = d_ids;
    fq.out_scores = d_scores;
    fq.out_n = d_n;
    fq.nq = nqg;
    fq.k = k;
    if (ix->sel_stats) {
      fq.stats_host = ix->sel_stats;
      fq.stats_seq = ++ix->sel_seq;
      fq.stats_level = 4u;
    }
    const int32_t rf = sq8_fallback_flagged(ix, d_q, q_stride, nqg, k, qmap, fq, st);
    if (rf != VDB_OK) return rf;
    ix->split_flags_off = o_flags;
    ix->split_flags_n = nqg;
    ix->split_flags_stream = st;
    if (ev) (void)hipEventRecord(ev->b, st);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
  }
  // VELESDB_WIDE_STAMPS=1 (probe builds): where a block of wide_rerank_verify spends its time, printed per batch (synchronises)
  static const bool g_wide_stamps = [] {
    const char* e = probe_env("VELESDB_WIDE_STAMPS");
    return e && e[0] == '1';
  }();
  static unsigned long long* stamp_buf = nullptr;
  if (g_wide_stamps && !l2) {
    if (!stamp_buf) VDB_HIP(hipMalloc(&stamp_buf, (size_t)65536 * 8 * 8));
    if (nqg <= 65536) wo.stamps = stamp_buf;
  }
  if (l2) launch_wide_rerank_l2(wa, wo, nqg, st);
  else launch_wide_rerank(ix->metric, wa, wo, nqg, g_wide_fuse, st);
  if (wo.stamps) {
    std::vector<unsigned long long> h((size_t)nqg * 8);
    VDB_HIP(hipStreamSynchronize(st));
    VDB_HIP(hipMemcpy(h.data(), stamp_buf, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (uint32_t b = 0; b < nqg; b++) {
      t0 = std::min(t0, h[(size_t)b * 8]);
      t1 = std::max(t1, h[(size_t)b * 8 + 6]);
    }
    double seg[6] = {0, 0, 0, 0, 0, 0}, start = 0, raw = 0, pool = 0;
    uint32_t late = 0;
    for (uint32_t b = 0; b < nqg; b++) {
      const unsigned long long* s = &h[(size_t)b * 8];
      for (int i = 0; i < 6; i++) seg[i] += (double)(s[i + 1] - s[i]) * 0.01;
      start += (double)(s[0] - t0) * 0.01;
      late += (s[0] - t0) > 500 ? 1u : 0u;  // started more than 5 us behind the first block
      raw += (double)(s[7] >> 32);
      pool += (double)(s[7] & 0xFFFFFFFFu);
    }
    fprintf(stderr,
            "[wide stamps] %u blocks, first start -> last end %.1f us; mean per block (us): list+bound+pool %.2f | query + first step %.2f | first "
            "chunk's chains %.2f | other chunks %.2f | rank+write %.2f | proof %.2f; mean start offset %.2f us, %u blocks started > 5 us late; "
            "mean list %.1f, pool %.1f\n",
            nqg, (double)(t1 - t0) * 0.01, seg[0] / nqg, seg[1] / nqg, seg[2] / nqg, seg[3] / nqg, seg[4] / nqg, seg[5] / nqg, start / nqg, late,
            raw / nqg, pool / nqg);
  }
  // unproven queries (an overflowed list, a pool beyond one block, non-finite data): listed on the device, answered by the exact
  // streaming kernel in gathered mode — one corpus pass per g_B listed queries, none when nothing is listed
  SweepArgs am{};
  am.rows = ix->rows.as<float>();
  am.norms = ix->norms.as<float>();
  am.alive = alive;
  am.queries = d_q;
  am.part_keys = ix->s_part_cnt.as<uint64_t>();
  am.row_stride = ix->row_stride;
  am.q_stride = q_stride;
  am.n_rows = n;
  am.dim = dim;
  am.nq = g_B;
  am.k = k;
  am.qmap = qmap;
  am.qcount = qcount;
  am.qcount_max = nqg;
  if (l2) {
    launch_sweep_f32(VDB_EUCLIDEAN, 8, am, g_blocks, st, (int)((nqg + 7) / 8));
  } else {
    e = launch_sweep_mfma(ix->metric, g_nqt, am, g_blocks, st, (int)((nqg + g_B - 1) / g_B));
    if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("gathered fallback launch: ") + hipGetErrorString(e));
  }
  MergeArgs mg{};
  mg.part_keys = am.part_keys;
  mg.ext_ids = ix->ext_ids.as<uint64_t>();
  mg.out_ids = reinterpret_cast<uint64_t*>(sd + o_gid);
  mg.out_scores = reinterpret_cast<float*>(sd + o_gsc);
  mg.out_n = reinterpret_cast<uint32_t*>(sd + o_gn);
  mg.n_lists = (uint32_t)g_blocks;
  mg.k = k;
  mg.active = qcount;
  launch_merge(!l2, mg, nqg, st);
  SelectFinishArgs fin{};
  fin.flags = flags;
  fin.qcount = qcount;
  fin.qslot = qslot;
  fin.g_ids = mg.out_ids;
  fin.g_scores = mg.out_scores;
  fin.g_n = mg.out_n;
  fin.out_ids = d_ids;
  fin.out_scores = d_scores;
  fin.out_n = d_n;
  fin.nq = nqg;
  fin.k = k;
  if (ix->sel_stats) {
    fin.stats_host = ix->sel_stats;
    fin.stats_seq = ++ix->sel_seq;
    fin.stats_level = 4u;
  }
  launch_select_finish(fin, st);
  ix->split_flags_off = o_flags;
  ix->split_flags_n = nqg;
  ix->split_flags_stream = st;
  if (ev) (void)hipEventRecord(ev->b, st);
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}


}  // namespace vdb
