// vdb_kernels.hpp — argument blocks and host-callable launchers of the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "vdb_gemm_schedule.hpp"

namespace vdb {

struct SweepArgs {
  const float* rows;       // [n_rows][row_stride] f32, row_stride % 4 == 0, 16-B aligned
  const float* norms;      // [n_rows] canonical sqrt(sum sq) (cosine only)
  const uint8_t* alive;    // [n_rows] 0 = soft-deleted (nullable = all alive)
  const float* queries;    // [nq][q_stride]
  uint64_t* part_keys;     // [nq][n_blocks][k], unused slots = invalid key
  uint32_t* part_cnt;      // unused (kept for ABI stability of the arg block)
  uint64_t row_stride;     // floats
  uint64_t q_stride;       // floats
  uint32_t n_rows;
  uint32_t dim;
  uint32_t nq;             // queries in this pass (<= B)
  uint32_t k;
  // gathered mode (sweep_topk_mfma_f32; the exact pass over the few queries a selection batch could not prove): block row
  // blockIdx.y serves the listed queries qmap[B y .. B y + B - 1] of `queries`, part_keys slots follow the list; the launch
  // does nothing when more than qcount_max queries are listed (the GEMM-structured fallback takes those batches) and the
  // GEMM-structured kernel does nothing when qcount is set and *qcount <= qcount_max
  const uint32_t* qmap;
  const uint32_t* qcount;
  uint32_t qcount_max;
};

struct MergeArgs {
  const uint64_t* part_keys;  // [nq][n_lists][k], unused slots = invalid key
  const uint32_t* part_cnt;   // unused
  const uint64_t* ext_ids;    // [n_rows] external ids (nullable -> row + row_base)
  uint64_t* out_ids;          // [nq][k]
  float* out_scores;          // [nq][k]
  uint32_t* out_n;            // [nq]
  uint64_t row_base;
  uint32_t n_lists;
  uint32_t k;                 // entries per partial list
  uint32_t k_out;             // 0 = k; otherwise the number of entries kept per query (out_* are [nq][k_out])
  const uint32_t* active;     // nullable: *active = number of leading queries that hold data (the other blocks exit)
  uint32_t active_max;        // with `active`: nothing to do at all when *active > active_max (0 = no such limit)
  uint32_t list_stride;       // lists per query in part_keys (0 = n_lists): merge only the first n_lists of them
  const uint32_t* skip_cnt;   // nullable: nothing to do when *skip_cnt <= skip_le (the pass that would have filled the lists did not run)
  uint32_t skip_le;
  const uint32_t* gate;       // nullable: query q is merged only when gate[q] != 0 (the lists of the others may hold anything)
  // selection stage, between two launches (sweep_split.hip): the next launch's bound in the same pass — reseed_tau[q] = key of
  // (the reseed_k-th best merged score lowered by 2 reseed_delta[q]), kKeyInvalid while fewer than reseed_k keys exist
  const float* reseed_delta;  // nullable
  uint64_t* reseed_tau;
  uint32_t reseed_k;
};

struct EuclidRerankArgs {
  const float* rows;            // f32 rows of the index
  const float* queries;
  const uint64_t* cand_rows;    // [nq][kp] internal rows, best approximate value first (merge_topk output)
  const float* cand_approx;     // [nq][kp] approximate squared distances
  const uint32_t* cand_n;       // [nq]
  const uint64_t* ext_ids;
  const uint32_t* norm_max_bits;  // device scalar: bits of max |v| over the rows (filled by launch_euclid_rerank)
  uint64_t* out_ids;            // [nq][k]
  float* out_scores;            // [nq][k] canonical Euclidean distances
  uint32_t* out_n;              // [nq]
  uint32_t* flags;              // [nq] 1 = not proven exact: re-run through the exact sweep
  uint64_t row_stride, q_stride;
  uint32_t dim, k, kp;
};
constexpr uint32_t kEuclidSlack = 16;  // extra candidates per query kept by the approximate selection

struct BitsArgs {
  const uint32_t* bits;     // [n_rows][words]
  const uint32_t* qbits;    // [nq][words]
  const uint8_t* alive;
  uint64_t* part_keys;      // [nq][n_waves][k]
  uint32_t* part_cnt;
  uint32_t n_rows;
  uint32_t words;           // multiple of 4
  uint32_t k;
};

struct PrepArgs {
  const float* rows;
  float* norms;      // nullable
  uint32_t* bits;    // nullable
  uint64_t row_stride;
  uint32_t row0;
  uint32_t n_rows;
  uint32_t dim;
  uint32_t words;
};

struct ScoreArgs {
  const float* query;
  const float* rows;  // [n_rows][dim]
  float* out;
  uint64_t n_rows;
  uint32_t dim;
  int32_t kind;       // 0 engine distance, 1 raw
  int32_t aligned16;  // rows, query 16-B aligned and dim % 4 == 0
};

// ---- HNSW traversal (hnsw_kernels.hip) ----------------------------------------------------
constexpr int kSearchRegSlots = 4;  // register-resident candidate list of the traversal kernel: 256 entries
constexpr int kBuildRegSlots = 8;   // ... of the construction kernel: 512 entries (ef_construction <= 448)
constexpr int kMaxLayers = 16;  // random_layer caps levels at 15 (native/graph.rs:401)
struct HnswLayerRef {
  const uint32_t* nbr;  // [capacity][stride] neighbour ids
  const uint32_t* cnt;  // [capacity] neighbour counts
  uint32_t stride;
  uint32_t pad;
};
struct HnswSearchArgs {
  const float* rows;
  const float* norms;       // cosine only
  const uint32_t* bits;     // hamming / jaccard only
  const uint8_t* alive;     // nullable
  const uint64_t* ext_ids;  // nullable -> node id
  const float* queries;     // [nq][q_stride]
  uint64_t row_stride, q_stride;
  HnswLayerRef layers[kMaxLayers];
  uint32_t* visited;        // [slots][vis_words] bitmaps, all zero between launches
  uint32_t* vlog;           // [slots][vlog_cap] ids whose bit was set (for the per-query clean-up)
  uint64_t vis_words;
  uint64_t* out_ids;        // [nq][k]
  float* out_scores;        // [nq][k]
  uint32_t* out_n;          // [nq]; 0xFFFFFFFF = candidate list overflow (caller must re-run with a larger cap)
  unsigned long long* stats;  // [3] += distance evaluations, expansions, expansions whose neighbour ids were prefetched (pf_ids)
  uint32_t dim, words, n_rows, nq, k, ef, cap, nbmax, vlog_cap, max_layer, entry_point;
  int32_t metric;
  uint32_t n_cus;       // launch sizing only
  uint32_t list_slots;  // 0: candidate list in LDS; kSearchRegSlots: in registers (ef + 64 <= slots * 64)
  uint32_t vis_log2;    // > 0: the visited set is an exact hash set of 2^vis_log2 entries in LDS at byte offset vis_off (VisSet,
  uint32_t vis_off;     // vdb_hnsw_device.hpp) instead of the HBM bitmap; a query that would pass 3/4 of it reports overflow
  uint32_t pf_ids;      // 1 = the neighbour list of the predicted next pop is requested with the current one's (hnsw_kernels.hip)
  uint32_t lat_spec;    // latency-mode kernel: 1 = all neighbours' rows are fetched beside the visited test (corpora beyond the
                        // Infinity Cache: the walk is a chain of memory round trips), 0 = the test (LDS set) first, then only
                        // the unvisited neighbours' rows (cache-resident corpora: the round trip is short and small graphs
                        // revisit most neighbours)
  uint32_t rerank_k;  // > 0: search_with_rerank (search.rs:118-160): the first rerank_k results are re-scored with the
                      // raw compute_distance, stable-sorted in the metric's order and cut to k
  // NativeHnsw::search_multi_entry (graph.rs:288-348): nullable; [nq][3] node ids drawn by the host from the graph's xorshift
  // stream (0xFFFFFFFF = no draw): further entry points of the layer-0 search beside the descent's result, duplicates skipped
  const uint32_t* extra_eps;
  uint32_t raw_small_ef;  // NativeHnsw-level call with ef_search < 4: the RAWEF instance (hnsw_kernels.hip)
};
size_t hnsw_lds_bytes(uint32_t cap, uint32_t nbmax, uint32_t dim, uint32_t words, int metric);  // without the visited set
// returns hipSuccess or the launch error; grid = slots blocks of 256 threads
hipError_t launch_hnsw_search(const HnswSearchArgs& a, int slots, hipStream_t st);

size_t sweep_lds_bytes(int B, uint32_t k, uint32_t dim, int cpl);
int sweep_cpl_for_dim(uint32_t dim);
void launch_sweep_f32(int metric, int B, const SweepArgs& a, int blocks, hipStream_t st, int groups = 1);
// large query tiles (B = 16 / 32) with the queries in LDS; dim % 256 == 0 and dim <= 1024 only
constexpr int kQldsWaves16 = 4;   // waves per block for B = 16 (3 blocks per CU)
constexpr int kQldsWaves32 = 16;  // waves per block for B = 32
size_t sweep_qlds_lds_bytes(int B, uint32_t k, uint32_t dim, int waves);
hipError_t launch_sweep_f32_qlds(int metric, int B, const SweepArgs& a, int blocks, hipStream_t st);
// MFMA f32 sweep (cosine / dot): nqt = 1 (16 queries per pass) or 2 (32)
constexpr int kMfmaWaves1 = 8;   // waves per block for one 16-query tile
constexpr int kMfmaWaves2 = 16;  // ... for two tiles (the 96 KiB query fragments fill most of the LDS)
size_t sweep_mfma_lds_bytes(int nqt, uint32_t k, uint32_t dim);
hipError_t launch_sweep_mfma(int metric, int nqt, const SweepArgs& a, int blocks, hipStream_t st, int groups = 1);
// GEMM-structured f32 sweep for large query batches (sweep_gemm.hip): 128-row x 32*nqf-query block tiles
struct GemmPlan {
  uint32_t nqt;   // query tiles
  uint32_t qper;  // queries per tile
  uint32_t G;     // row groups = partial top-k lists per query
  int nqf;        // 16-query accumulator tiles per wave (block tile = 32 * nqf queries; big: 256)
  int blocks;
  size_t lds;
  bool big;       // the 256-row x 256-query tile, one 8-wave block per CU (bf16 batches that fill 256-query tiles, k <= 16)
};
constexpr uint64_t kRowSlack = 256;         // rows allocated past the capacity of the row arrays (whole-tile reads)
constexpr uint32_t kGemmBigMinQueries = 224;
constexpr uint32_t kSelectMinQueries = 16;     // selection stage: smallest batch (VELESDB_SELECT_MIN_QUERIES overrides).  Round 3: 80 -> 16 — a
                                               // 16-query call 0.57 ms against 0.71 on the exact streaming kernel, 64 queries 0.62 against 1.07
                                               // (Euclidean 0.69 against 1.37): profiles/r03ab_select_min_queries.log
constexpr uint32_t kGemmMinQueries = 64;    // below this the streaming kernels (HBM-bound) are faster
constexpr uint32_t kGemmMaxK = 48;          // candidate buffers hold <= 64 keys per query (one per lane when compacted)
constexpr uint32_t kGemmMaxQueries = 1024;  // per launch (bounds the partial-list scratch)
size_t sweep_gemm_lds_bytes(int nqf, uint32_t k, bool big = false);
void sweep_gemm_plan(uint32_t nq, uint32_t n_rows, int n_cus, uint32_t k, GemmPlan* p, bool allow_big = false);
hipError_t launch_sweep_gemm(int metric, const GemmPlan& p, const SweepArgs& a, hipStream_t st,
                             const uint32_t* tile_needed = nullptr);
// bf16 variant of the same kernel (dim % 64 == 0): rows16 = the bf16 row copy, queries16 = launch_round_queries_bf16 output
void launch_round_queries_bf16(const float* q, uint64_t q_stride, uint16_t* out, uint64_t out_stride, uint32_t nq,
                               uint32_t dim, hipStream_t st);
hipError_t launch_sweep_gemm_bf16(int metric, const GemmPlan& p, const uint16_t* rows16, uint64_t row_stride,
                                  const float* norms, const uint8_t* alive, const uint16_t* queries16, uint64_t q_stride,
                                  uint64_t* part_keys, uint32_t n_rows, uint32_t dim, uint32_t nq, uint32_t k,
                                  hipStream_t st);
// bf16 GEMM-distance sweep for big batches (sweep_gemm_bf16.hip): 256 x 256 block tile, LDS-DMA staging, seeded thresholds
// (struct Bf16GemmPlan, sweep_gemm_bf16_plan: vdb_gemm_schedule.hpp)
constexpr uint32_t kGemmBf16MaxK = 10;          // candidate buffers of 12 keys per query
constexpr uint32_t kGemmBf16MinRows = 1u << 16; // below this the 128 x 128 kernel serves the batch alone
constexpr uint32_t kGemmBf16SeedRows = 1u << 14; // rows of the seeding pre-pass (k-th best key over a prefix of the corpus)
hipError_t launch_sweep_gemm_bf16_glds(int metric, const Bf16GemmPlan& p, const uint16_t* rows16, uint64_t row_stride,
                                       const float* norms, const uint8_t* alive, const uint16_t* queries16, uint64_t q_stride,
                                       const uint64_t* tau0, uint64_t* part_keys, uint32_t list_stride, uint32_t list_off,
                                       uint32_t dim, uint32_t nq, uint32_t k, hipStream_t st, bool split = false,
                                       const float* qnorms = nullptr, uint64_t* blk_tau = nullptr, const float* qnorms_half = nullptr);
// ---- ONE launch schedule for every user of the 256 x 256 selection kernel (the bf16 result path and the exact / SQ8 selection
// stage of index.hip, the bit metrics of bits_gemm.hip): a corpus is swept in a few launches of growing size, each starting from
// bounds re-seeded out of everything swept before it — a block's epilogue costs ~0.2 us per candidate it has to finish, so the
// rows swept under a weak bound are kept few.
// (kGemmMaxLaunches, struct GemmSchedule, gemm_schedule: vdb_gemm_schedule.hpp)
// the launches of a schedule: pre(j) in front of launch j, post(j, lists_written, last) behind it (the merge + re-seed between two
// launches belongs there); partial lists land at list_first + the row groups before the launch
template <class Pre, class Post>
static inline hipError_t run_gemm_schedule(const GemmSchedule& s, int metric, const uint16_t* rows16, uint64_t row_stride, const float* norms,
                                           const uint8_t* alive, const uint16_t* queries16, uint64_t q_stride, const uint64_t* tau0, uint64_t* part_keys,
                                           uint32_t list_stride, uint32_t list_first, uint32_t dim, uint32_t nq, uint32_t k, hipStream_t st, bool split,
                                           const float* qnorms, uint64_t* blk_tau, const float* qnorms_half, Pre&& pre, Post&& post);
// norms of the rounded queries of a result-mode batch (qnorms_half above)
void launch_query_norms_bf16(const uint16_t* q16, uint64_t q_stride, float* out, uint32_t nq, uint32_t dim, hipStream_t st);
void launch_seed_tau(const uint64_t* ids, const float* scores, const uint32_t* n, uint64_t* tau0, uint64_t* list,
                     uint32_t list_stride, uint32_t nq, uint32_t k, hipStream_t st, bool hib = true);
template <class Pre, class Post>
static inline hipError_t run_gemm_schedule(const GemmSchedule& s, int metric, const uint16_t* rows16, uint64_t row_stride, const float* norms,
                                           const uint8_t* alive, const uint16_t* queries16, uint64_t q_stride, const uint64_t* tau0, uint64_t* part_keys,
                                           uint32_t list_stride, uint32_t list_first, uint32_t dim, uint32_t nq, uint32_t k, hipStream_t st, bool split,
                                           const float* qnorms, uint64_t* blk_tau, const float* qnorms_half, Pre&& pre, Post&& post) {
  uint32_t list_off = list_first;
  for (int j = 0; j < s.n_launch; j++) {
    pre(j);
    const hipError_t e = launch_sweep_gemm_bf16_glds(metric, s.bp[j], rows16, row_stride, norms, alive, queries16, q_stride, tau0, part_keys, list_stride,
                                                     list_off, dim, nq, k, st, split, qnorms, blk_tau, qnorms_half);
    if (e != hipSuccess) return e;
    list_off += s.bp[j].G;
    post(j, list_off, j + 1 == s.n_launch);
  }
  return hipSuccess;
}
// bits_gemm.hip: four-bit image of packed bit rows (+ bit counts as floats) for the four-bit GEMM distance of Hamming / Jaccard
void launch_bits_expand(int metric, const uint32_t* bits, uint32_t words, uint8_t* img, uint32_t img_stride, float* cnt, uint32_t row0,
                        uint32_t n_rows, uint32_t dim, float fill, hipStream_t st);
uint32_t bits_image_stride(uint32_t dim);
// bf16 GEMM-distance sweep (cosine / dot over a bf16 copy of the rows): nqt in {1, 2, 4, 6} 16-query tiles
constexpr int kBf16WavesBig = 16;    // waves per block for nqt >= 4 (one block per CU)
constexpr int kBf16WavesSmall = 8;   // ... for nqt <= 2
size_t sweep_bf16_lds_bytes(int nqt, uint32_t k, uint32_t dim);
// rho_max_bits (nullable): device scalar, raised to the largest |x - bf16(x)| / |x| of the converted rows (select_eps_q)
void launch_prep_bf16(const float* rows, uint64_t row_stride, uint16_t* out, uint64_t out_stride, float* norms,
                      uint32_t row0, uint32_t n_rows, uint32_t dim, hipStream_t st, uint32_t* rho_max_bits = nullptr);
hipError_t launch_sweep_bf16(int metric, int nqt, const uint16_t* rows, uint64_t row_stride, const float* norms,
                             const uint8_t* alive, const float* queries, uint64_t q_stride, uint64_t* part_keys,
                             uint32_t n_rows, uint32_t dim, uint32_t nq, uint32_t k, int blocks, hipStream_t st);
void launch_merge(bool higher_is_better, const MergeArgs& m, uint32_t nq, hipStream_t st);
void launch_max_norm(const float* norms, uint32_t n_rows, uint32_t* out_bits, hipStream_t st);  // bits of max |v| (NaN propagates)
// ---- exact f32 Cosine / Dot batches through split-bf16 selection + exact re-scoring + proof (sweep_split.hip) ----
constexpr uint32_t kSplitPool = 32;  // candidates per query that are re-scored exactly (level 1: split selection)
constexpr uint32_t kSelect16Pool = 64;  // the same for level 2 (plain bf16 selection: a wider error band to cover)
constexpr uint32_t kSplitSeedRows = 4096;  // rows of the seed sweep
constexpr uint32_t kSeedIsSample = 0xFFFFFFFFu;  // "seed rows" argument of the approximate seed kernels: the seed only supplied bounds,
                                                 // its rows are swept again by the selection launches (pool slot 0 stays empty)
struct SplitRerankArgs {
  const float* rows;            // f32 rows of the index
  const float* norms;           // canonical row norms (cosine)
  const float* queries;         // original f32 queries
  const float* qnorms;          // canonical query norms
  const uint64_t* cand_rows;    // [nq][k2] internal rows, best pool score first
  const float* cand_scores;     // [nq][k2] pool scores
  const uint32_t* cand_n;       // [nq]
  const uint64_t* blk_tau;      // [nq][lists] the bound every selection block ended with
  const float* delta;           // [nq] error bound of an approximate score
  const uint64_t* ext_ids;
  uint64_t* out_ids;            // [nq][k]
  float* out_scores;            // [nq][k] exact scores (the exact kernel's bits)
  uint32_t* out_n;              // [nq]
  uint32_t* flags;              // [nq] 1 = not proven
  uint32_t* tile_needed;        // [ceil(nq / fb_qper)] query tiles of the exact fallback launch that hold a flagged query
  uint64_t row_stride, q_stride;
  uint32_t dim, dim_pad, k, k2, lists, fb_qper;
  // SQ8 storage mode (storage_modes.hip): candidates are re-scored with the reference's asymmetric distances over the
  // codes (dot_product_quantized_simd / cosine_similarity_quantized_simd, core/quantization.rs:410-554) instead of `rows`
  const uint32_t* norm_max_bits; // Euclidean instance (l2_rerank_verify): max row norm (launch_max_norm)
  const uint8_t* sq8_codes;     // nullptr = f32 rows
  const float* sq8_min;
  const float* sq8_max;
  const float* sq8_nsq;
  uint64_t sq8_stride;
  // the list of the unproven queries (sweep_split.hip list_unproven; nullable together)
  uint32_t* qcount;             // device word, zero before the launch: how many
  uint32_t* qmap;               // [nq] slot -> query, in the blocks' finishing order
  uint32_t* qslot;              // [nq] query -> slot (unproven queries only)
};
void launch_split_vectors(const float* src, uint64_t src_stride, uint16_t* out, float* norms, uint32_t row0, uint32_t n,
                          uint32_t dim, hipStream_t st);
void launch_split_seed(int metric, const uint64_t* ids, const float* scores, const uint32_t* n, const float* qnorms,
                       const uint32_t* norm_max_bits, uint64_t* tau0, float* delta, uint64_t* list, uint64_t* blk_tau,
                       uint32_t list_stride, uint32_t nq, uint32_t k, uint32_t klist, uint32_t dim, int level, hipStream_t st,
                       const float* rho_q = nullptr, const uint32_t* rho_max_bits = nullptr);
// level 2's measured rounding residuals (sweep_split.hip select_eps_q): rho_q[nq] = |q - bf16(q)| / |q| of a batch
void launch_query_round_error(const float* q, uint64_t q_stride, float* rho_q, uint32_t nq, uint32_t dim, hipStream_t st);
// the front of a level-2 / 3 batch in one launch: bf16 image rows, canonical norms, rho_q (nullable), cleared flag words
void launch_sel16_prep_queries(const float* q, uint64_t q_stride, uint16_t* img, uint64_t img_stride, float* qnorms, float* rho_q,
                               uint32_t* zero_words, uint32_t n_zero, uint32_t nq, uint32_t dim, hipStream_t st);
// Cosine batches over NORMALISED images (sweep_split.hip): the rows' image (v / |v| rounded to bf16, + the largest residual ratio) and the
// front of a batch (q / |q| image rows, canonical norms, residual ratios, cleared flag words)
void launch_seln_rows(const float* rows, uint64_t row_stride, const float* norms, uint16_t* out, uint64_t out_stride, uint32_t row0, uint32_t n,
                      uint32_t dim, uint32_t* rho_max_bits, hipStream_t st);
void launch_seln_prep_queries(const float* q, uint64_t q_stride, uint16_t* img, uint64_t img_stride, float* qnorms, float* rho_q,
                              uint32_t* zero_words, uint32_t n_zero, uint32_t nq, uint32_t dim, hipStream_t st);
// level 2's seed on the bf16 pipe: a plain GEMM of the first rows x the batch into keys [nq][seed_rows], and the seed kernel for
// approximate seed scores (tau = A_k - 2 delta; slot 0 of the pool with its bound)
void launch_seed_scores_bf16(int metric, const uint16_t* rows16, uint64_t row_stride, const float* norms, const uint8_t* alive,
                             const uint16_t* q16, uint64_t q_stride, const float* qnorms, uint64_t* keys, uint32_t seed_rows,
                             uint32_t nq, uint32_t dim, hipStream_t st);
void launch_split_seed_approx(int metric, const uint64_t* ids, const float* scores, const uint32_t* n, const float* qnorms,
                              const uint32_t* norm_max_bits, uint64_t* tau0, float* delta, uint64_t* list, uint64_t* blk_tau,
                              uint32_t list_stride, uint32_t nq, uint32_t k, uint32_t klist, uint32_t seed_rows, uint32_t dim, int level,
                              hipStream_t st, const float* rho_q = nullptr, const uint32_t* rho_max_bits = nullptr);
// pool_select.hip: the next launch's bound / the K2 best of a candidate pool [nq][list_stride][ks] by radix selection, without a merge
// (supported: <= 12 288 keys in the first n_lists lists of a query, k2 <= 128)
bool pool_select_supported(uint32_t n_lists, uint32_t ks, uint32_t k2);
void launch_pool_kth_reseed(const uint64_t* pool, uint32_t n_lists, uint32_t list_stride, uint32_t ks, uint32_t k, const float* delta, uint64_t* tau0,
                            uint32_t nq, hipStream_t st);
void launch_pool_topk(const uint64_t* pool, uint32_t n_lists, uint32_t list_stride, uint32_t ks, uint32_t k2, uint64_t* out_ids, float* out_scores,
                      uint32_t* out_n, uint32_t nq, hipStream_t st);
// the same straight from a SAMPLE seed's keys [nq][ngrp <= 256] (no merge in front): tau from their k-th best, delta, an empty pool slot 0
void launch_split_seed_sample(int metric, const uint64_t* keys, uint32_t ngrp, const float* qnorms, const uint32_t* norm_max_bits, uint64_t* tau0,
                              float* delta, uint64_t* list, uint64_t* blk_tau, uint32_t list_stride, uint32_t nq, uint32_t k, uint32_t klist, uint32_t dim,
                              int level, hipStream_t st, const float* rho_q = nullptr, const uint32_t* rho_max_bits = nullptr);
void launch_split_reseed(const uint64_t* ids, const float* scores, const uint32_t* n, const float* delta, uint64_t* tau0,
                         uint32_t nq, uint32_t k, uint32_t kout, hipStream_t st);
void launch_split_rerank(int metric, const SplitRerankArgs& a, uint32_t nq, hipStream_t st);
// Euclidean batches through the selection stage (sweep_split.hip): augmented images, seed, re-scoring + proof
void launch_l2_augment_rows(const float* rows, uint64_t row_stride, const float* norms, uint16_t* img, uint32_t dim_a, float* seed,
                            uint32_t dim_s, uint32_t seed_rows, uint32_t row0, uint32_t n, uint32_t dim, hipStream_t st,
                            uint32_t* rho_max_bits = nullptr);
void launch_l2_augment_queries(const float* q, uint64_t q_stride, uint16_t* img, uint32_t dim_a, float* qaug, uint32_t dim_s, uint32_t nq,
                               uint32_t dim, hipStream_t st);
void launch_l2_seed(const uint64_t* ids, const float* scores, const uint32_t* n, const float* qnorms, const uint32_t* norm_max_bits,
                    uint64_t* tau0, float* delta, uint64_t* list, uint64_t* blk_tau, uint32_t list_stride, uint32_t nq, uint32_t k,
                    uint32_t klist, uint32_t dim_a, float extra_rel, hipStream_t st, const float* rho_q = nullptr,
                    const uint32_t* rho_max_bits = nullptr, uint32_t approx_seed_rows = 0);
void launch_l2_rerank(const SplitRerankArgs& a, uint32_t nq, hipStream_t st);
// The end of a selection batch, one launch.  An unproven query (flags[q] != 0) takes its exact result: the GATHERED pass's slot
// qslot[q] when that pass ran — g_* given and (max_listed == 0 or *qcount <= max_listed) — otherwise the whole-tile fallback's own
// slot q (fb_* nullable: no such pass).  Block 0 posts {unproven, queries, seq, level} to pinned host memory (stats_host nullable;
// seq last, behind a system fence; no synchronisation).
struct SelectFinishArgs {
  const uint32_t* flags;
  const uint32_t* qcount;
  const uint32_t* qslot;
  uint32_t max_listed;
  const uint64_t* g_ids;
  const float* g_scores;
  const uint32_t* g_n;
  const uint64_t* fb_ids;
  const float* fb_scores;
  const uint32_t* fb_n;
  uint64_t* out_ids;
  float* out_scores;
  uint32_t* out_n;
  uint32_t nq, k;
  volatile uint32_t* stats_host;
  uint32_t stats_seq, stats_level;
};
void launch_select_finish(const SelectFinishArgs& a, hipStream_t st);
void launch_euclid_rerank(const EuclidRerankArgs& a, const float* norms, uint32_t n_rows, uint32_t nq, hipStream_t st);
void launch_sweep_bits(int metric, const BitsArgs& a, int blocks, uint32_t nq, hipStream_t st);
// B (8 or 32) queries per corpus pass; blocks = row blocks (= partial lists per query), grid.y = ceil(nq / B)
size_t sweep_bits_batch_lds_bytes(int B, uint32_t words, uint32_t k);
hipError_t launch_sweep_bits_batch(int metric, int B, const BitsArgs& a, int blocks, uint32_t nq, hipStream_t st);
// same popcount core, lock-free selection (candidate buffers + compaction); k <= kBitsTileMaxK
constexpr uint32_t kBitsTileMaxK = 48;
size_t sweep_bits_tile_lds_bytes(int B, uint32_t words);
hipError_t launch_sweep_bits_tile(int metric, int B, const BitsArgs& a, int blocks, uint32_t nq, hipStream_t st);
struct BitsPlan {
  int blocks;  // row blocks = partial top-k lists per query
  int B;       // queries per corpus pass (0: the per-query kernel)
  bool tile;   // lock-free selection (k <= kBitsTileMaxK)
};
// a call of one or two packed-bit queries in ONE launch (sweep.hip sweep_bits_fused): query packing, sweep, per-block lists and — in the
// block that draws the last ticket — the merge
struct BitsFusedArgs {
  const uint32_t* bits;  // [n_rows][words]
  const float* q;        // [nq][q_stride] f32 queries (device)
  uint64_t q_stride;
  const uint8_t* alive;  // nullable
  uint64_t* part_keys;   // [nq][blocks][k]
  uint32_t* tickets;     // [nq], zero on entry; the last block of a query leaves it zero
  uint32_t n_rows, words, dim, k;
  uint32_t sign_rule;    // how a query value becomes a bit: 0 = x > 0.5 (prep_rows: Hamming / Jaccard indexes), 1 = x >= 0 (sign_bits_rows: Binary storage mode)
  uint32_t probe_skip;   // probe builds only (VELESDB_BITS_FUSED_SKIP; results are WRONG with it): 1 = no last-block merge, 2 = no extraction, 4 = no loads
  MergeArgs m;           // the output side (part_keys / n_lists / k are filled by the launcher)
};
bool sweep_bits_fused_supported(uint32_t words, uint32_t nq, uint32_t k);
int sweep_bits_fused_blocks(uint64_t n_rows, int n_cus);
hipError_t launch_sweep_bits_fused(int metric, BitsFusedArgs a, int blocks, uint32_t nq, hipStream_t st);
BitsPlan plan_bits_sweep(uint64_t n_rows, int n_cus, uint32_t words, uint32_t nq, uint32_t k);
hipError_t launch_bits_plan(int metric, const BitsPlan& p, const BitsArgs& a, uint32_t nq, hipStream_t st);
// radix_sort.hip: hand-written stable LSD radix sort of (u64 key, u64 value) pairs by a list of digits (<= 8 bits each)
struct RadixDigit {
  uint32_t shift, bits;
};
size_t radix_sort_scratch_bytes(uint32_t n);
hipError_t radix_sort_pairs_u64(uint64_t* keys_a, uint64_t* vals_a, uint64_t* keys_b, uint64_t* vals_b, uint32_t n, const RadixDigit* digits,
                                int n_digits, void* scratch, bool* result_in_b, hipStream_t st);
void launch_prep_rows(const PrepArgs& a, hipStream_t st);
void launch_score_rows(int metric, const ScoreArgs& a, hipStream_t st);

}  // namespace vdb
