/*
 * vdb_oracle.h — C API of the CPU ORACLE for the velesdb-core HNSW hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (velesdb_amd/, include/) may
 * include, link, import or execute this.  Allowed users: tests/, bench.py's
 * cpu_baseline leg, __graft_entry__.smoke().
 *
 * The oracle is a from-scratch C++ restatement of the reference algorithm
 * (reference = cyberlife-coder/velesdb v1.4.1, Rust; it cannot be compiled here:
 * no rustc/cargo in the image).  Every function cites the reference file:line
 * it follows in vdb_oracle.cpp.
 *
 * PARITY PIN STATUS
 *   integer paths (Hamming, Jaccard counts, heap/graph logic, level RNG,
 *   file format): pinned by the reference's own known-answer tests
 *   (tests/golden/reference_kats.json lists each with its source line).
 *   f32 paths: the reference pins its kernels only to tolerances (1e-5 abs /
 *   1e-4 rel vs a naive scalar loop); the bit-level summation order inside the
 *   third-party `wide` 0.7.33 crate (f32x8::reduce_add, mul_add fusion) is
 *   NOT verifiable in this image => "parity unpinned" at the bit level for
 *   mode R, pinned at the reference's own tolerance.
 */
#ifndef VDB_ORACLE_H
#define VDB_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* DistanceMetric discriminants = reference on-disk order
 * (index/hnsw/index/constructors.rs:204-210). */
enum { VO_COSINE = 0, VO_EUCLIDEAN = 1, VO_DOT = 2, VO_HAMMING = 3, VO_JACCARD = 4 };

/* Arithmetic mode of the f32 kernels. */
enum {
  VO_MODE_R = 0,      /* reference production engine: SimdDistance -> simd::*_fast ->
                         simd_avx512 wide16 (4 x f32x8 accumulators, FMA)            */
  VO_MODE_C = 1,      /* canonical order shared bit-for-bit with the HIP kernels:
                         float4-chunk c -> lane c%64, per-lane fmaf chain, xor-butterfly
                         32,16,8,4,2,1; cosine uses sqrt(nq)*sqrt(nv) like mode R    */
  VO_MODE_SCALAR = 2, /* reference CpuDistance scalar engine (native/distance.rs:158-217) */
  VO_MODE_NATIVE = 3, /* reference NativeSimdDistance: simd_native.rs 16-lane AVX-512F shape */
  VO_MODE_M = 5,      /* matrix-core order of the HIP MFMA sweep (cosine / dot brute force only): ONE fmaf chain per
                         pair over k = 128U + 16m + 4kk + c for U, then m in 0..7, c in 0..3, kk in 0..3, the
                         vectors zero-padded to a multiple of 128; norms (cosine) in the canonical mode-C order; every
                         other kernel of mode M is mode C */
  VO_MODE_R_NOFMA = 4 /* mode R with wide::mul_add un-fused (a*b+c, two roundings): what
                         `wide` emits when compiled without target_feature=fma         */
};

/* Result order among equal distances. */
enum {
  VO_TIE_REFERENCE = 0, /* reference artefact: BinaryHeap backing-array order then stable sort */
  VO_TIE_CANONICAL = 1  /* declared canonical: (distance total-order asc, node id asc)        */
};

/* ---- distance kernels (reference: simd.rs facade, simd_avx512.rs, simd_explicit.rs) ---- */
float vo_dot(int mode, const float* a, const float* b, size_t n);
float vo_sql2(int mode, const float* a, const float* b, size_t n);
/* mode C's sums by the plain per-element loop: what the vectorised reduction must reproduce bit for bit */
float vo_dot_c_plain(const float* a, const float* b, size_t n);
float vo_sql2_c_plain(const float* a, const float* b, size_t n);
float vo_euclidean(int mode, const float* a, const float* b, size_t n);
float vo_cosine(int mode, const float* a, const float* b, size_t n);
float vo_norm_sq(int mode, const float* a, size_t n); /* canonical/wide sum of squares */
float vo_norm(const float* a, size_t n);              /* simd::norm (simd.rs:240-242) */
float vo_hamming(const float* a, const float* b, size_t n);
float vo_jaccard(const float* a, const float* b, size_t n);
uint32_t vo_hamming_binary(const uint64_t* a, const uint64_t* b, size_t n);
/* single-accumulator f32x8 variants (simd_explicit.rs:50-189) */
float vo_dot_simd8(const float* a, const float* b, size_t n);
float vo_sql2_simd8(const float* a, const float* b, size_t n);
float vo_cosine_simd8(const float* a, const float* b, size_t n);
/* DistanceEngine::distance (native/distance.rs:75-85): HNSW-internal distance */
float vo_distance(int metric, int mode, const float* a, const float* b, size_t n);
/* HnswIndex::compute_distance (index/search.rs:30-38): raw brute-force score */
float vo_compute_distance(int metric, int mode, const float* a, const float* b, size_t n);
void vo_batch_distance(int metric, int mode, const float* q, const float* rows, size_t nrows,
                       size_t dim, float* out);
void vo_batch_compute_distance(int metric, int mode, const float* q, const float* rows,
                               size_t nrows, size_t dim, float* out);
float vo_transform_score(int metric, float raw_distance); /* backend_adapter.rs:160-168 */
uint64_t vo_ef_search(int quality, uint64_t custom, uint64_t k); /* params.rs:309-319 */
int vo_total_cmp(float a, float b);                       /* f32::total_cmp */
void vo_sort_results(int metric, uint64_t* ids, float* scores, uint64_t n); /* DistanceMetric::sort_results, distance.rs:95-103 (stable) */
int vo_higher_is_better(int metric);                      /* distance.rs:76-82 */

/* ---- level RNG (graph.rs:368-403) ---- */
uint64_t vo_xorshift64_next(uint64_t* state); /* returns new state */
uint32_t vo_random_layer(uint64_t* state, double level_mult);

/* ---- Rust std BinaryHeap emulation, exposed for tests ---- */
/* pushes keys (dist,node) into a max-heap then returns backing-array order */
void vo_heap_order_after_pushes(const float* d, const uint64_t* node, size_t n, int min_heap,
                                uint64_t* out_nodes);

/* ---- NativeHnsw<D> (native/graph.rs) ---- */
typedef struct vo_hnsw vo_hnsw;
vo_hnsw* vo_hnsw_new(uint32_t dim, int metric, int mode, uint32_t M, uint32_t ef_construction);
void vo_hnsw_free(vo_hnsw*);
void vo_hnsw_set_alpha(vo_hnsw*, float alpha);
uint64_t vo_hnsw_insert(vo_hnsw*, const float* vec); /* returns node id */
/* order among equal distances in the candidate list that insert() feeds to select_neighbors:
 * VO_TIE_REFERENCE (default, heap artefact) or VO_TIE_CANONICAL ((distance, node id) ascending, what
 * the GPU construction kernels use). Identical whenever no two candidates are at the same distance. */
void vo_hnsw_set_build_tie(vo_hnsw*, int tie);
/* host threads of the batch-synchronous build's search phase (every node of a batch is searched against the graph as it was before the
 * batch: independent, read-only); the graph that results does not depend on it */
void vo_hnsw_set_build_threads(vo_hnsw*, uint32_t nthreads);
/* batch-synchronous insertion (deterministic stand-in for parallel_insert, backend_adapter.rs:110-123):
 * all n searches see the graph as it was before the call, links applied sources ascending. n==1 == insert */
void vo_hnsw_insert_batch_sync(vo_hnsw*, const float* vecs, uint64_t n);
uint32_t vo_build_batch_size(uint64_t linked, uint32_t max_batch);
void vo_hnsw_build_batched(vo_hnsw*, const float* vecs, uint64_t n, uint32_t max_batch);
uint64_t vo_hnsw_len(const vo_hnsw*);
uint32_t vo_hnsw_max_layer(const vo_hnsw*);
int64_t vo_hnsw_entry_point(const vo_hnsw*); /* -1 if none */
uint32_t vo_hnsw_num_layers(const vo_hnsw*);
/* returns count; writes up to cap ids */
uint32_t vo_hnsw_neighbors(const vo_hnsw*, uint32_t layer, uint64_t node, uint64_t* out,
                           uint32_t cap);
const float* vo_hnsw_vector(const vo_hnsw*, uint64_t node);
/* search(query,k,ef) (graph.rs:251-270); returns count */
uint32_t vo_hnsw_search(const vo_hnsw*, const float* q, uint32_t k, uint32_t ef, int tie,
                        uint64_t* out_nodes, float* out_dist);
/* batch of searches on nthreads host threads (rayon-style, batch.rs:180-194) with the reference's
 * neighbour prefetch; totals of distance evaluations / expansions are returned if non-NULL */
void vo_hnsw_search_batch(const vo_hnsw*, const float* queries, uint32_t nq, uint32_t k, uint32_t ef, int tie,
                          uint32_t nthreads, uint64_t* out_nodes, float* out_dist, uint32_t* out_n,
                          uint64_t* total_n_dist, uint64_t* total_n_expand);
/* NativeHnsw::search_multi_entry (graph.rs:288-348): advances the graph's xorshift stream by min(num_probes, 4) - 1 draws */
uint32_t vo_hnsw_search_multi_entry(vo_hnsw*, const float* q, uint32_t k, uint32_t ef, uint32_t num_probes, int tie,
                                    uint64_t* out_nodes, float* out_dist);
uint64_t vo_hnsw_rng_state(const vo_hnsw*);
/* search statistics of the last vo_hnsw_search on this thread: distance evals, expansions */
void vo_hnsw_last_stats(uint64_t* n_dist, uint64_t* n_expand);
uint64_t vo_hnsw_search_layer_single(const vo_hnsw*, const float* q, uint64_t entry,
                                     uint32_t layer);
uint32_t vo_hnsw_search_layer(const vo_hnsw*, const float* q, const uint64_t* eps, uint32_t neps,
                              uint32_t ef, uint32_t layer, int tie, uint64_t* out_nodes,
                              float* out_dist, uint32_t cap);
uint32_t vo_hnsw_select_neighbors(const vo_hnsw*, const uint64_t* cand, const float* cand_dist,
                                  uint32_t n, uint32_t max_neighbors, uint64_t* out);
int vo_hnsw_file_dump(const vo_hnsw*, const char* dir, const char* basename);
vo_hnsw* vo_hnsw_file_load(const char* dir, const char* basename, int metric, int mode);

/* ---- HnswIndex (index/hnsw/index/ *.rs): id mapping + quality + brute force ---- */
typedef struct vo_index vo_index;
vo_index* vo_index_new(uint32_t dim, int metric, int mode, uint32_t M, uint32_t ef_construction);
vo_index* vo_index_new_auto(uint32_t dim, int metric, int mode); /* HnswParams::auto */
void vo_index_free(vo_index*);
int vo_index_insert(vo_index*, uint64_t id, const float* vec); /* 1 inserted, 0 duplicate skipped */
int vo_index_remove(vo_index*, uint64_t id);
uint64_t vo_index_len(const vo_index*);
vo_hnsw* vo_index_graph(vo_index*);
/* quality: 0 Fast,1 Balanced,2 Accurate,3 Perfect,4 Custom(custom_ef) */
uint32_t vo_index_search_with_quality(const vo_index*, const float* q, uint32_t k, int quality,
                                      uint32_t custom_ef, int tie, uint64_t* out_ids,
                                      float* out_scores);
uint32_t vo_index_search_brute_force(const vo_index*, const float* q, uint32_t k,
                                     uint64_t* out_ids, float* out_scores);
/* search_batch_parallel semantics (always HNSW, batch.rs:180-194), nthreads host threads */
void vo_index_search_batch(const vo_index*, const float* queries, uint32_t nq, uint32_t k,
                           int quality, uint32_t custom_ef, int tie, uint32_t nthreads,
                           uint64_t* out_ids, float* out_scores, uint32_t* out_n);
uint32_t vo_index_search_with_rerank(const vo_index*, const float* q, uint32_t k,
                                     uint32_t rerank_k, uint64_t* out_ids, float* out_scores);
/* search_with_rerank_quality (search.rs:297-350) with a choice of tie order for the candidate search */
uint32_t vo_index_search_with_rerank_quality(const vo_index*, const float* q, uint32_t k, uint32_t rerank_k,
                                             int quality, uint32_t custom_ef, int tie, uint64_t* out_ids,
                                             float* out_scores);

/* ---- flat brute-force scan used by the CPU baseline (no index object) ---- */
/* exact top-k of rows by compute_distance, canonical tie order (score, row asc); nthreads>=1 */
void vo_scan_topk(int metric, int mode, const float* rows, uint64_t nrows, uint32_t dim,
                  const float* queries, uint32_t nq, uint32_t k, uint32_t nthreads,
                  uint64_t* out_rows, float* out_scores);

/* DualPrecisionHnsw (native/dual_precision.rs, native/quantization.rs): scalar quantiser, u8 codes, integer L2^2,
 * int8 graph traversal + exact f32 re-rank.  vo_sq_train: per-dimension min / scale = 255/range / inv_scale over n
 * vectors (the reference trains on the first min(1000, max_elements) inserts). */
void vo_sq_train(const float* vecs, uint64_t n, uint32_t dim, float* min_vals, float* scales, float* inv_scales);
void vo_sq_quantize(const float* vecs, uint64_t n, uint32_t dim, const float* min_vals, const float* scales,
                    uint8_t* codes);
uint32_t vo_sq_l2(const uint8_t* a, const uint8_t* b, uint32_t dim);
/* search_with_config(use_int8_traversal) -> search_int8_traversal: returns count; out = node ids + exact engine
 * distances, best first; counters = int8 distance evaluations / layer-0 expansions */
uint32_t vo_dual_search_int8(const vo_hnsw*, const uint8_t* codes, const float* min_vals, const float* scales,
                             const float* q, uint32_t k, uint32_t ef_search, uint32_t oversampling, int tie,
                             uint64_t* out_nodes, float* out_dist, uint64_t* n_dist_int8, uint64_t* n_expand);

/* half_precision.rs BF16 path: exact scan over bf16-rounded rows/queries, f32 sequential accumulation
 * (metric: VO_COSINE or VO_DOT); canonical tie order; out arrays are [nq][k] */
void vo_round_bf16(const float* in, float* out, uint64_t n);
void vo_scan_topk_bf16(int metric, const float* rows, uint64_t nrows, uint32_t dim, const float* queries,
                       uint32_t nq, uint32_t k, uint32_t nthreads, uint64_t* out_rows, float* out_scores);

int vo_cpu_has_avx512f(void);
/* page placement for many-thread runs (see the Pool notes in vdb_oracle.cpp): a copy of the corpus first-touched by the
 * threads vo_scan_topk will read it with; re-placement of a loaded graph's vectors in 2 MiB round-robin chunks */
float* vo_alloc_spread(const float* rows, uint64_t nrows, uint32_t dim, uint32_t nthreads);
void vo_free_spread(float* p);
void vo_hnsw_spread(vo_hnsw*, uint32_t nthreads);
/* ---- storage modes (core/quantization.rs): sign-bit and SQ8 (per-vector min/max) codes, asymmetric distances ---- */
void vo_binary_quantize(const float* v, uint32_t dim, uint8_t* out /* ceil(dim/8) */);
uint32_t vo_binary_hamming(const uint8_t* a, const uint8_t* b, uint32_t nbytes);
void vo_sq8_quantize(const float* v, uint32_t dim, uint8_t* data, float* out_min, float* out_max);
void vo_sq8_dequantize(const uint8_t* data, float mn, float mx, uint32_t dim, float* out);
float vo_sq8_dot(const float* q, const uint8_t* data, float mn, float mx, uint32_t dim);
float vo_sq8_l2sq(const float* q, const uint8_t* data, float mn, float mx, uint32_t dim, int simd);
float vo_sq8_cosine(const float* q, const uint8_t* data, float mn, float mx, uint32_t dim, int simd);
float vo_sq8_norm_sq(const uint8_t* data, float mn, float mx, uint32_t dim);
void vo_scan_topk_sq8(int metric, const float* rows, uint64_t nrows, uint32_t dim, const float* queries, uint32_t nq,
                      uint32_t k, uint32_t nthreads, uint64_t* out_rows, float* out_scores);
void vo_scan_topk_binary(const float* rows, uint64_t nrows, uint32_t dim, const float* queries, uint32_t nq, uint32_t k,
                         uint64_t* out_rows, float* out_scores);
/* Range-sharded exact search: merge of per-shard top-k lists = DistanceMetric::sort_results
 * (core/distance.rs:95-103: stable sort by total_cmp, descending for higher-is-better metrics) over the shard-major
 * concatenation, cut to k.  rec = [S][nq][k] wire records (id low, id high, score bits; a slot past a shard's count
 * carries id = ~0 and score bits 0xFFFFFFFF) — the checker of merge_shards_topk (csrc/shard_group.hip). */
void vo_merge_shard_records(const uint32_t* rec, uint32_t S, uint32_t nq, uint32_t k, int higher_is_better,
                            uint64_t* out_ids, float* out_scores, uint32_t* out_n);
const char* vo_build_info(void);

#ifdef __cplusplus
}
#endif
#endif
