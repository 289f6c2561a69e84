/*
 * velesdb_hip.h — C ABI of libvelesdb_hip.so: an MI355X (gfx950) implementation of
 * velesdb-core's HNSW similarity-search hot path.
 *
 * This is the drop-in boundary.  A Rust `velesdb-hip` shim binds these
 * symbols and implements the reference's own traits over the opaque handle:
 *   trait VectorIndex          crates/velesdb-core/src/index/mod.rs:30-83
 *   impl VectorIndex for HnswIndex   .../index/hnsw/index/trait_impl.rs:8-71
 *   HnswIndex inherent methods .../index/hnsw/index/search.rs, batch.rs, constructors.rs
 *   trait DistanceEngine       .../index/hnsw/native/distance.rs:14-28
 *   GpuAccelerator             crates/velesdb-core/src/gpu/gpu_backend.rs:33,136,157,355,397
 *
 * Conventions
 *   - every call returns an int32 status (0 = VDB_OK, >0 informational, <0 error); nothing
 *     unwinds across the boundary; vdb_hip_last_error() gives the thread-local message.
 *   - all buffers are caller-owned, plain pointers + sizes, row-major, little-endian.
 *   - "host" entry points take host pointers (what a Rust &[f32] is); "_dev" entry points take
 *     device pointers and a hipStream_t (as void*) and enqueue without synchronising.
 *   - a handle may be used from any thread (internally synchronised: Send + Sync).
 *   - there is NO CPU fallback: without a HIP device every compute call fails with
 *     VDB_ERR_NO_DEVICE (the reference's GpuAccelerator::new() == None, gpu_backend.rs:33).
 *   - result order: best first; equal scores are ordered by internal insertion index
 *     ascending (declared canonical tie-break; the reference's tie order is an artefact of
 *     heap / hash-map layout, SURVEY.md §8a note 8).
 */
#ifndef VELESDB_HIP_H
#define VELESDB_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vdb_hip_index vdb_hip_index; /* opaque */

/* DistanceMetric, same discriminants as the reference's on-disk order
 * (index/hnsw/index/constructors.rs:204-210; core/distance.rs:16-38). */
enum vdb_metric {
  VDB_COSINE = 0,
  VDB_EUCLIDEAN = 1,
  VDB_DOT = 2,
  VDB_HAMMING = 3,
  VDB_JACCARD = 4
};

enum vdb_status {
  VDB_OK = 0,
  VDB_DUPLICATE_IGNORED = 1,  /* insert of an existing id: no-op (trait_impl.rs:23-25) */
  VDB_ERR_INVALID_ARG = -1,
  VDB_ERR_DIM_MISMATCH = -2,  /* the shim panics "… dimension mismatch: expected {}, got {}" */
  VDB_ERR_NO_DEVICE = -3,
  VDB_ERR_HIP = -4,
  VDB_ERR_IO = -5,
  VDB_ERR_OOM = -6,
  VDB_ERR_UNSUPPORTED = -7,
  VDB_ERR_STATE = -8
};

/* search `mode` */
enum vdb_search_mode {
  VDB_SEARCH_AUTO = 0,  /* HnswIndex::search_with_quality (search.rs:59-94): brute force when
                           len<=100, else HNSW with `ef`; scores = transform_score           */
  VDB_SEARCH_BRUTE = 1, /* HnswIndex::search_brute_force (search.rs:176-219): exact scan, raw
                           similarity/distance scores, metric.sort_results order             */
  VDB_SEARCH_HNSW = 2,  /* always the graph (search_batch_parallel, batch.rs:180-194)        */
  VDB_SEARCH_HNSW_INT8 = 4, /* DualPrecisionHnsw::search_with_config(use_int8_traversal) (native/dual_precision.rs:
                           223-441): the graph is walked with integer L2^2 distances between u8 codes (4x fewer bytes
                           per visited node), the k * oversampling best are re-scored with the exact f32 distance.
                           `ef` = ef_search; needs vdb_hip_index_train_quantizer.  Scores as in VDB_SEARCH_HNSW. */
  VDB_SEARCH_BRUTE_SQ8 = 5, /* exact scan of the f32 queries over the SQ8 codes of the rows with the reference's asymmetric
                           distances (core/quantization.rs:410-554): Cosine -> cosine_similarity_quantized_simd, DotProduct
                           -> dot_product_quantized_simd (best = largest), Euclidean -> euclidean_squared_quantized_simd (the
                           SQUARED distance, best = smallest).  Needs vdb_hip_index_set_storage_mode(VDB_STORAGE_SQ8).  */
  VDB_SEARCH_BRUTE_BINARY = 6, /* exact scan by BinaryQuantizedVector::hamming_distance (quantization.rs:123-135) between
                           the sign bits of the query and of every row; scores = the distance as f32, smallest first.
                           Needs VDB_STORAGE_BINARY; any metric.                                              */
  VDB_SEARCH_BRUTE_BF16 = 3 /* exact scan over the bf16 copy of the rows with bf16-rounded queries and f32
                           accumulation on the matrix cores: half_precision::dot_product / cosine_similarity on
                           VectorData::BF16 (half_precision.rs:199-255).  Cosine / DotProduct only; needs
                           vdb_hip_index_enable_bf16.  Raw scores, best first.                       */
};

/* StorageMode (core/quantization.rs:17-29): which quantised copy of the rows the index keeps next to the f32 rows */
enum vdb_storage_mode {
  VDB_STORAGE_FULL = 0,  /* f32 only (default)                                                                  */
  VDB_STORAGE_SQ8 = 1,   /* QuantizedVector: one byte per dimension, per-vector min / max (quantization.rs:204-255) */
  VDB_STORAGE_BINARY = 2 /* BinaryQuantizedVector: bit = (x >= 0.0), LSB first (quantization.rs:48-86)           */
};

/* score convention of vdb_hip_batch_distance */
enum vdb_distance_kind {
  VDB_KIND_ENGINE = 0, /* DistanceEngine::distance (native/distance.rs:75-85): 1-cos, sqrt(l2),
                          -dot, hamming, 1-jaccard                                          */
  VDB_KIND_RAW = 1,    /* HnswIndex::compute_distance / GpuAccelerator::batch_* (search.rs:30-38;
                          gpu_backend.rs:157,355,397): cos, sqrt(l2), dot, hamming, jaccard.  With VDB_DOT this is also
                          simd::cosine_similarity_normalized / batch_cosine_normalized (simd_avx512.rs:390-422: the
                          cosine of pre-normalised vectors IS their dot product)                     */
  VDB_KIND_SQUARED = 2 /* simd::squared_l2_distance (simd.rs:207-211): VDB_EUCLIDEAN without the sqrt */
};

/* ---- device discovery: GpuAccelerator::new()/is_available() (gpu_backend.rs:33,136) ---- */
int32_t vdb_hip_device_count(int32_t* n);
int32_t vdb_hip_device_name(int32_t device, char* buf, size_t cap);

/* how a handle spreads over several GPUs of one node */
enum vdb_shard_mode {
  VDB_SHARD_REPLICA = 0, /* every device holds every row (and the graph); a query batch is split between the devices,
                            no collective — the mode of the graph path                                           */
  VDB_SHARD_RANGE = 1    /* contiguous ranges of the internal rows: row r lives on device min(r / C, n - 1),
                            C = ceil(max_elements / n); exact searches = per-shard top-k, ONE RCCL all-gather of
                            k (id, score) records per query and device, merge — BASELINE configs[4]              */
};

/* ---- lifecycle: HnswIndex::with_params (constructors.rs:117-160) ----
 * M = max_connections, M0 = 2M (native/graph.rs:62); max_elements is a capacity hint (storage grows; with
 * VDB_SHARD_RANGE it also fixes the rows per shard).  `devices` = n_devices HIP device ordinals (NULL / 0 = device 0).
 * One VectorIndex object whatever is behind it (index/mod.rs:30-83): with n_devices > 1 the handle owns one shard per
 * device and every entry point below works on the whole; results (ids, ranks, score bits, tie order) of the exact
 * search modes are those of the single-device index over the same rows.  The same device may be named more than once
 * (co-located shards: the exchange is then a device-to-device copy instead of RCCL — a test configuration).
 * HNSW modes on a VDB_SHARD_RANGE handle search one graph per shard and merge (recall differs from one big graph).
 * A single-device index (one shard) holds at most 2^32 - 512 rows: inserts past that answer VDB_ERR_UNSUPPORTED. */
int32_t vdb_hip_index_create(uint32_t dim, int32_t metric, uint32_t M, uint32_t ef_construction,
                             uint64_t max_elements, const int32_t* devices, int32_t n_devices, int32_t shard_mode,
                             vdb_hip_index** out);
/* One process per GPU (torchrun, MPI): every rank creates its own single-device index = ONE shard of a range-sharded
 * corpus (rank order = row order) and joins the group with the 128-byte id rank 0 generated and distributed out of
 * band.  From then on the exact search modes of this handle return the GLOBAL top-k on every rank: local top-k, one
 * ncclAllGather of nq * k 12-byte (u64 id, f32 score) records per rank, merge kernel.  Collective: every rank must
 * call search with the same queries, k and mode.  world == 1 is allowed (the collective degenerates). */
#define VDB_COMM_ID_BYTES 128
int32_t vdb_hip_comm_unique_id(uint8_t* id /* VDB_COMM_ID_BYTES */);
int32_t vdb_hip_index_join_group(vdb_hip_index* idx, const uint8_t* id, int32_t rank, int32_t world);
/* n_shards = devices behind the handle (1 for a plain index), shard_mode as created, rank / world of the process
 * group (0 / 1 without one), transport of the exchange: 0 none, 1 RCCL, 2 device-to-device copies */
int32_t vdb_hip_index_shard_info(vdb_hip_index* idx, int32_t* n_shards, int32_t* shard_mode, int32_t* rank,
                                 int32_t* world, int32_t* transport);
void vdb_hip_index_destroy(vdb_hip_index* idx);

/* ---- VectorIndex::insert (index/mod.rs:46, trait_impl.rs:10-36) ---- */
int32_t vdb_hip_index_insert(vdb_hip_index* idx, uint64_t id, const float* vec, uint32_t vec_len);
/* HnswIndex::insert_batch_sequential / insert_batch_parallel (batch.rs:83-149): rows are
 * dim floats each; *inserted = number of new ids (duplicates skipped). */
int32_t vdb_hip_index_insert_batch(vdb_hip_index* idx, const uint64_t* ids, const float* vecs_rowmajor,
                                   uint64_t n, uint64_t* inserted);
/* HnswIndex::insert_batch_parallel (batch.rs:83-108).  The reference inserts with rayon and is
 * non-deterministic; here the batch is inserted batch-synchronously (every node of a sub-batch
 * searches the graph as it was before the sub-batch, links applied sources ascending): deterministic,
 * restated in oracle/vdb_oracle.cpp hnsw_insert_batch_sync.  max_batch = largest sub-batch (0 =
 * default 2048; 1 = identical to insert_batch). */
int32_t vdb_hip_index_insert_batch_parallel(vdb_hip_index* idx, const uint64_t* ids, const float* vecs_rowmajor,
                                            uint64_t n, uint32_t max_batch, uint64_t* inserted);
/* ScalarQuantizer::train (native/quantization.rs:191-233) on the first sample_rows rows (0 = min(1000, rows), what
 * DualPrecisionHnsw uses) + u8 codes for every present and future row (+1 byte per element of HBM). */
int32_t vdb_hip_index_train_quantizer(vdb_hip_index* idx, uint32_t sample_rows);
/* DualPrecisionHnsw::is_quantizer_trained (native/dual_precision.rs:117-120): *trained = 1 once vdb_hip_index_train_quantizer ran
 * on this handle, else 0 (the codes are a derived image: they are not part of a saved directory, a loaded handle starts untrained). */
int32_t vdb_hip_index_quantizer_trained(const vdb_hip_index* idx, int32_t* trained);
/* DualPrecisionHnsw::search_with_config (native/dual_precision.rs:259-278) for a batch, the DualPrecisionConfig passed per call as
 * the reference passes it: the int8 graph traversal + exact f32 re-scoring of the k * oversampling_ratio best (dual_precision.rs:
 * 284-321) only when the quantiser is trained AND use_int8_traversal != 0 AND the index holds >= min_index_size vectors; otherwise
 * the plain f32 graph search — NativeHnsw::search with ef_search AS GIVEN (dual_precision.rs:269,274; graph.rs:251-270): the same
 * answer as VDB_SEARCH_HNSW when 0 < k <= ef_search; with ef_search < k at most ef_search results, ef_search = 0 acts as 1 (no
 * SearchQuality rule at this level; a device group keeps the HnswIndex rule).  oversampling_ratio = 0 or > 64 =>
 * VDB_ERR_INVALID_ARG (reference default 4, min_index_size default 10 000).  Ids / scores as VDB_SEARCH_HNSW reports them.  Does
 * not touch VDB_OPT_INT8_OVERSAMPLING. */
int32_t vdb_hip_index_search_with_config(vdb_hip_index* idx, const float* queries_rowmajor, uint32_t nq, uint32_t k,
                                         uint32_t ef_search, uint32_t oversampling_ratio, int32_t use_int8_traversal,
                                         uint64_t min_index_size, uint64_t* out_ids, float* out_scores, uint32_t* out_n);
/* DualPrecisionConfig::oversampling_ratio (default 4) of VDB_SEARCH_HNSW_INT8, process-wide */
int32_t vdb_hip_set_int8_oversampling(uint32_t ratio);
/* Collection StorageMode (quantization.rs:17-29; collection/core/crud.rs:66-82 quantises every upserted vector):
 * encodes every present and future row on the device (SQ8: +dim+12 bytes per row, Binary: +dim/8) for the
 * VDB_SEARCH_BRUTE_SQ8 / VDB_SEARCH_BRUTE_BINARY scans.  Switching modes drops the previous codes. */
int32_t vdb_hip_index_set_storage_mode(vdb_hip_index* idx, int32_t mode);
/* the stored code of one vector in the reference's serialisation: QuantizedVector::to_bytes (min f32, max f32, dim
 * bytes; quantization.rs:289-295) or BinaryQuantizedVector::to_bytes (dimension u32, ceil(dim/8) bytes; :155-169).
 * *len = bytes needed / written. */
int32_t vdb_hip_index_get_quantized(vdb_hip_index* idx, uint64_t id, uint8_t* out, size_t cap, size_t* len);
/* keeps a bf16 copy (round to nearest even, VectorData::from_f32_slice(.., BF16), half_precision.rs:94-101) of every
 * row next to the f32 rows, for VDB_SEARCH_BRUTE_BF16; +2 bytes per element of HBM */
int32_t vdb_hip_index_enable_bf16(vdb_hip_index* idx);
/* links every row that is not in the graph yet (rows that arrived through upload/upload_dev), same
 * schedule as insert_batch_parallel; afterwards the HNSW search modes are available. */
int32_t vdb_hip_index_build_graph(vdb_hip_index* idx, uint32_t max_batch);
/* bulk upload without graph construction: vectors become searchable by VDB_SEARCH_BRUTE at
 * once; the graph is absent until vdb_hip_index_build_graph / load_reference_files.
 * (HnswIndex keeps exact search available independently of the graph, search.rs:176-219.) */
int32_t vdb_hip_index_upload(vdb_hip_index* idx, const uint64_t* ids, const float* vecs_rowmajor,
                             uint64_t n, uint64_t* inserted);
/* same, rows already resident in HBM on the index's device (device pointer), ids = base..base+n-1 */
int32_t vdb_hip_index_upload_dev(vdb_hip_index* idx, uint64_t id_base, const float* d_vecs_rowmajor,
                                 uint64_t n, void* stream);

/* ---- VectorIndex::remove / len / dimension / metric (index/mod.rs:62-82) ---- */
int32_t vdb_hip_index_remove(vdb_hip_index* idx, uint64_t id, int32_t* removed); /* soft delete */
int32_t vdb_hip_index_len(const vdb_hip_index* idx, uint64_t* n);
int32_t vdb_hip_index_dimension(const vdb_hip_index* idx, uint32_t* dim);
int32_t vdb_hip_index_metric(const vdb_hip_index* idx, int32_t* metric);
/* HnswIndex::tombstone_count / vacuum (index/hnsw/index/vacuum.rs:45-52,110-184).  tombstone_ratio = count / node_count,
 * needs_vacuum = ratio > 0.2 (:60-76).  vacuum rebuilds the graph over the active vectors with HnswParams::auto(dim);
 * *count = vectors in the rebuilt index. */
int32_t vdb_hip_index_tombstone_count(const vdb_hip_index* idx, uint64_t* n);
int32_t vdb_hip_index_vacuum(vdb_hip_index* idx, uint64_t* count);
/* number of graph nodes (NativeHnsw::len, native/graph.rs:130-132; includes soft-deleted) */
int32_t vdb_hip_index_node_count(const vdb_hip_index* idx, uint64_t* n);

/* ---- VectorIndex::search + HnswIndex::search_with_quality/search_brute_force ----
 * ef = 0 => SearchQuality::Balanced rule max(128, 4k) (params.rs:309-319).
 * out_ids/out_scores hold k entries; *out_n <= k (soft-deleted rows can shorten HNSW results,
 * search.rs:86-91). */
int32_t vdb_hip_index_search(vdb_hip_index* idx, const float* query, uint32_t query_len, uint32_t k,
                             uint32_t ef, int32_t mode, uint64_t* out_ids, float* out_scores,
                             uint32_t* out_n);
/* HnswIndex::search_batch_parallel (batch.rs:159-197) for mode HNSW; one launch for all
 * queries.  out_ids/out_scores are nq*k, out_n is nq. */
int32_t vdb_hip_index_search_batch(vdb_hip_index* idx, const float* queries_rowmajor, uint32_t nq,
                                   uint32_t k, uint32_t ef, int32_t mode, uint64_t* out_ids,
                                   float* out_scores, uint32_t* out_n);
/* HnswIndex::search_with_rerank (search.rs:118-160) / search_with_rerank_quality (search.rs:297-350) for nq
 * queries: candidates = search_with_quality(query, rerank_k, quality) (ef = 0 => Accurate: max(512, 16*rerank_k);
 * otherwise Custom(ef)), re-scored with the raw compute_distance, stable-sorted in the metric's order, cut to
 * k.  Scores are RAW (similarity for Cosine/Dot/Jaccard, distance for Euclidean/Hamming), like search_brute_force. */
int32_t vdb_hip_index_search_rerank(vdb_hip_index* idx, const float* queries_rowmajor, uint32_t nq, uint32_t k,
                                    uint32_t rerank_k, uint32_t ef, uint64_t* out_ids, float* out_scores,
                                    uint32_t* out_n);
/* NativeHnsw::search_multi_entry (native/graph.rs:288-348; "improved recall on hard queries"): the layer-0 search starts from the
 * descent's result AND up to min(num_probes, 4) - 1 further nodes drawn from the graph's own xorshift stream (the stream that
 * draws insertion levels: like the reference, the call advances it — query i of the batch takes the draws nq sequential calls
 * would give it; duplicates are skipped; no draw when num_probes <= 1 or the graph has <= 10 nodes), same ef.  Ids / scores as
 * VDB_SEARCH_HNSW reports them.  `ef` is NativeHnsw's raw ef_search (graph.rs:343): no SearchQuality rule, no max(ef, k) — a call
 * with ef < k returns at most ef results per query (with more entry points than ef: as many as there are entry points, which is what
 * the reference's uncut `results` heap gives, graph.rs:463-468; ef = 0 therefore acts as ef = 1, NOT as "Balanced"). */
int32_t vdb_hip_index_search_multi_entry(vdb_hip_index* idx, const float* queries_rowmajor, uint32_t nq, uint32_t k, uint32_t ef,
                                         uint32_t num_probes, uint64_t* out_ids, float* out_scores, uint32_t* out_n);
/* device-resident variant: d_queries nq*dim f32 (16-byte aligned), outputs device buffers of nq*k / nq; enqueued on `stream`,
 * no host synchronisation — with two exceptions: (1) Euclidean VDB_SEARCH_BRUTE batches of >= 64 queries that the selection
 * stage does not take (dim % 64 != 0, dim < 128, k > 10, or fewer than 65 536 rows) read their per-query verdicts back once per
 * <= 1 024-query chunk; (2) the first search that needs a derived image of the rows (bf16 / split / augmented / SQ8-dequantised /
 * four-bit bit image) builds it on `stream` and waits for it, so that other search contexts may use it.
 * In HNSW mode d_out_n[i] == 0xFFFFFFFF marks a query whose LDS candidate list overflowed (needs very many exact distance
 * ties) or, in calls of at most one query per CU, whose walk visited more nodes than the LDS visited set holds (> ~24 000 at
 * ef <= 270); the host variant above re-runs such batches with a larger list and the HBM visited bitmaps by itself. */
int32_t vdb_hip_index_search_batch_dev(vdb_hip_index* idx, const float* d_queries, uint32_t nq,
                                       uint32_t k, uint32_t ef, int32_t mode, uint64_t* d_out_ids,
                                       float* d_out_scores, uint32_t* d_out_n, void* stream);

/* ---- DistanceEngine::batch_distance / GpuAccelerator::batch_{cosine_similarity,
 * euclidean_distance,dot_product} (native/distance.rs:21-24; gpu_backend.rs:157,355,397) ----
 * n rows of dim floats against one query; out has n floats, same order as the rows. */
int32_t vdb_hip_batch_distance(int32_t device, int32_t metric, int32_t kind, const float* query,
                               const float* vecs_rowmajor, uint64_t n, uint32_t dim, float* out);
int32_t vdb_hip_batch_distance_dev(int32_t metric, int32_t kind, const float* d_query,
                                   const float* d_vecs_rowmajor, uint64_t n, uint32_t dim, float* d_out,
                                   void* stream);

/* ---- the other free functions of the reference's SIMD module on the path, n vectors per call
 * (host pointers, row-major; vec_utils.hip) ---- */
/* simd::norm / simd_explicit::norm_simd (simd.rs:240-242; simd_explicit.rs:194-215): out[i] = |vecs[i]| */
int32_t vdb_hip_batch_norm(int32_t device, const float* vecs_rowmajor, uint64_t n, uint32_t dim, float* out);
/* simd::normalize_inplace (simd.rs:217-219; simd_explicit.rs:638-664): rows scaled to unit length in place; a zero
 * vector stays as it is */
int32_t vdb_hip_normalize_rows(int32_t device, float* vecs_rowmajor, uint64_t n, uint32_t dim);
/* simd_explicit::batch_dot_product (simd_explicit.rs:519-560): out[i * n + j] = dot(queries[i], vecs[j]) */
int32_t vdb_hip_batch_dot_product(int32_t device, const float* queries_rowmajor, uint32_t nq, const float* vecs_rowmajor,
                                  uint64_t n, uint32_t dim, float* out);
/* hamming_distance_binary(_fast) / jaccard_similarity_binary over packed u64 words (simd_explicit.rs:308-360,457-500):
 * one query of `words` u64 against n rows of `words` u64 */
int32_t vdb_hip_batch_hamming_binary(int32_t device, const uint64_t* query_words, const uint64_t* rows_words, uint64_t n,
                                     uint32_t words, uint32_t* out);
int32_t vdb_hip_batch_jaccard_binary(int32_t device, const uint64_t* query_words, const uint64_t* rows_words, uint64_t n,
                                     uint32_t words, float* out);

/* ---- persistence hand-off: NativeHnsw::file_dump / file_load, format v1
 * (native/backend_adapter.rs:184-381): <dir>/<basename>.vectors and .graph ---- */
int32_t vdb_hip_index_load_reference_files(vdb_hip_index* idx, const char* dir, const char* basename);
int32_t vdb_hip_index_save_reference_files(vdb_hip_index* idx, const char* dir, const char* basename);

/* HnswIndex::save / HnswIndex::load (index/hnsw/index/constructors.rs:190-287): a directory holding
 * native_hnsw.{vectors,graph} (above), native_mappings.bin (bincode 1.3.3: id_to_idx map, idx_to_id map, next_idx) and
 * native_meta.bin (bincode: dimension, metric, enable_vector_storage) — what a VelesDB collection keeps on disk for its
 * HNSW index.  load_dir creates the index (dimension / metric from the meta file, M / ef_construction from the graph
 * file, ids from the mappings; ids the reference removed are absent from the mappings and come back soft-deleted). */
int32_t vdb_hip_index_save_dir(vdb_hip_index* idx, const char* dir);
int32_t vdb_hip_index_load_dir(const char* dir, int32_t device, vdb_hip_index** out);
/* A flushed MmapStorage directory (core/storage/mmap.rs:96-160,402-455,602-626: vectors.idx = bincode
 * FxHashMap<u64 id, usize byte offset>, vectors.dat = raw f32 at those offsets) as an upload source: every vector
 * the store's index names is uploaded (no graph, as vdb_hip_index_upload) in ascending byte offset, i.e. in the order
 * the store first saw the ids.  The index's dimension must be the store's.  Ids already present are skipped.
 * VDB_ERR_IO: missing / truncated files, an offset past the end of vectors.dat ("Offset out of bounds", :563-568). */
int32_t vdb_hip_index_upload_vector_store(vdb_hip_index* idx, const char* dir, uint64_t* inserted);

/* ---- tuning options ----
 * Every option has a process-wide default (vdb_hip_set_* below) and a per-handle value: a handle follows the default until
 * vdb_hip_index_set_option gives it its own (value < 0: follow the default again).  A multi-device handle passes the option to
 * every shard.  Results never depend on an option. */
enum vdb_option {
  VDB_OPT_MAX_QUERY_TILE = 0,   /* vdb_hip_set_max_query_tile    */
  VDB_OPT_SWEEP_ENGINE = 1,     /* vdb_hip_set_sweep_engine      */
  VDB_OPT_SELECTOR_LEVEL = 2,   /* vdb_hip_set_split_selector    */
  VDB_OPT_INT8_OVERSAMPLING = 3,/* vdb_hip_set_int8_oversampling */
  VDB_OPT_KERNEL_TIMING = 4,    /* vdb_hip_set_kernel_timing     */
  /* The combining front of the host-pointer search entry points (vdb_hip_index_search / _search_batch / _search_rerank).  The
   * reference serves many threads that each search ONE query under a read lock (index/hnsw/index/search.rs:80; the server calls
   * collection.search once per request); a GPU serves that pattern at its own rate only when callers that arrive together share
   * a launch.  Calls of <= 64 queries queue up; one of the waiting callers becomes the leader, gathers the queued calls with the
   * same (k, ef, mode, rerank_k) into one batch of <= COMBINE_MAX_BATCH queries (default 256; 0 = every call launches alone),
   * runs it and hands every caller its slice.  At most COMBINE_INFLIGHT batches run at a time (default 0 = by kind of search:
   * two for graph walks — latency-bound, one CU per query, two launches overlap for free — and one for sweeps: an HBM-bound
   * pass gains nothing from a second launch beside it, and callers that split into groups wait for each other).  Batches
   * form from the calls that arrive while the launch in front of them runs; a leader that has evidence of company (the batch
   * that finished last carried several calls) waits until as many calls have arrived as that batch had callers, at most
   * COMBINE_WINDOW_US microseconds (default 100; 0 = never wait).  A lone caller is never delayed.
   * Results are per-query independent: bits do not depend on the batch a call lands in. */
  VDB_OPT_COMBINE_MAX_BATCH = 5,
  VDB_OPT_COMBINE_WINDOW_US = 6,
  VDB_OPT_COMBINE_INFLIGHT = 7,
  VDB_OPT_COUNT_ = 8
};
int32_t vdb_hip_index_set_option(vdb_hip_index* idx, int32_t option, int64_t value);
int32_t vdb_hip_index_get_option(vdb_hip_index* idx, int32_t option, int64_t* value); /* the effective value */

/* counters of the combining front since the handle was created: launches it issued, calls and queries they carried, the largest
 * batch (queries) — calls / launches is the average number of callers that shared a launch */
int32_t vdb_hip_index_combine_stats(vdb_hip_index* idx, uint64_t* launches, uint64_t* calls, uint64_t* queries,
                                    uint64_t* max_batch);

/* ---- introspection used by tests and the bench ---- */
/* neighbours of `node` on `layer`; returns count in *n, writes up to cap ids */
int32_t vdb_hip_index_get_neighbors(vdb_hip_index* idx, uint32_t layer, uint64_t node, uint32_t* out,
                                    uint32_t cap, uint32_t* n);
int32_t vdb_hip_index_graph_info(vdb_hip_index* idx, uint32_t* num_layers, uint32_t* max_layer,
                                 int64_t* entry_point);
/* counters of graph CONSTRUCTION (NativeHnsw::insert, graph.rs:158-237, with select_neighbors :526-581), cumulative since the
 * handle was created: rows whose distance to the inserted node (search_layer at ef_construction) or to a selected neighbour
 * (select_neighbors) was evaluated — the algorithmic gather traffic of the build is that x dim x 4 bytes —, the number of
 * distance phases (dependent memory round trips), the nodes inserted, and how many of the evaluated rows were select_neighbors
 * evaluations (the <= ef_construction candidate rows of a node re-read once per selected neighbour: cache hits, not HBM traffic).
 * Synchronises the device. */
int32_t vdb_hip_index_build_stats(vdb_hip_index* idx, uint64_t* rows_evaluated, uint64_t* distance_phases, uint64_t* nodes,
                                  uint64_t* select_rows);
/* counters of the last HNSW search batch: distance evaluations and expansions.
 * NOTE for every vdb_hip_index_last_* diagnostic below: with the combining front on (VDB_OPT_COMBINE_MAX_BATCH > 1, the default)
 * a host-pointer call of <= 64 queries may have travelled in ONE launch with other callers' requests; the diagnostics then describe
 * that whole combined launch (all its queries), not the caller's own call, and the next batch served by the same search context may
 * overwrite them before they are read.  A caller that wants per-call numbers sets VDB_OPT_COMBINE_MAX_BATCH to 0 on the handle
 * (bench.py reads them after single-threaded calls only). */
int32_t vdb_hip_index_last_search_stats(vdb_hip_index* idx, uint64_t* n_dist, uint64_t* n_expand);
/* of the last HNSW search batch's expansions: how many found their neighbour list already requested — the walk kernel asks for the
 * list of the nearest candidate it leaves unexpanded together with the list of the one it expands (the predicted next pop); a measure
 * of the prediction, not a count the reference has */
int32_t vdb_hip_index_last_prefetch_hits(vdb_hip_index* idx, uint64_t* hits);
/* average duration (ms) of the dominant kernel in the last search call, measured with HIP
 * events on the launch stream; 0 if timing is off.  Enable with vdb_hip_set_kernel_timing(1) — for measurements only: an event
 * record idles the GPU ~6 us on either side of the launch it brackets (a 1 024-query exact batch brackets four selection launches
 * and itself: ~50 us of a 1.7 ms batch). */
int32_t vdb_hip_set_kernel_timing(int32_t on);
/* tuning knob of the exact sweep: largest number of queries served by one corpus pass
 * (vector-ALU kernels: 1,2,4,8 register-resident tiles, 16,32 LDS-resident tiles; matrix-core streaming kernel:
 * 16, 32 or 48 queries; 128 = batches of >= 64 queries go to the GEMM-structured matrix-core kernel, up to 128
 * queries per block tile).  Default 128.  Results do not depend on it. */
int32_t vdb_hip_set_max_query_tile(uint32_t b);
/* arithmetic engine of the exact sweep for Cosine / DotProduct: 1 (default) = matrix-core kernel
 * (v_mfma_f32_16x16x4_f32: exact f32, one k-ordered fmaf chain per pair, oracle mode M); 0 = vector-ALU kernels
 * (canonical lane-chain order, oracle mode C — what Euclidean, the graph kernels and batch_distance always use).
 * Both are exact f32 arithmetic; scores differ in the last bits because the summation order does.  The choice
 * never depends on the batch size. */
int32_t vdb_hip_set_sweep_engine(int32_t engine);
/* Exact Cosine / DotProduct (and Euclidean, SQ8) batches of >= 16 queries, up to 1 024 per pass (round 2: 80 .. 256, or more that filled 256-query tiles to 7/8)
 * (k <= 10, >= 65 536 rows, dim % 32 == 0):
 * the matrix cores SELECT candidates on a reduced-precision image of rows and queries, the best candidates per query are
 * re-scored with the exact chain (oracle mode M), and every query's answer is PROVEN from an error bound or recomputed by
 * the exact kernel — same ids, ranks and score bits as the exact f32 matrix-core kernel for the whole batch.
 *   0 = no selection stage: the exact f32 matrix-core kernel;
 *   1 = split-bf16 selection (x = hi + lo, three bf16 MFMAs per product, error ~2^-15 + accumulation; 32 candidates);
 *       costs +4 bytes per element of HBM (the split image, built at first use);
 *   2 = plain bf16 selection first (one MFMA per product over the bf16 copy of the rows, error ~2^-7; 64
 *       candidates; +2 bytes per element, dim % 64 == 0), level 1 where that does not apply; a handle whose data defeats
 *       the wider bound (> 1/16 of a batch unproven: near-duplicate clusters) moves itself to level 1 for the next 64
 *       batches and then tries again.  Cosine (round 6): the image holds the NORMALISED rows v / |v| and the batch the
 *       normalised queries, so the selection is a DotProduct of unit vectors (no row norm in the kernel's bound).
 * 10 < k <= 128 (round 6; Cosine / DotProduct / Euclidean, level 2's eligibility): the WIDE selection — no block-local top-k at all:
 * every row whose approximate score passes the query's bound (k-th best approximate score seen so far - 2 x the error
 * bound, raised between the launches of the batch) becomes a candidate, all of them are re-scored exactly; a query is
 * unproven only when its candidate list overflows (4 096 per launch, 1 024 at the end) or its data is not finite.
 * Reported as level 4 by vdb_hip_index_last_select_level.  Larger k, other metrics at k > 10: the exact kernels.
 *   3 (default since round 6) = level 2, and Cosine / DotProduct / Euclidean batches over f32 rows and Cosine / DotProduct batches of the
 *       SQ8 storage mode take the WIDE selection at EVERY k <= 128 (k <= 10 included: faster than the block-local lists, ~30 instead
 *       of 64 rows to re-score); a handle with > 1/16 of a batch unproven there answers its next 64 batches by level 2's rules. */
int32_t vdb_hip_set_split_selector(int32_t level);
/* diagnostic: queries in the last split-selector batch (its last chunk of <= 1024) and how many of them the exact
 * fallback kernel answered because the selection could not be proven (near-ties inside the error bound, non-finite data) */
int32_t vdb_hip_index_last_split_stats(vdb_hip_index* idx, uint32_t* queries, uint32_t* unproven);
/* selection level (0 / 1 / 2; 3 = the SQ8 storage mode's block-local form; 4 = the WIDE selection; see vdb_hip_set_split_selector) the last
 * exact batch of this handle actually ran at */
int32_t vdb_hip_index_last_select_level(vdb_hip_index* idx, int32_t* level);
/* which kernel families served the last search call of this handle (a bit set; what a test asserts when it claims to have
 * driven a particular kernel — e.g. BASELINE configs[3] is VDB_KERNEL_GEMM_BF16_GLDS, which needs >= 65 536 rows) */
enum vdb_kernel_bit {
  VDB_KERNEL_SWEEP_VALU = 1,       /* sweep_topk_f32 / sweep_topk_f32_qlds (vector ALU, mode C)                       */
  VDB_KERNEL_SWEEP_MFMA_F32 = 2,   /* sweep_topk_mfma_f32 (streaming, <= 48 queries per corpus pass, mode M)           */
  VDB_KERNEL_GEMM_F32 = 4,         /* sweep_topk_gemm_f32 (GEMM-structured exact f32; also the selection stage's seed) */
  VDB_KERNEL_SWEEP_MFMA_BF16 = 8,  /* sweep_topk_mfma_bf16 (streaming over the bf16 rows, <= 96 queries per pass)      */
  VDB_KERNEL_GEMM_BF16 = 16,       /* register-staged bf16 GEMM kernels of sweep_gemm.hip (128 x 128 / 256 x 256)      */
  VDB_KERNEL_GEMM_BF16_GLDS = 32,  /* sweep_topk_gemm_bf16_glds reporting bf16 RESULTS (VDB_SEARCH_BRUTE_BF16)         */
  VDB_KERNEL_SELECT_BF16 = 64,     /* the same kernel as the selection stage of an exact / SQ8 batch (levels 2, 3)     */
  VDB_KERNEL_SELECT_SPLIT = 128,   /* ... its split-bf16 instance (level 1)                                            */
  VDB_KERNEL_BITS = 256,           /* packed-bit sweeps (Hamming / Jaccard / sign-bit codes)                           */
  VDB_KERNEL_SQ8 = 512,            /* sweep_topk_sq8                                                                   */
  VDB_KERNEL_HNSW = 1024,          /* hnsw_search_kernel                                                               */
  VDB_KERNEL_HNSW_INT8 = 2048,     /* hnsw_search_int8_kernel                                                          */
  VDB_KERNEL_BITS_GEMM = 4096      /* Hamming / Jaccard batches as a four-bit GEMM distance (sweep_topk_gemm_bf16_pp<.., FP4>) */
};
/* which kernels served THIS THREAD's last search on the handle: taken when that search's context was released (or, for a call the
 * combining front had another thread launch, handed back with the call's result), so a search of another thread that takes the same
 * context a moment later does not change it. */
int32_t vdb_hip_index_last_kernels(vdb_hip_index* idx, uint32_t* mask);
/* *mode = 1 if searches in VDB_SEARCH_BRUTE mode with this k run on the matrix-core kernel (mode M), else 0 */
int32_t vdb_hip_index_sweep_arith_mode(vdb_hip_index* idx, uint32_t k, int32_t* mode);
int32_t vdb_hip_index_last_kernel_ms(vdb_hip_index* idx, float* ms, uint32_t* launches);
/* with kernel timing on: the launches of the selection kernel (sweep_topk_gemm_bf16_glds) in the last search call — their
 * number and the SUM of their durations in ms (HIP events around each launch); 0 / 0 when the call had no selection stage */
int32_t vdb_hip_index_last_selection_ms(vdb_hip_index* idx, float* total_ms, uint32_t* launches);

const char* vdb_hip_last_error(void); /* thread-local, never NULL */
const char* vdb_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VELESDB_HIP_H */
