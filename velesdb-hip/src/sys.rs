//! Raw bindings of `include/velesdb_hip.h`: one declaration per symbol of the header, same order, same argument names.
//! `tests/test_abi_exports.py::test_rust_sys_matches_header` parses both files and fails on any difference (a missing or
//! extra symbol, an argument count, a pointer constness, an integer width).

#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_void};

/// `typedef struct vdb_hip_index vdb_hip_index;` — opaque.
#[repr(C)]
pub struct VdbHipIndex {
    _private: [u8; 0],
}

// enum vdb_metric (the reference's on-disk discriminants, index/hnsw/index/constructors.rs:204-210)
pub const VDB_COSINE: i32 = 0;
pub const VDB_EUCLIDEAN: i32 = 1;
pub const VDB_DOT: i32 = 2;
pub const VDB_HAMMING: i32 = 3;
pub const VDB_JACCARD: i32 = 4;

// enum vdb_status
pub const VDB_OK: i32 = 0;
pub const VDB_DUPLICATE_IGNORED: i32 = 1;
pub const VDB_ERR_INVALID_ARG: i32 = -1;
pub const VDB_ERR_DIM_MISMATCH: i32 = -2;
pub const VDB_ERR_NO_DEVICE: i32 = -3;
pub const VDB_ERR_HIP: i32 = -4;
pub const VDB_ERR_IO: i32 = -5;
pub const VDB_ERR_OOM: i32 = -6;
pub const VDB_ERR_UNSUPPORTED: i32 = -7;
pub const VDB_ERR_STATE: i32 = -8;

// enum vdb_search_mode
pub const VDB_SEARCH_AUTO: i32 = 0;
pub const VDB_SEARCH_BRUTE: i32 = 1;
pub const VDB_SEARCH_HNSW: i32 = 2;
pub const VDB_SEARCH_BRUTE_BF16: i32 = 3;
pub const VDB_SEARCH_HNSW_INT8: i32 = 4;
pub const VDB_SEARCH_BRUTE_SQ8: i32 = 5;
pub const VDB_SEARCH_BRUTE_BINARY: i32 = 6;

// enum vdb_storage_mode
pub const VDB_STORAGE_FULL: i32 = 0;
pub const VDB_STORAGE_SQ8: i32 = 1;
pub const VDB_STORAGE_BINARY: i32 = 2;

// enum vdb_distance_kind
pub const VDB_KIND_ENGINE: i32 = 0;
pub const VDB_KIND_RAW: i32 = 1;
pub const VDB_KIND_SQUARED: i32 = 2;

// enum vdb_shard_mode
pub const VDB_SHARD_REPLICA: i32 = 0;
pub const VDB_SHARD_RANGE: i32 = 1;

// enum vdb_option
pub const VDB_OPT_MAX_QUERY_TILE: i32 = 0;
pub const VDB_OPT_SWEEP_ENGINE: i32 = 1;
pub const VDB_OPT_SELECTOR_LEVEL: i32 = 2;
pub const VDB_OPT_INT8_OVERSAMPLING: i32 = 3;
pub const VDB_OPT_KERNEL_TIMING: i32 = 4;
pub const VDB_OPT_COMBINE_MAX_BATCH: i32 = 5;
pub const VDB_OPT_COMBINE_WINDOW_US: i32 = 6;
pub const VDB_OPT_COMBINE_INFLIGHT: i32 = 7;
#[allow(non_upper_case_globals)]
pub const VDB_OPT_COUNT_: i32 = 8;

// enum vdb_kernel_bit (vdb_hip_index_last_kernels)
pub const VDB_KERNEL_SWEEP_VALU: i32 = 1;
pub const VDB_KERNEL_SWEEP_MFMA_F32: i32 = 2;
pub const VDB_KERNEL_GEMM_F32: i32 = 4;
pub const VDB_KERNEL_SWEEP_MFMA_BF16: i32 = 8;
pub const VDB_KERNEL_GEMM_BF16: i32 = 16;
pub const VDB_KERNEL_GEMM_BF16_GLDS: i32 = 32;
pub const VDB_KERNEL_SELECT_BF16: i32 = 64;
pub const VDB_KERNEL_SELECT_SPLIT: i32 = 128;
pub const VDB_KERNEL_BITS: i32 = 256;
pub const VDB_KERNEL_SQ8: i32 = 512;
pub const VDB_KERNEL_HNSW: i32 = 1024;
pub const VDB_KERNEL_HNSW_INT8: i32 = 2048;
pub const VDB_KERNEL_BITS_GEMM: i32 = 4096;

pub const VDB_COMM_ID_BYTES: usize = 128;

extern "C" {
    pub fn vdb_hip_device_count(n: *mut i32) -> i32;
    pub fn vdb_hip_device_name(device: i32, buf: *mut c_char, cap: usize) -> i32;
    pub fn vdb_hip_index_create(dim: u32, metric: i32, M: u32, ef_construction: u32, max_elements: u64, devices: *const i32, n_devices: i32, shard_mode: i32, out: *mut *mut VdbHipIndex) -> i32;
    pub fn vdb_hip_comm_unique_id(id: *mut u8) -> i32;
    pub fn vdb_hip_index_join_group(idx: *mut VdbHipIndex, id: *const u8, rank: i32, world: i32) -> i32;
    pub fn vdb_hip_index_shard_info(idx: *mut VdbHipIndex, n_shards: *mut i32, shard_mode: *mut i32, rank: *mut i32, world: *mut i32, transport: *mut i32) -> i32;
    pub fn vdb_hip_index_destroy(idx: *mut VdbHipIndex);
    pub fn vdb_hip_index_insert(idx: *mut VdbHipIndex, id: u64, vec: *const f32, vec_len: u32) -> i32;
    pub fn vdb_hip_index_insert_batch(idx: *mut VdbHipIndex, ids: *const u64, vecs_rowmajor: *const f32, n: u64, inserted: *mut u64) -> i32;
    pub fn vdb_hip_index_insert_batch_parallel(idx: *mut VdbHipIndex, ids: *const u64, vecs_rowmajor: *const f32, n: u64, max_batch: u32, inserted: *mut u64) -> i32;
    pub fn vdb_hip_index_train_quantizer(idx: *mut VdbHipIndex, sample_rows: u32) -> i32;
    pub fn vdb_hip_index_quantizer_trained(idx: *const VdbHipIndex, trained: *mut i32) -> i32;
    pub fn vdb_hip_index_search_with_config(idx: *mut VdbHipIndex, queries_rowmajor: *const f32, nq: u32, k: u32, ef_search: u32, oversampling_ratio: u32, use_int8_traversal: i32, min_index_size: u64, out_ids: *mut u64, out_scores: *mut f32, out_n: *mut u32) -> i32;
    pub fn vdb_hip_set_int8_oversampling(ratio: u32) -> i32;
    pub fn vdb_hip_index_set_storage_mode(idx: *mut VdbHipIndex, mode: i32) -> i32;
    pub fn vdb_hip_index_get_quantized(idx: *mut VdbHipIndex, id: u64, out: *mut u8, cap: usize, len: *mut usize) -> i32;
    pub fn vdb_hip_index_enable_bf16(idx: *mut VdbHipIndex) -> i32;
    pub fn vdb_hip_index_build_graph(idx: *mut VdbHipIndex, max_batch: u32) -> i32;
    pub fn vdb_hip_index_upload(idx: *mut VdbHipIndex, ids: *const u64, vecs_rowmajor: *const f32, n: u64, inserted: *mut u64) -> i32;
    pub fn vdb_hip_index_upload_dev(idx: *mut VdbHipIndex, id_base: u64, d_vecs_rowmajor: *const f32, n: u64, stream: *mut c_void) -> i32;
    pub fn vdb_hip_index_remove(idx: *mut VdbHipIndex, id: u64, removed: *mut i32) -> i32;
    pub fn vdb_hip_index_len(idx: *const VdbHipIndex, n: *mut u64) -> i32;
    pub fn vdb_hip_index_dimension(idx: *const VdbHipIndex, dim: *mut u32) -> i32;
    pub fn vdb_hip_index_metric(idx: *const VdbHipIndex, metric: *mut i32) -> i32;
    pub fn vdb_hip_index_tombstone_count(idx: *const VdbHipIndex, n: *mut u64) -> i32;
    pub fn vdb_hip_index_vacuum(idx: *mut VdbHipIndex, count: *mut u64) -> i32;
    pub fn vdb_hip_index_node_count(idx: *const VdbHipIndex, n: *mut u64) -> i32;
    pub fn vdb_hip_index_search(idx: *mut VdbHipIndex, query: *const f32, query_len: u32, k: u32, ef: u32, mode: i32, out_ids: *mut u64, out_scores: *mut f32, out_n: *mut u32) -> i32;
    pub fn vdb_hip_index_search_batch(idx: *mut VdbHipIndex, queries_rowmajor: *const f32, nq: u32, k: u32, ef: u32, mode: i32, out_ids: *mut u64, out_scores: *mut f32, out_n: *mut u32) -> i32;
    pub fn vdb_hip_index_search_rerank(idx: *mut VdbHipIndex, queries_rowmajor: *const f32, nq: u32, k: u32, rerank_k: u32, ef: u32, out_ids: *mut u64, out_scores: *mut f32, out_n: *mut u32) -> i32;
    pub fn vdb_hip_index_search_multi_entry(idx: *mut VdbHipIndex, queries_rowmajor: *const f32, nq: u32, k: u32, ef: u32, num_probes: u32, out_ids: *mut u64, out_scores: *mut f32, out_n: *mut u32) -> i32;
    pub fn vdb_hip_index_search_batch_dev(idx: *mut VdbHipIndex, d_queries: *const f32, nq: u32, k: u32, ef: u32, mode: i32, d_out_ids: *mut u64, d_out_scores: *mut f32, d_out_n: *mut u32, stream: *mut c_void) -> i32;
    pub fn vdb_hip_batch_distance(device: i32, metric: i32, kind: i32, query: *const f32, vecs_rowmajor: *const f32, n: u64, dim: u32, out: *mut f32) -> i32;
    pub fn vdb_hip_batch_distance_dev(metric: i32, kind: i32, d_query: *const f32, d_vecs_rowmajor: *const f32, n: u64, dim: u32, d_out: *mut f32, stream: *mut c_void) -> i32;
    pub fn vdb_hip_batch_norm(device: i32, vecs_rowmajor: *const f32, n: u64, dim: u32, out: *mut f32) -> i32;
    pub fn vdb_hip_normalize_rows(device: i32, vecs_rowmajor: *mut f32, n: u64, dim: u32) -> i32;
    pub fn vdb_hip_batch_dot_product(device: i32, queries_rowmajor: *const f32, nq: u32, vecs_rowmajor: *const f32, n: u64, dim: u32, out: *mut f32) -> i32;
    pub fn vdb_hip_batch_hamming_binary(device: i32, query_words: *const u64, rows_words: *const u64, n: u64, words: u32, out: *mut u32) -> i32;
    pub fn vdb_hip_batch_jaccard_binary(device: i32, query_words: *const u64, rows_words: *const u64, n: u64, words: u32, out: *mut f32) -> i32;
    pub fn vdb_hip_index_load_reference_files(idx: *mut VdbHipIndex, dir: *const c_char, basename: *const c_char) -> i32;
    pub fn vdb_hip_index_save_reference_files(idx: *mut VdbHipIndex, dir: *const c_char, basename: *const c_char) -> i32;
    pub fn vdb_hip_index_save_dir(idx: *mut VdbHipIndex, dir: *const c_char) -> i32;
    pub fn vdb_hip_index_load_dir(dir: *const c_char, device: i32, out: *mut *mut VdbHipIndex) -> i32;
    pub fn vdb_hip_index_upload_vector_store(idx: *mut VdbHipIndex, dir: *const c_char, inserted: *mut u64) -> i32;
    pub fn vdb_hip_index_set_option(idx: *mut VdbHipIndex, option: i32, value: i64) -> i32;
    pub fn vdb_hip_index_get_option(idx: *mut VdbHipIndex, option: i32, value: *mut i64) -> i32;
    pub fn vdb_hip_index_get_neighbors(idx: *mut VdbHipIndex, layer: u32, node: u64, out: *mut u32, cap: u32, n: *mut u32) -> i32;
    pub fn vdb_hip_index_build_stats(idx: *mut VdbHipIndex, rows_evaluated: *mut u64, distance_phases: *mut u64, nodes: *mut u64, select_rows: *mut u64) -> i32;
    pub fn vdb_hip_index_graph_info(idx: *mut VdbHipIndex, num_layers: *mut u32, max_layer: *mut u32, entry_point: *mut i64) -> i32;
    pub fn vdb_hip_index_last_search_stats(idx: *mut VdbHipIndex, n_dist: *mut u64, n_expand: *mut u64) -> i32;
    pub fn vdb_hip_index_last_prefetch_hits(idx: *mut VdbHipIndex, hits: *mut u64) -> i32;
    pub fn vdb_hip_set_kernel_timing(on: i32) -> i32;
    pub fn vdb_hip_set_max_query_tile(b: u32) -> i32;
    pub fn vdb_hip_set_sweep_engine(engine: i32) -> i32;
    pub fn vdb_hip_set_split_selector(level: i32) -> i32;
    pub fn vdb_hip_index_last_split_stats(idx: *mut VdbHipIndex, queries: *mut u32, unproven: *mut u32) -> i32;
    pub fn vdb_hip_index_last_select_level(idx: *mut VdbHipIndex, level: *mut i32) -> i32;
    pub fn vdb_hip_index_last_kernels(idx: *mut VdbHipIndex, mask: *mut u32) -> i32;
    pub fn vdb_hip_index_combine_stats(idx: *mut VdbHipIndex, launches: *mut u64, calls: *mut u64, queries: *mut u64, max_batch: *mut u64) -> i32;
    pub fn vdb_hip_index_sweep_arith_mode(idx: *mut VdbHipIndex, k: u32, mode: *mut i32) -> i32;
    pub fn vdb_hip_index_last_kernel_ms(idx: *mut VdbHipIndex, ms: *mut f32, launches: *mut u32) -> i32;
    pub fn vdb_hip_index_last_selection_ms(idx: *mut VdbHipIndex, total_ms: *mut f32, launches: *mut u32) -> i32;
    pub fn vdb_hip_last_error() -> *const c_char;
    pub fn vdb_hip_version() -> *const c_char;
}
