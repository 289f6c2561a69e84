//! `velesdb-hip` — velesdb-core's HNSW / exact-search hot path on an AMD MI355X through `libvelesdb_hip.so`.
//!
//! What the crate provides, each over the C ABI of `include/velesdb_hip.h` (bound in [`sys`]):
//!
//! | here | replaces in velesdb-core |
//! |---|---|
//! | [`HipHnswIndex`] + `impl VectorIndex` | `HnswIndex` (`index/hnsw/index/*.rs`), `impl VectorIndex for HnswIndex` (`trait_impl.rs:8-71`) |
//! | [`HipDistance`] + `impl DistanceEngine` | `SimdDistance` / `NativeSimdDistance` (`index/hnsw/native/distance.rs:14-28,60-160`) |
//! | [`HipAccelerator`] | `GpuAccelerator` (`gpu/gpu_backend.rs:33,136,157,355,397`) |
//! | [`HipDualPrecisionHnsw`] + [`DualPrecisionConfig`] | `DualPrecisionHnsw` (`index/hnsw/native/dual_precision.rs:32-57,88-321`) |
//! | [`simd`] free functions | `simd::norm`, `normalize_inplace`, `simd_explicit::batch_dot_product`, `hamming_distance_binary`, `jaccard_similarity_binary` |
//!
//! Error behaviour follows the reference: dimension mismatches panic with the reference's messages
//! (`index/hnsw/index/search.rs:16-27`, `trait_impl.rs:12-18`), a duplicate id is silently skipped (`trait_impl.rs:23-25`),
//! persistence returns `io::Result`, `vacuum` returns `Result<usize, VacuumError>`; everything else the library reports
//! (no device, out of memory, a HIP error) panics with the library's message, as a failing allocation or a poisoned lock
//! does in the CPU index.  There is no CPU fallback in this crate: constructors return `None` without a HIP device and the
//! caller keeps `velesdb_core::HnswIndex`.

pub mod sys;

use std::ffi::{CStr, CString};
use std::io;
use std::os::raw::c_void;
use std::path::Path;
use std::ptr;

use velesdb_core::index::hnsw::native::DistanceEngine;
use velesdb_core::{DistanceMetric, HnswParams, SearchQuality, VectorIndex};

/// The thread-local message of the last failing call (`vdb_hip_last_error`, never NULL).
#[must_use]
pub fn last_error() -> String {
    // SAFETY: the library returns a pointer to a NUL-terminated thread-local buffer that stays valid until the next call
    // on this thread; it is copied before anything else is called.
    unsafe { CStr::from_ptr(sys::vdb_hip_last_error()) }.to_string_lossy().into_owned()
}

/// Library version string (`vdb_hip_version`).
#[must_use]
pub fn version() -> String {
    // SAFETY: static NUL-terminated string.
    unsafe { CStr::from_ptr(sys::vdb_hip_version()) }.to_string_lossy().into_owned()
}

/// Number of HIP devices the library sees (0 without a GPU or without the driver).
#[must_use]
pub fn device_count() -> usize {
    let mut n: i32 = 0;
    // SAFETY: `n` is a valid out pointer.
    let rc = unsafe { sys::vdb_hip_device_count(&mut n) };
    if rc < 0 || n < 0 {
        0
    } else {
        n as usize
    }
}

/// Name of a device ("AMD Instinct MI355X").
#[must_use]
pub fn device_name(device: usize) -> Option<String> {
    let mut buf = [0 as std::os::raw::c_char; 256];
    // SAFETY: `buf` holds `cap` bytes; the library NUL-terminates within `cap`.
    let rc = unsafe { sys::vdb_hip_device_name(device as i32, buf.as_mut_ptr(), buf.len()) };
    if rc < 0 {
        return None;
    }
    // SAFETY: NUL-terminated by the callee.
    Some(unsafe { CStr::from_ptr(buf.as_ptr()) }.to_string_lossy().into_owned())
}

fn metric_code(m: DistanceMetric) -> i32 {
    // the reference's on-disk order (index/hnsw/index/constructors.rs:204-210)
    match m {
        DistanceMetric::Cosine => sys::VDB_COSINE,
        DistanceMetric::Euclidean => sys::VDB_EUCLIDEAN,
        DistanceMetric::DotProduct => sys::VDB_DOT,
        DistanceMetric::Hamming => sys::VDB_HAMMING,
        DistanceMetric::Jaccard => sys::VDB_JACCARD,
    }
}

fn metric_from_code(c: i32) -> DistanceMetric {
    match c {
        sys::VDB_EUCLIDEAN => DistanceMetric::Euclidean,
        sys::VDB_DOT => DistanceMetric::DotProduct,
        sys::VDB_HAMMING => DistanceMetric::Hamming,
        sys::VDB_JACCARD => DistanceMetric::Jaccard,
        _ => DistanceMetric::Cosine,
    }
}

/// Panics with the library's message on an error status; returns the (non-negative) status otherwise.
fn check(rc: i32) -> i32 {
    assert!(rc >= 0, "velesdb-hip: {} (status {})", last_error(), rc);
    rc
}

fn io_check(rc: i32) -> io::Result<()> {
    if rc >= 0 {
        Ok(())
    } else {
        let kind = if rc == sys::VDB_ERR_IO { io::ErrorKind::InvalidData } else { io::ErrorKind::Other };
        Err(io::Error::new(kind, last_error()))
    }
}

fn c_path<P: AsRef<Path>>(p: P) -> io::Result<CString> {
    CString::new(p.as_ref().to_string_lossy().as_bytes())
        .map_err(|_| io::Error::new(io::ErrorKind::InvalidInput, "path contains a NUL byte"))
}

/// How a multi-GPU handle spreads its rows (`enum vdb_shard_mode`).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum ShardMode {
    /// Every device holds every row and the graph; a query batch is split between the devices.
    Replica,
    /// Contiguous row ranges per device; exact searches merge per-shard top-k after one RCCL all-gather.
    Range,
}

/// `HnswIndex::vacuum`'s error type (`index/hnsw/index/vacuum.rs:11-14`).  The GPU index always stores its vectors, so
/// the only variant the reference has can not occur here; it exists so that call sites compile unchanged.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum VacuumError {
    /// Vector storage is disabled, cannot rebuild index
    VectorStorageDisabled,
}

impl std::fmt::Display for VacuumError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "Cannot vacuum index: vector storage is disabled")
    }
}

impl std::error::Error for VacuumError {}

/// An HNSW index resident in the HBM of one or more MI355X GPUs.  Same inherent methods as `velesdb_core::HnswIndex`.
pub struct HipHnswIndex {
    h: *mut sys::VdbHipIndex,
    dimension: usize,
    metric: DistanceMetric,
}

// SAFETY: the handle is internally synchronised (one mutex per index, thread-local error strings); the C ABI documents
// "a handle may be used from any thread" (include/velesdb_hip.h conventions) and tests/test_gpu_hardening.py drives
// concurrent searches against a concurrent inserter.
unsafe impl Send for HipHnswIndex {}
// SAFETY: as above.
unsafe impl Sync for HipHnswIndex {}

impl HipHnswIndex {
    /// `HnswIndex::new` (`constructors.rs:29-32`): `HnswParams::auto(dimension)`.
    #[must_use]
    pub fn new(dimension: usize, metric: DistanceMetric) -> Option<Self> {
        Self::with_params(dimension, metric, HnswParams::auto(dimension))
    }

    /// `HnswIndex::with_params` (`constructors.rs:116-160`) on device 0.  `None` when no HIP device is present, like
    /// `GpuAccelerator::new()` (`gpu/gpu_backend.rs:33`).
    #[must_use]
    pub fn with_params(dimension: usize, metric: DistanceMetric, params: HnswParams) -> Option<Self> {
        Self::with_params_on(dimension, metric, params, &[0], ShardMode::Replica)
    }

    /// One index over several GPUs of a node (`devices` = HIP ordinals).  Still ONE `VectorIndex`.
    #[must_use]
    pub fn with_params_on(
        dimension: usize,
        metric: DistanceMetric,
        params: HnswParams,
        devices: &[i32],
        shard_mode: ShardMode,
    ) -> Option<Self> {
        if device_count() == 0 || devices.is_empty() {
            return None;
        }
        let mode = match shard_mode {
            ShardMode::Replica => sys::VDB_SHARD_REPLICA,
            ShardMode::Range => sys::VDB_SHARD_RANGE,
        };
        let mut h: *mut sys::VdbHipIndex = ptr::null_mut();
        // SAFETY: `devices` outlives the call; `h` is a valid out pointer.
        let rc = unsafe {
            sys::vdb_hip_index_create(
                dimension as u32,
                metric_code(metric),
                params.max_connections as u32,
                params.ef_construction as u32,
                params.max_elements as u64,
                devices.as_ptr(),
                devices.len() as i32,
                mode,
                &mut h,
            )
        };
        if rc != sys::VDB_OK || h.is_null() {
            return None;
        }
        let me = Self { h, dimension, metric };
        // HnswParams::storage_mode (params.rs:24-27): SQ8 / Binary collections quantise every stored vector
        let sm = match params.storage_mode {
            velesdb_core::quantization::StorageMode::Full => sys::VDB_STORAGE_FULL,
            velesdb_core::quantization::StorageMode::SQ8 => sys::VDB_STORAGE_SQ8,
            velesdb_core::quantization::StorageMode::Binary => sys::VDB_STORAGE_BINARY,
        };
        if sm != sys::VDB_STORAGE_FULL {
            // SAFETY: live handle.
            check(unsafe { sys::vdb_hip_index_set_storage_mode(me.h, sm) });
        }
        Some(me)
    }

    /// One process per GPU: joins the RCCL group of `world` ranks (`id` from [`comm_unique_id`] on rank 0, distributed
    /// out of band).  Afterwards the exact search modes return the global top-k on every rank.
    pub fn join_group(&self, id: &[u8; sys::VDB_COMM_ID_BYTES], rank: usize, world: usize) {
        // SAFETY: live handle, `id` has VDB_COMM_ID_BYTES bytes.
        check(unsafe { sys::vdb_hip_index_join_group(self.h, id.as_ptr(), rank as i32, world as i32) });
    }

    fn validate_dimension(&self, data: &[f32], data_type: &str) {
        // index/hnsw/index/search.rs:16-27
        assert_eq!(
            data.len(),
            self.dimension,
            "{data_type} dimension mismatch: expected {}, got {}",
            self.dimension,
            data.len()
        );
    }

    fn search_one(&self, query: &[f32], k: usize, ef: usize, mode: i32) -> Vec<(u64, f32)> {
        if k == 0 {
            return Vec::new();
        }
        let mut ids = vec![0u64; k];
        let mut scores = vec![0f32; k];
        let mut n: u32 = 0;
        // SAFETY: `ids` / `scores` hold k entries as the ABI requires; the query has `dimension` floats (validated by callers).
        check(unsafe {
            sys::vdb_hip_index_search(
                self.h,
                query.as_ptr(),
                query.len() as u32,
                k as u32,
                ef as u32,
                mode,
                ids.as_mut_ptr(),
                scores.as_mut_ptr(),
                &mut n,
            )
        });
        ids.truncate(n as usize);
        scores.truncate(n as usize);
        ids.into_iter().zip(scores).collect()
    }

    fn search_many(&self, queries: &[&[f32]], k: usize, ef: usize, mode: i32) -> Vec<Vec<(u64, f32)>> {
        if queries.is_empty() {
            return Vec::new();
        }
        if k == 0 {
            return vec![Vec::new(); queries.len()];
        }
        let nq = queries.len();
        let mut flat = Vec::with_capacity(nq * self.dimension);
        for q in queries {
            flat.extend_from_slice(q);
        }
        let mut ids = vec![0u64; nq * k];
        let mut scores = vec![0f32; nq * k];
        let mut counts = vec![0u32; nq];
        // SAFETY: buffer sizes are nq*dim / nq*k / nq as the ABI requires.
        check(unsafe {
            sys::vdb_hip_index_search_batch(
                self.h,
                flat.as_ptr(),
                nq as u32,
                k as u32,
                ef as u32,
                mode,
                ids.as_mut_ptr(),
                scores.as_mut_ptr(),
                counts.as_mut_ptr(),
            )
        });
        (0..nq)
            .map(|i| {
                let c = counts[i] as usize;
                (0..c).map(|j| (ids[i * k + j], scores[i * k + j])).collect()
            })
            .collect()
    }

    /// `HnswIndex::search_with_quality` (`search.rs:59-94`): `Perfect` = exact scan; otherwise mode AUTO (exact scan when
    /// `len() <= 100`, else the graph with `quality.ef_search(k)`; scores through `transform_score`).
    #[must_use]
    pub fn search_with_quality(&self, query: &[f32], k: usize, quality: SearchQuality) -> Vec<(u64, f32)> {
        self.validate_dimension(query, "Query");
        if matches!(quality, SearchQuality::Perfect) {
            return self.search_brute_force(query, k);
        }
        self.search_one(query, k, quality.ef_search(k), sys::VDB_SEARCH_AUTO)
    }

    /// `HnswIndex::search_brute_force` (`search.rs:176-219`): exact scan, raw scores, `metric.sort_results` order.
    #[must_use]
    pub fn search_brute_force(&self, query: &[f32], k: usize) -> Vec<(u64, f32)> {
        self.validate_dimension(query, "Query");
        self.search_one(query, k, 0, sys::VDB_SEARCH_BRUTE)
    }

    /// `search_brute_force_buffered` (`search.rs:366-369`), `brute_force_search_parallel` (`batch.rs:223-240`) and
    /// `search_brute_force_gpu` (`search.rs:229-290`) are the same exact scan here.
    #[must_use]
    pub fn search_brute_force_buffered(&self, query: &[f32], k: usize) -> Vec<(u64, f32)> {
        self.search_brute_force(query, k)
    }

    /// See [`Self::search_brute_force_buffered`].
    #[must_use]
    pub fn brute_force_search_parallel(&self, query: &[f32], k: usize) -> Vec<(u64, f32)> {
        self.search_brute_force(query, k)
    }

    /// See [`Self::search_brute_force_buffered`]; always `Some` (the reference returns `None` without its `gpu` feature).
    #[must_use]
    pub fn search_brute_force_gpu(&self, query: &[f32], k: usize) -> Option<Vec<(u64, f32)>> {
        Some(self.search_brute_force(query, k))
    }

    /// The exact scan for a whole batch of queries in one call: the library's headline path (one corpus pass serves up
    /// to 1 024 queries on the matrix cores).  Equal, query by query, to [`Self::search_brute_force`].
    #[must_use]
    pub fn search_batch_brute_force(&self, queries: &[&[f32]], k: usize) -> Vec<Vec<(u64, f32)>> {
        for q in queries {
            self.validate_dimension(q, "Query");
        }
        self.search_many(queries, k, 0, sys::VDB_SEARCH_BRUTE)
    }

    /// `HnswIndex::search_batch_parallel` (`batch.rs:159-197`): always the graph, one launch for all queries.
    #[must_use]
    pub fn search_batch_parallel(&self, queries: &[&[f32]], k: usize, quality: SearchQuality) -> Vec<Vec<(u64, f32)>> {
        for (i, query) in queries.iter().enumerate() {
            assert_eq!(
                query.len(),
                self.dimension,
                "Query {} dimension mismatch: expected {}, got {}",
                i,
                self.dimension,
                query.len()
            );
        }
        self.search_many(queries, k, quality.ef_search(k), sys::VDB_SEARCH_HNSW)
    }

    /// `HnswIndex::search_with_rerank` (`search.rs:118-160`): `rerank_k` candidates at `SearchQuality::Accurate`, re-scored
    /// with the raw distance, cut to `k`.
    #[must_use]
    pub fn search_with_rerank(&self, query: &[f32], k: usize, rerank_k: usize) -> Vec<(u64, f32)> {
        self.rerank(query, k, rerank_k, 0)
    }

    /// `HnswIndex::search_with_rerank_quality` (`search.rs:297-350`).
    #[must_use]
    pub fn search_with_rerank_quality(
        &self,
        query: &[f32],
        k: usize,
        rerank_k: usize,
        initial_quality: SearchQuality,
    ) -> Vec<(u64, f32)> {
        if matches!(initial_quality, SearchQuality::Perfect) {
            // the candidates of the Perfect profile are the exact scan's; its raw scores are what the re-ranking computes
            return self.search_brute_force(query, k);
        }
        self.rerank(query, k, rerank_k, initial_quality.ef_search(rerank_k))
    }

    /// `NativeHnsw::search_multi_entry` (`native/graph.rs:288-348`): the descent's result plus up to `num_probes.min(4) - 1`
    /// nodes drawn from the graph's own xorshift stream as entry points of one layer-0 search (advances that stream, as the
    /// reference does).
    #[must_use]
    pub fn search_multi_entry(&self, query: &[f32], k: usize, ef_search: usize, num_probes: usize) -> Vec<(u64, f32)> {
        self.validate_dimension(query, "Query");
        if k == 0 {
            return Vec::new();
        }
        let mut ids = vec![0u64; k];
        let mut scores = vec![0f32; k];
        let mut n: u32 = 0;
        // SAFETY: one query, k outputs.
        check(unsafe {
            sys::vdb_hip_index_search_multi_entry(
                self.h,
                query.as_ptr(),
                1,
                k as u32,
                ef_search as u32,
                num_probes as u32,
                ids.as_mut_ptr(),
                scores.as_mut_ptr(),
                &mut n,
            )
        });
        ids.truncate(n as usize);
        ids.into_iter().zip(scores).collect()
    }

    fn rerank(&self, query: &[f32], k: usize, rerank_k: usize, ef: usize) -> Vec<(u64, f32)> {
        self.validate_dimension(query, "Query");
        if k == 0 {
            return Vec::new();
        }
        let mut ids = vec![0u64; k];
        let mut scores = vec![0f32; k];
        let mut n: u32 = 0;
        // SAFETY: one query, k outputs.
        check(unsafe {
            sys::vdb_hip_index_search_rerank(
                self.h,
                query.as_ptr(),
                1,
                k as u32,
                rerank_k.max(k) as u32,
                ef as u32,
                ids.as_mut_ptr(),
                scores.as_mut_ptr(),
                &mut n,
            )
        });
        ids.truncate(n as usize);
        scores.truncate(n as usize);
        ids.into_iter().zip(scores).collect()
    }

    fn flatten<I>(&self, vectors: I) -> (Vec<u64>, Vec<f32>)
    where
        I: IntoIterator<Item = (u64, Vec<f32>)>,
    {
        let mut ids = Vec::new();
        let mut flat = Vec::new();
        for (id, v) in vectors {
            self.validate_dimension(&v, "Vector"); // batch.rs:26-34
            ids.push(id);
            flat.extend_from_slice(&v);
        }
        (ids, flat)
    }

    /// `HnswIndex::insert_batch_parallel` (`batch.rs:82-108`).  Returns the number of vectors inserted (duplicates are
    /// skipped).  The reference inserts with rayon in a non-deterministic order; the library inserts batch-synchronously
    /// and deterministically (header, `vdb_hip_index_insert_batch_parallel`).
    pub fn insert_batch_parallel<I>(&self, vectors: I) -> usize
    where
        I: IntoIterator<Item = (u64, Vec<f32>)>,
    {
        let (ids, flat) = self.flatten(vectors);
        if ids.is_empty() {
            return 0;
        }
        let mut inserted: u64 = 0;
        // SAFETY: `flat` holds ids.len() * dimension floats.
        check(unsafe {
            sys::vdb_hip_index_insert_batch_parallel(self.h, ids.as_ptr(), flat.as_ptr(), ids.len() as u64, 0, &mut inserted)
        });
        inserted as usize
    }

    /// `HnswIndex::insert_batch_sequential` (`batch.rs:120-149`): one vector after the other, the reference's own
    /// deterministic order.
    pub fn insert_batch_sequential<I>(&self, vectors: I) -> usize
    where
        I: IntoIterator<Item = (u64, Vec<f32>)>,
    {
        let (ids, flat) = self.flatten(vectors);
        if ids.is_empty() {
            return 0;
        }
        let mut inserted: u64 = 0;
        // SAFETY: as above.
        check(unsafe { sys::vdb_hip_index_insert_batch(self.h, ids.as_ptr(), flat.as_ptr(), ids.len() as u64, &mut inserted) });
        inserted as usize
    }

    /// Bulk load without graph construction: rows are searchable by the exact modes at once; [`Self::build_graph`] links
    /// them later.  (No counterpart in the reference: its `Collection::open` rebuilds the index by inserting.)
    pub fn upload(&self, ids: &[u64], vectors_rowmajor: &[f32]) -> usize {
        assert_eq!(
            vectors_rowmajor.len(),
            ids.len() * self.dimension,
            "Vector dimension mismatch: expected {} floats, got {}",
            ids.len() * self.dimension,
            vectors_rowmajor.len()
        );
        if ids.is_empty() {
            return 0;
        }
        let mut inserted: u64 = 0;
        // SAFETY: sizes checked above.
        check(unsafe { sys::vdb_hip_index_upload(self.h, ids.as_ptr(), vectors_rowmajor.as_ptr(), ids.len() as u64, &mut inserted) });
        inserted as usize
    }

    /// Links every row that is not in the graph yet.
    pub fn build_graph(&self) {
        // SAFETY: live handle.
        check(unsafe { sys::vdb_hip_index_build_graph(self.h, 0) });
    }

    /// `HnswIndex::set_searching_mode` (`search.rs:379-384`): a no-op kept for API compatibility, there as here.
    pub fn set_searching_mode(&self) {}

    /// `HnswIndex::has_vector_storage` (`constructors.rs:320-322`): the GPU index always keeps its vectors.
    #[must_use]
    pub fn has_vector_storage(&self) -> bool {
        true
    }

    /// `HnswIndex::tombstone_count` (`vacuum.rs:45-52`).
    #[must_use]
    pub fn tombstone_count(&self) -> usize {
        let mut n: u64 = 0;
        // SAFETY: live handle, valid out pointer.
        check(unsafe { sys::vdb_hip_index_tombstone_count(self.h, &mut n) });
        n as usize
    }

    /// `HnswIndex::tombstone_ratio` (`vacuum.rs:60-68`): tombstones / graph nodes.
    #[must_use]
    pub fn tombstone_ratio(&self) -> f64 {
        let mut nodes: u64 = 0;
        // SAFETY: live handle, valid out pointer.
        check(unsafe { sys::vdb_hip_index_node_count(self.h, &mut nodes) });
        if nodes == 0 {
            return 0.0;
        }
        self.tombstone_count() as f64 / nodes as f64
    }

    /// `HnswIndex::needs_vacuum` (`vacuum.rs:74-76`): ratio above 20 %.
    #[must_use]
    pub fn needs_vacuum(&self) -> bool {
        self.tombstone_ratio() > 0.2
    }

    /// `HnswIndex::vacuum` (`vacuum.rs:110-184`): rebuilds the graph over the active vectors.
    pub fn vacuum(&self) -> Result<usize, VacuumError> {
        let mut count: u64 = 0;
        // SAFETY: live handle, valid out pointer.
        check(unsafe { sys::vdb_hip_index_vacuum(self.h, &mut count) });
        Ok(count as usize)
    }

    /// `HnswIndex::save` (`constructors.rs:255-287`): `native_hnsw.{vectors,graph}`, `native_mappings.bin`, `native_meta.bin`.
    pub fn save<P: AsRef<Path>>(&self, path: P) -> Result<(), io::Error> {
        std::fs::create_dir_all(path.as_ref())?;
        let dir = c_path(path)?;
        // SAFETY: NUL-terminated path, live handle.
        io_check(unsafe { sys::vdb_hip_index_save_dir(self.h, dir.as_ptr()) })
    }

    /// `HnswIndex::load` (`constructors.rs:190-253`).  Dimension and metric are read from `native_meta.bin`; the
    /// arguments are checked against the files like the reference trusts them.
    pub fn load<P: AsRef<Path>>(path: P, dimension: usize, metric: DistanceMetric) -> io::Result<Self> {
        if device_count() == 0 {
            return Err(io::Error::new(io::ErrorKind::NotFound, "no HIP device"));
        }
        let dir = c_path(path)?;
        let mut h: *mut sys::VdbHipIndex = ptr::null_mut();
        // SAFETY: NUL-terminated path, valid out pointer.
        io_check(unsafe { sys::vdb_hip_index_load_dir(dir.as_ptr(), 0, &mut h) })?;
        let mut dim: u32 = 0;
        let mut mc: i32 = 0;
        // SAFETY: `h` was just created.
        unsafe {
            sys::vdb_hip_index_dimension(h, &mut dim);
            sys::vdb_hip_index_metric(h, &mut mc);
        }
        let me = Self { h, dimension: dim as usize, metric: metric_from_code(mc) };
        if me.dimension != dimension || me.metric != metric {
            return Err(io::Error::new(
                io::ErrorKind::InvalidData,
                format!(
                    "index files hold dimension {} / {:?}, caller expects {} / {:?}",
                    me.dimension, me.metric, dimension, metric
                ),
            ));
        }
        Ok(me)
    }

    /// Imports a flushed `MmapStorage` directory (`core/storage/mmap.rs`) with the store's ids; no graph is built.
    pub fn upload_vector_store<P: AsRef<Path>>(&self, path: P) -> io::Result<usize> {
        let dir = c_path(path)?;
        let mut inserted: u64 = 0;
        // SAFETY: NUL-terminated path, live handle, valid out pointer.
        io_check(unsafe { sys::vdb_hip_index_upload_vector_store(self.h, dir.as_ptr(), &mut inserted) })?;
        Ok(inserted as usize)
    }

    /// Device-resident search: `d_queries` / outputs are HIP device pointers, the call enqueues on `stream` and returns.
    ///
    /// # Safety
    /// The pointers must be device allocations of `nq * dimension` f32, `nq * k` u64, `nq * k` f32 and `nq` u32 that
    /// stay alive until `stream` has run the work; `stream` must be a live `hipStream_t` of the index's device.
    #[allow(clippy::too_many_arguments)]
    pub unsafe fn search_batch_dev(
        &self,
        d_queries: *const f32,
        nq: usize,
        k: usize,
        ef: usize,
        mode: i32,
        d_out_ids: *mut u64,
        d_out_scores: *mut f32,
        d_out_n: *mut u32,
        stream: *mut c_void,
    ) {
        check(sys::vdb_hip_index_search_batch_dev(
            self.h,
            d_queries,
            nq as u32,
            k as u32,
            ef as u32,
            mode,
            d_out_ids,
            d_out_scores,
            d_out_n,
            stream,
        ));
    }

    /// Per-handle tuning option (`sys::VDB_OPT_*`); a negative value returns the handle to the process-wide default.
    /// Results never depend on an option.
    pub fn set_option(&self, option: i32, value: i64) {
        // SAFETY: live handle.
        check(unsafe { sys::vdb_hip_index_set_option(self.h, option, value) });
    }

    /// The effective value of a tuning option.
    #[must_use]
    pub fn option(&self, option: i32) -> i64 {
        let mut v: i64 = 0;
        // SAFETY: live handle, valid out pointer.
        check(unsafe { sys::vdb_hip_index_get_option(self.h, option, &mut v) });
        v
    }

    /// Counters of the combining front of `search` / `search_batch_parallel` (many threads that each search one query share
    /// launches): `(launches, calls, queries, largest batch)` since the handle was created.
    #[must_use]
    pub fn combine_stats(&self) -> (u64, u64, u64, u64) {
        let (mut a, mut b, mut c, mut d) = (0u64, 0u64, 0u64, 0u64);
        // SAFETY: live handle, valid out pointers.
        check(unsafe { sys::vdb_hip_index_combine_stats(self.h, &mut a, &mut b, &mut c, &mut d) });
        (a, b, c, d)
    }

    /// Expansions of the last graph search batch whose neighbour list had been requested one pop ahead (the walk kernel's
    /// prediction of the next candidate; a measure, not a count the reference has).
    #[must_use]
    pub fn last_prefetch_hits(&self) -> u64 {
        let mut a = 0u64;
        // SAFETY: live handle, valid out pointer.
        check(unsafe { sys::vdb_hip_index_last_prefetch_hits(self.h, &mut a) });
        a
    }

    /// The raw handle, for the entry points this wrapper does not cover (`sys::*`).
    #[must_use]
    pub fn as_raw(&self) -> *mut sys::VdbHipIndex {
        self.h
    }
}

impl VectorIndex for HipHnswIndex {
    fn insert(&self, id: u64, vector: &[f32]) {
        // trait_impl.rs:12-18
        assert_eq!(
            vector.len(),
            self.dimension,
            "Vector dimension mismatch: expected {}, got {}",
            self.dimension,
            vector.len()
        );
        // SAFETY: live handle, `vector` has `dimension` floats.  Status 1 = duplicate id, silently skipped (trait_impl.rs:23-25).
        check(unsafe { sys::vdb_hip_index_insert(self.h, id, vector.as_ptr(), vector.len() as u32) });
    }

    fn search(&self, query: &[f32], k: usize) -> Vec<(u64, f32)> {
        // trait_impl.rs:38-41
        self.search_with_quality(query, k, SearchQuality::Balanced)
    }

    fn remove(&self, id: u64) -> bool {
        let mut removed: i32 = 0;
        // SAFETY: live handle, valid out pointer.
        check(unsafe { sys::vdb_hip_index_remove(self.h, id, &mut removed) });
        removed != 0
    }

    fn len(&self) -> usize {
        let mut n: u64 = 0;
        // SAFETY: live handle, valid out pointer.
        check(unsafe { sys::vdb_hip_index_len(self.h, &mut n) });
        n as usize
    }

    fn dimension(&self) -> usize {
        self.dimension
    }

    fn metric(&self) -> DistanceMetric {
        self.metric
    }
}

impl HipHnswIndex {
    /// `vdb_hip_index_shard_info`: (devices behind the handle, shard mode as created, rank, world, transport of the exchange:
    /// 0 none, 1 RCCL, 2 device-to-device copies).
    #[must_use]
    pub fn shard_info(&self) -> (usize, ShardMode, usize, usize, i32) {
        let (mut n, mut mode, mut rank, mut world, mut transport) = (0i32, 0i32, 0i32, 0i32, 0i32);
        // SAFETY: live handle, five valid out pointers.
        check(unsafe { sys::vdb_hip_index_shard_info(self.h, &mut n, &mut mode, &mut rank, &mut world, &mut transport) });
        let mode = if mode == sys::VDB_SHARD_RANGE { ShardMode::Range } else { ShardMode::Replica };
        (n as usize, mode, rank as usize, world as usize, transport)
    }

    /// Keeps a bf16 (round-to-nearest-even) copy of the rows for [`Self::search_batch_brute_force_bf16`]
    /// (`half_precision.rs:199-255` semantics: bf16 operands, f32 accumulation).
    pub fn enable_bf16(&self) {
        // SAFETY: live handle.
        check(unsafe { sys::vdb_hip_index_enable_bf16(self.h) });
    }

    fn search_batch_mode(&self, queries: &[&[f32]], k: usize, ef: usize, mode: i32) -> Vec<Vec<(u64, f32)>> {
        if queries.is_empty() || k == 0 {
            return queries.iter().map(|_| Vec::new()).collect();
        }
        let mut flat = Vec::with_capacity(queries.len() * self.dimension);
        for q in queries {
            self.validate_dimension(q, "Query");
            flat.extend_from_slice(q);
        }
        let nq = queries.len();
        let mut ids = vec![0u64; nq * k];
        let mut scores = vec![0f32; nq * k];
        let mut n = vec![0u32; nq];
        // SAFETY: nq row-major queries of `dimension` floats; nq * k outputs; nq counts.
        check(unsafe {
            sys::vdb_hip_index_search_batch(
                self.h,
                flat.as_ptr(),
                nq as u32,
                k as u32,
                ef as u32,
                mode,
                ids.as_mut_ptr(),
                scores.as_mut_ptr(),
                n.as_mut_ptr(),
            )
        });
        (0..nq).map(|i| (0..n[i] as usize).map(|j| (ids[i * k + j], scores[i * k + j])).collect()).collect()
    }

    /// Exact scan with bf16 rows and queries on the matrix cores (BASELINE configs[3]); needs [`Self::enable_bf16`].
    #[must_use]
    pub fn search_batch_brute_force_bf16(&self, queries: &[&[f32]], k: usize) -> Vec<Vec<(u64, f32)>> {
        self.search_batch_mode(queries, k, 0, sys::VDB_SEARCH_BRUTE_BF16)
    }

    /// `StorageMode::SQ8` collections: asymmetric distances against the stored codes (`quantization.rs:410-554`).
    #[must_use]
    pub fn search_batch_sq8(&self, queries: &[&[f32]], k: usize) -> Vec<Vec<(u64, f32)>> {
        self.search_batch_mode(queries, k, 0, sys::VDB_SEARCH_BRUTE_SQ8)
    }

    /// `StorageMode::Binary` collections: Hamming distance between sign bits (`quantization.rs:48-136`).
    #[must_use]
    pub fn search_batch_binary(&self, queries: &[&[f32]], k: usize) -> Vec<Vec<(u64, f32)>> {
        self.search_batch_mode(queries, k, 0, sys::VDB_SEARCH_BRUTE_BINARY)
    }

    /// The stored code of one vector in the reference's serialisation: `QuantizedVector::to_bytes`
    /// (`quantization.rs:289-295`) or `BinaryQuantizedVector::to_bytes` (`:155-169`).  `None` for an unknown id.
    #[must_use]
    pub fn get_quantized(&self, id: u64) -> Option<Vec<u8>> {
        let mut len: usize = 0;
        // SAFETY: a size query: no output buffer, `len` a valid out pointer.
        let rc = unsafe { sys::vdb_hip_index_get_quantized(self.h, id, ptr::null_mut(), 0, &mut len) };
        if rc < 0 || len == 0 {
            return None;
        }
        let mut out = vec![0u8; len];
        // SAFETY: `out` holds `len` bytes.
        check(unsafe { sys::vdb_hip_index_get_quantized(self.h, id, out.as_mut_ptr(), out.len(), &mut len) });
        out.truncate(len);
        Some(out)
    }

    /// Rows that already live in device memory (`d_vecs_rowmajor`: n x dimension f32, ids id_base ..), enqueued on `stream`.
    ///
    /// # Safety
    /// `d_vecs_rowmajor` must be a device pointer to at least `n * dimension` floats that stays valid until the work
    /// enqueued on `stream` (a `hipStream_t`, or null for the default stream) has completed.
    pub unsafe fn upload_dev(&self, id_base: u64, d_vecs_rowmajor: *const f32, n: usize, stream: *mut c_void) {
        check(sys::vdb_hip_index_upload_dev(self.h, id_base, d_vecs_rowmajor, n as u64, stream));
    }

    /// `NativeHnsw::file_dump` (`backend_adapter.rs:184-284`): `<dir>/<basename>.vectors` + `.graph`, reference format v1.
    pub fn file_dump<P: AsRef<Path>>(&self, dir: P, basename: &str) -> io::Result<()> {
        let d = c_path(dir)?;
        let b = CString::new(basename).map_err(|e| io::Error::new(io::ErrorKind::InvalidInput, e))?;
        // SAFETY: two NUL-terminated strings that outlive the call.
        io_check(unsafe { sys::vdb_hip_index_save_reference_files(self.h, d.as_ptr(), b.as_ptr()) })
    }

    /// `NativeHnsw::file_load` (`backend_adapter.rs:286-381`) into this (empty) index.
    pub fn file_load<P: AsRef<Path>>(&self, dir: P, basename: &str) -> io::Result<()> {
        let d = c_path(dir)?;
        let b = CString::new(basename).map_err(|e| io::Error::new(io::ErrorKind::InvalidInput, e))?;
        // SAFETY: as above.
        io_check(unsafe { sys::vdb_hip_index_load_reference_files(self.h, d.as_ptr(), b.as_ptr()) })
    }

    /// (layers, max layer, entry point) of the graph (`NativeHnsw` state, `graph.rs:36-70`).
    #[must_use]
    pub fn graph_info(&self) -> (usize, usize, Option<usize>) {
        let (mut nl, mut ml, mut ep) = (0u32, 0u32, -1i64);
        // SAFETY: live handle, three valid out pointers.
        check(unsafe { sys::vdb_hip_index_graph_info(self.h, &mut nl, &mut ml, &mut ep) });
        (nl as usize, ml as usize, if ep < 0 { None } else { Some(ep as usize) })
    }

    /// `Layer::get_neighbors` (`layer.rs:41-47`) of an internal node.
    #[must_use]
    pub fn neighbors(&self, layer: usize, node: usize) -> Vec<u32> {
        let mut n: u32 = 0;
        let mut out = vec![0u32; 256];
        // SAFETY: `out` holds `cap` ids.
        check(unsafe { sys::vdb_hip_index_get_neighbors(self.h, layer as u32, node as u64, out.as_mut_ptr(), out.len() as u32, &mut n) });
        if n as usize > out.len() {
            out.resize(n as usize, 0);
            // SAFETY: as above, with room for all of them.
            check(unsafe { sys::vdb_hip_index_get_neighbors(self.h, layer as u32, node as u64, out.as_mut_ptr(), out.len() as u32, &mut n) });
        }
        out.truncate(n as usize);
        out
    }
}

impl HipHnswIndex {
    /// The index side of `Collection::search_with_filter` (`collection/search/vector.rs:164-235`): post-filtering over an
    /// over-fetched candidate list — `candidates_k = max(4 k, k + 10)` through `VectorIndex::search`, the ids `keep` rejects
    /// dropped, the first `k` survivors kept, then a stable sort in the metric's order (`partial_cmp`, incomparable = Equal).
    /// Payload storage and the `Filter` type stay the caller's: `keep(id)` stands for `filter.matches(payload(id))`.
    #[must_use]
    pub fn search_filtered<F: Fn(u64) -> bool>(&self, query: &[f32], k: usize, keep: F) -> Vec<(u64, f32)> {
        let candidates_k = k.saturating_mul(4).max(k + 10); // vector.rs:182
        let mut out: Vec<(u64, f32)> = VectorIndex::search(self, query, candidates_k).into_iter().filter(|(id, _)| keep(*id)).take(k).collect();
        let higher_is_better = self.metric.higher_is_better();
        out.sort_by(|a, b| {
            let o = if higher_is_better { b.1.partial_cmp(&a.1) } else { a.1.partial_cmp(&b.1) };
            o.unwrap_or(std::cmp::Ordering::Equal)
        });
        out
    }

    /// Boxed as the trait object `Collection` holds: what a downstream crate's `hip` feature registers with the index factory
    /// hook of velesdb-core (see Cargo.toml: the dependency points from this crate to the core, never back).
    #[must_use]
    pub fn boxed(dimension: usize, metric: DistanceMetric, params: HnswParams) -> Option<Box<dyn VectorIndex>> {
        Self::with_params(dimension, metric, params).map(|ix| Box::new(ix) as Box<dyn VectorIndex>)
    }
}

/// The reference's second `impl VectorIndex` (`index/hnsw/native_index.rs:403-427`) over the same graph.  What differs from
/// [`HipHnswIndex`] is the search entry point: `search_with_quality` ALWAYS walks the graph with `ef = quality.ef_search(k)`
/// (`native_index.rs:230-249`) — no exact-scan shortcut for `Perfect` or for indexes of <= 100 vectors, scores always through
/// `transform_score`; removed ids are dropped after the cut.  Deviation kept from `HnswIndex`: a duplicate id is skipped (the
/// reference re-inserts the vector under the existing internal index, `native_index.rs:256-263`).
pub struct HipNativeHnswIndex(HipHnswIndex);

impl HipNativeHnswIndex {
    #[must_use]
    pub fn new(dimension: usize, metric: DistanceMetric) -> Option<Self> {
        HipHnswIndex::new(dimension, metric).map(Self)
    }

    #[must_use]
    pub fn with_params(dimension: usize, metric: DistanceMetric, params: HnswParams) -> Option<Self> {
        HipHnswIndex::with_params(dimension, metric, params).map(Self)
    }

    /// `native_index.rs:230-249`: the graph walk, whatever the quality and the size of the index.
    #[must_use]
    pub fn search_with_quality(&self, query: &[f32], k: usize, quality: SearchQuality) -> Vec<(u64, f32)> {
        let q = [query];
        self.0.search_batch_parallel(&q, k, quality).pop().unwrap_or_default()
    }

    /// `native_index.rs:275-295`
    pub fn insert_batch(&self, items: &[(u64, Vec<f32>)]) {
        let _ = self.0.insert_batch_parallel(items.iter().map(|(id, v)| (*id, v.clone())));
    }

    #[must_use]
    pub fn inner(&self) -> &HipHnswIndex {
        &self.0
    }
}

impl VectorIndex for HipNativeHnswIndex {
    fn insert(&self, id: u64, vector: &[f32]) {
        VectorIndex::insert(&self.0, id, vector);
    }

    fn search(&self, query: &[f32], k: usize) -> Vec<(u64, f32)> {
        // native_index.rs:225-227
        self.search_with_quality(query, k, SearchQuality::Balanced)
    }

    fn remove(&self, id: u64) -> bool {
        VectorIndex::remove(&self.0, id)
    }

    fn len(&self) -> usize {
        VectorIndex::len(&self.0)
    }

    fn dimension(&self) -> usize {
        VectorIndex::dimension(&self.0)
    }

    fn metric(&self) -> DistanceMetric {
        VectorIndex::metric(&self.0)
    }
}

impl Drop for HipHnswIndex {
    fn drop(&mut self) {
        // SAFETY: the handle was created by the library and is destroyed exactly once.
        unsafe { sys::vdb_hip_index_destroy(self.h) }
    }
}

/// `DualPrecisionConfig` (`native/dual_precision.rs:32-57`), passed with every call exactly as the reference passes it.
#[derive(Debug, Clone)]
pub struct DualPrecisionConfig {
    /// Candidates of the int8 walk that are re-scored exactly = `k * oversampling_ratio` (default 4).
    pub oversampling_ratio: usize,
    /// Use int8 quantised distances for the graph traversal (default true).
    pub use_int8_traversal: bool,
    /// Smaller indexes use the f32 graph search (default 10 000).
    pub min_index_size: usize,
    /// Accepted for source compatibility; the library keeps its timings in `vdb_hip_index_last_kernel_ms`.
    pub debug_timings: bool,
}

impl Default for DualPrecisionConfig {
    fn default() -> Self {
        // dual_precision.rs:47-56
        Self { oversampling_ratio: 4, use_int8_traversal: true, min_index_size: 10_000, debug_timings: false }
    }
}

/// `DualPrecisionHnsw` (`native/dual_precision.rs:59-321`): f32 graph + per-dimension min/max `ScalarQuantizer` codes
/// (`native/quantization.rs:191-251`), int8 traversal with exact f32 re-scoring.  Node ids are insertion order, as in the
/// reference (`insert` returns the `NodeId`).
pub struct HipDualPrecisionHnsw {
    inner: HipHnswIndex,
    training_sample_size: usize,
}

impl HipDualPrecisionHnsw {
    /// `DualPrecisionHnsw::new` (`dual_precision.rs:88-106`).  `None` without a HIP device.
    #[must_use]
    pub fn new(metric: DistanceMetric, dimension: usize, max_connections: usize, ef_construction: usize, max_elements: usize) -> Option<Self> {
        let mut params = HnswParams::auto(dimension);
        params.max_connections = max_connections;
        params.ef_construction = ef_construction;
        params.max_elements = max_elements;
        let inner = HipHnswIndex::with_params(dimension, metric, params)?;
        Some(Self { inner, training_sample_size: 1000.min(max_elements) })
    }

    /// `DualPrecisionHnsw::len` (`dual_precision.rs:108-111`)
    #[must_use]
    pub fn len(&self) -> usize {
        VectorIndex::len(&self.inner)
    }

    /// `DualPrecisionHnsw::is_empty` (`dual_precision.rs:113-116`)
    #[must_use]
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }

    /// `DualPrecisionHnsw::is_quantizer_trained` (`dual_precision.rs:118-121`), read from the handle.
    #[must_use]
    pub fn is_quantizer_trained(&self) -> bool {
        let mut t: i32 = 0;
        // SAFETY: live handle, valid out pointer.
        check(unsafe { sys::vdb_hip_index_quantizer_trained(self.inner.h, &mut t) });
        t != 0
    }

    /// `DualPrecisionHnsw::insert` (`dual_precision.rs:127-150`): the node id is the insertion order; the quantiser is trained
    /// once `training_sample_size` vectors are in, later rows are encoded as they arrive (the library does that itself).
    pub fn insert(&mut self, vector: Vec<f32>) -> usize {
        let node_id = self.len();
        VectorIndex::insert(&self.inner, node_id as u64, &vector);
        if !self.is_quantizer_trained() && self.len() >= self.training_sample_size {
            self.train(self.training_sample_size);
        }
        node_id
    }

    fn train(&self, sample_rows: usize) {
        // SAFETY: live handle.
        check(unsafe { sys::vdb_hip_index_train_quantizer(self.inner.h, sample_rows as u32) });
    }

    /// `DualPrecisionHnsw::force_train_quantizer` (`dual_precision.rs:181-185`): train on what is there.
    pub fn force_train_quantizer(&mut self) {
        if !self.is_quantizer_trained() && !self.is_empty() {
            self.train(self.len().min(self.training_sample_size.max(1)));
        }
    }

    /// `DualPrecisionHnsw::search` (`dual_precision.rs:191-200`).  Both branches are `NativeHnsw::search` with the caller's
    /// `ef_search` AS GIVEN (`graph.rs:251-270`: `search_layer(ef_search)`, cut to k — no `SearchQuality` rule, no `max(ef, k)`):
    /// without a quantiser directly; with one, `search_dual_precision` (`:209-243`) asks that search for
    /// `rerank_k = max(2 ef, 4 k)` candidates — it returns the at most `ef_search` it has —, re-computes their distances with the
    /// same f32 function the walk used, sorts them stably and keeps k: the first k of the same list.  The NativeHnsw-level entry
    /// point of the library with ONE entry point is that search (`vdb_hip_index_search_multi_entry`, `num_probes = 1`: no draw
    /// from the graph's stream, `graph.rs:303`).
    #[must_use]
    pub fn search(&self, query: &[f32], k: usize, ef_search: usize) -> Vec<(u64, f32)> {
        self.inner.search_multi_entry(query, k, ef_search, 1)
    }

    /// `DualPrecisionHnsw::search_with_config` (`dual_precision.rs:259-278`): int8 traversal only with a trained quantiser,
    /// `use_int8_traversal` and at least `min_index_size` vectors — decided inside the library from the call's own config.
    #[must_use]
    pub fn search_with_config(&self, query: &[f32], k: usize, ef_search: usize, config: &DualPrecisionConfig) -> Vec<(u64, f32)> {
        self.inner.validate_dimension(query, "Query");
        if k == 0 {
            return Vec::new();
        }
        let mut ids = vec![0u64; k];
        let mut scores = vec![0f32; k];
        let mut n: u32 = 0;
        // SAFETY: one query of `dimension` floats, k outputs.
        check(unsafe {
            sys::vdb_hip_index_search_with_config(
                self.inner.h,
                query.as_ptr(),
                1,
                k as u32,
                ef_search as u32,
                config.oversampling_ratio.max(1) as u32,
                i32::from(config.use_int8_traversal),
                config.min_index_size as u64,
                ids.as_mut_ptr(),
                scores.as_mut_ptr(),
                &mut n,
            )
        });
        ids.truncate(n as usize);
        scores.truncate(n as usize);
        ids.into_iter().zip(scores).collect()
    }

    /// The index under it (exact search, persistence, options).
    #[must_use]
    pub fn inner(&self) -> &HipHnswIndex {
        &self.inner
    }
}

/// Process-wide default of the int8 walk's oversampling for callers of `VDB_SEARCH_HNSW_INT8` that pass none
/// ([`HipDualPrecisionHnsw::search_with_config`] passes its own per call).
pub fn set_default_int8_oversampling(ratio: usize) {
    // SAFETY: no pointers.
    check(unsafe { sys::vdb_hip_set_int8_oversampling(ratio as u32) });
}

/// The free functions of the reference's SIMD module on the path, n vectors per call on device 0 (`vec_utils.hip`).
pub mod simd {
    use super::{check, sys};

    /// `simd::norm` / `simd_explicit::norm_simd` (`simd.rs:240-242`; `simd_explicit.rs:194-215`) for every row.
    #[must_use]
    pub fn batch_norm(vectors_rowmajor: &[f32], dimension: usize) -> Vec<f32> {
        assert!(dimension > 0 && vectors_rowmajor.len() % dimension == 0, "rows must be whole vectors");
        let n = vectors_rowmajor.len() / dimension;
        let mut out = vec![0f32; n];
        // SAFETY: n rows of `dimension` floats in, n floats out.
        check(unsafe { sys::vdb_hip_batch_norm(0, vectors_rowmajor.as_ptr(), n as u64, dimension as u32, out.as_mut_ptr()) });
        out
    }

    /// `simd::normalize_inplace` (`simd.rs:217-219`; `simd_explicit.rs:638-664`) for every row; a zero vector stays as it is.
    pub fn normalize_rows(vectors_rowmajor: &mut [f32], dimension: usize) {
        assert!(dimension > 0 && vectors_rowmajor.len() % dimension == 0, "rows must be whole vectors");
        let n = vectors_rowmajor.len() / dimension;
        // SAFETY: n rows of `dimension` floats, rewritten in place.
        check(unsafe { sys::vdb_hip_normalize_rows(0, vectors_rowmajor.as_mut_ptr(), n as u64, dimension as u32) });
    }

    /// `simd_explicit::batch_dot_product` (`simd_explicit.rs:519-560`): `out[i * n + j] = dot(queries[i], vectors[j])`.
    #[must_use]
    pub fn batch_dot_product(queries_rowmajor: &[f32], vectors_rowmajor: &[f32], dimension: usize) -> Vec<f32> {
        assert!(dimension > 0 && queries_rowmajor.len() % dimension == 0 && vectors_rowmajor.len() % dimension == 0, "rows must be whole vectors");
        let (nq, n) = (queries_rowmajor.len() / dimension, vectors_rowmajor.len() / dimension);
        let mut out = vec![0f32; nq * n];
        // SAFETY: nq and n rows of `dimension` floats in, nq * n floats out.
        check(unsafe {
            sys::vdb_hip_batch_dot_product(0, queries_rowmajor.as_ptr(), nq as u32, vectors_rowmajor.as_ptr(), n as u64, dimension as u32, out.as_mut_ptr())
        });
        out
    }

    /// `hamming_distance_binary(_fast)` (`simd_explicit.rs:308-360`): one packed query against n packed rows of `words` u64.
    #[must_use]
    pub fn batch_hamming_binary(query_words: &[u64], rows_words: &[u64]) -> Vec<u32> {
        let words = query_words.len();
        assert!(words > 0 && rows_words.len() % words == 0, "rows must be whole bit vectors");
        let n = rows_words.len() / words;
        let mut out = vec![0u32; n];
        // SAFETY: `words` u64 per vector, n outputs.
        check(unsafe { sys::vdb_hip_batch_hamming_binary(0, query_words.as_ptr(), rows_words.as_ptr(), n as u64, words as u32, out.as_mut_ptr()) });
        out
    }

    /// `jaccard_similarity_binary` (`simd_explicit.rs:457-500`): one packed query against n packed rows of `words` u64.
    #[must_use]
    pub fn batch_jaccard_binary(query_words: &[u64], rows_words: &[u64]) -> Vec<f32> {
        let words = query_words.len();
        assert!(words > 0 && rows_words.len() % words == 0, "rows must be whole bit vectors");
        let n = rows_words.len() / words;
        let mut out = vec![0f32; n];
        // SAFETY: as above.
        check(unsafe { sys::vdb_hip_batch_jaccard_binary(0, query_words.as_ptr(), rows_words.as_ptr(), n as u64, words as u32, out.as_mut_ptr()) });
        out
    }

    /// `DistanceEngine::batch_distance` over rows that already live in device memory, enqueued on `stream`.
    ///
    /// # Safety
    /// `d_query` (dim floats), `d_vectors_rowmajor` (n * dim floats) and `d_out` (n floats) must be device pointers that stay
    /// valid until the work enqueued on `stream` has completed.
    #[allow(clippy::too_many_arguments)]
    pub unsafe fn batch_distance_dev(
        metric: i32,
        kind: i32,
        d_query: *const f32,
        d_vectors_rowmajor: *const f32,
        n: usize,
        dimension: usize,
        d_out: *mut f32,
        stream: *mut std::os::raw::c_void,
    ) {
        check(sys::vdb_hip_batch_distance_dev(metric, kind, d_query, d_vectors_rowmajor, n as u64, dimension as u32, d_out, stream));
    }
}

/// 128-byte RCCL id for [`HipHnswIndex::join_group`]; rank 0 generates it and hands it to the other ranks.
#[must_use]
pub fn comm_unique_id() -> [u8; sys::VDB_COMM_ID_BYTES] {
    let mut id = [0u8; sys::VDB_COMM_ID_BYTES];
    // SAFETY: the buffer has VDB_COMM_ID_BYTES bytes.
    check(unsafe { sys::vdb_hip_comm_unique_id(id.as_mut_ptr()) });
    id
}

fn batch_distance_flat(device: i32, metric: DistanceMetric, kind: i32, query: &[f32], flat: &[f32], n: usize) -> Vec<f32> {
    let mut out = vec![0f32; n];
    if n == 0 {
        return out;
    }
    // SAFETY: `flat` holds n * query.len() floats (callers build it that way), `out` holds n.
    check(unsafe {
        sys::vdb_hip_batch_distance(device, metric_code(metric), kind, query.as_ptr(), flat.as_ptr(), n as u64, query.len() as u32, out.as_mut_ptr())
    });
    out
}

/// `DistanceEngine` over the GPU (`index/hnsw/native/distance.rs:14-28`): `distance` conventions of `SimdDistance`
/// (1 - cos, sqrt(l2), -dot, hamming, 1 - jaccard; `distance.rs:75-85`).
pub struct HipDistance {
    metric: DistanceMetric,
    device: i32,
}

impl HipDistance {
    /// `None` without a HIP device.
    #[must_use]
    pub fn new(metric: DistanceMetric) -> Option<Self> {
        (device_count() > 0).then_some(Self { metric, device: 0 })
    }
}

impl DistanceEngine for HipDistance {
    fn distance(&self, a: &[f32], b: &[f32]) -> f32 {
        assert_eq!(a.len(), b.len(), "Vector dimensions must match");
        batch_distance_flat(self.device, self.metric, sys::VDB_KIND_ENGINE, a, b, 1)[0]
    }

    fn batch_distance(&self, query: &[f32], candidates: &[&[f32]]) -> Vec<f32> {
        let mut flat = Vec::with_capacity(candidates.len() * query.len());
        for c in candidates {
            assert_eq!(c.len(), query.len(), "Vector dimensions must match");
            flat.extend_from_slice(c);
        }
        batch_distance_flat(self.device, self.metric, sys::VDB_KIND_ENGINE, query, &flat, candidates.len())
    }

    fn metric(&self) -> DistanceMetric {
        self.metric
    }
}

/// `GpuAccelerator` (`gpu/gpu_backend.rs`): batch kernels over a flat row-major matrix, raw scores.
pub struct HipAccelerator {
    device: i32,
}

impl HipAccelerator {
    /// `GpuAccelerator::new` (`gpu_backend.rs:33`).
    #[must_use]
    pub fn new() -> Option<Self> {
        (device_count() > 0).then_some(Self { device: 0 })
    }

    /// `GpuAccelerator::is_available` (`gpu_backend.rs:136`).
    #[must_use]
    pub fn is_available() -> bool {
        device_count() > 0
    }

    fn batch(&self, metric: DistanceMetric, vectors: &[f32], query: &[f32], dimension: usize) -> Vec<f32> {
        if dimension == 0 || vectors.is_empty() {
            return Vec::new(); // gpu_backend.rs:163-166
        }
        assert_eq!(query.len(), dimension, "Query dimension mismatch: expected {}, got {}", dimension, query.len());
        let n = vectors.len() / dimension;
        batch_distance_flat(self.device, metric, sys::VDB_KIND_RAW, query, &vectors[..n * dimension], n)
    }

    /// `GpuAccelerator::batch_cosine_similarity` (`gpu_backend.rs:157`).
    #[must_use]
    pub fn batch_cosine_similarity(&self, vectors: &[f32], query: &[f32], dimension: usize) -> Vec<f32> {
        self.batch(DistanceMetric::Cosine, vectors, query, dimension)
    }

    /// `GpuAccelerator::batch_euclidean_distance` (`gpu_backend.rs:355`).
    #[must_use]
    pub fn batch_euclidean_distance(&self, vectors: &[f32], query: &[f32], dimension: usize) -> Vec<f32> {
        self.batch(DistanceMetric::Euclidean, vectors, query, dimension)
    }

    /// `GpuAccelerator::batch_dot_product` (`gpu_backend.rs:397`).
    #[must_use]
    pub fn batch_dot_product(&self, vectors: &[f32], query: &[f32], dimension: usize) -> Vec<f32> {
        self.batch(DistanceMetric::DotProduct, vectors, query, dimension)
    }
}
