// vdb_index.hpp — host-side index object behind the C ABI (include/velesdb_hip.h).
// Mirrors the state of the reference's HnswIndex (index/hnsw/index/mod.rs:93-131):
// id mappings (sharded_mappings.rs:32-93), vector storage, and the graph — except that
// vectors and adjacency live in HBM, contiguous (the reference keeps Vec<Vec<f32>>,
// native/graph.rs:22).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/velesdb_hip.h"
#include "vdb_host_sync.hpp"

namespace vdb {

void set_last_error(const std::string& s);
int32_t fail(int32_t code, const std::string& msg);
bool hip_ok(hipError_t e, const char* what);

#define VDB_HIP(call)                                                  \
  do {                                                                 \
    hipError_t _e = (call);                                            \
    if (_e != hipSuccess) {                                            \
      return ::vdb::fail(VDB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
    }                                                                  \
  } while (0)

// growable device buffer (never shrinks); content preserved on growth when keep=true
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes, bool keep, hipStream_t st);
  void release();
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

// non-owning typed view of a region inside a DevBuf (the three result arrays of a search live in ONE allocation)
struct DevView {
  void* p = nullptr;
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

// growable pinned host buffer (hipHostMalloc): staging of the host-pointer entry points — queries go up and results come back
// through it with ONE truly asynchronous copy each way (copies from / to pageable caller memory are staged and synchronised by
// the runtime, one at a time)
struct HostStage {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes);
  void release();
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

struct Combiner;  // search_front.hip: the combining front of the host-pointer search entry points
Combiner* combiner_new();
void combiner_free(Combiner*);

struct GraphLayer {
  DevBuf nbr;  // [capacity][stride] u32
  DevBuf cnt;  // [capacity] u32
  DevBuf ndist;  // [capacity][stride] f32: distance node <-> neighbour (construction cache, hnsw_build.hip)
  uint32_t stride = 0;
};

struct EventPair {
  hipEvent_t a = nullptr, b = nullptr;
};

struct ShardGroup;  // shard_group.hip: the children of a multi-device handle
struct ProcComm;    // shard_group.hip: this process's membership of a one-process-per-GPU shard group (RCCL)
void shard_group_free(ShardGroup*);
void proc_comm_free(ProcComm*);

// nothing unwinds across the C ABI: every extern "C" body runs inside guarded()
template <class F>
static inline int32_t guarded(F&& f) noexcept {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    return fail(VDB_ERR_OOM, "out of host memory");
  } catch (const std::exception& e) {
    return fail(VDB_ERR_INVALID_ARG, std::string("internal exception: ") + e.what());
  } catch (...) {
    return fail(VDB_ERR_INVALID_ARG, "internal exception");
  }
}

// after taking ix->mu: select the device and order ix->stream behind a device-resident search still in flight on a
// caller's stream
// exclusive = the caller holds ix->mu exclusively (it is about to change the index): the primary's stream is also ordered behind
// whatever its search contexts still have in flight
// changes = false: an exclusive READER (get_neighbors, save_*: they use the primary's stream and scratch, nothing a search reads
// is touched) — search contexts keep their views
int32_t enter_index(vdb_hip_index* ix, bool exclusive = true, bool changes = true);
// the handle that owns the data a search context reads (itself unless it is a clone)
static inline vdb_hip_index* primary_of(vdb_hip_index* ix);
#define VDB_ENTER(ix)                        \
  do {                                       \
    int32_t _erc = ::vdb::enter_index(ix);   \
    if (_erc != VDB_OK) return _erc;         \
  } while (0)

#define VDB_ENTER_READONLY(ix)                          \
  do {                                                  \
    int32_t _erc = ::vdb::enter_index(ix, true, false); \
    if (_erc != VDB_OK) return _erc;                    \
  } while (0)

#define VDB_ENTER_SHARED(ix)                        \
  do {                                              \
    int32_t _erc = ::vdb::enter_index(ix, false);   \
    if (_erc != VDB_OK) return _erc;                \
  } while (0)

#define VDB_NO_GROUP(ix, what)                                                                     \
  do {                                                                                             \
    if ((ix)->group) return ::vdb::fail(VDB_ERR_UNSUPPORTED, what ": not available on a multi-device handle"); \
  } while (0)

}  // namespace vdb

struct vdb_hip_index {
  int device = 0;
  int n_cus = 256;
  uint32_t dim = 0;
  int metric = 0;
  uint32_t M = 0, M0 = 0, efc = 0;
  uint64_t row_stride = 0;  // floats
  uint32_t words = 0;       // packed-bit words per row (multiple of 4)
  uint64_t capacity = 0;    // rows allocated
  uint64_t n_rows = 0;      // rows/nodes present (including soft-deleted)
  hipStream_t stream = nullptr;

  vdb::DevBuf rows, norms, bits, alive, ext_ids;
  // optional bf16 copy of the rows for the GEMM-distance sweep (vdb_hip_index_enable_bf16)
  vdb::DevBuf rows_bf16, norms_bf16;
  // one u32: the f32 bit pattern of the largest rounding residual ratio |x - bf16(x)| / |x| over every row the bf16 copy ever held
  // (prep_bf16_rows raises it; never lowered: a bound) — the level-2 selection's measured error bound (sweep_split.hip)
  vdb::DevBuf bf16_rho;
  vdb::DevBuf l2_rho;  // the same for the Euclidean selection image (its first dim columns are the bf16-rounded rows)
  vdb::DevBuf sq8_rho;  // ... and for the SQ8 storage mode's image (bf16 of the dequantised rows)
  // optional scalar quantiser + u8 codes for the int8 traversal (hnsw_int8.hip)
  vdb::DevBuf sq_min, sq_scale, codes, codes_sq;
  uint32_t code_words = 0;
  bool quantizer_trained = false;
  bool bf16_enabled = false;
  uint64_t bf16_stride = 0;  // bf16 elements per row (multiple of 8)
  uint64_t bf16_rows = 0;    // rows converted so far
  // split-bf16 image of the f32 rows (hi + lo per element, 4 bytes like the f32 row: sweep_split.hip), built at the first
  // large exact Cosine / DotProduct batch and kept up to date from then on
  vdb::DevBuf rows_split;
  // per-handle options (vdb_hip_index_set_option): -1 = follow the process-wide default (vdb_hip_set_*)
  int32_t opt[VDB_OPT_COUNT_] = {-1, -1, -1, -1, -1, -1, -1, -1};
  bool split_enabled = false;
  bool sel_norms = false;    // canonical f32 norms are kept for every row whatever the metric (selection levels 1 / 2)
  // level 2 (plain bf16 selection) adaptivity: the verdict counts of finished batches arrive in pinned host memory
  // ({unproven, queries, sequence, level}); too many unproven queries park the handle at level 1 for a while
  volatile uint32_t* sel_stats = nullptr;
  uint32_t sel_seq = 0, sel_seq_seen = 0, sel16_hold = 0;
  uint32_t wide_hold = 0;   // batches with 10 < k answered by the exact kernels after a batch the WIDE selection could not prove (sweep_wide.hip)
  int last_select_level = 0;
  uint32_t last_kernels = 0;  // vdb_kernel_bit set of the last search call (vdb_hip_index_last_kernels)
  // Euclidean batches through the selection stage: augmented bf16 image [capacity][dim + 64], augmented f32 seed prefix
  vdb::DevBuf l2_img, l2_seed;
  uint64_t l2_rows = 0;     // rows converted so far
  // Cosine batches (level 2, round 6): bf16 image of the NORMALISED rows v / |v| [capacity][dim] and its largest rounding residual ratio —
  // the selection then runs as a DotProduct of unit vectors (no row norm in the kernel's bound): sweep_split.hip seln_rows_kernel
  vdb::DevBuf cosn_img, cosn_rho;
  uint64_t cosn_rows = 0;   // rows converted so far
  uint32_t l2_hold = 0;     // batches to answer on the f32 matrix-core path after a batch the selection could not prove
  uint64_t split_rows = 0;   // rows converted so far
  size_t split_flags_off = 0;     // where the last split batch left its per-query verdicts in s_seed
  uint32_t split_flags_n = 0;
  hipStream_t split_flags_stream = nullptr;
  // optional quantised copy of the rows per StorageMode (storage_modes.hip; core/quantization.rs)
  int32_t storage_mode = 0;      // VDB_STORAGE_FULL
  uint64_t sq8_stride = 0;       // bytes per SQ8 row (multiple of 16)
  vdb::DevBuf sq8_codes, sq8_min, sq8_max, sq8_nsq, sign_bits;
  // selection stage over SQ8 (level 3): bf16 image of the dequantised rows, their norms, f32 seed prefix.  The Binary storage mode
  // keeps the four-bit image of its sign-bit codes in the same two buffers (one storage mode at a time; sq8_img_rows = its progress)
  vdb::DevBuf sq8_img, sq8_nrm, sq8_seed;
  uint64_t sq8_img_rows = 0;
  // Hamming / Jaccard batches on the matrix cores (bits_gemm.hip): four-bit image (E2M1 values 0 / +-1) of the packed bit rows [capacity][stride bytes], the
  // rows' bit counts as floats; built at the first large batch, kept current by inserts from then on
  vdb::DevBuf bits_img, bits_cnt;
  uint64_t bits_img_rows = 0;
  uint32_t sq8_hold = 0;   // batches to answer with the exact SQ8 sweep after a batch the selection could not prove
  // graph
  std::vector<vdb::GraphLayer> layers;
  bool graph_valid = true;   // false once rows exist that are not linked into the graph
  int64_t entry_point = -1;
  uint32_t max_layer = 0;
  uint64_t graph_nodes = 0;  // nodes linked
  uint64_t rng_state = 0x5DEECE66D1A4B5B5ull;  // native/graph.rs:72

  // id mappings (host)
  std::unordered_map<uint64_t, uint64_t> id_to_idx;
  std::vector<uint64_t> idx_to_id;
  std::vector<uint8_t> idx_live;
  uint64_t live = 0;
  bool any_dead = false;
  bool raw_ef = false;  // transient, under the exclusive lock: the running call is NativeHnsw-level (search_multi_entry) — its ef is used as given

  // scratch
  vdb::DevBuf s_queries, s_part_keys, s_part_cnt, s_qbits, s_misc;
  vdb::DevBuf s_tickets;  // [2] u32, zero between calls: the block tickets of the one-launch packed-bit search (sweep_bits_fused)
  // results of a host-pointer search: ONE allocation [ids nq*k u64 | scores nq*k f32 | n nq u32] (reserve_out), so that one
  // copy brings everything back; the three views point into it
  vdb::DevBuf s_out;
  vdb::DevView s_out_ids, s_out_scores, s_out_n;
  size_t s_out_bytes = 0;            // bytes of the block the views currently describe
  vdb::HostStage h_in, h_out;        // pinned staging: queries up, the result block down
  vdb::DevBuf s_fb_keys;  // partial lists of the exact fallback launch behind the split-bf16 selection
  vdb::DevBuf s_seed;  // seeding pre-pass of the bf16 GEMM sweep: partial lists, merged prefix top-k, seed keys
  uint64_t euclid_fallbacks = 0;  // queries of Euclidean matrix-core batches re-run through the exact sweep (diagnostic)
  vdb::DevBuf s_visited, s_vlog, s_stats;  // HNSW traversal scratch (hnsw_kernels.hip)
  vdb::DevBuf s_build_stats;               // [3] u64, cumulative since creation: rows evaluated / distance phases / nodes of the insert kernel
  vdb::DevBuf s_levels, s_req_keys, s_req_vals, s_sort_tmp;  // construction scratch (hnsw_build.hip)
  bool ndist_valid = true;  // false for a graph loaded from files until the cache is recomputed
  uint64_t vis_words = 0;
  bool stats_pending = false;
  std::vector<vdb::EventPair> ev_pool;
  size_t ev_used = 0;
  std::vector<vdb::EventPair> sel_ev;  // kernel timing: one pair per launch of the selection kernel in the last search call
  size_t sel_ev_used = 0;
  uint64_t last_n_dist = 0, last_n_expand = 0, last_pf_hits = 0;

  // Reader / writer lock of the handle (the reference's RwLock around the graph, index/hnsw/index/search.rs:80): searches
  // hold it SHARED while they enqueue (and, host entry points, until their results are back), everything that changes the
  // index holds it exclusively.
  mutable vdb::IndexMutex mu;
  // Search contexts.  A search needs scratch buffers, event pools and a stream of its own; concurrent searches on one handle
  // each lease a CONTEXT: this object itself (context 0) or one of `ctx_clones` — handles that alias this one's data buffers
  // (rows, norms, images, graph: non-owning copies of the DevBufs, refreshed when `version` moved) and own only their scratch
  // and stream.  ctx_mu = "this context is in use"; pool_mu guards ctx_clones.  Clones are created on demand (a handle that
  // is only ever searched by one thread at a time never makes one).
  vdb_hip_index* primary = nullptr;            // set in a clone
  std::vector<vdb_hip_index*> ctx_clones;      // (primary only)
  std::mutex ctx_mu, pool_mu;
  uint64_t generation = 0;                     // unique per created handle (the thread-local "last context" is keyed on it)
  vdb::Combiner* combiner = nullptr;           // (primary only) callers of the host-pointer entry points that arrive together share one launch
  uint64_t version = 1, synced_version = 0;    // primary: bumped by every change; clone: the version its views were copied at
  // first-use construction of the selection images (split / bf16 / augmented / SQ8-dequantised) happens inside searches, i.e.
  // under the SHARED lock: serialised here, always on the primary's fields
  std::mutex img_mu;
  // Device-side ordering between streams (the scratch buffers, the rows and the graph are shared by every search of this
  // index): a device-resident search enqueued on a caller's stream records ev_foreign; whatever touches the index next on
  // another stream — ix->stream for every host entry point, another caller stream — waits for it first (VDB_ENTER /
  // vdb_hip_index_search_batch_dev), and a caller stream waits for the work pending on ix->stream (ev_own).
  hipEvent_t ev_foreign = nullptr, ev_own = nullptr;
  hipStream_t last_foreign = nullptr;
  bool foreign_pending = false;
  // work was enqueued on `stream` (enter_index: every host entry point) since a caller's stream last waited for it: a device-resident
  // search on a caller's stream records + waits for ev_own only then (two packets less per call in a loop of such searches)
  // (atomic: enter_index sets it under the SHARED lock too — read-only entry points — while a device-resident search clears it)
  std::atomic<bool> own_dirty{true};

  // multi-device handle (vdb_hip_index_create with n_devices > 1): this object then owns no device memory, only the
  // id mappings / counters above and the children; every entry point dispatches through the group (shard_group.hip)
  vdb::ShardGroup* group = nullptr;
  // one-process-per-GPU shard group joined with vdb_hip_index_join_group: exact searches end with ONE RCCL all-gather
  // of the per-shard top-k records and the merge kernel
  vdb::ProcComm* pcomm = nullptr;
};

namespace vdb {
static inline vdb_hip_index* primary_of(vdb_hip_index* ix) { return ix->primary ? ix->primary : ix; }
// rows per index: the tiled kernels count whole 256-row tiles of [0, n) in 32 bits — (n + 255) / 256 must not wrap
// (tests/gemm_schedule_model.cpp walks the launch schedule up to this limit)
constexpr uint64_t kMaxRowsPerIndex = 0xFFFFFE00ull;  // 2^32 - 512
// after a change to the index (exclusive lock held): search contexts refresh their views before their next search
static inline void mark_changed(vdb_hip_index* ix) { primary_of(ix)->version++; }
// copies the selection-image state (buffers + progress counters) of the primary into a search context
void copy_image_fields(vdb_hip_index* dst, const vdb_hip_index* src);
// a leased search context of `ix` (shared lock on ix->mu held by the caller): the primary if free, else a free / new clone
struct CtxLease {
  vdb_hip_index* ctx = nullptr;
  int32_t rc = VDB_OK;
  explicit CtxLease(vdb_hip_index* ix);
  ~CtxLease();
  CtxLease(const CtxLease&) = delete;
  CtxLease& operator=(const CtxLease&) = delete;
};
// one or two packed-bit queries per call take the one-launch kernel (index.hip: the default engine; probe builds: VELESDB_BITS_FUSED=0)
bool opt_bits_fused(const vdb_hip_index* ix);
// the context that served this thread's last search on `ix` (diagnostic getters read from it)
vdb_hip_index* last_context(vdb_hip_index* ix);
// index.hip
int32_t create_single(uint32_t dim, int32_t metric, uint32_t M, uint32_t ef_construction, uint64_t max_elements,
                      int32_t device, vdb_hip_index** out);
void destroy_single(vdb_hip_index* ix);
// search_with_quality dispatch for device-resident queries (enqueue only; see index.hip)
int32_t search_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k, uint32_t ef,
                   int32_t mode, uint64_t* d_ids, float* d_scores, uint32_t* d_n, hipStream_t st, uint32_t cap_mult = 1,
                   bool* used_hnsw = nullptr, uint32_t rerank_k = 0, const uint32_t* d_extra_eps = nullptr);
// host queries -> results in ix->s_out_ids / s_out_scores / s_out_n on the device (out_n also on the host); the caller
// holds ix->mu.  Re-runs HNSW batches whose candidate list overflowed.
int32_t search_to_device(vdb_hip_index* ix, const float* queries, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode,
                         uint32_t rerank_k, uint32_t* out_n);
// the result block of a context for nq queries x kk results: sizes s_out and points the three views into it
int32_t reserve_out(vdb_hip_index* ix, uint32_t nq, size_t kk, hipStream_t st);
// the pieces of search_to_device (search_front.hip uses them directly): rows of `queries` into the pinned staging buffer at query
// slot `at` (row_stride layout, padding zeroed); the staged rows up, the search, the whole result block back into h_out and ONE
// synchronisation (re-runs HNSW batches whose candidate list overflowed)
int32_t stage_queries(vdb_hip_index* ix, const float* queries, uint32_t at, uint32_t nq, uint32_t nq_total);
int32_t search_staged(vdb_hip_index* ix, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode, uint32_t rerank_k,
                      const uint32_t* d_extra_eps = nullptr);
// effective option values (index.hip)
int64_t opt_value(const vdb_hip_index* ix, int32_t option);
bool mode_higher_is_better(int metric, int32_t mode);
// shard_group.hip
int32_t group_create(vdb_hip_index* parent, const int32_t* devices, int32_t n_devices, int32_t shard_mode,
                     uint64_t max_elements);
int32_t group_insert(vdb_hip_index* ix, const uint64_t* ids, const float* vecs, uint64_t n, int kind, uint32_t max_batch,
                     uint64_t* inserted);  // kind 0 insert_batch, 1 insert_batch_parallel, 2 upload
int32_t group_remove(vdb_hip_index* ix, uint64_t id, int32_t* removed);
vdb_hip_index* group_shard(const vdb_hip_index* ix, size_t s);  // child s of a multi-device handle
size_t group_size(const vdb_hip_index* ix);
int group_mode(const vdb_hip_index* ix);
// the host-pointer search of one handle (single device, group, combining front): search_front.hip
int32_t search_batch_host(vdb_hip_index* ix, const float* queries, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode, uint32_t rerank_k,
                          uint64_t* out_ids, float* out_scores, uint32_t* out_n);
int32_t group_for_all(vdb_hip_index* ix, int op, uint32_t arg);
int32_t group_set_option(vdb_hip_index* ix, int32_t option, int64_t value);
vdb_hip_index* group_first_shard(vdb_hip_index* ix);  // op: 0 build_graph, 1 enable_bf16, 2 storage mode, 3 quantizer
int32_t group_search_host(vdb_hip_index* ix, const float* queries, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode,
                          uint32_t rerank_k, uint64_t* out_ids, float* out_scores, uint32_t* out_n);
int32_t group_search_dev(vdb_hip_index* ix, const float* d_q, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode,
                         uint64_t* d_ids, float* d_scores, uint32_t* d_n, hipStream_t st);
// after a local search whose results sit in (d_ids, d_scores, d_n) on `st`: all-gather the per-shard records over the
// process group and merge them into the same buffers (every rank ends with the global top-k)
int32_t pcomm_exchange_merge(vdb_hip_index* ix, uint32_t nq, uint32_t k, bool hib, uint64_t* d_ids, float* d_scores,
                             uint32_t* d_n, hipStream_t st);
int32_t ensure_capacity(vdb_hip_index* ix, uint64_t want);
int32_t append_host_rows(vdb_hip_index* ix, const uint64_t* ids, const float* vecs, uint64_t n, uint64_t* inserted,
                         uint64_t* first_row);
// graph.hip
int32_t ensure_layers(vdb_hip_index* ix, uint32_t num_layers);
EventPair* next_events(vdb_hip_index* ix);  // nullptr when kernel timing is off
// hnsw_kernels.hip; cap_mult scales the room for tie candidates beyond ef (1 = default)
int32_t hnsw_search_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k, uint32_t ef,
                        uint32_t cap_mult, uint64_t* d_ids, float* d_scores, uint32_t* d_n, hipStream_t st,
                        uint32_t rerank_k = 0, const uint32_t* d_extra_eps = nullptr);
// hnsw_build.hip; max_batch 1 = the reference's sequential insert, 0 = default batched schedule
int32_t graph_insert_rows(vdb_hip_index* ix, uint64_t first, uint64_t n, uint32_t max_batch);
int32_t ensure_traversal_scratch(vdb_hip_index* ix, hipStream_t st, int want_slots = 0);
// hnsw_int8.hip
int32_t quantizer_train(vdb_hip_index* ix, uint32_t sample_rows);
int32_t quantize_rows(vdb_hip_index* ix, uint64_t first, uint64_t n);
int32_t hnsw_search_int8_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k,
                             uint32_t ef_search, uint32_t oversampling, uint32_t cap_mult, uint64_t* d_ids,
                             float* d_scores, uint32_t* d_n, hipStream_t st);  // visited bitmaps + logs + stats
constexpr uint32_t kVlogCap = 16384;
constexpr int kTraversalSlotsPerCu = 8;  // visited bitmaps / id logs are sized for this many queries in flight per CU
// storage_modes.hip
int32_t storage_mode_append(vdb_hip_index* ix, uint64_t first, uint64_t n);
int32_t brute_sq8_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k, uint64_t* d_ids,
                      float* d_scores, uint32_t* d_n, hipStream_t st);
int32_t ensure_sq8_select(vdb_hip_index* ix, hipStream_t st);
struct SelectFinishArgs;  // vdb_kernels.hpp
int32_t sq8_fallback_flagged(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nqg, uint32_t k, const uint32_t* qmap,
                             const SelectFinishArgs& fin_in, hipStream_t st);
// select_stage.hip: selection + exact re-scoring + proof for a chunk of <= 1024 queries (level 1 / 2: f32 rows, 3: SQ8 storage mode)
int32_t brute_split_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nqg, uint32_t k, uint64_t* d_ids,
                        float* d_scores, uint32_t* d_n, hipStream_t st, int level);
int select_level_sq8(vdb_hip_index* ix, uint32_t nq_left, uint32_t k);
// bits_gemm.hip
uint32_t bits_gemm_chunk(const vdb_hip_index* ix, uint32_t nq_left, uint32_t k);
int32_t brute_bits_gemm_dev(vdb_hip_index* ix, int metric, const uint8_t* img, const float* cnt, const uint32_t* qbits, uint32_t nqg, uint32_t k,
                            uint64_t* d_ids, float* d_scores, uint32_t* d_n, hipStream_t st);
int32_t brute_binary_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k, uint64_t* d_ids,
                         float* d_scores, uint32_t* d_n, hipStream_t st);
}  // namespace vdb
