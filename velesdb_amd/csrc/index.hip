// index.hip — host side of libvelesdb_hip.so: the C ABI declared in include/velesdb_hip.h.
// Each entry point names the reference interface it replaces.  No CPU compute path exists
// here: every score, norm, bit-pack, top-k and traversal runs in a HIP kernel.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>

#include "vdb_index.hpp"
#include "vdb_probe_env.hpp"
#include "vdb_kernels.hpp"
#include "vdb_select_stage.hpp"

namespace vdb {

static thread_local std::string g_last_error;
// process-wide tuning / diagnostic knobs (results never depend on them): atomics, so that setting one while other
// threads search is a data-race-free read of either value
static std::atomic<int> g_timing{0};
static std::atomic<int> g_sweep_engine{1};  // 1 (default): MFMA kernel for cosine / dot (oracle mode M); 0: VALU kernels (mode C)
static std::atomic<uint32_t> g_max_tile{128};  // largest query tile of the exact sweep (vdb_hip_set_max_query_tile)
static std::atomic<int> g_split_selector{3};  // large exact Cosine / Dot batches: 0 exact kernel, 1 split-bf16 selection + exact re-scoring + proof,
                                              // 2 plain bf16 selection first (same proof, wider bound), level 1 when a handle's data defeats it
static std::atomic<uint32_t> g_int8_oversampling{4};  // DualPrecisionConfig::default().oversampling_ratio (dual_precision.rs:57)

// effective option values of a handle: its own (vdb_hip_index_set_option) or the process-wide default
static inline uint32_t opt_max_tile(const vdb_hip_index* ix) { return ix->opt[VDB_OPT_MAX_QUERY_TILE] >= 0 ? (uint32_t)ix->opt[VDB_OPT_MAX_QUERY_TILE] : (uint32_t)g_max_tile.load(); }
// one-launch packed-bit search for calls of one or two queries: the default engine's path (engine 0 keeps the three-launch form, so
// both stay under the parity tests); probe builds: VELESDB_BITS_FUSED=0 turns it off for A/B runs
bool opt_bits_fused(const vdb_hip_index* ix);
static inline int opt_engine(const vdb_hip_index* ix) { return ix->opt[VDB_OPT_SWEEP_ENGINE] >= 0 ? ix->opt[VDB_OPT_SWEEP_ENGINE] : g_sweep_engine.load(); }
bool opt_bits_fused(const vdb_hip_index* ix) {
  static const bool off = [] {
    const char* e = vdb::probe_env("VELESDB_BITS_FUSED");
    return e && e[0] == '0';
  }();
  return !off && opt_engine(ix) == 1;
}
static inline int opt_selector(const vdb_hip_index* ix) { return ix->opt[VDB_OPT_SELECTOR_LEVEL] >= 0 ? ix->opt[VDB_OPT_SELECTOR_LEVEL] : g_split_selector.load(); }
static inline uint32_t opt_oversampling(const vdb_hip_index* ix) { return ix->opt[VDB_OPT_INT8_OVERSAMPLING] >= 0 ? (uint32_t)ix->opt[VDB_OPT_INT8_OVERSAMPLING] : (uint32_t)g_int8_oversampling.load(); }
static inline bool opt_timing(const vdb_hip_index* ix) { return (ix->opt[VDB_OPT_KERNEL_TIMING] >= 0 ? ix->opt[VDB_OPT_KERNEL_TIMING] : g_timing.load()) != 0; }
// the combining front's defaults (include/velesdb_hip.h): 256 queries per combined launch, returning callers are awaited for <= 100 us, batches in flight by kind (0: walks 2, sweeps 1)
static constexpr int32_t kOptDefaultCombineMaxBatch = 256, kOptDefaultCombineWindowUs = 100, kOptDefaultCombineInflight = 0;
int64_t opt_value(const vdb_hip_index* ix, int32_t option) {
  switch (option) {
    case VDB_OPT_MAX_QUERY_TILE: return opt_max_tile(ix);
    case VDB_OPT_SWEEP_ENGINE: return opt_engine(ix);
    case VDB_OPT_SELECTOR_LEVEL: return opt_selector(ix);
    case VDB_OPT_INT8_OVERSAMPLING: return opt_oversampling(ix);
    case VDB_OPT_KERNEL_TIMING: return opt_timing(ix) ? 1 : 0;
    case VDB_OPT_COMBINE_MAX_BATCH: return ix->opt[option] >= 0 ? ix->opt[option] : kOptDefaultCombineMaxBatch;
    case VDB_OPT_COMBINE_WINDOW_US: return ix->opt[option] >= 0 ? ix->opt[option] : kOptDefaultCombineWindowUs;
    case VDB_OPT_COMBINE_INFLIGHT: return ix->opt[option] >= 0 ? ix->opt[option] : kOptDefaultCombineInflight;
    default: return -1;
  }
}


void set_last_error(const std::string& s) { g_last_error = s; }
int32_t fail(int32_t code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

hipError_t DevBuf::reserve(size_t bytes, bool keep, hipStream_t st) {
  if (bytes <= cap) return hipSuccess;
  size_t ncap = std::max(bytes, cap + cap / 2);
  ncap = (ncap + 255) & ~(size_t)255;
  void* np = nullptr;
  hipError_t e = hipMalloc(&np, ncap);
  if (e != hipSuccess) return e;
  if (keep && p && cap) {
    e = hipMemcpyAsync(np, p, cap, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
      (void)hipFree(np);
      return e;
    }
  }
  if (p) (void)hipFree(p);
  p = np;
  cap = ncap;
  return hipSuccess;
}
void DevBuf::release() {
  if (p) (void)hipFree(p);
  p = nullptr;
  cap = 0;
}
hipError_t HostStage::reserve(size_t bytes) {
  if (bytes <= cap) return hipSuccess;
  size_t ncap = std::max<size_t>(std::max(bytes, cap + cap / 2), 4096);
  ncap = (ncap + 4095) & ~(size_t)4095;
  void* np = nullptr;
  hipError_t e = hipHostMalloc(&np, ncap, hipHostMallocDefault);
  if (e != hipSuccess) return e;
  if (p) (void)hipHostFree(p);  // (contents are per call: nothing to keep)
  p = np;
  cap = ncap;
  return hipSuccess;
}
void HostStage::release() {
  if (p) (void)hipHostFree(p);
  p = nullptr;
  cap = 0;
}

static bool higher_is_better_host(int metric) {  // core/distance.rs:76-82
  return metric == VDB_COSINE || metric == VDB_DOT || metric == VDB_JACCARD;
}
static bool is_bits_metric(int metric) { return metric == VDB_HAMMING || metric == VDB_JACCARD; }

static int32_t check_device(int32_t* n_out) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    if (n_out) *n_out = 0;
    return fail(VDB_ERR_NO_DEVICE, "no HIP device visible (hipGetDeviceCount)");
  }
  if (n_out) *n_out = n;
  return VDB_OK;
}

int32_t enter_index(vdb_hip_index* ix, bool exclusive, bool changes) {
  VDB_HIP(hipSetDevice(ix->device));
  ix->own_dirty.store(true);  // (the caller is about to enqueue on ix->stream)
  if (ix->foreign_pending) {
    VDB_HIP(hipStreamWaitEvent(ix->stream, ix->ev_foreign, 0));
    ix->foreign_pending = false;
  }
  if (exclusive && changes) mark_changed(ix);  // (what searches read may change: contexts re-sync their views before their next search)
  // (exclusive callers only touch a primary: its search contexts may have device-resident searches in flight on callers' streams
  // or on their own; whatever changes the index waits for them — contexts are idle on the host side while mu is held exclusively)
  if (exclusive && !ix->primary)
    for (vdb_hip_index* c : ix->ctx_clones) {
      if (c->foreign_pending) {
        VDB_HIP(hipStreamWaitEvent(ix->stream, c->ev_foreign, 0));
        c->foreign_pending = false;
      }
      VDB_HIP(hipEventRecord(c->ev_own, c->stream));
      VDB_HIP(hipStreamWaitEvent(ix->stream, c->ev_own, 0));
    }
  return VDB_OK;
}

// ---- search contexts (vdb_index.hpp) ----------------------------------------------------------------------------------
void copy_image_fields(vdb_hip_index* c, const vdb_hip_index* p) {
  c->norms = p->norms;
  c->rows_split = p->rows_split;
  c->split_enabled = p->split_enabled;
  c->split_rows = p->split_rows;
  c->sel_norms = p->sel_norms;
  c->rows_bf16 = p->rows_bf16;
  c->norms_bf16 = p->norms_bf16;
  c->bf16_rho = p->bf16_rho;
  c->bf16_enabled = p->bf16_enabled;
  c->bf16_stride = p->bf16_stride;
  c->bf16_rows = p->bf16_rows;
  c->l2_img = p->l2_img;
  c->l2_rho = p->l2_rho;
  c->l2_seed = p->l2_seed;
  c->l2_rows = p->l2_rows;
  c->cosn_img = p->cosn_img;
  c->cosn_rho = p->cosn_rho;
  c->cosn_rows = p->cosn_rows;
  c->sq8_img = p->sq8_img;
  c->sq8_nrm = p->sq8_nrm;
  c->sq8_rho = p->sq8_rho;
  c->sq8_seed = p->sq8_seed;
  c->sq8_img_rows = p->sq8_img_rows;
  c->bits_img = p->bits_img;
  c->bits_cnt = p->bits_cnt;
  c->bits_img_rows = p->bits_img_rows;
}
// every field a search READS about the index (non-owning views of the device buffers); scratch, stream, events, diagnostics and
// the adaptive selection state stay the context's own
static void copy_data_fields(vdb_hip_index* c, const vdb_hip_index* p) {
  c->device = p->device;
  c->n_cus = p->n_cus;
  c->dim = p->dim;
  c->metric = p->metric;
  c->M = p->M;
  c->M0 = p->M0;
  c->efc = p->efc;
  c->row_stride = p->row_stride;
  c->words = p->words;
  c->capacity = p->capacity;
  c->n_rows = p->n_rows;
  c->rows = p->rows;
  c->bits = p->bits;
  c->alive = p->alive;
  c->ext_ids = p->ext_ids;
  c->sq_min = p->sq_min;
  c->sq_scale = p->sq_scale;
  c->codes = p->codes;
  c->codes_sq = p->codes_sq;
  c->code_words = p->code_words;
  c->quantizer_trained = p->quantizer_trained;
  for (int i = 0; i < VDB_OPT_COUNT_; i++) c->opt[i] = p->opt[i];
  c->storage_mode = p->storage_mode;
  c->sq8_stride = p->sq8_stride;
  c->sq8_codes = p->sq8_codes;
  c->sq8_min = p->sq8_min;
  c->sq8_max = p->sq8_max;
  c->sq8_nsq = p->sq8_nsq;
  c->sign_bits = p->sign_bits;
  c->layers = p->layers;
  c->graph_valid = p->graph_valid;
  c->entry_point = p->entry_point;
  c->max_layer = p->max_layer;
  c->graph_nodes = p->graph_nodes;
  c->live = p->live;
  c->any_dead = p->any_dead;
  c->ndist_valid = p->ndist_valid;
  copy_image_fields(c, p);
}
static thread_local vdb_hip_index* tl_ctx = nullptr;        // the context of this thread's last search ...
static thread_local vdb_hip_index* tl_ctx_owner = nullptr;  // ... the handle it belongs to ...
static thread_local uint64_t tl_ctx_gen = 0;                // ... and that handle's generation: a new handle at a recycled address never matches
static std::atomic<uint64_t> g_generation{0};
// (shared lock on ix->mu held by the caller: clones are only destroyed with the handle)
vdb_hip_index* last_context(vdb_hip_index* ix) {
  if (tl_ctx_owner != ix || tl_ctx_gen != ix->generation || !tl_ctx || tl_ctx == ix) return ix;
  std::lock_guard<std::mutex> pl(ix->pool_mu);
  for (vdb_hip_index* c : ix->ctx_clones)
    if (c == tl_ctx) return c;
  return ix;
}
// what this thread's last LEASED search ran, taken when the lease ends: another thread may lease the same context a moment later,
// and the diagnostic getter of this thread must still describe this thread's search (tests/test_gpu_hardening.py
// ::test_concurrent_searches_on_one_handle_overlap saw the mask of a neighbour's sweep, once, in round 5)
static thread_local bool tl_kernels_valid = false;
static thread_local uint32_t tl_kernels = 0;
uint32_t last_kernels_of_this_thread(vdb_hip_index* ix) {
  if (tl_kernels_valid && tl_ctx_owner == ix && tl_ctx_gen == ix->generation) return tl_kernels;
  return last_context(ix)->last_kernels;
}
void note_last_kernels(uint32_t mask) {  // (behind note_last_context: a search the combining front had another thread run)
  tl_kernels = mask;
  tl_kernels_valid = true;
}
void note_last_context(vdb_hip_index* handle, vdb_hip_index* ctx) {
  tl_ctx = ctx;
  tl_ctx_owner = handle;
  tl_ctx_gen = handle->generation;
  tl_kernels_valid = false;
}

// the primary + up to seven clones.  A context owns scratch and a stream; its graph-walk scratch is sized by the calls it has
// served (ensure_traversal_scratch: one visited bitmap + id log per query in flight), so contexts that serve small calls stay small
constexpr size_t kMaxSearchContexts = 8;
CtxLease::CtxLease(vdb_hip_index* ix) {
  if (ix->ctx_mu.try_lock()) {
    ctx = ix;
  } else {
    std::lock_guard<std::mutex> pl(ix->pool_mu);
    for (vdb_hip_index* c : ix->ctx_clones)
      if (c->ctx_mu.try_lock()) {
        ctx = c;
        break;
      }
    if (!ctx && ix->ctx_clones.size() + 1 < kMaxSearchContexts) {
      std::unique_ptr<vdb_hip_index> c(new vdb_hip_index());
      c->primary = ix;
      c->device = ix->device;
      hipError_t e = hipSetDevice(ix->device);
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_foreign, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_own, hipEventDisableTiming);
      if (e == hipSuccess) {
        c->ctx_mu.lock();
        ctx = c.get();
        ix->ctx_clones.push_back(c.release());
      } else {  // no second context: wait for the primary below
        if (c->stream) (void)hipStreamDestroy(c->stream);
        if (c->ev_foreign) (void)hipEventDestroy(c->ev_foreign);
        if (c->ev_own) (void)hipEventDestroy(c->ev_own);
        (void)hipGetLastError();
      }
    }
  }
  if (!ctx) {  // everything busy: queue for the primary
    ix->ctx_mu.lock();
    ctx = ix;
  }
  if (ctx != ix && ctx->synced_version != ix->version) {
    // the index changed since this context last looked: the change's device work (on the primary's stream) is complete before
    // a search on another stream reads it
    hipError_t e = hipSetDevice(ix->device);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    if (e != hipSuccess) rc = fail(VDB_ERR_HIP, std::string("search context: ") + hipGetErrorString(e));
    copy_data_fields(ctx, ix);
    ctx->synced_version = ix->version;
  }
  note_last_context(ix, ctx);
}
CtxLease::~CtxLease() {
  if (!ctx) return;
  if (tl_ctx == ctx) {
    tl_kernels = ctx->last_kernels;
    tl_kernels_valid = true;
  }
  ctx->ctx_mu.unlock();
}
// frees what a clone owns (scratch, stream, events); its views of the primary's buffers are dropped, not released
static void destroy_clone(vdb_hip_index* c) {
  const int dev = c->device;
  vdb_hip_index blank;
  copy_data_fields(c, &blank);  // every aliased DevBuf -> empty
  c->device = dev;
  c->primary = nullptr;
  destroy_single(c);
}

// ---- capacity management: all per-row arrays grow together ---------------------------------
int32_t ensure_capacity(vdb_hip_index* ix, uint64_t want) {
  if (want <= ix->capacity) return VDB_OK;
  uint64_t ncap = std::max<uint64_t>(want, ix->capacity + ix->capacity / 2);
  ncap = std::max<uint64_t>(ncap, 1024);
  hipStream_t st = ix->stream;
  hipError_t e;
  // + kRowSlack rows: tiled kernels read whole 128-row tiles (rows past n_rows are never reported)
  if ((e = ix->rows.reserve((ncap + kRowSlack) * ix->row_stride * 4, true, st)) != hipSuccess ||
      (e = ix->alive.reserve(ncap, true, st)) != hipSuccess ||
      (e = ix->ext_ids.reserve(ncap * 8, true, st)) != hipSuccess)
    return fail(e == hipErrorOutOfMemory ? VDB_ERR_OOM : VDB_ERR_HIP, std::string("grow: ") + hipGetErrorString(e));
  if ((ix->metric == VDB_COSINE || ix->metric == VDB_EUCLIDEAN || ix->split_enabled || ix->sel_norms) &&
      (e = ix->norms.reserve((ncap + kRowSlack) * 4, true, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("grow norms: ") + hipGetErrorString(e));
  if (ix->split_enabled && (e = ix->rows_split.reserve((ncap + kRowSlack) * (size_t)ix->dim * 4, true, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("grow split rows: ") + hipGetErrorString(e));
  if (is_bits_metric(ix->metric) && (e = ix->bits.reserve(ncap * ix->words * 4, true, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("grow bits: ") + hipGetErrorString(e));
  if (ix->quantizer_trained && ((e = ix->codes.reserve(ncap * ix->code_words * 4, true, st)) != hipSuccess ||
                                (e = ix->codes_sq.reserve(ncap * 4, true, st)) != hipSuccess))
    return fail(VDB_ERR_OOM, std::string("grow codes: ") + hipGetErrorString(e));
  if (ix->bf16_enabled && ((e = ix->rows_bf16.reserve((ncap + kRowSlack) * ix->bf16_stride * 2, true, st)) != hipSuccess ||
                           (e = ix->norms_bf16.reserve((ncap + kRowSlack) * 4, true, st)) != hipSuccess))
    return fail(VDB_ERR_OOM, std::string("grow bf16 rows: ") + hipGetErrorString(e));
  // the lazily built selection images grow HERE (exclusive lock), never inside a search (shared lock: other contexts hold views)
  if (ix->l2_img.cap && (e = ix->l2_img.reserve((ncap + kRowSlack) * (size_t)(ix->dim + 64) * 2, true, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("grow Euclidean selection image: ") + hipGetErrorString(e));
  if (ix->cosn_img.cap && (e = ix->cosn_img.reserve((ncap + kRowSlack) * (size_t)ix->dim * 2, true, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("grow normalised selection image: ") + hipGetErrorString(e));
  // (storage mode Binary keeps its four-bit sign image in sq8_img, bits_image_stride(dim) bytes per row — storage_modes.hip
  // ensure_sign_image; the SQ8 mode its dequantised bf16 image, dim [+ 64] two-byte elements per row)
  const size_t sq8_img_row = ix->storage_mode == VDB_STORAGE_BINARY ? (size_t)bits_image_stride(ix->dim)
                                                                    : (size_t)(ix->dim + (ix->metric == VDB_EUCLIDEAN ? 64 : 0)) * 2;
  if (ix->sq8_img.cap && ((e = ix->sq8_img.reserve((ncap + kRowSlack) * sq8_img_row, true, st)) != hipSuccess ||
                          (e = ix->sq8_nrm.reserve((ncap + kRowSlack) * 4, true, st)) != hipSuccess))
    return fail(VDB_ERR_OOM, std::string("grow SQ8 selection image: ") + hipGetErrorString(e));
  if (ix->bits_img.cap && ((e = ix->bits_img.reserve((ncap + kRowSlack) * (size_t)bits_image_stride(ix->dim), true, st)) != hipSuccess ||
                           (e = ix->bits_cnt.reserve((ncap + kRowSlack) * 4, true, st)) != hipSuccess))
    return fail(VDB_ERR_OOM, std::string("grow four-bit row image: ") + hipGetErrorString(e));
  for (auto& L : ix->layers) {
    if ((e = L.nbr.reserve(ncap * L.stride * 4, true, st)) != hipSuccess ||
        (e = L.cnt.reserve(ncap * 4, true, st)) != hipSuccess ||
        (ix->ndist_valid && (e = L.ndist.reserve(ncap * L.stride * 4, true, st)) != hipSuccess))
      return fail(VDB_ERR_OOM, std::string("grow graph: ") + hipGetErrorString(e));
    // rows beyond the old capacity have no links yet (Layer::ensure_capacity, layer.rs:26-30)
    if ((e = hipMemsetAsync(L.cnt.as<uint32_t>() + ix->capacity, 0, (ncap - ix->capacity) * 4, st)) != hipSuccess)
      return fail(VDB_ERR_HIP, std::string("grow graph: ") + hipGetErrorString(e));
  }
  ix->capacity = ncap;
  mark_changed(ix);
  return VDB_OK;
}

// appends n rows that are already laid out with row_stride on the device at rows[n_rows..]
static int32_t finish_append(vdb_hip_index* ix, uint64_t first, uint64_t n) {
  PrepArgs pa{};
  pa.rows = ix->rows.as<float>();
  // cosine: the kernels divide by them; Euclidean: |v|^2 of the matrix-core batch path (sweep_topk_gemm_f32<kEuclidean>)
  pa.norms = (ix->metric == VDB_COSINE || ix->metric == VDB_EUCLIDEAN || ix->split_enabled || ix->sel_norms) ? ix->norms.as<float>() : nullptr;
  pa.bits = is_bits_metric(ix->metric) ? ix->bits.as<uint32_t>() : nullptr;
  pa.row_stride = ix->row_stride;
  pa.row0 = (uint32_t)first;
  pa.n_rows = (uint32_t)n;
  pa.dim = ix->dim;
  pa.words = ix->words;
  if (pa.norms || pa.bits) launch_prep_rows(pa, ix->stream);
  if (ix->quantizer_trained) {
    int32_t rq = quantize_rows(ix, first, n);
    if (rq != VDB_OK) return rq;
  }
  if (ix->bf16_enabled) {
    launch_prep_bf16(ix->rows.as<float>(), ix->row_stride, ix->rows_bf16.as<uint16_t>(), ix->bf16_stride,
                     ix->norms_bf16.as<float>(), (uint32_t)first, (uint32_t)n, ix->dim, ix->stream, ix->bf16_rho.as<uint32_t>());
    ix->bf16_rows = first + n;
  }
  if (ix->split_enabled) {
    launch_split_vectors(ix->rows.as<float>(), ix->row_stride, ix->rows_split.as<uint16_t>(), nullptr, (uint32_t)first,
                         (uint32_t)n, ix->dim, ix->stream);
    ix->split_rows = first + n;
  }
  if (ix->bits_img.cap) {  // (behind prep_rows on the same stream: the packed bits of the new rows exist)
    launch_bits_expand(ix->metric, ix->bits.as<uint32_t>(), ix->words, ix->bits_img.as<uint8_t>(), bits_image_stride(ix->dim), ix->bits_cnt.as<float>(),
                       (uint32_t)first, (uint32_t)n, ix->dim, 0.0f, ix->stream);
    ix->bits_img_rows = first + n;
  }
  if (ix->storage_mode != VDB_STORAGE_FULL) {  // crud.rs:66-82: the quantised code is built with every upsert
    int32_t rs = storage_mode_append(ix, first, n);
    if (rs != VDB_OK) return rs;
  }
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}

// registers ids (duplicates skipped), copies rows H2D; returns the list of accepted source rows
int32_t append_host_rows(vdb_hip_index* ix, const uint64_t* ids, const float* vecs, uint64_t n,
                                uint64_t* inserted, uint64_t* first_row) {
  std::vector<uint64_t> src;
  src.reserve(n);
  std::vector<uint64_t> new_ids;
  for (uint64_t i = 0; i < n; i++) {
    if (ix->id_to_idx.count(ids[i])) continue;  // trait_impl.rs:23-25: duplicate id => skipped
    // duplicates inside the batch itself: first one wins
    ix->id_to_idx[ids[i]] = ix->n_rows + src.size();
    src.push_back(i);
    new_ids.push_back(ids[i]);
  }
  const uint64_t m = src.size();
  *first_row = ix->n_rows;
  if (inserted) *inserted = m;
  if (m == 0) return VDB_OK;
  if (ix->n_rows + m > kMaxRowsPerIndex) {
    for (uint64_t id : new_ids) ix->id_to_idx.erase(id);
    return fail(VDB_ERR_UNSUPPORTED, "more than 2^32-512 rows per index");
  }
  int32_t rc = ensure_capacity(ix, ix->n_rows + m);
  if (rc != VDB_OK) {
    for (uint64_t id : new_ids) ix->id_to_idx.erase(id);
    return rc;
  }
  float* drows = ix->rows.as<float>() + ix->n_rows * ix->row_stride;
  const bool contiguous = (m == n);
  hipError_t e = hipSuccess;
  if (contiguous) {
    e = hipMemcpy2DAsync(drows, ix->row_stride * 4, vecs, (size_t)ix->dim * 4, (size_t)ix->dim * 4, m,
                         hipMemcpyHostToDevice, ix->stream);
  } else {
    for (uint64_t j = 0; j < m && e == hipSuccess; j++)
      e = hipMemcpyAsync(drows + j * ix->row_stride, vecs + src[j] * ix->dim, (size_t)ix->dim * 4,
                         hipMemcpyHostToDevice, ix->stream);
  }
  if (e == hipSuccess && ix->row_stride != ix->dim) {
    // zero the padding floats so the storage is fully defined (never enters a chain)
    e = hipMemset2DAsync(drows + ix->dim, ix->row_stride * 4, 0, (ix->row_stride - ix->dim) * 4, m, ix->stream);
  }
  if (e == hipSuccess)
    e = hipMemcpyAsync(ix->ext_ids.as<uint64_t>() + ix->n_rows, new_ids.data(), m * 8, hipMemcpyHostToDevice,
                       ix->stream);
  if (e == hipSuccess) e = hipMemsetAsync(ix->alive.as<uint8_t>() + ix->n_rows, 1, m, ix->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);  // new_ids / vecs are caller memory
  if (e != hipSuccess) {
    for (uint64_t id : new_ids) ix->id_to_idx.erase(id);
    return fail(VDB_ERR_HIP, std::string("upload: ") + hipGetErrorString(e));
  }
  ix->idx_to_id.insert(ix->idx_to_id.end(), new_ids.begin(), new_ids.end());
  ix->idx_live.insert(ix->idx_live.end(), m, 1);
  ix->live += m;
  const uint64_t first = ix->n_rows;
  ix->n_rows += m;
  return finish_append(ix, first, m);
}

// (pointers into the pools stay valid: their capacity is reserved once)
EventPair* next_sel_events(vdb_hip_index* ix) {
  if (!opt_timing(ix)) return nullptr;
  if (ix->sel_ev.capacity() < 64) ix->sel_ev.reserve(64);
  if (ix->sel_ev_used == ix->sel_ev.size()) {
    if (ix->sel_ev.size() >= 64) return nullptr;
    EventPair p;
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
    ix->sel_ev.push_back(p);
  }
  return &ix->sel_ev[ix->sel_ev_used++];
}
EventPair* next_events(vdb_hip_index* ix) {
  if (!opt_timing(ix)) return nullptr;
  if (ix->ev_pool.capacity() < 8192) ix->ev_pool.reserve(8192);
  if (ix->ev_used == ix->ev_pool.size()) {
    if (ix->ev_pool.size() >= 8192) return nullptr;
    EventPair p;
    if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return nullptr;
    ix->ev_pool.push_back(p);
  }
  return &ix->ev_pool[ix->ev_used++];
}

static uint32_t pick_B(const vdb_hip_index* ix, uint32_t nq) {
  uint32_t b = nq >= 8 ? 8 : (nq >= 4 ? 4 : (nq >= 2 ? 2 : 1));
  return std::min(b, std::min<uint32_t>(opt_max_tile(ix), 8));
}
int blocks_for(const vdb_hip_index* ix, int B, uint32_t ngroups) {
  const int occ = (B == 1) ? 4 : (B == 8 ? 2 : 3);  // resident 256-thread blocks per CU (VGPR-limited)
  int64_t want = ((int64_t)ngroups + 3) / 4;
  int64_t cap = (int64_t)ix->n_cus * occ;
  return (int)std::max<int64_t>(1, std::min(want, cap));
}

// HnswIndex::search_brute_force (search.rs:176-219) for nq device-resident queries.
// Outputs are device buffers; nothing synchronises.
static int32_t brute_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k,
                         uint64_t* d_ids, float* d_scores, uint32_t* d_n, hipStream_t st) {
  if (nq == 0) return VDB_OK;
  if (k == 0 || ix->n_rows == 0) {
    VDB_HIP(hipMemsetAsync(d_n, 0, (size_t)nq * 4, st));
    return VDB_OK;
  }
  const bool hib = higher_is_better_host(ix->metric);
  const uint8_t* alive = ix->any_dead ? ix->alive.as<uint8_t>() : nullptr;
  if (is_bits_metric(ix->metric)) {
    if ((size_t)4 * k * 8 + (size_t)ix->words * 4 + 32 > 60 * 1024)
      return fail(VDB_ERR_UNSUPPORTED, "k too large for the fused top-k path");
    hipError_t e;
    // one or two queries: ONE launch — the blocks pack the query themselves, and the one that finishes last merges (sweep.hip)
    if (opt_bits_fused(ix) && sweep_bits_fused_supported(ix->words, nq, k)) {
      const int fblocks = sweep_bits_fused_blocks(ix->n_rows, ix->n_cus);
      if (ix->s_tickets.cap == 0) {
        if ((e = ix->s_tickets.reserve(16, false, st)) != hipSuccess) return fail(VDB_ERR_OOM, "ticket scratch");
        VDB_HIP(hipMemsetAsync(ix->s_tickets.p, 0, 16, st));
      }
      if ((e = ix->s_part_keys.reserve((size_t)nq * fblocks * k * 8, false, st)) != hipSuccess) return fail(VDB_ERR_OOM, "top-k scratch");
      BitsFusedArgs fa{};
      fa.bits = ix->bits.as<uint32_t>();
      fa.q = d_q;
      fa.q_stride = q_stride;
      fa.alive = alive;
      fa.part_keys = ix->s_part_keys.as<uint64_t>();
      fa.tickets = ix->s_tickets.as<uint32_t>();
      fa.n_rows = (uint32_t)ix->n_rows;
      fa.words = ix->words;
      fa.dim = ix->dim;
      fa.k = k;
      fa.m.ext_ids = ix->ext_ids.as<uint64_t>();
      fa.m.out_ids = d_ids;
      fa.m.out_scores = d_scores;
      fa.m.out_n = d_n;
      ix->last_kernels |= VDB_KERNEL_BITS;
      EventPair* evf = next_events(ix);
      if (evf) (void)hipEventRecord(evf->a, st);
      const hipError_t ef = launch_sweep_bits_fused(ix->metric, fa, fblocks, nq, st);
      if (ef != hipSuccess) return fail(VDB_ERR_HIP, std::string("one-launch packed-bit search: ") + hipGetErrorString(ef));
      if (evf) (void)hipEventRecord(evf->b, st);
      return VDB_OK;
    }
    // pack the queries with the same kernel that packs rows
    e = ix->s_qbits.reserve((size_t)nq * ix->words * 4, false, st);
    if (e != hipSuccess) return fail(VDB_ERR_OOM, "qbits scratch");
    PrepArgs pa{};
    pa.rows = d_q;
    pa.bits = ix->s_qbits.as<uint32_t>();
    pa.row_stride = q_stride;
    pa.n_rows = nq;
    pa.dim = ix->dim;
    pa.words = ix->words;
    launch_prep_rows(pa, st);
    // large batches: the dot products as a four-bit GEMM on the matrix cores, exact (bits_gemm.hip); what is left of the batch
    // (and every other shape) keeps the vector-ALU kernels below — the same keys either way
    uint32_t qdone = 0;
    while (opt_engine(ix) == 1 && opt_max_tile(ix) >= 128) {
      const uint32_t nqg = bits_gemm_chunk(ix, nq - qdone, k);
      if (!nqg) break;
      int32_t rg = ensure_bits_image(ix, st);
      if (rg == VDB_OK)
        rg = brute_bits_gemm_dev(ix, ix->metric, ix->bits_img.as<uint8_t>(), ix->bits_cnt.as<float>(), ix->s_qbits.as<uint32_t>() + (size_t)qdone * ix->words,
                                 nqg, k, d_ids + (size_t)qdone * k, d_scores + (size_t)qdone * k, d_n + qdone, st);
      if (rg != VDB_OK) return rg;
      qdone += nqg;
    }
    if (qdone == nq) return VDB_OK;
    d_ids += (size_t)qdone * k;
    d_scores += (size_t)qdone * k;
    d_n += qdone;
    nq -= qdone;
    const uint32_t* qbits_rest = ix->s_qbits.as<uint32_t>() + (size_t)qdone * ix->words;
    const BitsPlan bp = plan_bits_sweep(ix->n_rows, ix->n_cus, ix->words, nq, k);
    const int blocks = bp.blocks;
    const uint32_t nw = (uint32_t)blocks;  // one list per block
    if ((e = ix->s_part_keys.reserve((size_t)nq * nw * k * 8, false, st)) != hipSuccess ||
        (e = ix->s_part_cnt.reserve((size_t)nq * nw * 4, false, st)) != hipSuccess)
      return fail(VDB_ERR_OOM, "top-k scratch");
    BitsArgs ba{};
    ba.bits = ix->bits.as<uint32_t>();
    ba.qbits = qbits_rest;
    ba.alive = alive;
    ba.part_keys = ix->s_part_keys.as<uint64_t>();
    ba.part_cnt = ix->s_part_cnt.as<uint32_t>();
    ba.n_rows = (uint32_t)ix->n_rows;
    ba.words = ix->words;
    ba.k = k;
    ix->last_kernels |= VDB_KERNEL_BITS;
    EventPair* ev = next_events(ix);
    if (ev) (void)hipEventRecord(ev->a, st);
    {
      hipError_t eb = launch_bits_plan(ix->metric, bp, ba, nq, st);
      if (eb != hipSuccess) return fail(VDB_ERR_HIP, std::string("packed-bit sweep launch: ") + hipGetErrorString(eb));
    }
    if (ev) (void)hipEventRecord(ev->b, st);
    MergeArgs m{};
    m.part_keys = ba.part_keys;
    m.part_cnt = ba.part_cnt;
    m.ext_ids = ix->ext_ids.as<uint64_t>();
    m.out_ids = d_ids;
    m.out_scores = d_scores;
    m.out_n = d_n;
    m.n_lists = nw;
    m.k = k;
    launch_merge(hib, m, nq, st);
    VDB_HIP(hipGetLastError());
    return VDB_OK;
  }
  uint32_t euclid_skip_until = 0;  // Euclidean chunk without proofs: [q0, this) goes through the exact tiles
  for (uint32_t q0 = 0; q0 < nq;) {
    uint32_t B = pick_B(ix, nq - q0);
    const int cpl = sweep_cpl_for_dim(ix->dim);
    // matrix-core engine (cosine / dot): one or two 16-query tiles per corpus pass
    int mfma_nqt = 0;
    if (opt_engine(ix) == 1 && (ix->metric == VDB_COSINE || ix->metric == VDB_DOT)) {
      int want = (nq - q0 > 32 && opt_max_tile(ix) >= 48) ? 3 : ((nq - q0 > 16 && opt_max_tile(ix) >= 32) ? 2 : 1);
      for (; want >= 1; want--)
        if (sweep_mfma_lds_bytes(want, k, ix->dim) <= 160 * 1024) break;
      mfma_nqt = want;  // 0: does not fit the LDS (very large dim or k): VALU kernels
    }
    // large Cosine / DotProduct batches over a large corpus: split-bf16 selection + exact re-scoring + proof (same bits
    // as the exact matrix-core kernel below, which remains the fallback for unproven queries and every other shape)
    // ... and with 10 < k <= 128: the WIDE selection (sweep_wide.hip) — no block-local lists, every row above the bound is a candidate
    if ((mfma_nqt || ix->metric == VDB_EUCLIDEAN) && q0 >= euclid_skip_until && select_level_wide(ix, nq - q0, k)) {
      const uint32_t nqg = select_chunk(nq - q0);
      const int32_t rcw = brute_wide_dev(ix, d_q + (size_t)q0 * q_stride, q_stride, nqg, k, d_ids + (size_t)q0 * k, d_scores + (size_t)q0 * k, d_n + q0, st);
      if (rcw != VDB_OK) return rcw;
      q0 += nqg;
      continue;
    }
    const int sel_level = mfma_nqt ? select_level(ix, nq - q0, k) : 0;
    if (sel_level) {
      const uint32_t nqg = select_chunk(nq - q0);
      const int32_t rcs = brute_split_dev(ix, d_q + (size_t)q0 * q_stride, q_stride, nqg, k, d_ids + (size_t)q0 * k,
                                          d_scores + (size_t)q0 * k, d_n + q0, st, sel_level);
      if (rcs != VDB_OK) return rcs;
      q0 += nqg;
      continue;
    }
    // Euclidean batches, first choice: the selection stage of the Cosine / DotProduct batches over the augmented form
    // s = q.v - |v|^2 / 2 (bf16 matrix pipe, canonical re-scoring, proof, unproven queries gathered on the device — no host
    // synchronisation); the f32 matrix-core path below remains for the shapes it does not take and for handles it parks
    if (ix->metric == VDB_EUCLIDEAN && q0 >= euclid_skip_until && select_level_l2(ix, nq - q0, k)) {
      const uint32_t nqg = select_chunk(nq - q0);
      const int32_t rcs = brute_split_dev(ix, d_q + (size_t)q0 * q_stride, q_stride, nqg, k, d_ids + (size_t)q0 * k,
                                          d_scores + (size_t)q0 * k, d_n + q0, st, 2);
      if (rcs != VDB_OK) return rcs;
      q0 += nqg;
      continue;
    }
    // Euclidean batches: approximate selection of k + slack candidates on the matrix cores, canonical re-scoring, proof
    // of exactness per query; the (rare) unproven queries go through the exact vector-ALU sweep below
    if (ix->metric == VDB_EUCLIDEAN && opt_engine(ix) == 1 && opt_max_tile(ix) >= 128 && nq - q0 >= kGemmMinQueries &&
        k + kEuclidSlack <= kGemmMaxK && q0 >= euclid_skip_until) {
      const uint32_t nqg = std::min<uint32_t>(nq - q0, kGemmMaxQueries);
      // k <= 10: 16 candidates — the 32-entry candidate buffers then leave room for two blocks per CU; the per-query
      // verdict catches the (rare) query whose near-ties are wider than the slack
      const uint32_t kp = k + 6 <= 16 ? 16 : k + kEuclidSlack;
      GemmPlan gp;
      sweep_gemm_plan(nqg, (uint32_t)ix->n_rows, ix->n_cus, kp, &gp);
      if (gp.lds <= 160 * 1024) {
        hipError_t e3;
        // scratch: norm max (16 B) | flags [nqg] | cand n [nqg] | cand approx [nqg][kp] | cand rows [nqg][kp] (8-B aligned)
        const size_t off_flags = 16, off_n = off_flags + (size_t)nqg * 4, off_ap = off_n + (size_t)nqg * 4;
        const size_t off_rows = (off_ap + (size_t)nqg * kp * 4 + 15) & ~(size_t)15;
        if ((e3 = ix->s_part_keys.reserve((size_t)nqg * gp.G * kp * 8, false, st)) != hipSuccess ||
            (e3 = ix->s_misc.reserve(off_rows + (size_t)nqg * kp * 8, false, st)) != hipSuccess)
          return fail(VDB_ERR_OOM, "top-k scratch");
        unsigned char* sc = ix->s_misc.as<unsigned char>();
        SweepArgs ag{};
        ag.rows = ix->rows.as<float>();
        ag.norms = ix->norms.as<float>();
        ag.alive = alive;
        ag.queries = d_q + (size_t)q0 * q_stride;
        ag.part_keys = ix->s_part_keys.as<uint64_t>();
        ag.row_stride = ix->row_stride;
        ag.q_stride = q_stride;
        ag.n_rows = (uint32_t)ix->n_rows;
        ag.dim = ix->dim;
        ag.nq = nqg;
        ag.k = kp;
        EventPair* evg = next_events(ix);
        if (evg) (void)hipEventRecord(evg->a, st);
        ix->last_kernels |= VDB_KERNEL_GEMM_F32;
        e3 = launch_sweep_gemm(VDB_EUCLIDEAN, gp, ag, st);
        if (evg) (void)hipEventRecord(evg->b, st);
        if (e3 != hipSuccess) return fail(VDB_ERR_HIP, std::string("gemm sweep launch: ") + hipGetErrorString(e3));
        MergeArgs mg{};
        mg.part_keys = ag.part_keys;
        mg.ext_ids = nullptr;  // internal rows
        mg.out_ids = reinterpret_cast<uint64_t*>(sc + off_rows);
        mg.out_scores = reinterpret_cast<float*>(sc + off_ap);
        mg.out_n = reinterpret_cast<uint32_t*>(sc + off_n);
        mg.n_lists = gp.G;
        mg.k = kp;
        launch_merge(false, mg, nqg, st);
        EuclidRerankArgs ra{};
        ra.rows = ag.rows;
        ra.queries = ag.queries;
        ra.cand_rows = mg.out_ids;
        ra.cand_approx = mg.out_scores;
        ra.cand_n = mg.out_n;
        ra.ext_ids = ix->ext_ids.as<uint64_t>();
        ra.norm_max_bits = reinterpret_cast<const uint32_t*>(sc);
        ra.out_ids = d_ids + (size_t)q0 * k;
        ra.out_scores = d_scores + (size_t)q0 * k;
        ra.out_n = d_n + q0;
        ra.flags = reinterpret_cast<uint32_t*>(sc + off_flags);
        ra.row_stride = ix->row_stride;
        ra.q_stride = q_stride;
        ra.dim = ix->dim;
        ra.k = k;
        ra.kp = kp;
        launch_euclid_rerank(ra, ag.norms, ag.n_rows, nqg, st);
        VDB_HIP(hipGetLastError());
        std::vector<uint32_t> flags(nqg);
        VDB_HIP(hipMemcpyAsync(flags.data(), ra.flags, (size_t)nqg * 4, hipMemcpyDeviceToHost, st));
        VDB_HIP(hipStreamSynchronize(st));
        uint32_t n_flagged = 0;
        for (uint32_t i = 0; i < nqg; i++) n_flagged += flags[i] ? 1u : 0u;
        if (n_flagged > nqg / 8) {
          // tie-heavy data (duplicates, low-cardinality values): most queries lack a proof — the whole chunk goes through
          // the exact vector-ALU tiles below instead of one sweep per query
          ix->euclid_fallbacks += nqg;
          euclid_skip_until = q0 + nqg;
          continue;
        }
        for (uint32_t i = 0; i < nqg; i++) {
          if (!flags[i]) continue;
          const int32_t rc1 = brute_dev(ix, d_q + (size_t)(q0 + i) * q_stride, q_stride, 1, k, d_ids + (size_t)(q0 + i) * k,
                                        d_scores + (size_t)(q0 + i) * k, d_n + q0 + i, st);
          if (rc1 != VDB_OK) return rc1;
          ix->euclid_fallbacks++;
        }
        q0 += nqg;
        continue;
      }
    }
    // large batches: GEMM-structured matrix-core kernel, the whole batch in one launch (sweep_gemm.hip)
    if (mfma_nqt && opt_max_tile(ix) >= 128 && nq - q0 >= kGemmMinQueries && k <= kGemmMaxK) {
      const uint32_t nqg = std::min<uint32_t>(nq - q0, kGemmMaxQueries);
      GemmPlan gp;
      sweep_gemm_plan(nqg, (uint32_t)ix->n_rows, ix->n_cus, k, &gp);
      if (gp.lds <= 160 * 1024) {
        hipError_t e3;
        if ((e3 = ix->s_part_keys.reserve((size_t)nqg * gp.G * k * 8, false, st)) != hipSuccess)
          return fail(VDB_ERR_OOM, "top-k scratch");
        SweepArgs ag{};
        ag.rows = ix->rows.as<float>();
        ag.norms = ix->norms.as<float>();
        ag.alive = alive;
        ag.queries = d_q + (size_t)q0 * q_stride;
        ag.part_keys = ix->s_part_keys.as<uint64_t>();
        ag.row_stride = ix->row_stride;
        ag.q_stride = q_stride;
        ag.n_rows = (uint32_t)ix->n_rows;
        ag.dim = ix->dim;
        ag.nq = nqg;
        ag.k = k;
        EventPair* evg = next_events(ix);
        if (evg) (void)hipEventRecord(evg->a, st);
        ix->last_kernels |= VDB_KERNEL_GEMM_F32;
        e3 = launch_sweep_gemm(ix->metric, gp, ag, st);
        if (evg) (void)hipEventRecord(evg->b, st);
        if (e3 != hipSuccess) return fail(VDB_ERR_HIP, std::string("gemm sweep launch: ") + hipGetErrorString(e3));
        MergeArgs mg{};
        mg.part_keys = ag.part_keys;
        mg.ext_ids = ix->ext_ids.as<uint64_t>();
        mg.out_ids = d_ids + (size_t)q0 * k;
        mg.out_scores = d_scores + (size_t)q0 * k;
        mg.out_n = d_n + q0;
        mg.n_lists = gp.G;
        mg.k = k;
        launch_merge(hib, mg, nqg, st);
        q0 += nqg;
        continue;
      }
    }
    if (mfma_nqt) {
      const uint32_t Bm = (uint32_t)mfma_nqt * 16;
      const uint32_t tile_m = std::min<uint32_t>(Bm, nq - q0);
      const int waves = mfma_nqt >= 2 ? kMfmaWaves2 : kMfmaWaves1;
      const size_t lds = sweep_mfma_lds_bytes(mfma_nqt, k, ix->dim);
      const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds, (size_t)(16 / waves)));
      const uint32_t ntiles = (uint32_t)((ix->n_rows + 15) / 16);
      const int blocks_m = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ntiles + waves - 1) / waves,
                                                                     (int64_t)ix->n_cus * per_cu));
      hipError_t e2;
      if ((e2 = ix->s_part_keys.reserve((size_t)Bm * blocks_m * k * 8, false, st)) != hipSuccess)
        return fail(VDB_ERR_OOM, "top-k scratch");
      SweepArgs am{};
      am.rows = ix->rows.as<float>();
      am.norms = ix->norms.as<float>();
      am.alive = alive;
      am.queries = d_q + (size_t)q0 * q_stride;
      am.part_keys = ix->s_part_keys.as<uint64_t>();
      am.row_stride = ix->row_stride;
      am.q_stride = q_stride;
      am.n_rows = (uint32_t)ix->n_rows;
      am.dim = ix->dim;
      am.nq = tile_m;
      am.k = k;
      EventPair* evm = next_events(ix);
      if (evm) (void)hipEventRecord(evm->a, st);
      ix->last_kernels |= VDB_KERNEL_SWEEP_MFMA_F32;
      e2 = launch_sweep_mfma(ix->metric, mfma_nqt, am, blocks_m, st);
      if (evm) (void)hipEventRecord(evm->b, st);
      if (e2 != hipSuccess) return fail(VDB_ERR_HIP, std::string("mfma sweep launch: ") + hipGetErrorString(e2));
      MergeArgs mm{};
      mm.part_keys = am.part_keys;
      mm.ext_ids = ix->ext_ids.as<uint64_t>();
      mm.out_ids = d_ids + (size_t)q0 * k;
      mm.out_scores = d_scores + (size_t)q0 * k;
      mm.out_n = d_n + q0;
      mm.n_lists = (uint32_t)blocks_m;
      mm.k = k;
      launch_merge(hib, mm, tile_m, st);
      q0 += tile_m;
      continue;
    }
    // large tiles: queries in LDS, 16 or 32 per corpus pass (dims that are a multiple of 256, <= 1024)
    bool qlds = false;
    const uint32_t max_tile = opt_max_tile(ix);
    if (cpl > 0 && nq - q0 >= 12 && max_tile >= 16) {
      const uint32_t want = (nq - q0 >= 24 && max_tile >= 32) ? 32 : 16;
      for (uint32_t b = want; b >= 16; b /= 2) {
        const int waves = b == 32 ? kQldsWaves32 : kQldsWaves16;
        if (sweep_qlds_lds_bytes((int)b, k, ix->dim, waves) <= 160 * 1024) {
          B = b;
          qlds = true;
          break;
        }
      }
    }
    // generic dims keep the query tile in LDS: shrink the tile until it fits the default 64 KiB window
    while (!qlds && B > 1 && sweep_lds_bytes((int)B, k, ix->dim, cpl) > 60 * 1024) B /= 2;
    const uint32_t tile = std::min<uint32_t>(B, nq - q0);
    if (!qlds && sweep_lds_bytes((int)B, k, ix->dim, cpl) > 60 * 1024)
      return fail(VDB_ERR_UNSUPPORTED, "k (x dim) too large for the fused top-k path");
    const uint32_t rpg = 64 / B;
    const uint32_t ngroups = (uint32_t)((ix->n_rows + rpg - 1) / rpg);
    int blocks;
    if (qlds) {
      const int waves = B == 32 ? kQldsWaves32 : kQldsWaves16;
      const size_t lds = sweep_qlds_lds_bytes((int)B, k, ix->dim, waves);
      const int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / lds, (size_t)(B == 32 ? 1 : 3)));
      blocks = (int)std::max<int64_t>(1, std::min<int64_t>(((int64_t)ngroups + waves - 1) / waves,
                                                           (int64_t)ix->n_cus * per_cu));
    } else {
      blocks = blocks_for(ix, (int)B, ngroups);
    }
    const uint32_t nw = (uint32_t)blocks;  // one list per block
    hipError_t e;
    if ((e = ix->s_part_keys.reserve((size_t)B * nw * k * 8, false, st)) != hipSuccess ||
        (e = ix->s_part_cnt.reserve((size_t)B * nw * 4, false, st)) != hipSuccess)
      return fail(VDB_ERR_OOM, "top-k scratch");
    SweepArgs a{};
    a.rows = ix->rows.as<float>();
    a.norms = ix->norms.as<float>();
    a.alive = alive;
    a.queries = d_q + (size_t)q0 * q_stride;
    a.part_keys = ix->s_part_keys.as<uint64_t>();
    a.part_cnt = ix->s_part_cnt.as<uint32_t>();
    a.row_stride = ix->row_stride;
    a.q_stride = q_stride;
    a.n_rows = (uint32_t)ix->n_rows;
    a.dim = ix->dim;
    a.nq = tile;
    a.k = k;
    ix->last_kernels |= VDB_KERNEL_SWEEP_VALU;
    EventPair* ev = next_events(ix);
    if (ev) (void)hipEventRecord(ev->a, st);
    if (qlds) {
      hipError_t le = launch_sweep_f32_qlds(ix->metric, (int)B, a, blocks, st);
      if (le != hipSuccess) return fail(VDB_ERR_HIP, std::string("sweep launch: ") + hipGetErrorString(le));
    } else {
      launch_sweep_f32(ix->metric, (int)B, a, blocks, st);
    }
    if (ev) (void)hipEventRecord(ev->b, st);
    MergeArgs m{};
    m.part_keys = a.part_keys;
    m.part_cnt = a.part_cnt;
    m.ext_ids = ix->ext_ids.as<uint64_t>();
    m.out_ids = d_ids + (size_t)q0 * k;
    m.out_scores = d_scores + (size_t)q0 * k;
    m.out_n = d_n + q0;
    m.n_lists = nw;
    m.k = k;
    launch_merge(hib, m, tile, st);
    q0 += tile;
  }
  VDB_HIP(hipGetLastError());
  return VDB_OK;
}

static uint32_t balanced_ef(uint32_t k) { return std::max<uint32_t>(128, k * 4); }  // params.rs:313

// dispatch of search_with_quality (search.rs:59-94) for device-resident queries
int32_t search_dev(vdb_hip_index* ix, const float* d_q, uint64_t q_stride, uint32_t nq, uint32_t k, uint32_t ef,
                   int32_t mode, uint64_t* d_ids, float* d_scores, uint32_t* d_n, hipStream_t st, uint32_t cap_mult,
                   bool* used_hnsw, uint32_t rerank_k, const uint32_t* d_extra_eps) {
  if (used_hnsw) *used_hnsw = false;
  ix->ev_used = 0;
  ix->sel_ev_used = 0;
  // diagnostics describe THIS call: brute_split_dev / the sweep paths overwrite them when they run
  ix->last_select_level = 0;
  ix->split_flags_n = 0;
  ix->last_kernels = 0;
  if (ix->n_rows == 0) {  // empty index: no entry point => empty result (native/graph.rs:252-255)
    if (nq) VDB_HIP(hipMemsetAsync(d_n, 0, (size_t)nq * 4, st));
    return VDB_OK;
  }
  if (mode == VDB_SEARCH_BRUTE) return brute_dev(ix, d_q, q_stride, nq, k, d_ids, d_scores, d_n, st);
  if (mode == VDB_SEARCH_BRUTE_BF16) return brute_bf16_dev(ix, d_q, q_stride, nq, k, d_ids, d_scores, d_n, st);
  if (mode == VDB_SEARCH_BRUTE_SQ8) {
    // large Cosine / DotProduct batches: bf16 selection over the dequantised rows + the reference chain for the candidates +
    // proof (level 3); everything else, and what a handle's data defeats, on the exact SQ8 sweep
    if (ix->storage_mode != VDB_STORAGE_SQ8) return fail(VDB_ERR_STATE, "SQ8 search: set the storage mode to SQ8 first");
    uint32_t q0 = 0;
    while (q0 < nq && k > 0 && ix->n_rows > 0) {
      if (select_level_wide(ix, nq - q0, k, /*sq8=*/true)) {  // 10 < k <= 128: the WIDE selection over the dequantised image (sweep_wide.hip)
        const uint32_t nqw = select_chunk(nq - q0, kSelectMinQueriesSq8);
        const int32_t rcw = brute_wide_dev(ix, d_q + (size_t)q0 * q_stride, q_stride, nqw, k, d_ids + (size_t)q0 * k, d_scores + (size_t)q0 * k, d_n + q0, st, true);
        if (rcw != VDB_OK) return rcw;
        q0 += nqw;
        continue;
      }
      if (select_level_sq8(ix, nq - q0, k) != 3) break;
      const uint32_t nqg = select_chunk(nq - q0, kSelectMinQueriesSq8);
      const int32_t rcs = brute_split_dev(ix, d_q + (size_t)q0 * q_stride, q_stride, nqg, k, d_ids + (size_t)q0 * k,
                                          d_scores + (size_t)q0 * k, d_n + q0, st, 3);
      if (rcs != VDB_OK) return rcs;
      q0 += nqg;
    }
    if (q0 == nq) return VDB_OK;
    ix->last_kernels |= VDB_KERNEL_SQ8;
    return brute_sq8_dev(ix, d_q + (size_t)q0 * q_stride, q_stride, nq - q0, k, d_ids + (size_t)q0 * k, d_scores + (size_t)q0 * k,
                         d_n + q0, st);
  }
  if (mode == VDB_SEARCH_BRUTE_BINARY) ix->last_kernels |= VDB_KERNEL_BITS;
  if (mode == VDB_SEARCH_BRUTE_BINARY) return brute_binary_dev(ix, d_q, q_stride, nq, k, d_ids, d_scores, d_n, st);
  if (mode == VDB_SEARCH_AUTO && ix->live <= 100 && ix->n_rows > 0)  // search.rs:75-77
    return brute_dev(ix, d_q, q_stride, nq, k, d_ids, d_scores, d_n, st);
  if (mode == VDB_SEARCH_HNSW_INT8) {
    if (used_hnsw) *used_hnsw = true;
    ix->last_kernels |= VDB_KERNEL_HNSW_INT8;
    // (rerank_k in this mode: the call's own DualPrecisionConfig::oversampling_ratio, vdb_hip_index_search_with_config; 0 = the handle's option)
    return hnsw_search_int8_dev(ix, d_q, q_stride, nq, k, ef == 0 ? balanced_ef(k) : ef, rerank_k ? rerank_k : opt_oversampling(ix), cap_mult,
                                d_ids, d_scores, d_n, st);
  }
  if (mode != VDB_SEARCH_AUTO && mode != VDB_SEARCH_HNSW) return fail(VDB_ERR_INVALID_ARG, "bad search mode");
  if (rerank_k) {
    // search_with_rerank(_quality): the candidate search runs with k = rerank_k (search.rs:124,310)
    if (ef == 0) ef = std::max<uint32_t>(512, rerank_k * 16);  // SearchQuality::Accurate, params.rs:314
    ef = std::max(ef, rerank_k);
  } else {
    if (ef == 0 && !ix->raw_ef) ef = balanced_ef(k);
    if (!ix->raw_ef) ef = std::max(ef, k);  // SearchQuality::Custom(ef) = max(ef, k), params.rs:317 (NativeHnsw-level calls pass theirs as is)
  }
  if (used_hnsw) *used_hnsw = true;
  ix->last_kernels |= VDB_KERNEL_HNSW;
  return hnsw_search_dev(ix, d_q, q_stride, nq, k, ef, cap_mult, d_ids, d_scores, d_n, st, rerank_k, d_extra_eps);
}

// one single-device index: HnswIndex::with_params — index/hnsw/index/constructors.rs:117-160
int32_t create_single(uint32_t dim, int32_t metric, uint32_t M, uint32_t ef_construction, uint64_t max_elements,
                      int32_t device, vdb_hip_index** out) {
  *out = nullptr;
  if (dim == 0 || metric < 0 || metric > 4 || M < 2) return fail(VDB_ERR_INVALID_ARG, "bad dim/metric/M");
  int32_t ndev = 0;
  int32_t rc = check_device(&ndev);
  if (rc != VDB_OK) return rc;
  if (device < 0 || device >= ndev) return fail(VDB_ERR_INVALID_ARG, "bad device ordinal");
  VDB_HIP(hipSetDevice(device));
  std::unique_ptr<vdb_hip_index, void (*)(vdb_hip_index*)> ix(new vdb_hip_index(), destroy_single);
  ix->device = device;
  ix->generation = ++g_generation;
  ix->combiner = combiner_new();
  hipDeviceProp_t p;
  VDB_HIP(hipGetDeviceProperties(&p, device));
  ix->n_cus = p.multiProcessorCount;
  ix->dim = dim;
  ix->metric = metric;
  ix->M = M;
  ix->M0 = M * 2;  // native/graph.rs:62
  ix->efc = ef_construction;
  ix->row_stride = ((uint64_t)dim + 3) / 4 * 4;
  ix->words = ((dim + 31) / 32 + 3) / 4 * 4;
  VDB_HIP(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
  VDB_HIP(hipEventCreateWithFlags(&ix->ev_foreign, hipEventDisableTiming));
  VDB_HIP(hipEventCreateWithFlags(&ix->ev_own, hipEventDisableTiming));
  GraphLayer l0;
  l0.stride = ix->M0;
  ix->layers.push_back(l0);  // graph.rs:68 vec![Layer::new(max_elements)]
  rc = ensure_capacity(ix.get(), std::max<uint64_t>(max_elements, 1));
  if (rc != VDB_OK) return rc;
  *out = ix.release();
  return VDB_OK;
}

// EVERY device buffer a single-device handle can own — the one list destroy_single frees (a buffer missing here leaks with
// every destroyed handle; tests/test_gpu_hardening.py::test_destroy_returns_all_device_memory watches hipMemGetInfo)
std::vector<DevBuf*> index_buffers(vdb_hip_index* ix) {
  std::vector<DevBuf*> v = {
      &ix->rows, &ix->norms, &ix->bits, &ix->alive, &ix->ext_ids,           // rows
      &ix->rows_bf16, &ix->norms_bf16, &ix->bf16_rho, &ix->rows_split,      // bf16 copy, split-bf16 image
      &ix->l2_img, &ix->l2_seed, &ix->l2_rho,                               // Euclidean selection images
      &ix->cosn_img, &ix->cosn_rho,                                         // Cosine selection image (normalised rows)
      &ix->sq_min, &ix->sq_scale, &ix->codes, &ix->codes_sq,                // int8 traversal
      &ix->sq8_codes, &ix->sq8_min, &ix->sq8_max, &ix->sq8_nsq, &ix->sign_bits,  // storage modes
      &ix->sq8_img, &ix->sq8_nrm, &ix->sq8_seed, &ix->sq8_rho,              // SQ8 selection images
      &ix->bits_img, &ix->bits_cnt,                                         // four-bit image of the bit rows (Hamming / Jaccard GEMM)
      &ix->s_queries, &ix->s_part_keys, &ix->s_part_cnt, &ix->s_out, &ix->s_qbits, &ix->s_tickets,
      &ix->s_misc, &ix->s_fb_keys, &ix->s_seed, &ix->s_visited, &ix->s_vlog, &ix->s_stats, &ix->s_build_stats, &ix->s_levels, &ix->s_req_keys,
      &ix->s_req_vals, &ix->s_sort_tmp};
  for (auto& L : ix->layers) {
    v.push_back(&L.nbr);
    v.push_back(&L.cnt);
    v.push_back(&L.ndist);
  }
  return v;
}

void destroy_single(vdb_hip_index* ix) {
  if (!ix) return;
  (void)hipSetDevice(ix->device);
  if (ix->stream) (void)hipStreamSynchronize(ix->stream);
  for (vdb_hip_index* c : ix->ctx_clones) destroy_clone(c);
  ix->ctx_clones.clear();
  proc_comm_free(ix->pcomm);
  for (DevBuf* b : index_buffers(ix)) b->release();
  ix->h_in.release();
  ix->h_out.release();
  combiner_free(ix->combiner);
  for (auto* pool : {&ix->ev_pool, &ix->sel_ev})
    for (auto& e : *pool) {
      (void)hipEventDestroy(e.a);
      (void)hipEventDestroy(e.b);
    }
  if (ix->sel_stats) (void)hipHostFree(const_cast<uint32_t*>(ix->sel_stats));
  if (ix->ev_foreign) (void)hipEventDestroy(ix->ev_foreign);
  if (ix->ev_own) (void)hipEventDestroy(ix->ev_own);
  if (ix->stream) (void)hipStreamDestroy(ix->stream);
  delete ix;
}

// direction of the scores a search mode reports (the merge of per-shard results needs it): the metric's own
// (core/distance.rs:76-82) except the sign-bit scan, which reports Hamming distances whatever the metric
bool mode_higher_is_better(int metric, int32_t mode) {
  if (mode == VDB_SEARCH_BRUTE_BINARY) return false;
  return higher_is_better_host(metric);
}

// the result block of a context: [ids nq*kk u64 | scores nq*kk f32 | n nq u32] in ONE allocation
int32_t reserve_out(vdb_hip_index* ix, uint32_t nq, size_t kk, hipStream_t st) {
  const size_t o_sc = (size_t)nq * kk * 8, o_n = o_sc + (size_t)nq * kk * 4, total = o_n + (size_t)nq * 4;
  hipError_t e = ix->s_out.reserve(total, false, st);
  if (e != hipSuccess) return fail(VDB_ERR_OOM, std::string("search scratch: ") + hipGetErrorString(e));
  unsigned char* b = ix->s_out.as<unsigned char>();
  ix->s_out_ids.p = b;
  ix->s_out_scores.p = b + o_sc;
  ix->s_out_n.p = b + o_n;
  ix->s_out_bytes = total;
  return VDB_OK;
}

// rows of `queries` (nq x dim, caller memory) into the context's pinned staging buffer at query slot `at` of a batch of
// nq_total (row_stride layout; the padding floats zeroed so the device rows are fully defined)
int32_t stage_queries(vdb_hip_index* ix, const float* queries, uint32_t at, uint32_t nq, uint32_t nq_total) {
  hipError_t e = ix->h_in.reserve((size_t)nq_total * ix->row_stride * 4);
  if (e != hipSuccess) return fail(VDB_ERR_OOM, std::string("pinned query staging: ") + hipGetErrorString(e));
  float* h = ix->h_in.as<float>() + (size_t)at * ix->row_stride;
  if (ix->row_stride == ix->dim) {
    std::memcpy(h, queries, (size_t)nq * ix->dim * 4);
  } else {
    for (uint32_t i = 0; i < nq; i++) {
      std::memcpy(h + (size_t)i * ix->row_stride, queries + (size_t)i * ix->dim, (size_t)ix->dim * 4);
      std::memset(h + (size_t)i * ix->row_stride + ix->dim, 0, (ix->row_stride - ix->dim) * 4);
    }
  }
  return VDB_OK;
}

// The staged rows (h_in) up, the search, the whole result block back into h_out, ONE synchronisation.  HnswIndex::
// search_batch_parallel (batch.rs:159-197) / search_with_quality / search_brute_force.  staged = false: the queries already sit
// in ix->s_queries (row_stride layout).  The traversal kernel reports (out_n = 0xFFFFFFFF) a query whose candidate list
// overflowed its LDS capacity (only possible with many exact distance ties); such a batch is re-run with more room.
static int32_t search_block(vdb_hip_index* ix, bool staged, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode, uint32_t rerank_k,
                            const uint32_t* d_extra_eps = nullptr) {
  hipStream_t st = ix->stream;
  const size_t kk = std::max<uint32_t>(k, 1);
  hipError_t e;
  if ((e = ix->s_queries.reserve((size_t)nq * ix->row_stride * 4, false, st)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("search scratch: ") + hipGetErrorString(e));
  int32_t rc = reserve_out(ix, nq, kk, st);
  if (rc != VDB_OK) return rc;
  if ((e = ix->h_out.reserve(ix->s_out_bytes)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("pinned result staging: ") + hipGetErrorString(e));
  float* dq = ix->s_queries.as<float>();
  if (staged) VDB_HIP(hipMemcpyAsync(dq, ix->h_in.p, (size_t)nq * ix->row_stride * 4, hipMemcpyHostToDevice, st));
  uint32_t* h_n = reinterpret_cast<uint32_t*>(ix->h_out.as<unsigned char>() + ((unsigned char*)ix->s_out_n.p - (unsigned char*)ix->s_out.p));
  for (uint32_t cap_mult = 1;; cap_mult *= 4) {
    bool used_hnsw = false;
    rc = search_dev(ix, dq, ix->row_stride, nq, k, ef, mode, ix->s_out_ids.as<uint64_t>(), ix->s_out_scores.as<float>(),
                    ix->s_out_n.as<uint32_t>(), st, cap_mult, &used_hnsw, rerank_k, d_extra_eps);
    if (rc != VDB_OK) {
      (void)hipStreamSynchronize(st);
      return rc;
    }
    VDB_HIP(hipMemcpyAsync(ix->h_out.p, ix->s_out.p, ix->s_out_bytes, hipMemcpyDeviceToHost, st));
    VDB_HIP(hipStreamSynchronize(st));
    bool overflow = false;
    if (used_hnsw)
      for (uint32_t i = 0; i < nq; i++) overflow |= h_n[i] == 0xFFFFFFFFu;
    if (!overflow) {
      // rerank over the <=100-vector exact shortcut: at most rerank_k candidates exist (search.rs:124)
      if (rerank_k && !used_hnsw) {
        bool cut = false;
        for (uint32_t i = 0; i < nq; i++) {
          cut |= h_n[i] > rerank_k;
          h_n[i] = std::min(h_n[i], rerank_k);
        }
        if (cut) VDB_HIP(hipMemcpyAsync(ix->s_out_n.p, h_n, (size_t)nq * 4, hipMemcpyHostToDevice, st));
      }
      break;
    }
  }
  return VDB_OK;
}
int32_t search_staged(vdb_hip_index* ix, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode, uint32_t rerank_k, const uint32_t* d_extra_eps) {
  return search_block(ix, true, nq, k, ef, mode, rerank_k, d_extra_eps);
}

// host queries -> results in ix->s_out_* on the device AND in ix->h_out (same layout); out_n also in the caller's array.
// queries == nullptr: they already sit in ix->s_queries.  The caller holds ix->mu.
// Large uploads skip the pinned staging (a host-side copy of every byte in front of the DMA): the runtime pipelines a copy from
// pageable memory through its own staging chunks.
// (1 M x 768, 1 024 queries = 3 MB per call: 0.16 ms over the device-resident call from pageable memory, 0.26 ms through the pinned
// buffer; 256 queries = 0.75 MB: 0.08 ms either way; one query: 33 us)
constexpr size_t kStageMaxBytes = (size_t)1 << 20;
int32_t search_to_device(vdb_hip_index* ix, const float* queries, uint32_t nq, uint32_t k, uint32_t ef, int32_t mode,
                         uint32_t rerank_k, uint32_t* out_n) {
  bool staged = false;
  if (queries) {
    const size_t bytes = (size_t)nq * ix->row_stride * 4;
    if (bytes <= kStageMaxBytes) {
      const int32_t rs = stage_queries(ix, queries, 0, nq, nq);
      if (rs != VDB_OK) return rs;
      staged = true;
    } else {
      hipStream_t st = ix->stream;
      if (ix->s_queries.reserve(bytes, false, st) != hipSuccess) return fail(VDB_ERR_OOM, "search scratch");
      float* dq = ix->s_queries.as<float>();
      if (ix->row_stride != ix->dim) {
        VDB_HIP(hipMemsetAsync(dq, 0, bytes, st));
        VDB_HIP(hipMemcpy2DAsync(dq, ix->row_stride * 4, queries, (size_t)ix->dim * 4, (size_t)ix->dim * 4, nq, hipMemcpyHostToDevice, st));
      } else {
        VDB_HIP(hipMemcpyAsync(dq, queries, bytes, hipMemcpyHostToDevice, st));  // (dense rows: one linear copy)
      }
    }
  }
  const int32_t rc = search_block(ix, staged, nq, k, ef, mode, rerank_k);
  if (rc != VDB_OK) return rc;
  if (out_n)
    std::memcpy(out_n, ix->h_out.as<unsigned char>() + ((unsigned char*)ix->s_out_n.p - (unsigned char*)ix->s_out.p), (size_t)nq * 4);
  return VDB_OK;
}

}  // namespace vdb

using namespace vdb;

// =============================================================================================
// the counters of the context's last graph search, device -> host (ctx_mu held; synchronises)
static int32_t fetch_search_stats(vdb_hip_index* ix) {
  unsigned long long h[3] = {0, 0, 0};
  VDB_ENTER_SHARED(ix);
  VDB_HIP(hipDeviceSynchronize());
  VDB_HIP(hipMemcpy(h, ix->s_stats.p, 24, hipMemcpyDeviceToHost));
  ix->last_n_dist = h[0];
  ix->last_n_expand = h[1];
  ix->last_pf_hits = h[2];
  ix->stats_pending = false;
  return VDB_OK;
}

extern "C" {

const char* vdb_hip_last_error(void) { return g_last_error.c_str(); }
const char* vdb_hip_version(void) { return "velesdb-hip 0.1.0 (gfx950)"; }

int32_t vdb_hip_set_max_query_tile(uint32_t b) {
  return vdb::guarded([&]() -> int32_t {
  if (b != 1 && b != 2 && b != 4 && b != 8 && b != 16 && b != 32 && b != 48 && b != 128)
    return fail(VDB_ERR_INVALID_ARG, "tile must be 1, 2, 4, 8, 16, 32, 48 or 128");
  g_max_tile = b;
  return VDB_OK;
  });
}

int32_t vdb_hip_set_sweep_engine(int32_t engine) {
  return vdb::guarded([&]() -> int32_t {
  if (engine != 0 && engine != 1) return fail(VDB_ERR_INVALID_ARG, "engine must be 0 (VALU) or 1 (MFMA)");
  g_sweep_engine = engine;
  return VDB_OK;
  });
}

// diagnostic: how many queries of the last split-selector batch (its last <= 1024-query chunk) were answered by the exact
// fallback kernel because their selection could not be proven
int32_t vdb_hip_index_last_split_stats(vdb_hip_index* ix, uint32_t* queries, uint32_t* unproven) {
  return vdb::guarded([&]() -> int32_t {
    if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
    VDB_NO_GROUP(ix, "last_split_stats");
    std::shared_lock<vdb::IndexMutex> g(ix->mu);
    ix = last_context(ix);  // the search context that served this thread's last search (the handle itself unless searches overlapped)
    std::lock_guard<std::mutex> cg(ix->ctx_mu);
    VDB_HIP(hipSetDevice(ix->device));
    uint32_t nq = ix->split_flags_n, bad = 0;
    if (nq) {
      std::vector<uint32_t> h(nq);
      VDB_HIP(hipStreamSynchronize(ix->split_flags_stream));
      VDB_HIP(hipMemcpy(h.data(), ix->s_seed.as<unsigned char>() + ix->split_flags_off, (size_t)nq * 4, hipMemcpyDeviceToHost));
      for (uint32_t f : h) bad += f ? 1u : 0u;
    }
    if (queries) *queries = nq;
    if (unproven) *unproven = bad;
    return VDB_OK;
  });
}

int32_t vdb_hip_index_last_select_level(vdb_hip_index* ix, int32_t* level) {
  return vdb::guarded([&]() -> int32_t {
    if (!ix || !level) return fail(VDB_ERR_INVALID_ARG, "null argument");
    VDB_NO_GROUP(ix, "last_select_level");
    std::shared_lock<vdb::IndexMutex> g(ix->mu);
    ix = last_context(ix);  // the search context that served this thread's last search (the handle itself unless searches overlapped)
    std::lock_guard<std::mutex> cg(ix->ctx_mu);
    *level = ix->last_select_level;
    return VDB_OK;
  });
}

int32_t vdb_hip_index_last_kernels(vdb_hip_index* ix, uint32_t* mask) {
  return vdb::guarded([&]() -> int32_t {
    if (!ix || !mask) return fail(VDB_ERR_INVALID_ARG, "null argument");
    std::shared_lock<vdb::IndexMutex> g(ix->mu);
    // a multi-device handle: what any shard ran; a plain handle: the context of this thread's last search
    uint32_t m = ix->group ? ix->last_kernels : vdb::last_kernels_of_this_thread(ix);
    if (ix->group)
      for (size_t s = 0; s < group_size(ix); s++) m |= group_shard(ix, s)->last_kernels;
    *mask = m;
    return VDB_OK;
  });
}

int32_t vdb_hip_index_set_option(vdb_hip_index* ix, int32_t option, int64_t value) {
  return vdb::guarded([&]() -> int32_t {
    if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
    if (option < 0 || option >= VDB_OPT_COUNT_) return fail(VDB_ERR_INVALID_ARG, "unknown option");
    if (ix->group) return group_set_option(ix, option, value);
    std::lock_guard<vdb::IndexMutex> g(ix->mu);
    mark_changed(ix);
    int32_t v = -1;  // negative: back to the process-wide default
    if (value >= 0) {
      switch (option) {
        case VDB_OPT_MAX_QUERY_TILE:
          if (!(value == 1 || value == 2 || value == 4 || value == 8 || value == 16 || value == 32 || value == 48 || value == 128))
            return fail(VDB_ERR_INVALID_ARG, "query tile: 1, 2, 4, 8, 16, 32, 48 or 128");
          v = (int32_t)value;
          break;
        case VDB_OPT_SWEEP_ENGINE:
          if (value > 1) return fail(VDB_ERR_INVALID_ARG, "sweep engine: 0 or 1");
          v = (int32_t)value;
          break;
        case VDB_OPT_SELECTOR_LEVEL: v = value >= 3 ? 3 : (int32_t)value; break;
        case VDB_OPT_INT8_OVERSAMPLING:
          if (value < 1 || value > 64) return fail(VDB_ERR_INVALID_ARG, "oversampling ratio: 1..64");
          v = (int32_t)value;
          break;
        case VDB_OPT_COMBINE_MAX_BATCH:
          if (value > 1024) return fail(VDB_ERR_INVALID_ARG, "combined batch: 0 (off) .. 1024 queries");
          v = (int32_t)value;
          break;
        case VDB_OPT_COMBINE_WINDOW_US:
          if (value > 10000) return fail(VDB_ERR_INVALID_ARG, "combining window: 0 .. 10000 us");
          v = (int32_t)value;
          break;
        case VDB_OPT_COMBINE_INFLIGHT:
          if (value > 7) return fail(VDB_ERR_INVALID_ARG, "combined batches in flight: 0 (by kind of search) .. 7");
          v = (int32_t)value;
          break;
        default: v = value ? 1 : 0; break;
      }
    }
    ix->opt[option] = v;
    return VDB_OK;
  });
}

int32_t vdb_hip_index_get_option(vdb_hip_index* ix, int32_t option, int64_t* value) {
  return vdb::guarded([&]() -> int32_t {
    if (!ix || !value) return fail(VDB_ERR_INVALID_ARG, "null argument");
    if (option < 0 || option >= VDB_OPT_COUNT_) return fail(VDB_ERR_INVALID_ARG, "unknown option");
    vdb_hip_index* c = ix->group ? group_first_shard(ix) : ix;
    std::shared_lock<vdb::IndexMutex> g(c->mu);
    *value = opt_value(c, option);
    return VDB_OK;
  });
}

int32_t vdb_hip_set_split_selector(int32_t level) {
  g_split_selector = level <= 0 ? 0 : (level >= 3 ? 3 : level);
  return VDB_OK;
}

int32_t vdb_hip_set_kernel_timing(int32_t on) {
  return vdb::guarded([&]() -> int32_t {
  g_timing = on ? 1 : 0;
  return VDB_OK;
  });
}

// GpuAccelerator::new() / is_available() — gpu/gpu_backend.rs:33,136
int32_t vdb_hip_device_count(int32_t* n) {
  return vdb::guarded([&]() -> int32_t {
  if (!n) return fail(VDB_ERR_INVALID_ARG, "n is null");
  int32_t rc = check_device(n);
  return rc == VDB_ERR_NO_DEVICE ? VDB_OK : rc;  // 0 devices is an answer, not an error
  });
}
int32_t vdb_hip_device_name(int32_t device, char* buf, size_t cap) {
  return vdb::guarded([&]() -> int32_t {
  if (!buf || cap == 0) return fail(VDB_ERR_INVALID_ARG, "buf is null");
  int32_t rc = check_device(nullptr);
  if (rc != VDB_OK) return rc;
  hipDeviceProp_t p;
  VDB_HIP(hipGetDeviceProperties(&p, device));
  std::snprintf(buf, cap, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
  return VDB_OK;
  });
}

// HnswIndex::with_params — index/hnsw/index/constructors.rs:117-160
int32_t vdb_hip_index_create(uint32_t dim, int32_t metric, uint32_t M, uint32_t ef_construction,
                             uint64_t max_elements, const int32_t* devices, int32_t n_devices, int32_t shard_mode,
                             vdb_hip_index** out) {
  return vdb::guarded([&]() -> int32_t {
  if (!out) return fail(VDB_ERR_INVALID_ARG, "out is null");
  *out = nullptr;
  if (n_devices < 0 || n_devices > 64 || (n_devices > 0 && !devices)) return fail(VDB_ERR_INVALID_ARG, "bad device list");
  if (shard_mode != VDB_SHARD_REPLICA && shard_mode != VDB_SHARD_RANGE) return fail(VDB_ERR_INVALID_ARG, "bad shard mode");
  const int32_t dev0 = n_devices > 0 ? devices[0] : 0;
  if (n_devices <= 1) return create_single(dim, metric, M, ef_construction, max_elements, dev0, out);
  if (dim == 0 || metric < 0 || metric > 4 || M < 2) return fail(VDB_ERR_INVALID_ARG, "bad dim/metric/M");
  int32_t ndev = 0;
  int32_t rc = check_device(&ndev);
  if (rc != VDB_OK) return rc;
  for (int32_t i = 0; i < n_devices; i++)
    if (devices[i] < 0 || devices[i] >= ndev) return fail(VDB_ERR_INVALID_ARG, "bad device ordinal");
  std::unique_ptr<vdb_hip_index, void (*)(vdb_hip_index*)> ix(new vdb_hip_index(), vdb_hip_index_destroy);
  ix->dim = dim;
  ix->metric = metric;
  ix->M = M;
  ix->M0 = M * 2;
  ix->efc = ef_construction;
  ix->row_stride = ((uint64_t)dim + 3) / 4 * 4;
  rc = group_create(ix.get(), devices, n_devices, shard_mode, max_elements);
  if (rc != VDB_OK) return rc;
  *out = ix.release();
  return VDB_OK;
  });
}

void vdb_hip_index_destroy(vdb_hip_index* ix) {
  if (!ix) return;
  if (ix->group) {
    shard_group_free(ix->group);
    delete ix;
    return;
  }
  destroy_single(ix);
}

// VectorIndex::insert — index/mod.rs:46; trait_impl.rs:10-36
int32_t vdb_hip_index_insert(vdb_hip_index* ix, uint64_t id, const float* vec, uint32_t vec_len) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !vec) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (vec_len != ix->dim)
    return fail(VDB_ERR_DIM_MISMATCH, "Vector dimension mismatch: expected " + std::to_string(ix->dim) + ", got " +
                                          std::to_string(vec_len));
  if (ix->group) {
    uint64_t gi = 0;
    int32_t grc = group_insert(ix, &id, vec, 1, 0, 1, &gi);
    return grc != VDB_OK ? grc : (gi ? VDB_OK : VDB_DUPLICATE_IGNORED);
  }
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  VDB_ENTER(ix);
  uint64_t ins = 0, first = 0;
  int32_t rc = append_host_rows(ix, &id, vec, 1, &ins, &first);
  if (rc != VDB_OK) return rc;
  if (ins == 0) return VDB_DUPLICATE_IGNORED;
  if (ix->graph_valid) {
    rc = graph_insert_rows(ix, first, 1, 1);
    if (rc != VDB_OK) return rc;
  }
  return VDB_OK;
  });
}

// HnswIndex::insert_batch_sequential — batch.rs:128-149 (deterministic order)
int32_t vdb_hip_index_insert_batch(vdb_hip_index* ix, const uint64_t* ids, const float* vecs, uint64_t n,
                                   uint64_t* inserted) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || (n && (!ids || !vecs))) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) return group_insert(ix, ids, vecs, n, 0, 1, inserted);
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  VDB_ENTER(ix);
  uint64_t ins = 0, first = 0;
  int32_t rc = append_host_rows(ix, ids, vecs, n, &ins, &first);
  if (inserted) *inserted = ins;
  if (rc != VDB_OK) return rc;
  if (ins && ix->graph_valid) rc = graph_insert_rows(ix, first, ins, 1);
  return rc;
  });
}

// HnswIndex::insert_batch_parallel — batch.rs:83-108 (rayon in the reference, non-deterministic there;
// here batch-synchronous and deterministic, hnsw_build.hip)
int32_t vdb_hip_index_insert_batch_parallel(vdb_hip_index* ix, const uint64_t* ids, const float* vecs, uint64_t n,
                                            uint32_t max_batch, uint64_t* inserted) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || (n && (!ids || !vecs))) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) return group_insert(ix, ids, vecs, n, 1, max_batch, inserted);
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  VDB_ENTER(ix);
  uint64_t ins = 0, first = 0;
  int32_t rc = append_host_rows(ix, ids, vecs, n, &ins, &first);
  if (inserted) *inserted = ins;
  if (rc != VDB_OK) return rc;
  if (ins && ix->graph_valid) rc = graph_insert_rows(ix, first, ins, max_batch);
  return rc;
  });
}

// ScalarQuantizer::train + quantisation of every row (native/quantization.rs:191-252; DualPrecisionHnsw trains on its
// first min(1000, max_elements) inserts, dual_precision.rs:95,134-157)
int32_t vdb_hip_index_train_quantizer(vdb_hip_index* ix, uint32_t sample_rows) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) return group_for_all(ix, 3, sample_rows);
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  VDB_ENTER(ix);
  int32_t rc = quantizer_train(ix, sample_rows);
  if (rc == VDB_OK) VDB_HIP(hipStreamSynchronize(ix->stream));
  return rc;
  });
}
int32_t vdb_hip_set_int8_oversampling(uint32_t ratio) {
  return vdb::guarded([&]() -> int32_t {
  if (ratio == 0 || ratio > 64) return fail(VDB_ERR_INVALID_ARG, "oversampling ratio must be 1..64");
  g_int8_oversampling = ratio;
  return VDB_OK;
  });
}

// keeps a bf16 copy of the rows (round to nearest even) for VDB_SEARCH_BRUTE_BF16; existing rows are converted now,
// later inserts / uploads as they arrive
int32_t vdb_hip_index_enable_bf16(vdb_hip_index* ix) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) return group_for_all(ix, 1, 0);
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  if (ix->bf16_enabled) return VDB_OK;
  if (ix->metric != VDB_COSINE && ix->metric != VDB_DOT)
    return fail(VDB_ERR_UNSUPPORTED, "bf16 sweep: Cosine and DotProduct only");
  VDB_ENTER(ix);
  ix->bf16_stride = ((uint64_t)ix->dim + 7) / 8 * 8;
  hipError_t e;
  if ((e = ix->rows_bf16.reserve((std::max<uint64_t>(ix->capacity, 1) + kRowSlack) * ix->bf16_stride * 2, false, ix->stream)) != hipSuccess ||
      (e = ix->norms_bf16.reserve((std::max<uint64_t>(ix->capacity, 1) + kRowSlack) * 4, false, ix->stream)) != hipSuccess)
    return fail(VDB_ERR_OOM, std::string("bf16 rows: ") + hipGetErrorString(e));
  {
    const int32_t rr = reset_bf16_rho(ix, ix->stream);
    if (rr != VDB_OK) return rr;
  }
  ix->bf16_enabled = true;
  if (ix->n_rows) {
    launch_prep_bf16(ix->rows.as<float>(), ix->row_stride, ix->rows_bf16.as<uint16_t>(), ix->bf16_stride,
                     ix->norms_bf16.as<float>(), 0, (uint32_t)ix->n_rows, ix->dim, ix->stream, ix->bf16_rho.as<uint32_t>());
    VDB_HIP(hipGetLastError());
    VDB_HIP(hipStreamSynchronize(ix->stream));
  }
  ix->bf16_rows = ix->n_rows;
  return VDB_OK;
  });
}

// links every row that is not in the graph yet (rows that arrived through upload / upload_dev)
int32_t vdb_hip_index_build_graph(vdb_hip_index* ix, uint32_t max_batch) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) return group_for_all(ix, 0, max_batch);
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  VDB_ENTER(ix);
  int32_t rc = VDB_OK;
  if (ix->graph_nodes < ix->n_rows) rc = graph_insert_rows(ix, ix->graph_nodes, ix->n_rows - ix->graph_nodes, max_batch);
  if (rc == VDB_OK) ix->graph_valid = true;
  return rc;
  });
}

int32_t vdb_hip_index_upload(vdb_hip_index* ix, const uint64_t* ids, const float* vecs, uint64_t n,
                             uint64_t* inserted) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || (n && (!ids || !vecs))) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) return group_insert(ix, ids, vecs, n, 2, 0, inserted);
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  VDB_ENTER(ix);
  uint64_t ins = 0, first = 0;
  int32_t rc = append_host_rows(ix, ids, vecs, n, &ins, &first);
  if (inserted) *inserted = ins;
  if (ins) ix->graph_valid = false;
  if (rc == VDB_OK) VDB_HIP(hipStreamSynchronize(ix->stream));
  return rc;
  });
}

int32_t vdb_hip_index_upload_dev(vdb_hip_index* ix, uint64_t id_base, const float* d_vecs, uint64_t n, void* stream) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || (n && !d_vecs)) return fail(VDB_ERR_INVALID_ARG, "null argument");
  VDB_NO_GROUP(ix, "upload_dev (rows resident on one device)");
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  VDB_ENTER(ix);
  if (n == 0) return VDB_OK;
  for (uint64_t i = 0; i < n; i++)
    if (ix->id_to_idx.count(id_base + i)) return fail(VDB_ERR_INVALID_ARG, "upload_dev: id range overlaps existing ids");
  if (ix->n_rows + n > kMaxRowsPerIndex) return fail(VDB_ERR_UNSUPPORTED, "more than 2^32-512 rows per index");
  int32_t rc = ensure_capacity(ix, ix->n_rows + n);
  if (rc != VDB_OK) return rc;
  hipStream_t caller = reinterpret_cast<hipStream_t>(stream);
  VDB_HIP(hipStreamSynchronize(caller));  // d_vecs must be complete before our stream reads it
  float* drows = ix->rows.as<float>() + ix->n_rows * ix->row_stride;
  VDB_HIP(hipMemcpy2DAsync(drows, ix->row_stride * 4, d_vecs, (size_t)ix->dim * 4, (size_t)ix->dim * 4, n,
                           hipMemcpyDeviceToDevice, ix->stream));
  if (ix->row_stride != ix->dim)
    VDB_HIP(hipMemset2DAsync(drows + ix->dim, ix->row_stride * 4, 0, (ix->row_stride - ix->dim) * 4, n, ix->stream));
  std::vector<uint64_t> new_ids(n);
  for (uint64_t i = 0; i < n; i++) {
    new_ids[i] = id_base + i;
    ix->id_to_idx[id_base + i] = ix->n_rows + i;
  }
  VDB_HIP(hipMemcpyAsync(ix->ext_ids.as<uint64_t>() + ix->n_rows, new_ids.data(), n * 8, hipMemcpyHostToDevice,
                         ix->stream));
  VDB_HIP(hipMemsetAsync(ix->alive.as<uint8_t>() + ix->n_rows, 1, n, ix->stream));
  ix->idx_to_id.insert(ix->idx_to_id.end(), new_ids.begin(), new_ids.end());
  ix->idx_live.insert(ix->idx_live.end(), n, 1);
  ix->live += n;
  const uint64_t first = ix->n_rows;
  ix->n_rows += n;
  ix->graph_valid = false;
  rc = finish_append(ix, first, n);
  if (rc != VDB_OK) return rc;
  VDB_HIP(hipStreamSynchronize(ix->stream));
  return VDB_OK;
  });
}

// VectorIndex::remove — soft delete (trait_impl.rs:54-58)
int32_t vdb_hip_index_remove(vdb_hip_index* ix, uint64_t id, int32_t* removed) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) return group_remove(ix, id, removed);
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  auto it = ix->id_to_idx.find(id);
  if (it == ix->id_to_idx.end()) {
    if (removed) *removed = 0;
    return VDB_OK;
  }
  const uint64_t idx = it->second;
  VDB_ENTER(ix);
  VDB_HIP(hipMemsetAsync(ix->alive.as<uint8_t>() + idx, 0, 1, ix->stream));
  VDB_HIP(hipStreamSynchronize(ix->stream));
  ix->idx_live[idx] = 0;
  ix->id_to_idx.erase(it);
  ix->live--;
  ix->any_dead = true;
  if (removed) *removed = 1;
  return VDB_OK;
  });
}

int32_t vdb_hip_index_len(const vdb_hip_index* ix, uint64_t* n) {  // trait_impl.rs:60-62 mappings.len()
  if (!ix || !n) return fail(VDB_ERR_INVALID_ARG, "null argument");
  std::shared_lock<vdb::IndexMutex> g(ix->mu);  // (a pure read: searches go on)
  *n = ix->live;
  return VDB_OK;
}
// HnswIndex::tombstone_count (index/hnsw/index/vacuum.rs:45-52): entries removed from the mappings but still in the
// graph = next_idx - len.  tombstone_ratio / needs_vacuum (:60-76) follow from it and vdb_hip_index_node_count.
int32_t vdb_hip_index_tombstone_count(const vdb_hip_index* ix, uint64_t* n) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !n) return fail(VDB_ERR_INVALID_ARG, "null argument");
  std::shared_lock<vdb::IndexMutex> g(ix->mu);  // (a pure read: searches go on)
  *n = ix->n_rows - ix->live;
  return VDB_OK;
  });
}

// HnswIndex::vacuum (vacuum.rs:110-184): rebuild the graph over the active vectors only, with HnswParams::auto of the
// dimension (params.rs:41-57), new internal indices 0..count-1.  The reference collects the active vectors in
// hash-map order and inserts them with rayon (non-deterministic); here: ascending old internal index, the
// deterministic batch-synchronous construction (vdb_hip_index_insert_batch_parallel).  Quantised / bf16 copies are
// re-encoded with the rows.
int32_t vdb_hip_index_vacuum(vdb_hip_index* ix, uint64_t* count) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
  VDB_NO_GROUP(ix, "vacuum");
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  VDB_ENTER(ix);
  const uint64_t n_old = ix->n_rows, n_live = ix->live;
  if (count) *count = n_live;
  if (n_live == 0) return VDB_OK;  // vacuum.rs:123-125: nothing to rebuild
  // 1. snapshot of the active vectors (vacuum.rs:115-121)
  std::vector<float> all((size_t)n_old * ix->dim);
  VDB_HIP(hipMemcpy2DAsync(all.data(), (size_t)ix->dim * 4, ix->rows.p, ix->row_stride * 4, (size_t)ix->dim * 4, n_old,
                           hipMemcpyDeviceToHost, ix->stream));
  VDB_HIP(hipStreamSynchronize(ix->stream));
  std::vector<float> vecs((size_t)n_live * ix->dim);
  std::vector<uint64_t> ids(n_live);
  uint64_t w = 0;
  for (uint64_t i = 0; i < n_old; i++)
    if (ix->idx_live[i]) {
      std::memcpy(vecs.data() + (size_t)w * ix->dim, all.data() + (size_t)i * ix->dim, (size_t)ix->dim * 4);
      ids[w++] = ix->idx_to_id[i];
    }
  all.clear();
  all.shrink_to_fit();
  // 2. fresh graph state with auto parameters (vacuum.rs:127-134).  The one large allocation of the rebuild — the new layer-0
  // arrays at the index's capacity — is made BEFORE anything is torn down: when the device is out of memory the call fails
  // here with the index untouched (the rows keep their buffer, the id maps are host memory, upper layers are small).
  const uint32_t M_new = ix->dim <= 256 ? 24 : 32, M0_new = M_new * 2;
  GraphLayer l0;
  l0.stride = M0_new;
  {
    const uint64_t cap0 = std::max<uint64_t>(ix->capacity, 1);
    hipError_t ea;
    if ((ea = l0.nbr.reserve(cap0 * M0_new * 4, false, ix->stream)) != hipSuccess ||
        (ea = l0.cnt.reserve(cap0 * 4, false, ix->stream)) != hipSuccess ||
        (ea = l0.ndist.reserve(cap0 * M0_new * 4, false, ix->stream)) != hipSuccess) {
      l0.nbr.release();
      l0.cnt.release();
      l0.ndist.release();
      return fail(VDB_ERR_OOM, std::string("vacuum: the rebuilt graph does not fit (index unchanged): ") + hipGetErrorString(ea));
    }
  }
  ix->M = M_new;
  ix->M0 = M0_new;
  ix->efc = ix->dim <= 256 ? 300 : 400;
  for (auto& L : ix->layers) {
    L.nbr.release();
    L.cnt.release();
    L.ndist.release();
  }
  ix->layers.clear();
  ix->layers.push_back(l0);
  ix->entry_point = -1;
  ix->max_layer = 0;
  ix->graph_nodes = 0;
  ix->graph_valid = true;
  ix->ndist_valid = true;
  ix->rng_state = 0x5DEECE66D1A4B5B5ull;
  ix->id_to_idx.clear();
  ix->idx_to_id.clear();
  ix->idx_live.clear();
  ix->live = 0;
  ix->any_dead = false;
  ix->n_rows = 0;
  ix->bf16_rows = 0;
  ix->split_rows = 0;
  ix->l2_rows = 0;
  ix->cosn_rows = 0;
  ix->bits_img_rows = 0;
  const uint64_t cap = ix->capacity;
  ix->capacity = 0;  // re-reserve the per-row arrays (rows keep their buffer; the new layer arrays are allocated)
  int32_t rc = ensure_capacity(ix, std::max<uint64_t>(cap, 1));
  if (rc != VDB_OK) return rc;
  // 3./4./5. re-register, re-store, link (vacuum.rs:136-154)
  uint64_t ins = 0, first = 0;
  rc = append_host_rows(ix, ids.data(), vecs.data(), n_live, &ins, &first);
  if (rc != VDB_OK) return rc;
  rc = graph_insert_rows(ix, 0, n_live, 0);
  if (rc != VDB_OK) return rc;
  VDB_HIP(hipStreamSynchronize(ix->stream));
  return VDB_OK;
  });
}

int32_t vdb_hip_index_node_count(const vdb_hip_index* ix, uint64_t* n) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !n) return fail(VDB_ERR_INVALID_ARG, "null argument");
  std::shared_lock<vdb::IndexMutex> g(ix->mu);  // (a pure read: searches go on)
  *n = ix->n_rows;
  return VDB_OK;
  });
}
int32_t vdb_hip_index_dimension(const vdb_hip_index* ix, uint32_t* dim) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !dim) return fail(VDB_ERR_INVALID_ARG, "null argument");
  *dim = ix->dim;
  return VDB_OK;
  });
}
int32_t vdb_hip_index_metric(const vdb_hip_index* ix, int32_t* metric) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !metric) return fail(VDB_ERR_INVALID_ARG, "null argument");
  *metric = ix->metric;
  return VDB_OK;
  });
}

int32_t vdb_hip_index_search_batch_dev(vdb_hip_index* ix, const float* d_queries, uint32_t nq, uint32_t k,
                                       uint32_t ef, int32_t mode, uint64_t* d_out_ids, float* d_out_scores,
                                       uint32_t* d_out_n, void* stream) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || (nq && (!d_queries || !d_out_ids || !d_out_scores || !d_out_n)))
    return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group)
    return group_search_dev(ix, d_queries, nq, k, ef, mode, d_out_ids, d_out_scores, d_out_n,
                            reinterpret_cast<hipStream_t>(stream));
  // searches share the handle (search.rs:80 takes the read lock); a member of a process group searches collectively on one
  // gather buffer: those calls stay exclusive
  std::shared_lock<vdb::IndexMutex> rd(ix->mu, std::defer_lock);
  std::unique_lock<vdb::IndexMutex> wr(ix->mu, std::defer_lock);
  if (ix->pcomm) wr.lock(); else rd.lock();
  CtxLease lease(ix);
  if (lease.rc != VDB_OK) return lease.rc;
  vdb_hip_index* const handle = ix;
  ix = lease.ctx;
  (void)handle;
  VDB_HIP(hipSetDevice(ix->device));
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (st != ix->stream) {
    // the kernels below share this index's scratch, rows and graph with everything enqueued before: order the caller's
    // stream behind the work pending on ix->stream and behind an earlier device-resident search on ANOTHER stream
    // (cleared BEFORE the event is recorded: a mark set by an entry point that enqueues beside this call is then kept for the next one)
    const bool dirty = ix->own_dirty.exchange(false);
    if (dirty || ix->last_foreign != st) {  // (a stream that has not waited since: it is ordered behind ev_foreign below at best)
      VDB_HIP(hipEventRecord(ix->ev_own, ix->stream));
      VDB_HIP(hipStreamWaitEvent(st, ix->ev_own, 0));
    }
    if (ix->foreign_pending && ix->last_foreign != st) VDB_HIP(hipStreamWaitEvent(st, ix->ev_foreign, 0));
  } else {
    VDB_ENTER_SHARED(ix);
  }
  int32_t rc = search_dev(ix, d_queries, ix->dim, nq, k, ef, mode, d_out_ids, d_out_scores, d_out_n, st);
  if (rc == VDB_OK && ix->pcomm && nq && k) {
    const int32_t m = (mode == VDB_SEARCH_AUTO) ? (ix->live <= 100 ? VDB_SEARCH_BRUTE : VDB_SEARCH_HNSW) : mode;
    rc = pcomm_exchange_merge(ix, nq, k, mode_higher_is_better(ix->metric, m), d_out_ids, d_out_scores, d_out_n, st);
  }
  if (st != ix->stream) {
    VDB_HIP(hipEventRecord(ix->ev_foreign, st));
    ix->last_foreign = st;
    ix->foreign_pending = true;
  }
  return rc;
  });
}

// which summation order the exact sweep uses for this index and k (tests / bench pick the oracle mode by it)
int32_t vdb_hip_index_sweep_arith_mode(vdb_hip_index* ix, uint32_t k, int32_t* mode) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !mode) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) return vdb_hip_index_sweep_arith_mode(group_shard(ix, 0), k, mode);
  const bool mfma = opt_engine(ix) == 1 && (ix->metric == VDB_COSINE || ix->metric == VDB_DOT) &&
                    sweep_mfma_lds_bytes(1, k, ix->dim) <= 160 * 1024;
  *mode = mfma ? 1 : 0;
  return VDB_OK;
  });
}

// DistanceEngine::batch_distance (native/distance.rs:21-24) /
// GpuAccelerator::batch_cosine_similarity|batch_euclidean_distance|batch_dot_product
// (gpu/gpu_backend.rs:157,355,397)
int32_t vdb_hip_batch_distance_dev(int32_t metric, int32_t kind, const float* d_query, const float* d_vecs,
                                   uint64_t n, uint32_t dim, float* d_out, void* stream) {
  return vdb::guarded([&]() -> int32_t {
  if (metric < 0 || metric > 4 || kind < 0 || kind > 2 || (kind == VDB_KIND_SQUARED && metric != VDB_EUCLIDEAN))
    return fail(VDB_ERR_INVALID_ARG, "bad metric/kind");
  if (n == 0 || dim == 0) return VDB_OK;  // gpu_backend.rs:163-169: empty in, empty out
  if (!d_query || !d_vecs || !d_out) return fail(VDB_ERR_INVALID_ARG, "null argument");
  ScoreArgs a{};
  a.query = d_query;
  a.rows = d_vecs;
  a.out = d_out;
  a.n_rows = n;
  a.dim = dim;
  a.kind = kind;
  a.aligned16 = (dim % 4 == 0 && ((uintptr_t)d_query % 16) == 0 && ((uintptr_t)d_vecs % 16) == 0) ? 1 : 0;
  launch_score_rows(metric, a, reinterpret_cast<hipStream_t>(stream));
  VDB_HIP(hipGetLastError());
  return VDB_OK;
  });
}

int32_t vdb_hip_batch_distance(int32_t device, int32_t metric, int32_t kind, const float* query, const float* vecs,
                               uint64_t n, uint32_t dim, float* out) {
  return vdb::guarded([&]() -> int32_t {
  if (metric < 0 || metric > 4 || kind < 0 || kind > 2 || (kind == VDB_KIND_SQUARED && metric != VDB_EUCLIDEAN))
    return fail(VDB_ERR_INVALID_ARG, "bad metric/kind");
  int32_t ndev = 0;
  int32_t rc = check_device(&ndev);
  if (rc != VDB_OK) return rc;
  if (n == 0 || dim == 0) return VDB_OK;
  if (!query || !vecs || !out) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (device < 0 || device >= ndev) return fail(VDB_ERR_INVALID_ARG, "bad device ordinal");
  VDB_HIP(hipSetDevice(device));
  // persistent, growing staging buffers per device (the reference's wgpu path allocates and frees three buffers per
  // call, gpu/gpu_backend.rs:157-296 — the pattern SURVEY 2.3 says not to reproduce); one call per device at a time
  struct Staging {
    std::mutex mu;
    DevBuf q, v, out;
    hipStream_t st = nullptr;
  };
  static Staging staging[64];
  if (device >= 64) return fail(VDB_ERR_INVALID_ARG, "bad device ordinal");
  Staging& sg = staging[device];
  std::lock_guard<std::mutex> lk(sg.mu);
  if (!sg.st) VDB_HIP(hipStreamCreateWithFlags(&sg.st, hipStreamNonBlocking));
  hipError_t e = sg.q.reserve((size_t)dim * 4, false, sg.st);
  if (e == hipSuccess) e = sg.v.reserve((size_t)n * dim * 4, false, sg.st);
  if (e == hipSuccess) e = sg.out.reserve((size_t)n * 4, false, sg.st);
  if (e == hipSuccess) e = hipMemcpyAsync(sg.q.p, query, (size_t)dim * 4, hipMemcpyHostToDevice, sg.st);
  if (e == hipSuccess) e = hipMemcpyAsync(sg.v.p, vecs, (size_t)n * dim * 4, hipMemcpyHostToDevice, sg.st);
  rc = VDB_OK;
  if (e == hipSuccess)
    rc = vdb_hip_batch_distance_dev(metric, kind, sg.q.as<float>(), sg.v.as<float>(), n, dim, sg.out.as<float>(), sg.st);
  if (e == hipSuccess && rc == VDB_OK) e = hipMemcpyAsync(out, sg.out.p, (size_t)n * 4, hipMemcpyDeviceToHost, sg.st);
  if (e == hipSuccess) e = hipStreamSynchronize(sg.st);
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? VDB_ERR_OOM : VDB_ERR_HIP,
                                   std::string("batch_distance: ") + hipGetErrorString(e));
  return rc;
  });
}

int32_t vdb_hip_index_last_kernel_ms(vdb_hip_index* ix, float* ms, uint32_t* launches) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !ms) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) return vdb_hip_index_last_kernel_ms(group_shard(ix, 0), ms, launches);
  std::shared_lock<vdb::IndexMutex> g(ix->mu);
    ix = last_context(ix);  // the search context that served this thread's last search (the handle itself unless searches overlapped)
    std::lock_guard<std::mutex> cg(ix->ctx_mu);
  double total = 0.0;
  uint32_t cnt = 0;
  for (size_t i = 0; i < ix->ev_used; i++) {
    float t = 0.f;
    if (hipEventSynchronize(ix->ev_pool[i].b) == hipSuccess &&
        hipEventElapsedTime(&t, ix->ev_pool[i].a, ix->ev_pool[i].b) == hipSuccess) {
      total += t;
      cnt++;
    }
  }
  *ms = cnt ? (float)(total / cnt) : 0.0f;
  if (launches) *launches = cnt;
  return VDB_OK;
  });
}

int32_t vdb_hip_index_last_selection_ms(vdb_hip_index* ix, float* total_ms, uint32_t* launches) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !total_ms) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) return vdb_hip_index_last_selection_ms(group_shard(ix, 0), total_ms, launches);
  std::shared_lock<vdb::IndexMutex> g(ix->mu);
    ix = last_context(ix);  // the search context that served this thread's last search (the handle itself unless searches overlapped)
    std::lock_guard<std::mutex> cg(ix->ctx_mu);
  double total = 0.0;
  uint32_t cnt = 0;
  for (size_t i = 0; i < ix->sel_ev_used; i++) {
    float t = 0.f;
    if (hipEventSynchronize(ix->sel_ev[i].b) == hipSuccess && hipEventElapsedTime(&t, ix->sel_ev[i].a, ix->sel_ev[i].b) == hipSuccess) {
      total += t;
      cnt++;
    }
  }
  *total_ms = (float)total;
  if (launches) *launches = cnt;
  return VDB_OK;
  });
}

// construction counters (hnsw_build.hip), cumulative since the handle was created; synchronises the device
int32_t vdb_hip_index_build_stats(vdb_hip_index* ix, uint64_t* rows_evaluated, uint64_t* distance_phases, uint64_t* nodes, uint64_t* select_rows) {
  return vdb::guarded([&]() -> int32_t {
    if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
    if (ix->group) return fail(VDB_ERR_UNSUPPORTED, "build_stats: ask the shards");
    std::lock_guard<vdb::IndexMutex> g(ix->mu);
    VDB_ENTER(ix);
    unsigned long long h[4] = {0, 0, 0, 0};
    if (ix->s_build_stats.p) {
      VDB_HIP(hipDeviceSynchronize());
      VDB_HIP(hipMemcpy(h, ix->s_build_stats.p, 32, hipMemcpyDeviceToHost));
    }
    if (rows_evaluated) *rows_evaluated = h[0];
    if (distance_phases) *distance_phases = h[1];
    if (nodes) *nodes = h[2];
    if (select_rows) *select_rows = h[3];
    return VDB_OK;
  });
}

int32_t vdb_hip_index_last_search_stats(vdb_hip_index* ix, uint64_t* n_dist, uint64_t* n_expand) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) {  // sum over the shards (replicas: each searched its slice of the batch)
    uint64_t a = 0, b = 0;
    for (size_t s = 0; s < group_size(ix); s++) {
      uint64_t x = 0, y = 0;
      int32_t rc = vdb_hip_index_last_search_stats(group_shard(ix, s), &x, &y);
      if (rc != VDB_OK) return rc;
      a += x;
      b += y;
    }
    if (n_dist) *n_dist = a;
    if (n_expand) *n_expand = b;
    return VDB_OK;
  }
  std::shared_lock<vdb::IndexMutex> g(ix->mu);
    ix = last_context(ix);  // the search context that served this thread's last search (the handle itself unless searches overlapped)
    std::lock_guard<std::mutex> cg(ix->ctx_mu);
  if (ix->stats_pending) {
    const int32_t rc = fetch_search_stats(ix);
    if (rc != VDB_OK) return rc;
  }
  if (n_dist) *n_dist = ix->last_n_dist;
  if (n_expand) *n_expand = ix->last_n_expand;
  return VDB_OK;
  });
}

int32_t vdb_hip_index_last_prefetch_hits(vdb_hip_index* ix, uint64_t* hits) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !hits) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) {
    uint64_t a = 0;
    for (size_t s = 0; s < group_size(ix); s++) {
      uint64_t x = 0;
      int32_t rc = vdb_hip_index_last_prefetch_hits(group_shard(ix, s), &x);
      if (rc != VDB_OK) return rc;
      a += x;
    }
    *hits = a;
    return VDB_OK;
  }
  std::shared_lock<vdb::IndexMutex> g(ix->mu);
  ix = last_context(ix);
  std::lock_guard<std::mutex> cg(ix->ctx_mu);
  if (ix->stats_pending) {
    const int32_t rc = fetch_search_stats(ix);
    if (rc != VDB_OK) return rc;
  }
  *hits = ix->last_pf_hits;
  return VDB_OK;
  });
}

}  // extern "C"
