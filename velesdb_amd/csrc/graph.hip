// graph.hip — graph side of the index: persistence hand-off in the reference's format v1
// (NativeHnsw::file_dump / file_load, native/backend_adapter.rs:184-381), introspection, and the
// launch paths of the traversal / construction kernels (hnsw_kernels.hip).
#include <algorithm>
#include <filesystem>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "vdb_index.hpp"
#include "vdb_kernels.hpp"

using namespace vdb;

namespace vdb {

int32_t ensure_layers(vdb_hip_index* ix, uint32_t num_layers) {
  while (ix->layers.size() < num_layers) {
    GraphLayer L;
    L.stride = ix->M;  // upper layers: M links (graph.rs:204-208)
    hipError_t e = L.nbr.reserve(ix->capacity * L.stride * 4, false, ix->stream);
    if (e == hipSuccess) e = L.cnt.reserve(ix->capacity * 4, false, ix->stream);
    if (e == hipSuccess) e = hipMemsetAsync(L.cnt.p, 0, L.cnt.cap, ix->stream);
    if (e == hipSuccess && ix->ndist_valid) e = L.ndist.reserve(ix->capacity * L.stride * 4, false, ix->stream);
    if (e != hipSuccess) return fail(VDB_ERR_OOM, std::string("graph layer: ") + hipGetErrorString(e));
    ix->layers.push_back(L);
  }
  return VDB_OK;
}

}  // namespace vdb

extern "C" {

int32_t vdb_hip_index_graph_info(vdb_hip_index* ix, uint32_t* num_layers, uint32_t* max_layer,
                                 int64_t* entry_point) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) {  // replicas hold identical graphs; a range-sharded handle has one graph per shard
    if (group_mode(ix) != VDB_SHARD_REPLICA) return fail(VDB_ERR_UNSUPPORTED, "graph_info: one graph per shard on a range-sharded handle");
    return vdb_hip_index_graph_info(group_shard(ix, 0), num_layers, max_layer, entry_point);
  }
  std::shared_lock<vdb::IndexMutex> g(ix->mu);
  if (num_layers) *num_layers = (uint32_t)ix->layers.size();
  if (max_layer) *max_layer = ix->max_layer;
  if (entry_point) *entry_point = ix->graph_valid ? ix->entry_point : -1;
  return VDB_OK;
  });
}

int32_t vdb_hip_index_get_neighbors(vdb_hip_index* ix, uint32_t layer, uint64_t node, uint32_t* out, uint32_t cap,
                                    uint32_t* n) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !n) return fail(VDB_ERR_INVALID_ARG, "null argument");
  if (ix->group) {
    if (group_mode(ix) != VDB_SHARD_REPLICA) return fail(VDB_ERR_UNSUPPORTED, "get_neighbors: one graph per shard on a range-sharded handle");
    return vdb_hip_index_get_neighbors(group_shard(ix, 0), layer, node, out, cap, n);
  }
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  *n = 0;
  if (layer >= ix->layers.size() || node >= ix->n_rows) return VDB_OK;  // layer.rs:33-39: empty
  VDB_ENTER_READONLY(ix);
  GraphLayer& L = ix->layers[layer];
  uint32_t c = 0;
  VDB_HIP(hipMemcpyAsync(&c, L.cnt.as<uint32_t>() + node, 4, hipMemcpyDeviceToHost, ix->stream));
  VDB_HIP(hipStreamSynchronize(ix->stream));
  *n = c;
  uint32_t w = std::min(c, cap);
  if (w && out) {
    VDB_HIP(hipMemcpyAsync(out, L.nbr.as<uint32_t>() + node * L.stride, (size_t)w * 4, hipMemcpyDeviceToHost,
                           ix->stream));
    VDB_HIP(hipStreamSynchronize(ix->stream));
  }
  return VDB_OK;
  });
}

// NativeHnsw::file_load — native/backend_adapter.rs:273-381.  The index must be empty; dim must
// match; M/M0/ef_construction are taken from the file like the reference does.  External ids are
// the node ids (the reference keeps its id mappings in a separate bincode file that is out of
// scope; HnswIndex::load re-associates them, constructors.rs:190-253).
int32_t vdb_hip_index_load_reference_files(vdb_hip_index* ix, const char* dir, const char* basename) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !dir || !basename) return fail(VDB_ERR_INVALID_ARG, "null argument");
  VDB_NO_GROUP(ix, "load_reference_files");
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  if (ix->n_rows != 0) return fail(VDB_ERR_STATE, "load_reference_files needs an empty index");
  VDB_ENTER(ix);
  const std::string vp = std::string(dir) + "/" + basename + ".vectors";
  const std::string gp = std::string(dir) + "/" + basename + ".graph";
  // Everything is parsed and validated into locals first; the index is only touched once both files are known to be
  // consistent (a corrupt header must neither size a host buffer beyond the file nor leave a half-loaded index).
  struct Closer {
    FILE* f;
    ~Closer() {
      if (f) std::fclose(f);
    }
  };
  auto file_size = [](FILE* f) -> uint64_t {
    const long cur = std::ftell(f);
    std::fseek(f, 0, SEEK_END);
    const long end = std::ftell(f);
    std::fseek(f, cur, SEEK_SET);
    return end < 0 ? 0 : (uint64_t)end;
  };
  uint32_t version = 0, dim = 0;
  uint64_t count = 0;
  std::vector<float> vecs;
  {
    Closer c{std::fopen(vp.c_str(), "rb")};
    if (!c.f) return fail(VDB_ERR_IO, "cannot open " + vp);
    bool ok = std::fread(&version, 4, 1, c.f) == 1 && std::fread(&count, 8, 1, c.f) == 1 && std::fread(&dim, 4, 1, c.f) == 1;
    if (!ok || version != 1)
      return fail(VDB_ERR_IO, "Unsupported version: " + std::to_string(version));  // backend_adapter.rs:285-290
    if (count && dim != ix->dim)
      return fail(VDB_ERR_DIM_MISMATCH, "file dimension " + std::to_string(dim) + " != index dimension " +
                                            std::to_string(ix->dim));
    const uint64_t payload = file_size(c.f) - 16;
    if (count > kMaxRowsPerIndex || (dim && count > payload / ((uint64_t)dim * 4))) return fail(VDB_ERR_IO, "truncated " + vp);
    vecs.resize((size_t)count * dim);
    if (std::fread(vecs.data(), 4, vecs.size(), c.f) != vecs.size()) return fail(VDB_ERR_IO, "truncated " + vp);
  }
  uint32_t num_layers = 0, M = 0, M0 = 0, efc = 0, max_layer = 0;
  uint64_t ep = 0, count2 = 0;
  std::vector<std::vector<uint32_t>> h_nbr, h_cnt;
  {
    Closer c{std::fopen(gp.c_str(), "rb")};
    if (!c.f) return fail(VDB_ERR_IO, "cannot open " + gp);
    bool ok = std::fread(&version, 4, 1, c.f) == 1 && version == 1 && std::fread(&num_layers, 4, 1, c.f) == 1 &&
              std::fread(&M, 4, 1, c.f) == 1 && std::fread(&M0, 4, 1, c.f) == 1 && std::fread(&efc, 4, 1, c.f) == 1 &&
              std::fread(&ep, 8, 1, c.f) == 1 && std::fread(&max_layer, 4, 1, c.f) == 1 && std::fread(&count2, 8, 1, c.f) == 1;
    // the traversal and construction kernels index rows[entry_point] and layers[max_layer] without further checks
    if (!ok || num_layers == 0 || num_layers > (uint32_t)kMaxLayers || M < 2 || M > 4096 || M0 < M || M0 > 8192 ||
        max_layer >= num_layers || count2 != count || (count && ep >= count))
      return fail(VDB_ERR_IO, "bad graph header in " + gp);
    const uint64_t fsz = file_size(c.f);
    h_nbr.resize(num_layers);
    h_cnt.resize(num_layers);
    std::vector<uint32_t> tmp;
    for (uint32_t l = 0; l < num_layers; l++) {
      const uint32_t stride = l == 0 ? M0 : M;
      uint64_t nn = 0;
      if (std::fread(&nn, 8, 1, c.f) != 1) return fail(VDB_ERR_IO, "truncated " + gp);
      if (nn > fsz / 4) return fail(VDB_ERR_IO, "bad layer size in " + gp);  // every node costs >= 4 bytes
      h_nbr[l].assign((size_t)count * stride, 0);
      h_cnt[l].assign((size_t)count, 0);
      for (uint64_t i = 0; i < nn; i++) {
        uint32_t kk = 0;
        if (std::fread(&kk, 4, 1, c.f) != 1) return fail(VDB_ERR_IO, "truncated " + gp);
        if (kk > stride) return fail(VDB_ERR_IO, "node with more neighbours than the layer's max_connections");
        tmp.resize(kk);
        if (kk && std::fread(tmp.data(), 4, kk, c.f) != kk) return fail(VDB_ERR_IO, "truncated " + gp);
        if (i >= count) continue;
        for (uint32_t j = 0; j < kk; j++) {
          if (tmp[j] >= count) return fail(VDB_ERR_IO, "neighbour id out of range");
          h_nbr[l][(size_t)i * stride + j] = tmp[j];
        }
        h_cnt[l][i] = kk;
      }
    }
  }
  // ---- commit: vectors + ids, then the layers (adopting the file's parameters, backend_adapter.rs:368-379) ----
  std::vector<uint64_t> ids(count);
  for (uint64_t i = 0; i < count; i++) ids[i] = i;
  int32_t rc = ensure_capacity(ix, std::max<uint64_t>(count, 1));
  if (rc != VDB_OK) return rc;
  uint64_t ins = 0, first = 0;
  rc = append_host_rows(ix, ids.data(), vecs.data(), count, &ins, &first);
  if (rc != VDB_OK) return rc;
  std::vector<GraphLayer> layers;
  auto drop = [&]() {
    for (auto& L : layers) {
      L.nbr.release();
      L.cnt.release();
    }
  };
  for (uint32_t l = 0; l < num_layers; l++) {
    GraphLayer L;
    L.stride = l == 0 ? M0 : M;
    hipError_t e = L.nbr.reserve(std::max<size_t>(ix->capacity * L.stride * 4, 4), false, ix->stream);
    if (e == hipSuccess) e = L.cnt.reserve(ix->capacity * 4, false, ix->stream);
    if (e == hipSuccess) e = hipMemsetAsync(L.cnt.p, 0, L.cnt.cap, ix->stream);
    if (e == hipSuccess && count)
      e = hipMemcpyAsync(L.nbr.p, h_nbr[l].data(), h_nbr[l].size() * 4, hipMemcpyHostToDevice, ix->stream);
    if (e == hipSuccess && count)
      e = hipMemcpyAsync(L.cnt.p, h_cnt[l].data(), h_cnt[l].size() * 4, hipMemcpyHostToDevice, ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
    layers.push_back(L);
    if (e != hipSuccess) {
      drop();
      ix->graph_valid = false;  // the rows are in; the graph is not
      return fail(e == hipErrorOutOfMemory ? VDB_ERR_OOM : VDB_ERR_HIP, std::string("graph upload: ") + hipGetErrorString(e));
    }
  }
  for (auto& L : ix->layers) {
    L.nbr.release();
    L.cnt.release();
    L.ndist.release();
  }
  ix->layers = std::move(layers);
  ix->M = M;
  ix->M0 = M0;
  ix->efc = efc;
  ix->ndist_valid = false;  // the files carry no distances; recomputed before the first insert
  ix->entry_point = count ? (int64_t)ep : -1;
  ix->max_layer = max_layer;
  ix->graph_nodes = count;
  ix->graph_valid = true;
  ix->rng_state = 0x5DEECE66D1A4B5B5ull;  // backend_adapter.rs:373
  return VDB_OK;
  });
}

// NativeHnsw::file_dump — native/backend_adapter.rs:184-261
int32_t vdb_hip_index_save_reference_files(vdb_hip_index* ix, const char* dir, const char* basename) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !dir || !basename) return fail(VDB_ERR_INVALID_ARG, "null argument");
  VDB_NO_GROUP(ix, "save_reference_files");
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  if (!ix->graph_valid) return fail(VDB_ERR_STATE, "graph not built for all rows");
  VDB_ENTER_READONLY(ix);
  const uint64_t count = ix->n_rows;
  std::vector<float> vecs((size_t)count * ix->dim);
  if (count) {
    VDB_HIP(hipMemcpy2DAsync(vecs.data(), (size_t)ix->dim * 4, ix->rows.p, ix->row_stride * 4, (size_t)ix->dim * 4,
                             count, hipMemcpyDeviceToHost, ix->stream));
    VDB_HIP(hipStreamSynchronize(ix->stream));
  }
  const std::string vp = std::string(dir) + "/" + basename + ".vectors";
  const std::string gp = std::string(dir) + "/" + basename + ".graph";
  FILE* f = std::fopen(vp.c_str(), "wb");
  if (!f) return fail(VDB_ERR_IO, "cannot create " + vp);
  uint32_t version = 1, dim = count ? ix->dim : 0;
  std::fwrite(&version, 4, 1, f);
  std::fwrite(&count, 8, 1, f);
  std::fwrite(&dim, 4, 1, f);
  std::fwrite(vecs.data(), 4, vecs.size(), f);
  std::fclose(f);
  f = std::fopen(gp.c_str(), "wb");
  if (!f) return fail(VDB_ERR_IO, "cannot create " + gp);
  uint32_t num_layers = (uint32_t)ix->layers.size(), M = ix->M, M0 = ix->M0, efc = ix->efc, max_layer = ix->max_layer;
  uint64_t ep = ix->entry_point < 0 ? 0 : (uint64_t)ix->entry_point;
  std::fwrite(&version, 4, 1, f);
  std::fwrite(&num_layers, 4, 1, f);
  std::fwrite(&M, 4, 1, f);
  std::fwrite(&M0, 4, 1, f);
  std::fwrite(&efc, 4, 1, f);
  std::fwrite(&ep, 8, 1, f);
  std::fwrite(&max_layer, 4, 1, f);
  std::fwrite(&count, 8, 1, f);
  for (auto& L : ix->layers) {
    std::vector<uint32_t> nbr((size_t)count * L.stride), cnt(count);
    if (count) {
      VDB_HIP(hipMemcpyAsync(nbr.data(), L.nbr.p, nbr.size() * 4, hipMemcpyDeviceToHost, ix->stream));
      VDB_HIP(hipMemcpyAsync(cnt.data(), L.cnt.p, cnt.size() * 4, hipMemcpyDeviceToHost, ix->stream));
      VDB_HIP(hipStreamSynchronize(ix->stream));
    }
    uint64_t nn = count;
    std::fwrite(&nn, 8, 1, f);
    for (uint64_t i = 0; i < count; i++) {
      std::fwrite(&cnt[i], 4, 1, f);
      std::fwrite(nbr.data() + (size_t)i * L.stride, 4, cnt[i], f);
    }
  }
  std::fclose(f);
  return VDB_OK;
  });
}


// ---- HnswIndex::save / HnswIndex::load (index/hnsw/index/constructors.rs:190-287) --------------------------
// A directory with
//   native_hnsw.vectors / native_hnsw.graph   NativeHnsw::file_dump, format v1 (above)
//   native_mappings.bin   bincode 1.3.3 (Cargo.lock:393-396), default options = fixed-width little-endian integers,
//                         u64 sequence lengths: (HashMap<u64, usize> id_to_idx, HashMap<usize, u64> idx_to_id,
//                         usize next_idx) -> len u64, (key u64, value u64)*, len u64, (key u64, value u64)*, u64.
//                         Entry order is the writer's hash-iteration order and carries no meaning; removed ids are
//                         simply absent from both maps (sharded_mappings.rs:115-122).
//   native_meta.bin       (usize dimension, u8 metric, bool enable_vector_storage) -> u64, u8, u8
// The loaded index keeps the vectors the .vectors file carries, so exact search stays exact after a load (the
// reference comes back with an empty ShardedVectors and falls back to HNSW there, SURVEY 8a note 9).
namespace {
bool put_u64(FILE* f, uint64_t v) { return std::fwrite(&v, 8, 1, f) == 1; }
bool get_u64(FILE* f, uint64_t* v) { return std::fread(v, 8, 1, f) == 1; }
}  // namespace

int32_t vdb_hip_index_save_dir(vdb_hip_index* ix, const char* dir) {
  return vdb::guarded([&]() -> int32_t {
  if (!ix || !dir) return fail(VDB_ERR_INVALID_ARG, "null argument");
  VDB_NO_GROUP(ix, "save_dir");
  {  // HnswIndex::save starts with std::fs::create_dir_all(path) (constructors.rs:257)
    std::error_code ec;
    std::filesystem::create_directories(dir, ec);
    if (ec) return fail(VDB_ERR_IO, std::string("cannot create directory ") + dir + ": " + ec.message());
  }
  int32_t rc = vdb_hip_index_save_reference_files(ix, dir, "native_hnsw");
  if (rc != VDB_OK) return rc;
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  const std::string mp = std::string(dir) + "/native_mappings.bin";
  FILE* f = std::fopen(mp.c_str(), "wb");
  if (!f) return fail(VDB_ERR_IO, "cannot create " + mp);
  bool ok = put_u64(f, ix->id_to_idx.size());
  for (uint64_t idx = 0; idx < ix->n_rows && ok; idx++)  // ascending internal index: any order is valid bincode
    if (ix->idx_live[idx]) ok = put_u64(f, ix->idx_to_id[idx]) && put_u64(f, idx);
  ok = ok && put_u64(f, ix->id_to_idx.size());
  for (uint64_t idx = 0; idx < ix->n_rows && ok; idx++)
    if (ix->idx_live[idx]) ok = put_u64(f, idx) && put_u64(f, ix->idx_to_id[idx]);
  ok = ok && put_u64(f, ix->n_rows);  // next_idx: one index per node ever inserted
  std::fclose(f);
  if (!ok) return fail(VDB_ERR_IO, "short write to " + mp);
  const std::string tp = std::string(dir) + "/native_meta.bin";
  f = std::fopen(tp.c_str(), "wb");
  if (!f) return fail(VDB_ERR_IO, "cannot create " + tp);
  const uint8_t metric = (uint8_t)ix->metric, storage = 1;
  ok = put_u64(f, ix->dim) && std::fwrite(&metric, 1, 1, f) == 1 && std::fwrite(&storage, 1, 1, f) == 1;
  std::fclose(f);
  return ok ? VDB_OK : fail(VDB_ERR_IO, "short write to " + tp);
  });
}

int32_t vdb_hip_index_load_dir(const char* dir, int32_t device, vdb_hip_index** out) {
  return vdb::guarded([&]() -> int32_t {
  if (!dir || !out) return fail(VDB_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  const std::string tp = std::string(dir) + "/native_meta.bin";
  FILE* f = std::fopen(tp.c_str(), "rb");
  if (!f) return fail(VDB_ERR_IO, "cannot open " + tp);
  uint64_t dim = 0;
  uint8_t metric = 0, storage = 0;
  bool ok = get_u64(f, &dim) && std::fread(&metric, 1, 1, f) == 1 && std::fread(&storage, 1, 1, f) == 1;
  std::fclose(f);
  if (!ok || dim == 0 || dim > 0xFFFFFFFFull) return fail(VDB_ERR_IO, "bad " + tp);
  if (metric > 4) return fail(VDB_ERR_IO, "Unknown distance metric");  // constructors.rs:211-216
  vdb_hip_index* ix = nullptr;
  int32_t rc = create_single((uint32_t)dim, (int32_t)metric, 16, 200, 1024, device, &ix);  // M / efc: from the graph file
  if (rc != VDB_OK) return rc;
  rc = vdb_hip_index_load_reference_files(ix, dir, "native_hnsw");
  if (rc != VDB_OK) {
    vdb_hip_index_destroy(ix);
    return rc;
  }
  const std::string mp = std::string(dir) + "/native_mappings.bin";
  f = std::fopen(mp.c_str(), "rb");
  if (!f) {
    vdb_hip_index_destroy(ix);
    return fail(VDB_ERR_IO, "cannot open " + mp);
  }
  uint64_t count = 0;
  (void)vdb_hip_index_node_count(ix, &count);
  std::vector<uint64_t> ids(count, 0);
  std::vector<uint8_t> live(count, 0);
  std::unordered_map<uint64_t, uint64_t> id_to_idx;
  uint64_t n1 = 0, n2 = 0, next_idx = 0;
  ok = get_u64(f, &n1) && n1 <= count;
  for (uint64_t i = 0; i < n1 && ok; i++) {
    uint64_t id = 0, idx = 0;
    ok = get_u64(f, &id) && get_u64(f, &idx) && idx < count;
    if (ok) id_to_idx[id] = idx;
  }
  ok = ok && get_u64(f, &n2) && n2 <= count;
  for (uint64_t i = 0; i < n2 && ok; i++) {
    uint64_t id = 0, idx = 0;
    ok = get_u64(f, &idx) && get_u64(f, &id) && idx < count;
    if (ok) {
      ids[idx] = id;
      live[idx] = 1;
    }
  }
  ok = ok && get_u64(f, &next_idx);
  std::fclose(f);
  // the two maps must be inverse of each other (they are written from one registry)
  if (ok) ok = id_to_idx.size() == n1 && n1 == n2;
  for (auto it = id_to_idx.begin(); ok && it != id_to_idx.end(); ++it) ok = live[it->second] && ids[it->second] == it->first;
  if (!ok) {
    vdb_hip_index_destroy(ix);
    return fail(VDB_ERR_IO, "bad " + mp);
  }
  std::lock_guard<vdb::IndexMutex> g(ix->mu);
  ix->id_to_idx.swap(id_to_idx);
  ix->idx_to_id = ids;
  ix->idx_live = live;
  ix->live = n1;
  ix->any_dead = n1 != count;
  hipError_t e = hipSuccess;
  if (count) {
    e = hipMemcpyAsync(ix->ext_ids.p, ids.data(), count * 8, hipMemcpyHostToDevice, ix->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ix->alive.p, live.data(), count, hipMemcpyHostToDevice, ix->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ix->stream);
  }
  if (e != hipSuccess) return fail(VDB_ERR_HIP, std::string("mappings upload: ") + hipGetErrorString(e));  // (index leaked on a device error)
  *out = ix;
  return VDB_OK;
  });
}

// ---- MmapStorage directory as an upload source (core/storage/mmap.rs:96-160 open, :402-455 store, :602-626 flush) ----
//   vectors.idx   bincode FxHashMap<u64, usize>: u64 count, then count x (u64 id, u64 byte offset into vectors.dat)
//   vectors.dat   raw little-endian f32, `dimension` values at each offset (pre-sized: 16 MiB, then grown)
//   vectors.wal   append-only log; MmapStorage::new does not replay it, so a flushed store = idx + dat
// Rows are uploaded in ascending byte offset = the order in which the store first saw each id (offsets are handed
// out by a monotonic counter, :434-436; an update rewrites its slot in place), which makes the internal row order —
// the tie-break of the exact search — a function of the files alone, not of hash-map iteration order.
int32_t vdb_hip_index_upload_vector_store(vdb_hip_index* ix, const char* dir, uint64_t* inserted) {
  return vdb::guarded([&]() -> int32_t {
  if (inserted) *inserted = 0;
  if (!ix || !dir) return fail(VDB_ERR_INVALID_ARG, "null argument");
  VDB_NO_GROUP(ix, "upload_vector_store");
  const std::string ip = std::string(dir) + "/vectors.idx", dp = std::string(dir) + "/vectors.dat";
  FILE* f = std::fopen(ip.c_str(), "rb");
  if (!f) return fail(VDB_ERR_IO, "cannot open " + ip);
  std::fseek(f, 0, SEEK_END);
  const long idx_bytes = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  uint64_t count = 0;
  bool ok = get_u64(f, &count) && idx_bytes >= 8 && count == (uint64_t)(idx_bytes - 8) / 16 && (idx_bytes - 8) % 16 == 0;
  std::vector<std::pair<uint64_t, uint64_t>> ent;  // (offset, id)
  if (ok) ent.reserve(count);
  for (uint64_t i = 0; i < count && ok; i++) {
    uint64_t id = 0, off = 0;
    ok = get_u64(f, &id) && get_u64(f, &off);
    if (ok) ent.emplace_back(off, id);
  }
  std::fclose(f);
  if (!ok) return fail(VDB_ERR_IO, "bad " + ip);
  FILE* d = std::fopen(dp.c_str(), "rb");
  if (!d) return fail(VDB_ERR_IO, "cannot open " + dp);
  std::fseek(d, 0, SEEK_END);
  const uint64_t dat_bytes = (uint64_t)std::ftell(d);
  const uint64_t vbytes = (uint64_t)ix->dim * 4;
  std::sort(ent.begin(), ent.end());
  for (size_t i = 0; i < ent.size() && ok; i++) {
    ok = ent[i].first % 4 == 0 && ent[i].first + vbytes <= dat_bytes;            // "Offset out of bounds" (mmap.rs:563-568)
    if (ok && i) ok = ent[i].first >= ent[i - 1].first + vbytes && ent[i].second != ent[i - 1].second;  // slots never overlap
  }
  if (!ok) {
    std::fclose(d);
    return fail(VDB_ERR_IO, "vectors.idx does not describe vectors.dat (offset out of bounds / overlapping slots)");
  }
  // chunks of <= 64 MiB through the ordinary upload path (duplicate ids already in the index are skipped there)
  const size_t chunk = std::max<size_t>(1, (size_t)((64u << 20) / vbytes));
  std::vector<float> rows;
  std::vector<uint64_t> ids;
  uint64_t total = 0;
  int32_t rc = VDB_OK;
  for (size_t base = 0; base < ent.size() && rc == VDB_OK; base += chunk) {
    const size_t n = std::min(chunk, ent.size() - base);
    rows.resize(n * ix->dim);
    ids.resize(n);
    for (size_t i = 0; i < n && ok; i++) {
      ids[i] = ent[base + i].second;
      ok = std::fseek(d, (long)ent[base + i].first, SEEK_SET) == 0 &&
           std::fread(rows.data() + i * ix->dim, 4, ix->dim, d) == ix->dim;
    }
    if (!ok) {
      rc = fail(VDB_ERR_IO, "short read from " + dp);
      break;
    }
    uint64_t ins = 0;
    rc = vdb_hip_index_upload(ix, ids.data(), rows.data(), n, &ins);
    total += ins;
  }
  std::fclose(d);
  if (inserted) *inserted = total;
  return rc;
  });
}

}  // extern "C"
