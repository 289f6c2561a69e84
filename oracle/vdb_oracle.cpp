// vdb_oracle.cpp — CPU ORACLE: a from-scratch restatement of the velesdb-core
// (v1.4.1) HNSW similarity-search hot path.  TEST INFRASTRUCTURE ONLY — see
// vdb_oracle.h for who may use it and for the parity-pin statement.
//
// Path abbreviations in citations (all under the reference checkout):
//   core/   = crates/velesdb-core/src/
//   hnsw/   = crates/velesdb-core/src/index/hnsw/
//   native/ = crates/velesdb-core/src/index/hnsw/native/
//
// Build: see oracle/Makefile (g++ -O3 -mavx2 -mfma -ffp-contract=off).
// -ffp-contract=off matters: where the reference writes `x += a*b` (Rust never
// contracts) the product and the sum must round separately; fusion happens only
// where the reference calls mul_add, and there we call fmaf / vfmadd explicitly.
#include "vdb_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#if defined(__AVX2__) && defined(__FMA__)
#include <immintrin.h>

#include <condition_variable>
#include <functional>
#include <mutex>

// ---------------------------------------------------------------------------
// Host-side execution helpers of the CPU baseline (the reference runs its batch paths on a rayon pool,
// index/hnsw/index/batch.rs:159-244): one persistent pool instead of a thread spawn per call, and page placement by the
// threads that will read the pages (a 3 GB corpus first-touched by one thread lives on ONE NUMA node of the host, and 256
// threads scanning it are then bound by that node's memory controllers and the socket link).
// ---------------------------------------------------------------------------
namespace {
class Pool {
 public:
  static Pool& get() {
    static Pool p;
    return p;
  }
  // runs fn(t) for t in [0, n) on n pool threads (n <= 1: inline); returns when all are done
  void run(uint32_t n, const std::function<void(uint32_t)>& fn) {
    if (n <= 1) {
      fn(0);
      return;
    }
    std::unique_lock<std::mutex> call(call_mu_);  // one parallel region at a time
    {
      std::unique_lock<std::mutex> lk(mu_);
      while (workers_.size() < n) {
        const uint32_t id = (uint32_t)workers_.size();
        workers_.emplace_back([this, id] { loop(id); });
      }
      fn_ = &fn;
      n_ = n;
      pending_ = n;
      gen_++;
    }
    cv_.notify_all();
    std::unique_lock<std::mutex> lk(mu_);
    done_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  Pool() = default;
  ~Pool() {
    {
      std::unique_lock<std::mutex> lk(mu_);
      stop_ = true;
      gen_++;
    }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  void loop(uint32_t id) {
    uint64_t seen = 0;
    for (;;) {
      const std::function<void(uint32_t)>* fn = nullptr;
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
        if (stop_) return;
        seen = gen_;
        if (id < n_) fn = fn_;
      }
      if (fn) {
        (*fn)(id);
        std::unique_lock<std::mutex> lk(mu_);
        if (--pending_ == 0) done_.notify_all();
      }
    }
  }
  std::mutex mu_, call_mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> workers_;
  const std::function<void(uint32_t)>* fn_ = nullptr;
  uint32_t n_ = 0, pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

// allocator that leaves floats uninitialised: resize() does not touch the pages, the first writer does
template <class T>
struct NoInitAlloc {
  typedef T value_type;
  NoInitAlloc() = default;
  template <class U>
  NoInitAlloc(const NoInitAlloc<U>&) {}
  T* allocate(size_t n) { return static_cast<T*>(::operator new(n * sizeof(T))); }
  void deallocate(T* p, size_t) { ::operator delete(p); }
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if (sizeof...(A)) ::new ((void*)p) U(std::forward<A>(a)...);
  }
  template <class U>
  bool operator==(const NoInitAlloc<U>&) const { return true; }
  template <class U>
  bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
}  // namespace

#define VO_HAVE_AVX2 1
#else
#define VO_HAVE_AVX2 0
#endif

namespace {

// ---------------------------------------------------------------------------
// f32::total_cmp  (native/ordered_float.rs:31-36 uses it for every heap key;
// native/graph.rs:518 and core/distance.rs:98-101 for the final sorts)
// ---------------------------------------------------------------------------
inline int32_t total_key(float f) {
  int32_t b;
  std::memcpy(&b, &f, 4);
  // Rust: left ^= (((left >> 31) as u32) >> 1) as i32  -> flips magnitude bits of negatives
  b ^= (int32_t)(((uint32_t)(b >> 31)) >> 1);
  return b;
}
inline int total_cmp(float a, float b) {
  int32_t x = total_key(a), y = total_key(b);
  return x < y ? -1 : (x > y ? 1 : 0);
}

// ---------------------------------------------------------------------------
// `wide` 0.7.33 f32x8 (third-party, absent from the checkout; Cargo.lock:7405).
// Restated from the crate's published AVX code path:
//   mul_add  -> vfmadd (fused) when built with target_feature fma (the repo forces
//               -C target-cpu=native, .cargo/config.toml:33-34); a*b+c otherwise
//   reduce_add (AVX) -> ((l0+l4)+(l2+l6)) + ((l1+l5)+(l3+l7))
// ---------------------------------------------------------------------------
#if VO_HAVE_AVX2
struct f32x8 {
  __m256 v;
};
inline f32x8 zero8() { return {_mm256_setzero_ps()}; }
inline f32x8 load8(const float* p) { return {_mm256_loadu_ps(p)}; }
template <bool FMA>
inline f32x8 mul_add(f32x8 a, f32x8 b, f32x8 c) {
  if (FMA) return {_mm256_fmadd_ps(a.v, b.v, c.v)};
  return {_mm256_add_ps(_mm256_mul_ps(a.v, b.v), c.v)};
}
inline f32x8 add8(f32x8 a, f32x8 b) { return {_mm256_add_ps(a.v, b.v)}; }
inline f32x8 sub8(f32x8 a, f32x8 b) { return {_mm256_sub_ps(a.v, b.v)}; }
inline float reduce_add(f32x8 a) {
  alignas(32) float l[8];
  _mm256_store_ps(l, a.v);
  return ((l[0] + l[4]) + (l[2] + l[6])) + ((l[1] + l[5]) + (l[3] + l[7]));
}
#else
struct f32x8 {
  float v[8];
};
inline f32x8 zero8() {
  f32x8 r;
  for (int i = 0; i < 8; i++) r.v[i] = 0.f;
  return r;
}
inline f32x8 load8(const float* p) {
  f32x8 r;
  for (int i = 0; i < 8; i++) r.v[i] = p[i];
  return r;
}
template <bool FMA>
inline f32x8 mul_add(f32x8 a, f32x8 b, f32x8 c) {
  f32x8 r;
  for (int i = 0; i < 8; i++) r.v[i] = FMA ? std::fmaf(a.v[i], b.v[i], c.v[i]) : a.v[i] * b.v[i] + c.v[i];
  return r;
}
inline f32x8 add8(f32x8 a, f32x8 b) {
  f32x8 r;
  for (int i = 0; i < 8; i++) r.v[i] = a.v[i] + b.v[i];
  return r;
}
inline f32x8 sub8(f32x8 a, f32x8 b) {
  f32x8 r;
  for (int i = 0; i < 8; i++) r.v[i] = a.v[i] - b.v[i];
  return r;
}
inline float reduce_add(f32x8 a) {
  const float* l = a.v;
  return ((l[0] + l[4]) + (l[2] + l[6])) + ((l[1] + l[5]) + (l[3] + l[7]));
}
#endif

// ---------------------------------------------------------------------------
// simd_explicit.rs:50-189 — single-accumulator f32x8 kernels (used for len<16)
// ---------------------------------------------------------------------------
template <bool FMA>
float dot_simd8(const float* a, const float* b, size_t len) {  // simd_explicit.rs:50-78
  size_t simd_len = len / 8, rem = len % 8;
  f32x8 sum = zero8();
  for (size_t i = 0; i < simd_len; i++) sum = mul_add<FMA>(load8(a + i * 8), load8(b + i * 8), sum);
  float result = reduce_add(sum);
  size_t base = simd_len * 8;
  for (size_t i = 0; i < rem; i++) result += a[base + i] * b[base + i];
  return result;
}
template <bool FMA>
float sql2_simd8(const float* a, const float* b, size_t len) {  // simd_explicit.rs:103-129
  size_t simd_len = len / 8, rem = len % 8;
  f32x8 sum = zero8();
  for (size_t i = 0; i < simd_len; i++) {
    f32x8 d = sub8(load8(a + i * 8), load8(b + i * 8));
    sum = mul_add<FMA>(d, d, sum);
  }
  float result = reduce_add(sum);
  size_t base = simd_len * 8;
  for (size_t i = 0; i < rem; i++) {
    float d = a[base + i] - b[base + i];
    result += d * d;
  }
  return result;
}
template <bool FMA>
float cosine_simd8(const float* a, const float* b, size_t len) {  // simd_explicit.rs:143-189
  size_t simd_len = len / 8, rem = len % 8;
  f32x8 ds = zero8(), nas = zero8(), nbs = zero8();
  for (size_t i = 0; i < simd_len; i++) {
    f32x8 va = load8(a + i * 8), vb = load8(b + i * 8);
    ds = mul_add<FMA>(va, vb, ds);
    nas = mul_add<FMA>(va, va, nas);
    nbs = mul_add<FMA>(vb, vb, nbs);
  }
  float dot = reduce_add(ds), na = reduce_add(nas), nb = reduce_add(nbs);
  size_t base = simd_len * 8;
  for (size_t i = 0; i < rem; i++) {
    float ai = a[base + i], bi = b[base + i];
    dot += ai * bi;
    na += ai * ai;
    nb += bi * bi;
  }
  float norm_a = std::sqrt(na), norm_b = std::sqrt(nb);
  if (norm_a == 0.0f || norm_b == 0.0f) return 0.0f;
  return dot / (norm_a * norm_b);
}

// ---------------------------------------------------------------------------
// simd_avx512.rs:150-352 — the "wide16" kernels the production engine executes:
// 4 x f32x8 accumulators (32 floats / iteration), pairwise combine, then 8-wide
// and scalar tails.
// ---------------------------------------------------------------------------
template <bool FMA>
float dot_wide16(const float* a, const float* b, size_t len) {  // simd_avx512.rs:150-204
  size_t simd_len = len / 32;
  f32x8 s0 = zero8(), s1 = zero8(), s2 = zero8(), s3 = zero8();
  for (size_t i = 0; i < simd_len; i++) {
    size_t o = i * 32;
    s0 = mul_add<FMA>(load8(a + o), load8(b + o), s0);
    s1 = mul_add<FMA>(load8(a + o + 8), load8(b + o + 8), s1);
    s2 = mul_add<FMA>(load8(a + o + 16), load8(b + o + 16), s2);
    s3 = mul_add<FMA>(load8(a + o + 24), load8(b + o + 24), s3);
  }
  float result = reduce_add(add8(add8(s0, s1), add8(s2, s3)));  // :182-184
  size_t pos = simd_len * 32;
  while (pos + 8 <= len) {  // :190-195
    result += reduce_add(mul_add<FMA>(load8(a + pos), load8(b + pos), zero8()));
    pos += 8;
  }
  while (pos < len) {  // :198-201
    result += a[pos] * b[pos];
    pos++;
  }
  return result;
}
template <bool FMA>
float sql2_wide16(const float* a, const float* b, size_t len) {  // simd_avx512.rs:208-264
  size_t simd_len = len / 32;
  f32x8 s0 = zero8(), s1 = zero8(), s2 = zero8(), s3 = zero8();
  for (size_t i = 0; i < simd_len; i++) {
    size_t o = i * 32;
    f32x8 d0 = sub8(load8(a + o), load8(b + o));
    s0 = mul_add<FMA>(d0, d0, s0);
    f32x8 d1 = sub8(load8(a + o + 8), load8(b + o + 8));
    s1 = mul_add<FMA>(d1, d1, s1);
    f32x8 d2 = sub8(load8(a + o + 16), load8(b + o + 16));
    s2 = mul_add<FMA>(d2, d2, s2);
    f32x8 d3 = sub8(load8(a + o + 24), load8(b + o + 24));
    s3 = mul_add<FMA>(d3, d3, s3);
  }
  float result = reduce_add(add8(add8(s0, s1), add8(s2, s3)));
  size_t pos = simd_len * 32;
  while (pos + 8 <= len) {
    f32x8 d = sub8(load8(a + pos), load8(b + pos));
    result += reduce_add(mul_add<FMA>(d, d, zero8()));
    pos += 8;
  }
  while (pos < len) {
    float d = a[pos] - b[pos];
    result += d * d;
    pos++;
  }
  return result;
}
template <bool FMA>
float cosine_wide16(const float* a, const float* b, size_t len) {  // simd_avx512.rs:271-352
  size_t simd_len = len / 32;
  f32x8 d0 = zero8(), d1 = zero8(), d2 = zero8(), d3 = zero8();
  f32x8 na0 = zero8(), na1 = zero8(), na2 = zero8(), na3 = zero8();
  f32x8 nb0 = zero8(), nb1 = zero8(), nb2 = zero8(), nb3 = zero8();
  for (size_t i = 0; i < simd_len; i++) {
    size_t o = i * 32;
    f32x8 va0 = load8(a + o), vb0 = load8(b + o);
    d0 = mul_add<FMA>(va0, vb0, d0);
    na0 = mul_add<FMA>(va0, va0, na0);
    nb0 = mul_add<FMA>(vb0, vb0, nb0);
    f32x8 va1 = load8(a + o + 8), vb1 = load8(b + o + 8);
    d1 = mul_add<FMA>(va1, vb1, d1);
    na1 = mul_add<FMA>(va1, va1, na1);
    nb1 = mul_add<FMA>(vb1, vb1, nb1);
    f32x8 va2 = load8(a + o + 16), vb2 = load8(b + o + 16);
    d2 = mul_add<FMA>(va2, vb2, d2);
    na2 = mul_add<FMA>(va2, va2, na2);
    nb2 = mul_add<FMA>(vb2, vb2, nb2);
    f32x8 va3 = load8(a + o + 24), vb3 = load8(b + o + 24);
    d3 = mul_add<FMA>(va3, vb3, d3);
    na3 = mul_add<FMA>(va3, va3, na3);
    nb3 = mul_add<FMA>(vb3, vb3, nb3);
  }
  float dot = reduce_add(add8(add8(d0, d1), add8(d2, d3)));      // :318
  float na = reduce_add(add8(add8(na0, na1), add8(na2, na3)));   // :319
  float nb = reduce_add(add8(add8(nb0, nb1), add8(nb2, nb3)));   // :320
  size_t pos = simd_len * 32;
  while (pos + 8 <= len) {  // :326-333
    f32x8 va = load8(a + pos), vb = load8(b + pos);
    dot += reduce_add(mul_add<FMA>(va, vb, zero8()));
    na += reduce_add(mul_add<FMA>(va, va, zero8()));
    nb += reduce_add(mul_add<FMA>(vb, vb, zero8()));
    pos += 8;
  }
  while (pos < len) {  // :335-342
    float ai = a[pos], bi = b[pos];
    dot += ai * bi;
    na += ai * ai;
    nb += bi * bi;
    pos++;
  }
  float norm_a = std::sqrt(na), norm_b = std::sqrt(nb);
  if (norm_a == 0.0f || norm_b == 0.0f) return 0.0f;  // :347-349
  return dot / (norm_a * norm_b);
}
// simd_avx512.rs:87-138 — *_auto dispatch: wide16 for len >= 16, f32x8 otherwise
template <bool FMA>
float dot_auto(const float* a, const float* b, size_t n) {
  return n >= 16 ? dot_wide16<FMA>(a, b, n) : dot_simd8<FMA>(a, b, n);
}
template <bool FMA>
float sql2_auto(const float* a, const float* b, size_t n) {
  return n >= 16 ? sql2_wide16<FMA>(a, b, n) : sql2_simd8<FMA>(a, b, n);
}
template <bool FMA>
float cosine_auto(const float* a, const float* b, size_t n) {
  return n >= 16 ? cosine_wide16<FMA>(a, b, n) : cosine_simd8<FMA>(a, b, n);
}
template <bool FMA>
float normsq_wide(const float* a, size_t n) {
  return dot_auto<FMA>(a, a, n);
}

// ---------------------------------------------------------------------------
// MODE C — the canonical order shared bit-for-bit with the HIP kernels
// (velesdb_amd/csrc/*.hip).  Definition, for a vector pair of length n:
//   * element i belongs to float4-chunk c = i/4; chunk c belongs to lane c % 64;
//   * each of the 64 lanes runs ONE fmaf chain, starting from +0.0f, over its
//     elements in increasing i (elements i >= n do not exist: no padding terms);
//   * lanes are combined by the xor butterfly s = 32,16,8,4,2,1:
//     t[l] <- t[l] + t[l ^ s] for all l simultaneously; the value is t[0].
// f32 add is commutative, so every lane holds the same bits after each stage and
// a kernel may implement the butterfly as a transposed reduction.
// Both orders (R and C) satisfy every tolerance the reference's tests state.
// ---------------------------------------------------------------------------
enum { OP_DOT = 0, OP_SQL2 = 1 };
inline float butterfly64(float* t) {
  for (int s = 32; s >= 1; s >>= 1) {
    float u[64];
    for (int l = 0; l < 64; l++) u[l] = t[l] + t[l ^ s];
    std::memcpy(t, u, sizeof(u));
  }
  return t[0];
}
// The whole 64-chunk blocks run eight lanes per AVX2 register (a 8 x 4 transposition per operand and block row; the lanes of a
// register sit in the order kLanePerm leaves them in, undone before the tail and the butterfly): the same fmaf per lane in the
// same order — vfmadd is the fused operation std::fmaf is — at a fifth of the scalar loop's time.  The butterfly only follows
// lane 0's value: every lane holds the same bits after each stage (f32 add commutes), so t[0] = ((v0+v4)+(v2+v6)) + ... as written.
template <int OP>
float reduceC(const float* a, const float* b, size_t n) {
  alignas(32) float t[64];
  size_t full = n / 4;  // complete chunks
  size_t c = 0;
  if (full >= 64) {
    static const int kLanePerm[8] = {0, 2, 4, 6, 1, 3, 5, 7};  // register position p holds lane 8 g + kLanePerm[p]
    __m256 acc[8];
    for (int g = 0; g < 8; g++) acc[g] = _mm256_setzero_ps();
    // whole 64-chunk blocks: lane = c % 64 = index within block
    for (; c + 64 <= full; c += 64) {
      for (int g = 0; g < 8; g++) {
        const float* pa = a + c * 4 + g * 32;
        const float* pb = b + c * 4 + g * 32;
        __m256 x[4], y[4];
        {
          const __m256 r0 = _mm256_loadu_ps(pa), r1 = _mm256_loadu_ps(pa + 8), r2 = _mm256_loadu_ps(pa + 16), r3 = _mm256_loadu_ps(pa + 24);
          const __m256 t0 = _mm256_unpacklo_ps(r0, r1), t1 = _mm256_unpackhi_ps(r0, r1), t2 = _mm256_unpacklo_ps(r2, r3), t3 = _mm256_unpackhi_ps(r2, r3);
          x[0] = _mm256_castpd_ps(_mm256_unpacklo_pd(_mm256_castps_pd(t0), _mm256_castps_pd(t2)));
          x[1] = _mm256_castpd_ps(_mm256_unpackhi_pd(_mm256_castps_pd(t0), _mm256_castps_pd(t2)));
          x[2] = _mm256_castpd_ps(_mm256_unpacklo_pd(_mm256_castps_pd(t1), _mm256_castps_pd(t3)));
          x[3] = _mm256_castpd_ps(_mm256_unpackhi_pd(_mm256_castps_pd(t1), _mm256_castps_pd(t3)));
        }
        {
          const __m256 r0 = _mm256_loadu_ps(pb), r1 = _mm256_loadu_ps(pb + 8), r2 = _mm256_loadu_ps(pb + 16), r3 = _mm256_loadu_ps(pb + 24);
          const __m256 t0 = _mm256_unpacklo_ps(r0, r1), t1 = _mm256_unpackhi_ps(r0, r1), t2 = _mm256_unpacklo_ps(r2, r3), t3 = _mm256_unpackhi_ps(r2, r3);
          y[0] = _mm256_castpd_ps(_mm256_unpacklo_pd(_mm256_castps_pd(t0), _mm256_castps_pd(t2)));
          y[1] = _mm256_castpd_ps(_mm256_unpackhi_pd(_mm256_castps_pd(t0), _mm256_castps_pd(t2)));
          y[2] = _mm256_castpd_ps(_mm256_unpacklo_pd(_mm256_castps_pd(t1), _mm256_castps_pd(t3)));
          y[3] = _mm256_castpd_ps(_mm256_unpackhi_pd(_mm256_castps_pd(t1), _mm256_castps_pd(t3)));
        }
        __m256 v = acc[g];
        for (int e = 0; e < 4; e++) {
          if (OP == OP_SQL2) {
            const __m256 d = _mm256_sub_ps(x[e], y[e]);
            v = _mm256_fmadd_ps(d, d, v);
          } else {
            v = _mm256_fmadd_ps(x[e], y[e], v);
          }
        }
        acc[g] = v;
      }
    }
    for (int g = 0; g < 8; g++) {
      alignas(32) float tmp[8];
      _mm256_store_ps(tmp, acc[g]);
      for (int p = 0; p < 8; p++) t[8 * g + kLanePerm[p]] = tmp[p];
    }
  } else {
    for (int l = 0; l < 64; l++) t[l] = 0.0f;
  }
  for (size_t i = c * 4; i < n; i++) {
    int l = (int)((i / 4) % 64);
    float x = a[i], y = b[i];
    if (OP == OP_SQL2) {
      float d = x - y;
      t[l] = std::fmaf(d, d, t[l]);
    } else {
      t[l] = std::fmaf(x, y, t[l]);
    }
  }
  // xor butterfly 32, 16, 8 across registers, 4, 2, 1 inside one
  __m256 v[8];
  for (int g = 0; g < 8; g++) v[g] = _mm256_load_ps(t + 8 * g);
  for (int g = 0; g < 4; g++) v[g] = _mm256_add_ps(v[g], v[g + 4]);
  for (int g = 0; g < 2; g++) v[g] = _mm256_add_ps(v[g], v[g + 2]);
  __m256 w = _mm256_add_ps(v[0], v[1]);
  w = _mm256_add_ps(w, _mm256_permute2f128_ps(w, w, 0x01));
  w = _mm256_add_ps(w, _mm256_shuffle_ps(w, w, 0x4E));  // l ^ 2
  w = _mm256_add_ps(w, _mm256_shuffle_ps(w, w, 0xB1));  // l ^ 1
  return _mm256_cvtss_f32(w);
}
// (the plain statement of the same thing — tests/test_oracle_kernels.py holds the two against each other, bit for bit)
template <int OP>
float reduceC_plain(const float* a, const float* b, size_t n) {
  float t[64];
  for (int l = 0; l < 64; l++) t[l] = 0.0f;
  for (size_t i = 0; i < n; i++) {
    int l = (int)((i / 4) % 64);
    float x = a[i], y = b[i];
    if (OP == OP_SQL2) {
      float d = x - y;
      t[l] = std::fmaf(d, d, t[l]);
    } else {
      t[l] = std::fmaf(x, y, t[l]);
    }
  }
  return butterfly64(t);
}
inline float dotC(const float* a, const float* b, size_t n) { return reduceC<OP_DOT>(a, b, n); }
// matrix-core order (v_mfma_f32_16x16x4_f32 in velesdb_amd/csrc/sweep.hip sweep_topk_mfma_f32): the instruction
// adds its four k-slots as a k-ordered fmaf chain, one rounding per product; the kernel feeds slot kk of step
// (U, m, c) with element 128U + 16m + 4kk + c and pads the vectors with zeros to a multiple of 128
inline float dotM(const float* a, const float* b, size_t n) {
  float acc = 0.0f;
  const size_t KU = (n + 127) / 128;
  for (size_t U = 0; U < KU; U++)
    for (size_t m = 0; m < 8; m++)
      for (size_t c = 0; c < 4; c++)
        for (size_t kk = 0; kk < 4; kk++) {
          const size_t k = 128 * U + 16 * m + 4 * kk + c;
          const float x = k < n ? a[k] : 0.0f, y = k < n ? b[k] : 0.0f;
          acc = std::fmaf(x, y, acc);
        }
  return acc;
}
inline float sql2C(const float* a, const float* b, size_t n) { return reduceC<OP_SQL2>(a, b, n); }
inline float normsqC(const float* a, size_t n) { return reduceC<OP_DOT>(a, a, n); }
inline float cosineC(const float* a, const float* b, size_t n) {
  // same formula as simd_avx512.rs:344-351 on canonical sums
  float dot = dotC(a, b, n);
  float norm_a = std::sqrt(normsqC(a, n)), norm_b = std::sqrt(normsqC(b, n));
  if (norm_a == 0.0f || norm_b == 0.0f) return 0.0f;
  return dot / (norm_a * norm_b);
}

// ---------------------------------------------------------------------------
// simd_native.rs:37-119 — true AVX-512F shape: ONE 16-lane accumulator, masked
// tail, _mm512_reduce_add_ps.  Restated lane-wise (identical bits to the
// intrinsic sequence: vfmadd per lane; reduce = 512->256->128->64->32 halving).
// simd_native.rs:447-469 — cosine_similarity_native is a plain scalar loop.
// ---------------------------------------------------------------------------
inline float reduce16(const float* l) {
  float h8[8], h4[4];
  for (int i = 0; i < 8; i++) h8[i] = l[i] + l[i + 8];
  for (int i = 0; i < 4; i++) h4[i] = h8[i] + h8[i + 4];
  return (h4[0] + h4[2]) + (h4[1] + h4[3]);
}
// the intrinsic sequence itself (simd_native.rs:40-77,86-119): compiled for avx512f whatever the flags of this file are,
// called only when the host has it (vo_cpu_has_avx512f); bit-identical to the lane-wise restatement below
template <int OP>
__attribute__((target("avx512f"))) float native16_avx512(const float* a, const float* b, size_t n) {
  __m512 acc = _mm512_setzero_ps();
  const size_t blocks = n / 16, rem = n % 16;
  for (size_t k = 0; k < blocks; k++) {
    const __m512 x = _mm512_loadu_ps(a + k * 16), y = _mm512_loadu_ps(b + k * 16);
    if (OP == OP_SQL2) {
      const __m512 d = _mm512_sub_ps(x, y);
      acc = _mm512_fmadd_ps(d, d, acc);
    } else {
      acc = _mm512_fmadd_ps(x, y, acc);
    }
  }
  if (rem) {
    const __mmask16 m = (__mmask16)((1u << rem) - 1u);
    const __m512 x = _mm512_maskz_loadu_ps(m, a + blocks * 16), y = _mm512_maskz_loadu_ps(m, b + blocks * 16);
    if (OP == OP_SQL2) {
      const __m512 d = _mm512_sub_ps(x, y);
      acc = _mm512_fmadd_ps(d, d, acc);
    } else {
      acc = _mm512_fmadd_ps(x, y, acc);
    }
  }
  // _mm512_reduce_add_ps: 512 -> 256 -> 128 -> 64 -> 32 halving, as reduce16
  alignas(64) float l[16];
  _mm512_store_ps(l, acc);
  float h8[8], h4[4];
  for (int i = 0; i < 8; i++) h8[i] = l[i] + l[i + 8];
  for (int i = 0; i < 4; i++) h4[i] = h8[i] + h8[i + 4];
  return (h4[0] + h4[2]) + (h4[1] + h4[3]);
}
static const bool g_has_avx512f = __builtin_cpu_supports("avx512f");

template <int OP>
float native16(const float* a, const float* b, size_t n) {
  if (n < 16) {  // scalar fallback arms of *_native (simd_native.rs:399,417-424)
    float s = 0.0f;
    for (size_t i = 0; i < n; i++) {
      if (OP == OP_SQL2) {
        float d = a[i] - b[i];
        s += d * d;
      } else {
        s += a[i] * b[i];
      }
    }
    return s;
  }
  if (g_has_avx512f) return native16_avx512<OP>(a, b, n);
  float acc[16];
  for (int i = 0; i < 16; i++) acc[i] = 0.0f;
  size_t blocks = n / 16, rem = n % 16;
  for (size_t k = 0; k < blocks; k++)
    for (int i = 0; i < 16; i++) {
      float x = a[k * 16 + i], y = b[k * 16 + i];
      if (OP == OP_SQL2) {
        float d = x - y;
        acc[i] = std::fmaf(d, d, acc[i]);
      } else {
        acc[i] = std::fmaf(x, y, acc[i]);
      }
    }
  if (rem) {  // maskz load: missing lanes are 0 and still go through the fmadd
    size_t base = blocks * 16;
    for (int i = 0; i < 16; i++) {
      float x = (size_t)i < rem ? a[base + i] : 0.0f, y = (size_t)i < rem ? b[base + i] : 0.0f;
      if (OP == OP_SQL2) {
        float d = x - y;
        acc[i] = std::fmaf(d, d, acc[i]);
      } else {
        acc[i] = std::fmaf(x, y, acc[i]);
      }
    }
  }
  return reduce16(acc);
}
inline float cosine_native(const float* a, const float* b, size_t n) {  // simd_native.rs:447-469
  float dot = 0.f, na = 0.f, nb = 0.f;
  for (size_t i = 0; i < n; i++) {
    dot += a[i] * b[i];
    na += a[i] * a[i];
    nb += b[i] * b[i];
  }
  float norm_a = std::sqrt(na), norm_b = std::sqrt(nb);
  if (norm_a == 0.0f || norm_b == 0.0f) return 0.0f;
  return dot / (norm_a * norm_b);
}

// ---------------------------------------------------------------------------
// Integer-valued kernels (exact; fully pinned by the reference's KATs)
// ---------------------------------------------------------------------------
// simd_explicit.rs:234-287 — Hamming over f32: positions where (a>0.5)!=(b>0.5)
inline uint32_t hamming_u32(const float* a, const float* b, size_t n) {
  uint32_t c = 0;
  for (size_t i = 0; i < n; i++) c += (uint32_t)((a[i] > 0.5f) != (b[i] > 0.5f));
  return c;
}
// simd_explicit.rs:372-443 — Jaccard over f32 thresholded at 0.5; empty union -> 1.0
inline float jaccard_sim(const float* a, const float* b, size_t n) {
  uint32_t inter = 0, uni = 0;
  for (size_t i = 0; i < n; i++) {
    bool x = a[i] > 0.5f, y = b[i] > 0.5f;
    inter += (uint32_t)(x && y);
    uni += (uint32_t)(x || y);
  }
  if (uni == 0) return 1.0f;
  return (float)inter / (float)uni;
}

// ---------------------------------------------------------------------------
// native/distance.rs:158-217 — CpuDistance scalar engine (returns DISTANCES)
// ---------------------------------------------------------------------------
inline float scalar_engine_distance(int metric, const float* a, const float* b, size_t n) {
  switch (metric) {
    case VO_COSINE: {  // :159-176
      float dot = 0.f, na = 0.f, nb = 0.f;
      for (size_t i = 0; i < n; i++) {
        dot += a[i] * b[i];
        na += a[i] * a[i];
        nb += b[i] * b[i];
      }
      float denom = std::sqrt(na * nb);
      return denom == 0.0f ? 1.0f : 1.0f - (dot / denom);
    }
    case VO_EUCLIDEAN: {  // :178-185  (x-y).powi(2) == (x-y)*(x-y)
      float s = 0.f;
      for (size_t i = 0; i < n; i++) {
        float d = a[i] - b[i];
        s += d * d;
      }
      return std::sqrt(s);
    }
    case VO_DOT: {  // :187-191
      float s = 0.f;
      for (size_t i = 0; i < n; i++) s += a[i] * b[i];
      return -s;
    }
    case VO_HAMMING: {  // :193-200  bit-pattern inequality (a different function from SimdDistance's)
      uint32_t c = 0;
      for (size_t i = 0; i < n; i++) {
        uint32_t x, y;
        std::memcpy(&x, a + i, 4);
        std::memcpy(&y, b + i, 4);
        c += (x ^ y) != 0;
      }
      return (float)c;
    }
    default: {  // Jaccard :202-217 (min/max form)
      float inter = 0.f, uni = 0.f;
      for (size_t i = 0; i < n; i++) {
        inter += std::min(a[i], b[i]);
        uni += std::max(a[i], b[i]);
      }
      return uni == 0.0f ? 1.0f : 1.0f - (inter / uni);
    }
  }
}

// ---------------------------------------------------------------------------
// mode dispatch for the similarity primitives
// ---------------------------------------------------------------------------
inline float k_dot(int mode, const float* a, const float* b, size_t n) {
  switch (mode) {
    case VO_MODE_C: return dotC(a, b, n);
    case VO_MODE_M: return dotM(a, b, n);
    case VO_MODE_NATIVE: return native16<OP_DOT>(a, b, n);
    case VO_MODE_R_NOFMA: return dot_auto<false>(a, b, n);
    case VO_MODE_SCALAR: {
      float s = 0.f;
      for (size_t i = 0; i < n; i++) s += a[i] * b[i];
      return s;
    }
    default: return dot_auto<true>(a, b, n);
  }
}
inline float k_sql2(int mode, const float* a, const float* b, size_t n) {
  switch (mode) {
    case VO_MODE_C: return sql2C(a, b, n);
    case VO_MODE_M: return sql2C(a, b, n);
    case VO_MODE_NATIVE: return native16<OP_SQL2>(a, b, n);
    case VO_MODE_R_NOFMA: return sql2_auto<false>(a, b, n);
    case VO_MODE_SCALAR: {
      float s = 0.f;
      for (size_t i = 0; i < n; i++) {
        float d = a[i] - b[i];
        s += d * d;
      }
      return s;
    }
    default: return sql2_auto<true>(a, b, n);
  }
}
inline float k_cosine(int mode, const float* a, const float* b, size_t n) {
  switch (mode) {
    case VO_MODE_C: return cosineC(a, b, n);
    case VO_MODE_M: {
      float dot = dotM(a, b, n);
      float norm_a = std::sqrt(normsqC(a, n)), norm_b = std::sqrt(normsqC(b, n));
      if (norm_a == 0.0f || norm_b == 0.0f) return 0.0f;
      return dot / (norm_a * norm_b);
    }
    case VO_MODE_NATIVE: return cosine_native(a, b, n);
    case VO_MODE_R_NOFMA: return cosine_auto<false>(a, b, n);
    case VO_MODE_SCALAR: return 1.0f - scalar_engine_distance(VO_COSINE, a, b, n);
    default: return cosine_auto<true>(a, b, n);
  }
}

// DistanceEngine::distance — native/distance.rs:75-85 (SimdDistance), :126-136
// (NativeSimdDistance), :44-52 (CpuDistance)
inline float engine_distance(int metric, int mode, const float* a, const float* b, size_t n) {
  if (mode == VO_MODE_SCALAR) return scalar_engine_distance(metric, a, b, n);
  switch (metric) {
    case VO_COSINE: return 1.0f - k_cosine(mode, a, b, n);
    case VO_EUCLIDEAN: return std::sqrt(k_sql2(mode, a, b, n));
    case VO_DOT: return -k_dot(mode, a, b, n);
    case VO_HAMMING: return (float)hamming_u32(a, b, n);
    default: return 1.0f - jaccard_sim(a, b, n);
  }
}
// HnswIndex::compute_distance — hnsw/index/search.rs:30-38 (raw *_fast values)
inline float index_compute_distance(int metric, int mode, const float* a, const float* b, size_t n) {
  int m = (mode == VO_MODE_SCALAR) ? VO_MODE_R : mode;  // the index always uses simd::*_fast
  switch (metric) {
    case VO_COSINE: return k_cosine(m, a, b, n);
    case VO_EUCLIDEAN: return std::sqrt(k_sql2(m, a, b, n));
    case VO_DOT: return k_dot(m, a, b, n);
    case VO_HAMMING: return (float)hamming_u32(a, b, n);
    default: return jaccard_sim(a, b, n);
  }
}
inline bool higher_is_better(int metric) {  // core/distance.rs:76-82
  return metric == VO_COSINE || metric == VO_DOT || metric == VO_JACCARD;
}
inline float transform_score(int metric, float d) {  // native/backend_adapter.rs:160-168
  switch (metric) {
    case VO_COSINE: {
      float s = 1.0f - d;  // f32::clamp(0,1): NaN stays NaN
      if (s < 0.0f) s = 0.0f;
      if (s > 1.0f) s = 1.0f;
      return s;
    }
    case VO_DOT: return -d;
    default: return d;
  }
}

// ---------------------------------------------------------------------------
// Rust alloc::collections::BinaryHeap (std source is not in the checkout; restated
// from its published algorithm): push = append + sift_up; pop = take last, swap
// with root, sift_down_to_bottom (always descend to the greater child, `<=` picks
// the right one on ties) then sift_up; into_iter()/into_vec() = backing-array order.
// Key = (OrderedFloat, NodeId) tuple order (native/graph.rs:449-450); MIN = Reverse<>.
// ---------------------------------------------------------------------------
struct HeapItem {
  float d;
  uint64_t node;
};
inline int item_cmp(const HeapItem& a, const HeapItem& b) {
  int c = total_cmp(a.d, b.d);
  if (c) return c;
  return a.node < b.node ? -1 : (a.node > b.node ? 1 : 0);
}
template <bool MIN>
struct RustHeap {
  std::vector<HeapItem> data;
  static bool le(const HeapItem& a, const HeapItem& b) {  // a <= b in heap order
    int c = item_cmp(a, b);
    return MIN ? c >= 0 : c <= 0;
  }
  size_t size() const { return data.size(); }
  bool empty() const { return data.empty(); }
  const HeapItem& peek() const { return data[0]; }
  size_t sift_up(size_t start, size_t pos) {
    HeapItem elem = data[pos];
    while (pos > start) {
      size_t parent = (pos - 1) / 2;
      if (le(elem, data[parent])) break;
      data[pos] = data[parent];
      pos = parent;
    }
    data[pos] = elem;
    return pos;
  }
  void push(HeapItem it) {
    size_t old = data.size();
    data.push_back(it);
    sift_up(0, old);
  }
  void sift_down_to_bottom(size_t pos) {
    size_t end = data.size(), start = pos;
    HeapItem elem = data[pos];
    size_t child = 2 * pos + 1;
    while (end >= 2 && child <= end - 2) {
      child += le(data[child], data[child + 1]) ? 1 : 0;
      data[pos] = data[child];
      pos = child;
      child = 2 * pos + 1;
    }
    if (child == end - 1) {
      data[pos] = data[child];
      pos = child;
    }
    data[pos] = elem;
    sift_up(start, pos);
  }
  HeapItem pop() {
    HeapItem item = data.back();
    data.pop_back();
    if (!data.empty()) {
      std::swap(item, data[0]);
      sift_down_to_bottom(0);
    }
    return item;
  }
};

thread_local uint64_t tl_n_dist = 0, tl_n_expand = 0;

}  // namespace

// ===========================================================================
// NativeHnsw<D> — native/graph.rs, native/layer.rs, native/backend_adapter.rs
// ===========================================================================
struct vo_hnsw {
  uint32_t dim = 0;
  int metric = 0, mode = 0;
  std::vector<float, NoInitAlloc<float>> vectors;            // graph.rs:22 (flattened; pages placed by their first writer)
  std::vector<std::vector<std::vector<uint64_t>>> layers;    // graph.rs:24, layer.rs:12-15
  int64_t entry_point = -1;                                  // graph.rs:26
  size_t max_layer = 0, count = 0;                           // graph.rs:28-30
  uint64_t rng_state = 0x5DEECE66D1A4B5B5ULL;                // graph.rs:72
  size_t M = 0, M0 = 0, efc = 0;                             // graph.rs:34-38, M0 = 2M :62
  double level_mult = 0.0;                                   // graph.rs:63
  float alpha = 1.0f;                                        // graph.rs:77
  int build_tie = VO_TIE_REFERENCE;                          // order among equal distances fed to select_neighbors
  uint32_t build_threads = 1;                                // host threads of hnsw_insert_batch_sync's search phase (results do not depend on it)
  // Cosine in modes C / M: sqrt(canonical sum of squares) of every stored vector, computed once where the vector is appended
  // (note_vectors) instead of inside every distance — the value cosineC computes, so the bits of a distance do not change
  std::vector<float> cnorm;
  void note_vectors();
  bool norm_of(const float* p, float* out) const {
    const float* base = vectors.data();
    if (p < base || p >= base + vectors.size()) return false;
    const size_t off = (size_t)(p - base);
    if (off % dim != 0 || off / dim >= cnorm.size()) return false;
    *out = cnorm[off / dim];
    return true;
  }

  const float* vec(uint64_t id) const { return vectors.data() + (size_t)id * dim; }
  float dist(const float* a, const float* b) const {
    tl_n_dist++;
    if (metric == VO_COSINE && (mode == VO_MODE_C || mode == VO_MODE_M) && !cnorm.empty()) return cosine_distance_cached(a, b);
    return engine_distance(metric, mode, a, b, dim);
  }
  float cosine_distance_cached(const float* a, const float* b) const;
  const std::vector<uint64_t>& nbrs(size_t layer, uint64_t node) const {  // layer.rs:33-39
    static const std::vector<uint64_t> empty;
    if (layer >= layers.size() || node >= layers[layer].size()) return empty;
    return layers[layer][node];
  }
};

void vo_hnsw::note_vectors() {
  if (metric != VO_COSINE || (mode != VO_MODE_C && mode != VO_MODE_M)) return;
  const size_t nvec = vectors.size() / dim;
  if (cnorm.size() > nvec) cnorm.clear();
  for (size_t i = cnorm.size(); i < nvec; i++) cnorm.push_back(std::sqrt(normsqC(vec(i), dim)));
}
// k_cosine's modes C / M with the stored vectors' norms looked up (same values, same formula => same bits)
float vo_hnsw::cosine_distance_cached(const float* a, const float* b) const {
  float na, nb;
  if (!norm_of(a, &na)) na = std::sqrt(normsqC(a, dim));
  if (!norm_of(b, &nb)) nb = std::sqrt(normsqC(b, dim));
  const float dot = mode == VO_MODE_M ? dotM(a, b, dim) : dotC(a, b, dim);
  const float cosv = (na == 0.0f || nb == 0.0f) ? 0.0f : dot / (na * nb);
  return 1.0f - cosv;
}

namespace {

// graph.rs:368-403
uint64_t xorshift_next(uint64_t& state) {
  uint64_t s = state;
  if (s == 0) s = 0x853c49e6748fea9bULL;
  s ^= s << 13;
  s ^= s >> 7;
  s ^= s << 17;
  state = s;
  return s;
}
uint32_t random_layer(uint64_t& state, double level_mult) {
  uint64_t s = xorshift_next(state);
  double uniform = (double)s / (double)UINT64_MAX;  // (state as f64) / (u64::MAX as f64)
  double safe = std::max(uniform, std::numeric_limits<double>::min());
  double lv = std::floor(-std::log(safe) * level_mult);
  // Rust `as usize` saturates; level >= 0 here
  size_t level = lv >= 18446744073709551615.0 ? SIZE_MAX : (size_t)lv;
  return (uint32_t)std::min<size_t>(level, 15);
}

// graph.rs:405-428
uint64_t search_layer_single(const vo_hnsw& g, const float* q, uint64_t entry, size_t layer) {
  uint64_t best = entry;
  float best_dist = g.dist(q, g.vec(entry));
  for (;;) {
    const std::vector<uint64_t> neighbors = g.nbrs(layer, best);  // clone, as get_neighbors does
    bool improved = false;
    for (uint64_t nb : neighbors) {
      float d = g.dist(q, g.vec(nb));
      if (d < best_dist) {
        best = nb;
        best_dist = d;
        improved = true;
      }
    }
    if (!improved) break;
  }
  return best;
}

// graph.rs:438-520.  Prefetching (:482-497) is performance-only and omitted.
std::vector<std::pair<uint64_t, float>> search_layer(const vo_hnsw& g, const float* q,
                                                     const std::vector<uint64_t>& eps, size_t ef,
                                                     size_t layer, int tie) {
  // visited set (membership only): stamps of the calling thread — an epoch per call, so one array serves every graph the thread
  // searches, and the batch-synchronous build may run its searches on several threads
  thread_local std::vector<uint32_t> stamp;
  thread_local uint32_t epoch = 0;
  size_t nvec = g.vectors.size() / g.dim;
  if (stamp.size() < nvec) stamp.resize(nvec, 0);
  if (++epoch == 0) {
    std::fill(stamp.begin(), stamp.end(), 0);
    epoch = 1;
  }
  auto visit = [&](uint64_t n) -> bool {  // FxHashSet::insert -> true if newly inserted
    if (stamp[n] == epoch) return false;
    stamp[n] = epoch;
    return true;
  };
  RustHeap<true> candidates;   // BinaryHeap<Reverse<(OrderedFloat, NodeId)>>
  RustHeap<false> results;     // BinaryHeap<(OrderedFloat, NodeId)>
  for (uint64_t ep : eps) {    // :464-469 (pushes happen even for a repeated ep)
    float d = g.dist(q, g.vec(ep));
    candidates.push({d, ep});
    results.push({d, ep});
    visit(ep);
  }
  while (!candidates.empty()) {  // :471
    HeapItem c = candidates.pop();
    float furthest = results.empty() ? std::numeric_limits<float>::max() : results.peek().d;
    if (c.d > furthest && results.size() >= ef) break;  // :474 raw f32 compare
    tl_n_expand++;
    const std::vector<uint64_t>& neighbors = g.nbrs(layer, c.node);  // :478
    for (uint64_t nb : neighbors) {
      if (visit(nb)) {  // :499
        float d = g.dist(q, g.vec(nb));
        float far = results.empty() ? std::numeric_limits<float>::max() : results.peek().d;
        if (d < far || results.size() < ef) {  // :503
          candidates.push({d, nb});
          results.push({d, nb});
          if (results.size() > ef) results.pop();  // :507-509
        }
      }
    }
  }
  std::vector<std::pair<uint64_t, float>> out;  // :516-519
  out.reserve(results.size());
  for (const HeapItem& it : results.data) out.emplace_back(it.node, it.d);
  if (tie == VO_TIE_CANONICAL) {
    std::sort(out.begin(), out.end(), [](const auto& a, const auto& b) {
      int c = total_cmp(a.second, b.second);
      return c ? c < 0 : a.first < b.first;
    });
  } else {
    std::stable_sort(out.begin(), out.end(),
                     [](const auto& a, const auto& b) { return total_cmp(a.second, b.second) < 0; });
  }
  return out;
}

// graph.rs:526-581
std::vector<uint64_t> select_neighbors(const vo_hnsw& g,
                                       const std::vector<std::pair<uint64_t, float>>& cand,
                                       size_t max_neighbors) {
  std::vector<uint64_t> selected;
  if (cand.empty()) return selected;
  if (cand.size() <= max_neighbors) {
    for (auto& c : cand) selected.push_back(c.first);
    return selected;
  }
  for (auto& c : cand) {
    if (selected.size() >= max_neighbors) break;
    const float* cv = g.vec(c.first);
    bool diverse = true;
    for (uint64_t s : selected) {  // .all() short-circuits on the first failure
      float ds = g.dist(cv, g.vec(s));
      if (!(g.alpha * c.second <= ds)) {
        diverse = false;
        break;
      }
    }
    if (diverse || selected.empty()) selected.push_back(c.first);
  }
  if (selected.size() < max_neighbors) {  // :569-578 back-fill
    for (auto& c : cand) {
      if (selected.size() >= max_neighbors) break;
      if (std::find(selected.begin(), selected.end(), c.first) == selected.end())
        selected.push_back(c.first);
    }
  }
  return selected;
}

// graph.rs:592-639
void add_bidirectional_connection(vo_hnsw& g, uint64_t new_node, uint64_t neighbor, size_t layer,
                                  size_t max_conn) {
  std::vector<uint64_t>& cur = g.layers[layer][neighbor];
  if (cur.size() < max_conn) {
    cur.push_back(new_node);
    return;
  }
  std::vector<uint64_t> all = cur;
  all.push_back(new_node);
  const float* nv = g.vec(neighbor);
  std::vector<std::pair<uint64_t, float>> with_dist;
  with_dist.reserve(all.size());
  for (uint64_t n : all) with_dist.emplace_back(n, g.dist(nv, g.vec(n)));
  std::stable_sort(with_dist.begin(), with_dist.end(),
                   [](const auto& a, const auto& b) { return total_cmp(a.second, b.second) < 0; });
  std::vector<uint64_t> pruned;
  for (size_t i = 0; i < with_dist.size() && i < max_conn; i++) pruned.push_back(with_dist[i].first);
  cur = std::move(pruned);
}

// graph.rs:158-237
uint64_t hnsw_insert(vo_hnsw& g, const float* v) {
  uint64_t node_id = g.vectors.size() / g.dim;  // :160-165 id = insertion order
  g.vectors.insert(g.vectors.end(), v, v + g.dim);
  g.note_vectors();
  size_t node_layer = random_layer(g.rng_state, g.level_mult);  // :168
  while (g.layers.size() <= node_layer) g.layers.emplace_back();  // :171-179
  for (auto& L : g.layers)
    if (L.size() <= node_id) L.resize(node_id + 1);
  const float* qv = g.vec(node_id);
  if (g.entry_point >= 0) {
    uint64_t cur = (uint64_t)g.entry_point;
    size_t max_layer = g.max_layer;
    for (size_t l = max_layer; l >= node_layer + 1 && l > 0; l--)  // :189-192
      cur = search_layer_single(g, qv, cur, l);
    for (size_t li = node_layer + 1; li-- > 0;) {  // :195-223  (0..=node_layer).rev()
      auto neighbors = search_layer(g, qv, {cur}, g.efc, li, g.build_tie);
      size_t max_conn = li == 0 ? g.M0 : g.M;
      std::vector<uint64_t> selected = select_neighbors(g, neighbors, max_conn);
      g.layers[li][node_id] = selected;  // set_neighbors :213
      for (uint64_t nb : selected) add_bidirectional_connection(g, node_id, nb, li, max_conn);
      if (!neighbors.empty()) cur = neighbors[0].first;  // :220-222
    }
  } else {
    g.entry_point = (int64_t)node_id;  // :226
  }
  if (node_layer > g.max_layer) {  // :230-233
    g.max_layer = node_layer;
    g.entry_point = (int64_t)node_id;
  }
  g.count++;
  return node_id;
}

// ---------------------------------------------------------------------------
// Batch-synchronous insertion: the deterministic stand-in for NativeHnsw::parallel_insert
// (backend_adapter.rs:110-123, rayon; non-deterministic in the reference).  Every node of the
// batch runs the searches + select_neighbors of insert() (graph.rs:183-223) against the graph as
// it was BEFORE the batch (same entry point / max layer for all); then the links are applied with
// the reference's own rules (set_neighbors, add_bidirectional_connection), sources in ascending
// node order.  A batch of one node is exactly hnsw_insert().  This is the algorithm of the GPU's
// batched construction (velesdb_amd/csrc/hnsw_build.hip), restated here so it can be checked
// link for link.
// ---------------------------------------------------------------------------
void hnsw_insert_batch_sync(vo_hnsw& g, const float* vecs, size_t n) {
  if (n == 0) return;
  const uint64_t first = g.vectors.size() / g.dim;
  g.vectors.insert(g.vectors.end(), vecs, vecs + n * g.dim);
  g.note_vectors();
  std::vector<size_t> level(n);
  size_t top = 0;
  for (size_t b = 0; b < n; b++) {
    level[b] = random_layer(g.rng_state, g.level_mult);
    top = std::max(top, level[b]);
  }
  while (g.layers.size() <= top) g.layers.emplace_back();
  for (auto& L : g.layers)
    if (L.size() < first + n) L.resize(first + n);
  struct Sel {
    uint64_t node;
    float dist;
  };
  std::vector<std::vector<std::vector<Sel>>> sel(n);  // [b][layer] -> selected (id, dist to the new node)
  const int64_t ep0 = g.entry_point;
  const size_t max0 = g.max_layer;
  // every node of the batch against the graph as it was before the batch: read-only on g, sel[b] is the node's own => the
  // nodes may be searched on several host threads (build_threads; the result is the same for any number)
  auto search_one = [&](size_t b) {
    const uint64_t node_id = first + b;
    const float* qv = g.vec(node_id);
    const size_t node_layer = level[b];
    sel[b].resize(node_layer + 1);
    uint64_t cur = (uint64_t)ep0;
    for (size_t l = max0; l >= node_layer + 1 && l > 0; l--) cur = search_layer_single(g, qv, cur, l);
    for (size_t li = node_layer + 1; li-- > 0;) {
      auto neighbors = search_layer(g, qv, {cur}, g.efc, li, g.build_tie);
      size_t max_conn = li == 0 ? g.M0 : g.M;
      std::vector<uint64_t> selected = select_neighbors(g, neighbors, max_conn);
      for (uint64_t s : selected) {
        float d = 0.f;
        for (auto& p : neighbors)
          if (p.first == s) d = p.second;
        sel[b][li].push_back({s, d});
      }
      if (!neighbors.empty()) cur = neighbors[0].first;
    }
  };
  if (ep0 >= 0) {
    const uint32_t nt = (uint32_t)std::min<size_t>(std::max<uint32_t>(g.build_threads, 1), n);
    if (nt <= 1) {
      for (size_t b = 0; b < n; b++) search_one(b);
    } else {
      std::atomic<size_t> next{0};
      Pool::get().run(nt, [&](uint32_t) {
        for (size_t b; (b = next.fetch_add(1)) < n;) search_one(b);
      });
    }
  }
  // Back-links.  A target is a node of the graph before the batch (nothing of the batch was visible to the searches), and its list
  // only depends on the sources that select it, in ascending source order: with several build threads the (layer, target) groups
  // are applied side by side, each in that order — the lists the sequential loop below leaves.
  const uint32_t apply_threads = (uint32_t)std::min<size_t>(std::max<uint32_t>(g.build_threads, 1), n);
  const bool grouped = ep0 >= 0 && apply_threads > 1;
  if (grouped) {
    struct Op {
      uint64_t key;  // layer << 40 | target
      uint64_t src;
    };
    std::vector<Op> ops;
    for (size_t b = 0; b < n; b++)
      for (size_t li = level[b] + 1; li-- > 0;) {
        std::vector<uint64_t> ids;
        for (auto& s : sel[b][li]) {
          ids.push_back(s.node);
          ops.push_back({((uint64_t)li << 40) | s.node, first + b});
        }
        g.layers[li][first + b] = ids;  // set_neighbors :213 (the new node's own list: nobody else's target)
      }
    std::stable_sort(ops.begin(), ops.end(), [](const Op& x, const Op& y) { return x.key < y.key; });  // sources stay ascending inside a group
    std::vector<size_t> starts;
    for (size_t i = 0; i < ops.size(); i++)
      if (i == 0 || ops[i].key != ops[i - 1].key) starts.push_back(i);
    starts.push_back(ops.size());
    std::atomic<size_t> next{0};
    Pool::get().run(apply_threads, [&](uint32_t) {
      for (size_t gi; (gi = next.fetch_add(1)) + 1 < starts.size();) {
        const size_t li = (size_t)(ops[starts[gi]].key >> 40);
        const uint64_t target = ops[starts[gi]].key & ((1ull << 40) - 1);
        const size_t max_conn = li == 0 ? g.M0 : g.M;
        for (size_t i = starts[gi]; i < starts[gi + 1]; i++) add_bidirectional_connection(g, ops[i].src, target, li, max_conn);
      }
    });
  }
  for (size_t b = 0; b < n; b++) {
    const uint64_t node_id = first + b;
    if (grouped) {
    } else if (ep0 >= 0) {
      for (size_t li = level[b] + 1; li-- > 0;) {
        size_t max_conn = li == 0 ? g.M0 : g.M;
        std::vector<uint64_t> ids;
        for (auto& s : sel[b][li]) ids.push_back(s.node);
        g.layers[li][node_id] = ids;
        for (uint64_t nb : ids) add_bidirectional_connection(g, node_id, nb, li, max_conn);
      }
    } else if (g.entry_point < 0) {
      g.entry_point = (int64_t)node_id;  // first node of an empty graph (graph.rs:226)
    }
    if (level[b] > g.max_layer) {
      g.max_layer = level[b];
      g.entry_point = (int64_t)node_id;
    }
    g.count++;
  }
}

// deterministic batch schedule of the batched build: one node at a time while the graph is tiny,
// then a sixteenth of the linked nodes per batch, capped
uint32_t build_batch_size(uint64_t linked, uint32_t max_batch) {
  uint64_t b = linked / 16;
  if (b < 1) b = 1;
  if (b > max_batch) b = max_batch;
  return (uint32_t)b;
}

// graph.rs:251-270
std::vector<std::pair<uint64_t, float>> hnsw_search(const vo_hnsw& g, const float* q, size_t k,
                                                    size_t ef, int tie) {
  std::vector<std::pair<uint64_t, float>> out;
  if (g.entry_point < 0) return out;
  uint64_t cur = (uint64_t)g.entry_point;
  for (size_t l = g.max_layer; l >= 1; l--) cur = search_layer_single(g, q, cur, l);
  out = search_layer(g, q, {cur}, ef, 0, tie);
  if (out.size() > k) out.resize(k);
  return out;
}

// graph.rs:288-348: the descent's result plus up to num_probes.min(4) - 1 ids drawn from the graph's OWN xorshift stream (the one
// random_layer advances — without its zero-state reseed, :320-334), duplicates skipped, all of them entry points of ONE
// search_layer with the full ef
std::vector<std::pair<uint64_t, float>> hnsw_search_multi_entry(vo_hnsw& g, const float* q, size_t k, size_t ef, size_t num_probes,
                                                                int tie) {
  std::vector<std::pair<uint64_t, float>> out;
  if (g.entry_point < 0) return out;
  const uint64_t count = g.count;
  if (count == 0) return out;
  uint64_t cur = (uint64_t)g.entry_point;
  for (size_t l = g.max_layer; l >= 1; l--) cur = search_layer_single(g, q, cur, l);
  std::vector<uint64_t> eps{cur};
  if (num_probes > 1 && count > 10) {
    for (size_t p = 1; p < std::min<size_t>(num_probes, 4); p++) {
      uint64_t s = g.rng_state;
      s ^= s << 13;
      s ^= s >> 7;
      s ^= s << 17;
      g.rng_state = s;
      const uint64_t id = s % count;
      if (std::find(eps.begin(), eps.end(), id) == eps.end()) eps.push_back(id);
    }
  }
  out = search_layer(g, q, eps, ef, 0, tie);
  if (out.size() > k) out.resize(k);
  return out;
}

}  // namespace


// ===========================================================================
// HnswIndex — hnsw/index/{mod,trait_impl,search,batch}.rs + sharded_mappings.rs
// ===========================================================================
struct vo_index {
  vo_hnsw g;
  std::unordered_map<uint64_t, uint64_t> id_to_idx;  // sharded_mappings.rs:32-39
  std::vector<uint64_t> idx_to_id;
  std::vector<uint8_t> idx_live;
  uint64_t next_idx = 0;
  size_t live = 0;
};

namespace {

void hnsw_init(vo_hnsw& g, uint32_t dim, int metric, int mode, uint32_t M, uint32_t efc) {
  g.dim = dim;
  g.metric = metric;
  g.mode = mode;
  g.M = M;
  g.M0 = (size_t)M * 2;                        // graph.rs:62
  g.efc = efc;
  g.level_mult = 1.0 / std::log((double)M);    // graph.rs:63
  g.layers.emplace_back();                      // graph.rs:68 vec![Layer::new(..)]
}

size_t ef_search(int quality, size_t custom, size_t k) {  // hnsw/params.rs:309-319
  switch (quality) {
    case 0: return std::max<size_t>(64, k * 2);
    case 1: return std::max<size_t>(128, k * 4);
    case 2: return std::max<size_t>(512, k * 16);
    case 3: return std::max<size_t>(4096, k * 100);
    default: return std::max(custom, k);
  }
}

// core/distance.rs:95-103 — stable sort, direction by metric.  Ties: the reference
// order among equal scores is shard/hash-iteration order (sharded_vectors.rs:229-241),
// an artefact we do not reproduce; rows are visited in idx order => (score, idx asc).
void sort_results(int metric, std::vector<std::pair<uint64_t, float>>& r) {
  if (higher_is_better(metric))
    std::stable_sort(r.begin(), r.end(),
                     [](const auto& a, const auto& b) { return total_cmp(b.second, a.second) < 0; });
  else
    std::stable_sort(r.begin(), r.end(),
                     [](const auto& a, const auto& b) { return total_cmp(a.second, b.second) < 0; });
}

// hnsw/index/search.rs:176-219
std::vector<std::pair<uint64_t, float>> index_brute_force(const vo_index& ix, const float* q, size_t k) {
  std::vector<std::pair<uint64_t, float>> res;
  const vo_hnsw& g = ix.g;
  size_t n = g.vectors.size() / g.dim;
  res.reserve(n);
  for (size_t idx = 0; idx < n; idx++) {
    if (!ix.idx_live[idx]) continue;  // mappings.get_id(idx) == None -> skipped (:205)
    res.emplace_back(ix.idx_to_id[idx], index_compute_distance(g.metric, g.mode, q, g.vec(idx), g.dim));
  }
  sort_results(g.metric, res);
  if (res.size() > k) res.resize(k);
  return res;
}

// hnsw/index/search.rs:79-93 — map node -> external id (dropping soft-deleted) + transform_score
std::vector<std::pair<uint64_t, float>> index_hnsw_search(const vo_index& ix, const float* q, size_t k,
                                                         size_t ef, int tie) {
  auto nb = hnsw_search(ix.g, q, k, ef, tie);
  std::vector<std::pair<uint64_t, float>> res;
  for (auto& p : nb)
    if (p.first < ix.idx_live.size() && ix.idx_live[p.first])
      res.emplace_back(ix.idx_to_id[p.first], transform_score(ix.g.metric, p.second));
  return res;
}

// hnsw/index/search.rs:59-94
std::vector<std::pair<uint64_t, float>> index_search_with_quality(const vo_index& ix, const float* q,
                                                                  size_t k, int quality, size_t custom,
                                                                  int tie) {
  if (quality == 3) return index_brute_force(ix, q, k);                      // :68-70
  if (ix.live <= 100 && !ix.g.vectors.empty()) return index_brute_force(ix, q, k);  // :75-77
  return index_hnsw_search(ix, q, k, ef_search(quality, custom, k), tie);
}

}  // namespace

extern "C" {

float vo_dot(int mode, const float* a, const float* b, size_t n) { return k_dot(mode, a, b, n); }
float vo_sql2(int mode, const float* a, const float* b, size_t n) { return k_sql2(mode, a, b, n); }
/* mode C's sums by the plain per-element loop (the AVX2 form of reduceC is held against it) */
float vo_dot_c_plain(const float* a, const float* b, size_t n) { return reduceC_plain<OP_DOT>(a, b, n); }
float vo_sql2_c_plain(const float* a, const float* b, size_t n) { return reduceC_plain<OP_SQL2>(a, b, n); }
float vo_euclidean(int mode, const float* a, const float* b, size_t n) { return std::sqrt(k_sql2(mode, a, b, n)); }
float vo_cosine(int mode, const float* a, const float* b, size_t n) { return k_cosine(mode, a, b, n); }
float vo_norm_sq(int mode, const float* a, size_t n) { return k_dot(mode, a, a, n); }
float vo_norm(const float* a, size_t n) {  // core/simd.rs:240-242  v.iter().map(|x| x*x).sum().sqrt()
  float s = 0.f;
  for (size_t i = 0; i < n; i++) s += a[i] * a[i];
  return std::sqrt(s);
}
float vo_hamming(const float* a, const float* b, size_t n) { return (float)hamming_u32(a, b, n); }
float vo_jaccard(const float* a, const float* b, size_t n) { return jaccard_sim(a, b, n); }
uint32_t vo_hamming_binary(const uint64_t* a, const uint64_t* b, size_t n) {  // simd_explicit.rs:308-317
  uint32_t c = 0;
  for (size_t i = 0; i < n; i++) c += (uint32_t)__builtin_popcountll(a[i] ^ b[i]);
  return c;
}
float vo_dot_simd8(const float* a, const float* b, size_t n) { return dot_simd8<true>(a, b, n); }
float vo_sql2_simd8(const float* a, const float* b, size_t n) { return sql2_simd8<true>(a, b, n); }
float vo_cosine_simd8(const float* a, const float* b, size_t n) { return cosine_simd8<true>(a, b, n); }
float vo_distance(int metric, int mode, const float* a, const float* b, size_t n) {
  return engine_distance(metric, mode, a, b, n);
}
float vo_compute_distance(int metric, int mode, const float* a, const float* b, size_t n) {
  return index_compute_distance(metric, mode, a, b, n);
}
void vo_batch_distance(int metric, int mode, const float* q, const float* rows, size_t nrows, size_t dim,
                       float* out) {  // native/distance.rs:87-102 (order-preserving loop)
  for (size_t i = 0; i < nrows; i++) out[i] = engine_distance(metric, mode, q, rows + i * dim, dim);
}
void vo_batch_compute_distance(int metric, int mode, const float* q, const float* rows, size_t nrows,
                               size_t dim, float* out) {
  for (size_t i = 0; i < nrows; i++) out[i] = index_compute_distance(metric, mode, q, rows + i * dim, dim);
}
float vo_transform_score(int metric, float d) { return transform_score(metric, d); }
int vo_higher_is_better(int metric) { return higher_is_better(metric) ? 1 : 0; }
uint64_t vo_ef_search(int quality, uint64_t custom, uint64_t k) { return ef_search(quality, custom, k); }
int vo_total_cmp(float a, float b) { return total_cmp(a, b); }
// DistanceMetric::sort_results (core/distance.rs:95-103) on caller-supplied (id, score) pairs, in place
void vo_sort_results(int metric, uint64_t* ids, float* scores, uint64_t n) {
  std::vector<std::pair<uint64_t, float>> r(n);
  for (uint64_t i = 0; i < n; i++) r[i] = {ids[i], scores[i]};
  sort_results(metric, r);
  for (uint64_t i = 0; i < n; i++) {
    ids[i] = r[i].first;
    scores[i] = r[i].second;
  }
}

uint64_t vo_xorshift64_next(uint64_t* state) { return xorshift_next(*state); }
uint32_t vo_random_layer(uint64_t* state, double level_mult) { return random_layer(*state, level_mult); }

void vo_heap_order_after_pushes(const float* d, const uint64_t* node, size_t n, int min_heap,
                                uint64_t* out_nodes) {
  if (min_heap) {
    RustHeap<true> h;
    for (size_t i = 0; i < n; i++) h.push({d[i], node[i]});
    for (size_t i = 0; i < n; i++) out_nodes[i] = h.data[i].node;
  } else {
    RustHeap<false> h;
    for (size_t i = 0; i < n; i++) h.push({d[i], node[i]});
    for (size_t i = 0; i < n; i++) out_nodes[i] = h.data[i].node;
  }
}

vo_hnsw* vo_hnsw_new(uint32_t dim, int metric, int mode, uint32_t M, uint32_t efc) {
  vo_hnsw* g = new vo_hnsw();
  hnsw_init(*g, dim, metric, mode, M, efc);
  return g;
}
void vo_hnsw_free(vo_hnsw* g) { delete g; }
void vo_hnsw_set_alpha(vo_hnsw* g, float alpha) { g->alpha = alpha; }
void vo_hnsw_set_build_tie(vo_hnsw* g, int tie) { g->build_tie = tie; }
void vo_hnsw_set_build_threads(vo_hnsw* g, uint32_t nthreads) { g->build_threads = nthreads < 1 ? 1 : nthreads; }
void vo_hnsw_insert_batch_sync(vo_hnsw* g, const float* vecs, uint64_t n) { hnsw_insert_batch_sync(*g, vecs, n); }
uint32_t vo_build_batch_size(uint64_t linked, uint32_t max_batch) { return build_batch_size(linked, max_batch); }
/* whole build with the batched schedule: rows [0,n) appended to the graph */
void vo_hnsw_build_batched(vo_hnsw* g, const float* vecs, uint64_t n, uint32_t max_batch) {
  uint64_t pos = 0;
  while (pos < n) {
    uint64_t b = build_batch_size(g->count, max_batch);
    if (g->entry_point < 0) b = 1;
    if (b > n - pos) b = n - pos;
    hnsw_insert_batch_sync(*g, vecs + pos * g->dim, b);
    pos += b;
  }
}
uint64_t vo_hnsw_insert(vo_hnsw* g, const float* v) { return hnsw_insert(*g, v); }
uint64_t vo_hnsw_len(const vo_hnsw* g) { return g->count; }
uint32_t vo_hnsw_max_layer(const vo_hnsw* g) { return (uint32_t)g->max_layer; }
int64_t vo_hnsw_entry_point(const vo_hnsw* g) { return g->entry_point; }
uint32_t vo_hnsw_num_layers(const vo_hnsw* g) { return (uint32_t)g->layers.size(); }
uint32_t vo_hnsw_neighbors(const vo_hnsw* g, uint32_t layer, uint64_t node, uint64_t* out, uint32_t cap) {
  const auto& v = g->nbrs(layer, node);
  for (size_t i = 0; i < v.size() && i < cap; i++) out[i] = v[i];
  return (uint32_t)v.size();
}
const float* vo_hnsw_vector(const vo_hnsw* g, uint64_t node) { return g->vec(node); }
uint32_t vo_hnsw_search(const vo_hnsw* g, const float* q, uint32_t k, uint32_t ef, int tie,
                        uint64_t* out_nodes, float* out_dist) {
  tl_n_dist = 0;
  tl_n_expand = 0;
  auto r = hnsw_search(*g, q, k, ef, tie);
  for (size_t i = 0; i < r.size(); i++) {
    out_nodes[i] = r[i].first;
    out_dist[i] = r[i].second;
  }
  return (uint32_t)r.size();
}
uint32_t vo_hnsw_search_multi_entry(vo_hnsw* g, const float* q, uint32_t k, uint32_t ef, uint32_t num_probes, int tie, uint64_t* out_nodes,
                                    float* out_dist) {
  tl_n_dist = tl_n_expand = 0;
  auto r = hnsw_search_multi_entry(*g, q, k, ef, num_probes, tie);
  for (size_t i = 0; i < r.size(); i++) {
    out_nodes[i] = r[i].first;
    out_dist[i] = r[i].second;
  }
  return (uint32_t)r.size();
}
uint64_t vo_hnsw_rng_state(const vo_hnsw* g) { return g->rng_state; }
void vo_hnsw_last_stats(uint64_t* n_dist, uint64_t* n_expand) {
  *n_dist = tl_n_dist;
  *n_expand = tl_n_expand;
}
uint64_t vo_hnsw_search_layer_single(const vo_hnsw* g, const float* q, uint64_t entry, uint32_t layer) {
  return search_layer_single(*g, q, entry, layer);
}
uint32_t vo_hnsw_search_layer(const vo_hnsw* g, const float* q, const uint64_t* eps, uint32_t neps,
                              uint32_t ef, uint32_t layer, int tie, uint64_t* out_nodes, float* out_dist,
                              uint32_t cap) {
  std::vector<uint64_t> e(eps, eps + neps);
  auto r = search_layer(*g, q, e, ef, layer, tie);
  for (size_t i = 0; i < r.size() && i < cap; i++) {
    out_nodes[i] = r[i].first;
    out_dist[i] = r[i].second;
  }
  return (uint32_t)r.size();
}
uint32_t vo_hnsw_select_neighbors(const vo_hnsw* g, const uint64_t* cand, const float* cand_dist, uint32_t n,
                                  uint32_t max_neighbors, uint64_t* out) {
  std::vector<std::pair<uint64_t, float>> c;
  for (uint32_t i = 0; i < n; i++) c.emplace_back(cand[i], cand_dist[i]);
  auto s = select_neighbors(*g, c, max_neighbors);
  for (size_t i = 0; i < s.size(); i++) out[i] = s[i];
  return (uint32_t)s.size();
}

// native/backend_adapter.rs:184-261 — file_dump, format v1 (little-endian)
int vo_hnsw_file_dump(const vo_hnsw* g, const char* dir, const char* basename) {
  std::string vp = std::string(dir) + "/" + basename + ".vectors";
  std::string gp = std::string(dir) + "/" + basename + ".graph";
  FILE* f = std::fopen(vp.c_str(), "wb");
  if (!f) return -1;
  uint32_t version = 1, dim = g->vectors.empty() ? 0 : g->dim;
  uint64_t count = g->vectors.size() / g->dim;
  std::fwrite(&version, 4, 1, f);
  std::fwrite(&count, 8, 1, f);
  std::fwrite(&dim, 4, 1, f);
  std::fwrite(g->vectors.data(), 4, g->vectors.size(), f);
  std::fclose(f);
  f = std::fopen(gp.c_str(), "wb");
  if (!f) return -1;
  uint32_t num_layers = (uint32_t)g->layers.size(), M = (uint32_t)g->M, M0 = (uint32_t)g->M0,
           efc = (uint32_t)g->efc, max_layer = (uint32_t)g->max_layer;
  uint64_t ep = g->entry_point < 0 ? 0 : (uint64_t)g->entry_point;  // unwrap_or(0)
  std::fwrite(&version, 4, 1, f);
  std::fwrite(&num_layers, 4, 1, f);
  std::fwrite(&M, 4, 1, f);
  std::fwrite(&M0, 4, 1, f);
  std::fwrite(&efc, 4, 1, f);
  std::fwrite(&ep, 8, 1, f);
  std::fwrite(&max_layer, 4, 1, f);
  std::fwrite(&count, 8, 1, f);
  for (const auto& L : g->layers) {
    uint64_t nn = L.size();
    std::fwrite(&nn, 8, 1, f);
    for (const auto& nb : L) {
      uint32_t k = (uint32_t)nb.size();
      std::fwrite(&k, 4, 1, f);
      for (uint64_t x : nb) {
        uint32_t y = (uint32_t)x;
        std::fwrite(&y, 4, 1, f);
      }
    }
  }
  std::fclose(f);
  return 0;
}
// Re-places the vector storage: 2 MiB chunks copied (= first touched) round-robin by the pool's threads, so that the
// random 3 KB reads of a many-thread search batch spread over every memory controller of the host instead of hitting the
// one NUMA node of the thread that loaded the file.
void vo_hnsw_spread(vo_hnsw* g, uint32_t nthreads) {
  if (!g || nthreads <= 1 || g->vectors.empty()) return;
  std::vector<float, NoInitAlloc<float>> fresh;
  fresh.resize(g->vectors.size());
  const size_t chunk = (size_t)512 * 1024, n = g->vectors.size(), nchunks = (n + chunk - 1) / chunk;
  const float* src = g->vectors.data();
  float* dst = fresh.data();
  Pool::get().run(nthreads, [&](uint32_t t) {
    for (size_t c = t; c < nchunks; c += nthreads) {
      const size_t lo = c * chunk, hi = std::min(n, lo + chunk);
      std::memcpy(dst + lo, src + lo, (hi - lo) * sizeof(float));
    }
  });
  g->vectors.swap(fresh);
}

// native/backend_adapter.rs:273-381 — file_load (alpha reset to 1.0, rng reseeded)
vo_hnsw* vo_hnsw_file_load(const char* dir, const char* basename, int metric, int mode) {
  std::string vp = std::string(dir) + "/" + basename + ".vectors";
  std::string gp = std::string(dir) + "/" + basename + ".graph";
  FILE* f = std::fopen(vp.c_str(), "rb");
  if (!f) return nullptr;
  uint32_t version = 0, dim = 0;
  uint64_t count = 0;
  bool ok = std::fread(&version, 4, 1, f) == 1 && version == 1 && std::fread(&count, 8, 1, f) == 1 &&
            std::fread(&dim, 4, 1, f) == 1;
  std::unique_ptr<vo_hnsw> g(new vo_hnsw());
  if (ok) {
    g->vectors.resize((size_t)count * dim);
    ok = std::fread(g->vectors.data(), 4, g->vectors.size(), f) == g->vectors.size();
  }
  std::fclose(f);
  if (!ok) return nullptr;
  f = std::fopen(gp.c_str(), "rb");
  if (!f) return nullptr;
  uint32_t num_layers = 0, M = 0, M0 = 0, efc = 0, max_layer = 0;
  uint64_t ep = 0, count2 = 0;
  ok = std::fread(&version, 4, 1, f) == 1 && version == 1 && std::fread(&num_layers, 4, 1, f) == 1 &&
       std::fread(&M, 4, 1, f) == 1 && std::fread(&M0, 4, 1, f) == 1 && std::fread(&efc, 4, 1, f) == 1 &&
       std::fread(&ep, 8, 1, f) == 1 && std::fread(&max_layer, 4, 1, f) == 1 &&
       std::fread(&count2, 8, 1, f) == 1;
  if (ok) {
    g->layers.resize(num_layers);
    for (uint32_t l = 0; l < num_layers && ok; l++) {
      uint64_t nn = 0;
      ok = std::fread(&nn, 8, 1, f) == 1;
      if (!ok) break;
      g->layers[l].resize(nn);
      for (uint64_t i = 0; i < nn && ok; i++) {
        uint32_t k = 0;
        ok = std::fread(&k, 4, 1, f) == 1;
        std::vector<uint32_t> tmp(k);
        if (ok && k) ok = std::fread(tmp.data(), 4, k, f) == k;
        g->layers[l][i].assign(tmp.begin(), tmp.end());
      }
    }
  }
  std::fclose(f);
  if (!ok) return nullptr;
  g->dim = dim;
  g->metric = metric;
  g->mode = mode;
  g->M = M;
  g->M0 = M0;
  g->efc = efc;
  g->entry_point = (int64_t)ep;  // Some(entry_point), backend_adapter.rs:371
  g->max_layer = max_layer;
  g->count = count;
  g->level_mult = 1.0 / std::log((double)M);
  g->alpha = 1.0f;
  g->note_vectors();
  return g.release();
}

vo_index* vo_index_new(uint32_t dim, int metric, int mode, uint32_t M, uint32_t efc) {
  vo_index* ix = new vo_index();
  hnsw_init(ix->g, dim, metric, mode, M, efc);
  return ix;
}
vo_index* vo_index_new_auto(uint32_t dim, int metric, int mode) {  // hnsw/params.rs:41-57
  return dim <= 256 ? vo_index_new(dim, metric, mode, 24, 300) : vo_index_new(dim, metric, mode, 32, 400);
}
void vo_index_free(vo_index* ix) { delete ix; }
int vo_index_insert(vo_index* ix, uint64_t id, const float* v) {  // hnsw/index/trait_impl.rs:10-36
  if (ix->id_to_idx.count(id)) return 0;  // duplicate id: silently skipped (:23-25)
  uint64_t idx = ix->next_idx++;
  ix->id_to_idx[id] = idx;
  if (ix->idx_to_id.size() <= idx) {
    ix->idx_to_id.resize(idx + 1);
    ix->idx_live.resize(idx + 1, 0);
  }
  ix->idx_to_id[idx] = id;
  ix->idx_live[idx] = 1;
  ix->live++;
  hnsw_insert(ix->g, v);  // node id == idx under sequential insertion
  return 1;
}
int vo_index_remove(vo_index* ix, uint64_t id) {  // trait_impl.rs:54-58 soft delete
  auto it = ix->id_to_idx.find(id);
  if (it == ix->id_to_idx.end()) return 0;
  ix->idx_live[it->second] = 0;
  ix->id_to_idx.erase(it);
  ix->live--;
  return 1;
}
uint64_t vo_index_len(const vo_index* ix) { return ix->live; }
vo_hnsw* vo_index_graph(vo_index* ix) { return &ix->g; }

static uint32_t emit(const std::vector<std::pair<uint64_t, float>>& r, uint64_t* ids, float* sc) {
  for (size_t i = 0; i < r.size(); i++) {
    ids[i] = r[i].first;
    sc[i] = r[i].second;
  }
  return (uint32_t)r.size();
}
uint32_t vo_index_search_with_quality(const vo_index* ix, const float* q, uint32_t k, int quality,
                                      uint32_t custom_ef, int tie, uint64_t* out_ids, float* out_scores) {
  tl_n_dist = 0;
  tl_n_expand = 0;
  return emit(index_search_with_quality(*ix, q, k, quality, custom_ef, tie), out_ids, out_scores);
}
uint32_t vo_index_search_brute_force(const vo_index* ix, const float* q, uint32_t k, uint64_t* out_ids,
                                     float* out_scores) {
  return emit(index_brute_force(*ix, q, k), out_ids, out_scores);
}
// hnsw/index/search.rs:118-160
uint32_t vo_index_search_with_rerank_quality(const vo_index* ix, const float* q, uint32_t k, uint32_t rerank_k,
                                             int quality, uint32_t custom_ef, int tie, uint64_t* out_ids,
                                             float* out_scores) {  // hnsw/index/search.rs:297-350
  if (quality == 3) quality = 2;  // Perfect -> Accurate (:305-310)
  auto cand = index_search_with_quality(*ix, q, rerank_k, quality, custom_ef, tie);
  std::vector<std::pair<uint64_t, float>> rr;
  for (auto& c : cand) {
    auto it = ix->id_to_idx.find(c.first);
    if (it == ix->id_to_idx.end()) continue;
    rr.emplace_back(c.first, index_compute_distance(ix->g.metric, ix->g.mode, q, ix->g.vec(it->second), ix->g.dim));
  }
  sort_results(ix->g.metric, rr);
  if (rr.size() > k) rr.resize(k);
  return emit(rr, out_ids, out_scores);
}
uint32_t vo_index_search_with_rerank(const vo_index* ix, const float* q, uint32_t k, uint32_t rerank_k,
                                     uint64_t* out_ids, float* out_scores) {
  auto cand = index_search_with_quality(*ix, q, rerank_k, 2, 0, VO_TIE_REFERENCE);
  std::vector<std::pair<uint64_t, float>> rr;
  for (auto& c : cand) {
    auto it = ix->id_to_idx.find(c.first);
    if (it == ix->id_to_idx.end()) continue;
    rr.emplace_back(c.first, index_compute_distance(ix->g.metric, ix->g.mode, q, ix->g.vec(it->second), ix->g.dim));
  }
  sort_results(ix->g.metric, rr);
  if (rr.size() > k) rr.resize(k);
  return emit(rr, out_ids, out_scores);
}

// hnsw/index/batch.rs:159-197 — always HNSW (no Perfect / <=100 shortcut), one query per
// worker.  Each worker searches through a private shallow view (own visited stamps).
void vo_index_search_batch(const vo_index* ix, const float* queries, uint32_t nq, uint32_t k, int quality,
                           uint32_t custom_ef, int tie, uint32_t nthreads, uint64_t* out_ids,
                           float* out_scores, uint32_t* out_n) {
  size_t ef = ef_search(quality, custom_ef, k);
  if (nthreads < 1) nthreads = 1;
  std::atomic<uint32_t> next(0);
  auto worker = [&]() {
    // private visited array: copy only the scalar fields, alias the big arrays read-only
    std::vector<uint32_t> stamp(ix->g.vectors.size() / ix->g.dim, 0);
    uint32_t epoch = 0;
    for (;;) {
      uint32_t qi = next.fetch_add(1);
      if (qi >= nq) break;
      const float* q = queries + (size_t)qi * ix->g.dim;
      const vo_hnsw& g = ix->g;
      // inline re-statement of hnsw_search with the private stamp array
      std::vector<std::pair<uint64_t, float>> res;
      if (g.entry_point >= 0) {
        uint64_t cur = (uint64_t)g.entry_point;
        for (size_t l = g.max_layer; l >= 1; l--) cur = search_layer_single(g, q, cur, l);
        if (++epoch == 0) {
          std::fill(stamp.begin(), stamp.end(), 0);
          epoch = 1;
        }
        RustHeap<true> candidates;
        RustHeap<false> results;
        {
          float d = g.dist(q, g.vec(cur));
          candidates.push({d, cur});
          results.push({d, cur});
          stamp[cur] = epoch;
        }
        while (!candidates.empty()) {
          HeapItem c = candidates.pop();
          float furthest = results.empty() ? std::numeric_limits<float>::max() : results.peek().d;
          if (c.d > furthest && results.size() >= ef) break;
          for (uint64_t nb : g.nbrs(0, c.node)) {
            if (stamp[nb] == epoch) continue;
            stamp[nb] = epoch;
            float d = g.dist(q, g.vec(nb));
            float far = results.empty() ? std::numeric_limits<float>::max() : results.peek().d;
            if (d < far || results.size() < ef) {
              candidates.push({d, nb});
              results.push({d, nb});
              if (results.size() > ef) results.pop();
            }
          }
        }
        for (const HeapItem& it : results.data) res.emplace_back(it.node, it.d);
        if (tie == VO_TIE_CANONICAL)
          std::sort(res.begin(), res.end(), [](const auto& a, const auto& b) {
            int c = total_cmp(a.second, b.second);
            return c ? c < 0 : a.first < b.first;
          });
        else
          std::stable_sort(res.begin(), res.end(),
                           [](const auto& a, const auto& b) { return total_cmp(a.second, b.second) < 0; });
        if (res.size() > k) res.resize(k);
      }
      uint32_t n = 0;
      for (auto& p : res)
        if (p.first < ix->idx_live.size() && ix->idx_live[p.first]) {
          out_ids[(size_t)qi * k + n] = ix->idx_to_id[p.first];
          out_scores[(size_t)qi * k + n] = transform_score(g.metric, p.second);
          n++;
        }
      out_n[qi] = n;
    }
  };
  Pool::get().run(nthreads, [&](uint32_t) { worker(); });
}

// NativeHnsw::search over a batch of queries with nthreads host threads (one search per thread at a
// time, like rayon in search_batch_parallel, batch.rs:180-194), on a bare graph (e.g. one loaded from
// the reference's files).  Used by bench.py's CPU leg for the graph path; includes the reference's
// software prefetch of upcoming neighbour vectors (graph.rs:482-497: first cache line, T0, 16 ahead
// when dim >= 384) — a performance hint with no effect on results.
void vo_hnsw_search_batch(const vo_hnsw* gp, const float* queries, uint32_t nq, uint32_t k, uint32_t ef_in,
                          int tie, uint32_t nthreads, uint64_t* out_nodes, float* out_dist, uint32_t* out_n,
                          uint64_t* total_n_dist, uint64_t* total_n_expand) {
  const vo_hnsw& g = *gp;
  const size_t ef = std::max<size_t>(ef_in, k);
  if (nthreads < 1) nthreads = 1;
  std::atomic<uint32_t> next(0);
  std::atomic<uint64_t> nd_total(0), ne_total(0);
  const size_t pd = std::min<size_t>(16, std::max<size_t>(4, (size_t)g.dim * 4 / 64));  // core/simd.rs:40-51
  auto worker = [&]() {
    std::vector<uint32_t> stamp(g.vectors.size() / g.dim, 0);
    uint32_t epoch = 0;
    tl_n_dist = 0;
    uint64_t n_expand = 0;
    for (;;) {
      uint32_t qi = next.fetch_add(1);
      if (qi >= nq) break;
      const float* q = queries + (size_t)qi * g.dim;
      std::vector<std::pair<uint64_t, float>> res;
      if (g.entry_point >= 0) {
        uint64_t cur = (uint64_t)g.entry_point;
        for (size_t l = g.max_layer; l >= 1; l--) cur = search_layer_single(g, q, cur, l);
        if (++epoch == 0) {
          std::fill(stamp.begin(), stamp.end(), 0);
          epoch = 1;
        }
        RustHeap<true> candidates;
        RustHeap<false> results;
        {
          float d = g.dist(q, g.vec(cur));
          candidates.push({d, cur});
          results.push({d, cur});
          stamp[cur] = epoch;
        }
        while (!candidates.empty()) {
          HeapItem c = candidates.pop();
          float furthest = results.empty() ? std::numeric_limits<float>::max() : results.peek().d;
          if (c.d > furthest && results.size() >= ef) break;
          n_expand++;
          const std::vector<uint64_t>& nbs = g.nbrs(0, c.node);
          const bool pf = g.dim >= 384 && nbs.size() > pd;
          if (pf)
            for (size_t i = 0; i < pd; i++) __builtin_prefetch(g.vec(nbs[i]), 0, 3);
          for (size_t i = 0; i < nbs.size(); i++) {
            if (g.dim >= 384 && i + pd < nbs.size()) __builtin_prefetch(g.vec(nbs[i + pd]), 0, 3);
            const uint64_t nb = nbs[i];
            if (stamp[nb] == epoch) continue;
            stamp[nb] = epoch;
            float d = g.dist(q, g.vec(nb));
            float far = results.empty() ? std::numeric_limits<float>::max() : results.peek().d;
            if (d < far || results.size() < ef) {
              candidates.push({d, nb});
              results.push({d, nb});
              if (results.size() > ef) results.pop();
            }
          }
        }
        for (const HeapItem& it : results.data) res.emplace_back(it.node, it.d);
        if (tie == VO_TIE_CANONICAL)
          std::sort(res.begin(), res.end(), [](const auto& a, const auto& b) {
            int c = total_cmp(a.second, b.second);
            return c ? c < 0 : a.first < b.first;
          });
        else
          std::stable_sort(res.begin(), res.end(),
                           [](const auto& a, const auto& b) { return total_cmp(a.second, b.second) < 0; });
        if (res.size() > k) res.resize(k);
      }
      for (size_t i = 0; i < res.size(); i++) {
        out_nodes[(size_t)qi * k + i] = res[i].first;
        out_dist[(size_t)qi * k + i] = res[i].second;
      }
      out_n[qi] = (uint32_t)res.size();
    }
    nd_total += tl_n_dist;
    ne_total += n_expand;
  };
  Pool::get().run(nthreads, [&](uint32_t) { worker(); });
  if (total_n_dist) *total_n_dist = nd_total.load();
  if (total_n_expand) *total_n_expand = ne_total.load();
}

// Flat exact scan used by bench.py's cpu_baseline leg and by the large-N parity tests:
// search_brute_force semantics (hnsw/index/search.rs:197-218) over a row-major corpus,
// rows split over nthreads like brute_force_search_parallel (hnsw/index/batch.rs:223-244).
void vo_scan_topk(int metric, int mode, const float* rows, uint64_t nrows, uint32_t dim, const float* queries,
                  uint32_t nq, uint32_t k, uint32_t nthreads, uint64_t* out_rows, float* out_scores) {
  if (nthreads < 1) nthreads = 1;
  bool hib = higher_is_better(metric);
  auto better = [hib](const std::pair<uint64_t, float>& a, const std::pair<uint64_t, float>& b) {
    int c = hib ? total_cmp(b.second, a.second) : total_cmp(a.second, b.second);
    return c ? c < 0 : a.first < b.first;
  };
  // brute_force_search_parallel (batch.rs:223-244): the rows of ONE query are split over the pool's threads (rayon in the
  // reference); the partition is static, so thread t always reads the same rows (and finds them on its own NUMA node
  // when the caller placed them with vo_alloc_spread)
  std::vector<std::vector<std::pair<uint64_t, float>>> part(nthreads);
  for (uint32_t qi = 0; qi < nq; qi++) {
    const float* q = queries + (size_t)qi * dim;
    auto work = [&](uint32_t t) {
      uint64_t lo = nrows * t / nthreads, hi = nrows * (t + 1) / nthreads;
      auto& v = part[t];
      v.clear();
      v.reserve((size_t)(hi - lo));
      for (uint64_t r = lo; r < hi; r++)
        v.emplace_back(r, index_compute_distance(metric, mode, q, rows + (size_t)r * dim, dim));
      size_t kk = std::min<size_t>(k, v.size());
      std::partial_sort(v.begin(), v.begin() + kk, v.end(), better);
      v.resize(kk);
    };
    Pool::get().run(nthreads, work);
    std::vector<std::pair<uint64_t, float>> all;
    for (auto& v : part) all.insert(all.end(), v.begin(), v.end());
    std::sort(all.begin(), all.end(), better);
    for (uint32_t i = 0; i < k; i++) {
      if (i < all.size()) {
        out_rows[(size_t)qi * k + i] = all[i].first;
        out_scores[(size_t)qi * k + i] = all[i].second;
      } else {
        out_rows[(size_t)qi * k + i] = UINT64_MAX;
        out_scores[(size_t)qi * k + i] = std::numeric_limits<float>::quiet_NaN();
      }
    }
  }
}

// A copy of `rows` whose pages are first touched by the pool thread that vo_scan_topk(…, nthreads) will read them with
// (same static partition); freed with vo_free_spread.
float* vo_alloc_spread(const float* rows, uint64_t nrows, uint32_t dim, uint32_t nthreads) {
  if (nthreads < 1) nthreads = 1;
  float* p = static_cast<float*>(::operator new((size_t)nrows * dim * sizeof(float) + 64));
  Pool::get().run(nthreads, [&](uint32_t t) {
    const uint64_t lo = nrows * t / nthreads, hi = nrows * (t + 1) / nthreads;
    std::memcpy(p + (size_t)lo * dim, rows + (size_t)lo * dim, (size_t)(hi - lo) * dim * sizeof(float));
  });
  return p;
}
void vo_free_spread(float* p) { ::operator delete(p); }

// Range-sharded exact search: merge of per-shard top-k record lists.  The unsharded result is the stable sort of ALL rows
// by score (core/distance.rs:95-103) in row order; shards are contiguous row ranges and every shard's list is already in
// that order, so a stable sort of the shard-major concatenation by score alone reproduces it.
void vo_merge_shard_records(const uint32_t* rec, uint32_t S, uint32_t nq, uint32_t k, int hib, uint64_t* out_ids,
                            float* out_scores, uint32_t* out_n) {
  for (uint32_t q = 0; q < nq; q++) {
    std::vector<std::pair<uint64_t, float>> all;
    for (uint32_t s = 0; s < S; s++)
      for (uint32_t p = 0; p < k; p++) {
        const uint32_t* r = rec + (((size_t)s * nq + q) * k + p) * 3;
        if (r[0] == 0xFFFFFFFFu && r[1] == 0xFFFFFFFFu && r[2] == 0xFFFFFFFFu) continue;
        float f;
        std::memcpy(&f, &r[2], 4);
        all.emplace_back(((uint64_t)r[1] << 32) | r[0], f);
      }
    std::stable_sort(all.begin(), all.end(), [hib](const auto& a, const auto& b) {
      return hib ? total_cmp(b.second, a.second) < 0 : total_cmp(a.second, b.second) < 0;
    });
    const uint32_t n = (uint32_t)std::min<size_t>(all.size(), k);
    out_n[q] = n;
    for (uint32_t i = 0; i < k; i++) {
      out_ids[(size_t)q * k + i] = i < n ? all[i].first : UINT64_MAX;
      out_scores[(size_t)q * k + i] = i < n ? all[i].second : std::numeric_limits<float>::quiet_NaN();
    }
  }
}

// ---------------------------------------------------------------------------
// DualPrecisionHnsw — native/dual_precision.rs + native/quantization.rs: per-dimension scalar quantiser trained
// on the first min(1000, n) inserted vectors (quantization.rs:191-233), u8 codes (:236-252, f32::round = half away
// from zero), integer L2^2 between codes (:42-91), int8 graph traversal (dual_precision.rs:284-441) and exact f32
// re-ranking of k * oversampling candidates (:253-282).
// ---------------------------------------------------------------------------
void vo_sq_train(const float* vecs, uint64_t n, uint32_t dim, float* min_vals, float* scales, float* inv_scales) {
  std::vector<float> mx(dim, std::numeric_limits<float>::lowest());
  for (uint32_t i = 0; i < dim; i++) min_vals[i] = std::numeric_limits<float>::max();
  for (uint64_t r = 0; r < n; r++)
    for (uint32_t i = 0; i < dim; i++) {
      const float v = vecs[(size_t)r * dim + i];
      min_vals[i] = std::fmin(min_vals[i], v);  // f32::min / f32::max: the non-NaN operand wins
      mx[i] = std::fmax(mx[i], v);
    }
  for (uint32_t i = 0; i < dim; i++) {
    const float range = mx[i] - min_vals[i];
    scales[i] = std::fabs(range) < 1e-10f ? 1.0f : 255.0f / range;
    inv_scales[i] = 1.0f / scales[i];
  }
}
void vo_sq_quantize(const float* vecs, uint64_t n, uint32_t dim, const float* min_vals, const float* scales,
                    uint8_t* codes) {
  for (uint64_t r = 0; r < n; r++)
    for (uint32_t i = 0; i < dim; i++) {
      float q = std::round((vecs[(size_t)r * dim + i] - min_vals[i]) * scales[i]);
      q = q < 0.0f ? 0.0f : (q > 255.0f ? 255.0f : q);  // clamp; NaN -> `as u8` saturating cast = 0
      codes[(size_t)r * dim + i] = std::isnan(q) ? 0 : (uint8_t)q;
    }
}
uint32_t vo_sq_l2(const uint8_t* a, const uint8_t* b, uint32_t dim) {
  uint32_t s = 0;
  for (uint32_t i = 0; i < dim; i++) {
    const int32_t d = (int32_t)a[i] - (int32_t)b[i];
    s += (uint32_t)(d * d);
  }
  return s;
}
// search_int8_traversal (dual_precision.rs:253-282) on graph g with the code store `codes` ([count][dim]).
// Output: node ids + EXACT engine distances (inner.compute_distance), best first.
uint32_t vo_dual_search_int8(const vo_hnsw* gp, const uint8_t* codes, const float* min_vals, const float* scales,
                             const float* q, uint32_t k, uint32_t ef_search, uint32_t oversampling, int tie,
                             uint64_t* out_nodes, float* out_dist, uint64_t* n_dist_int8, uint64_t* n_expand) {
  const vo_hnsw& g = *gp;
  const uint32_t dim = g.dim;
  uint64_t nd = 0, ne = 0;
  if (g.entry_point < 0) return 0;
  std::vector<uint8_t> qc(dim);
  vo_sq_quantize(q, 1, dim, min_vals, scales, qc.data());
  auto dist = [&](uint64_t node) {
    nd++;
    return vo_sq_l2(qc.data(), codes + (size_t)node * dim, dim);
  };
  const size_t cand_k = (size_t)k * oversampling;
  // greedy descent with int8 distances (:407-441)
  uint64_t cur = (uint64_t)g.entry_point;
  for (size_t l = g.max_layer; l >= 1; l--) {
    uint64_t current = cur;
    uint32_t current_dist = dist(current);
    for (;;) {
      const std::vector<uint64_t> nbs = g.nbrs(l, current);
      bool improved = false;
      for (uint64_t nb : nbs) {
        const uint32_t d = dist(nb);
        if (d < current_dist) {
          current = nb;
          current_dist = d;
          improved = true;
        }
      }
      if (!improved) break;
    }
    cur = current;
  }
  // layer 0 beam (:316-384)
  struct IKey {
    uint32_t d;
    uint64_t node;
    bool operator<(const IKey& o) const { return d != o.d ? d < o.d : node < o.node; }
  };
  std::vector<uint8_t> visited(g.vectors.size() / dim, 0);
  std::vector<IKey> candidates, results;  // kept as sorted vectors: pops follow the (dist, node) total order
  auto push_sorted = [](std::vector<IKey>& v, IKey x) { v.insert(std::upper_bound(v.begin(), v.end(), x), x); };
  {
    const uint32_t d = dist(cur);
    push_sorted(candidates, {d, cur});
    push_sorted(results, {d, cur});
    visited[cur] = 1;
  }
  const size_t ef = std::max<size_t>(ef_search, cand_k);
  while (!candidates.empty()) {
    const IKey c = candidates.front();
    candidates.erase(candidates.begin());
    const uint32_t furthest = results.empty() ? UINT32_MAX : results.back().d;
    if (c.d > furthest && results.size() >= ef) break;
    ne++;
    for (uint64_t nb : g.nbrs(0, c.node)) {
      if (visited[nb]) continue;
      visited[nb] = 1;
      const uint32_t d = dist(nb);
      const uint32_t far = results.empty() ? UINT32_MAX : results.back().d;
      if (d < far || results.size() < ef) {
        push_sorted(candidates, {d, nb});
        push_sorted(results, {d, nb});
        if (results.size() > ef) results.pop_back();
      }
    }
  }
  (void)tie;  // results is already in canonical (dist, node) order; the reference's order among equal integer
              // distances is the heap artefact (into_iter + stable sort_by_key), not reproduced
  if (results.size() > cand_k) results.resize(cand_k);
  // exact re-ranking (:267-281): inner.compute_distance = DistanceEngine::distance, stable sort by total_cmp
  std::vector<std::pair<uint64_t, float>> rr;
  for (const IKey& c : results) rr.emplace_back(c.node, engine_distance(g.metric, g.mode, q, g.vec(c.node), dim));
  std::stable_sort(rr.begin(), rr.end(), [](const auto& a, const auto& b) { return total_cmp(a.second, b.second) < 0; });
  if (rr.size() > k) rr.resize(k);
  for (size_t i = 0; i < rr.size(); i++) {
    out_nodes[i] = rr[i].first;
    out_dist[i] = rr[i].second;
  }
  if (n_dist_int8) *n_dist_int8 = nd;
  if (n_expand) *n_expand = ne;
  return (uint32_t)rr.size();
}

// ---------------------------------------------------------------------------
// half_precision.rs — VectorData::BF16: `half::bf16::from_f32` (round to nearest even; NaN stays NaN),
// dot_product (:199-233) = sequential f32 sum of x.to_f32() * y.to_f32(); cosine_similarity (:237-254) =
// dot / (sqrt(norm_squared(a)) * sqrt(norm_squared(b))), 0.0 when a norm is below f32::EPSILON;
// norm_squared (:290-311) sequential.  Exact scan + top-k (score descending, row ascending among equals).
// ---------------------------------------------------------------------------
static inline uint16_t bf16_from_f32(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
void vo_round_bf16(const float* in, float* out, uint64_t n) {
  for (uint64_t i = 0; i < n; i++) out[i] = bf16_to_f32(bf16_from_f32(in[i]));
}
void vo_scan_topk_bf16(int metric, const float* rows, uint64_t nrows, uint32_t dim, const float* queries,
                       uint32_t nq, uint32_t k, uint32_t nthreads, uint64_t* out_rows, float* out_scores) {
  std::vector<float> rr((size_t)nrows * dim), qq((size_t)nq * dim), rn(nrows);
  vo_round_bf16(rows, rr.data(), (uint64_t)nrows * dim);
  vo_round_bf16(queries, qq.data(), (uint64_t)nq * dim);
  auto nsq = [&](const float* v) {
    float s = 0.f;
    for (uint32_t i = 0; i < dim; i++) s += v[i] * v[i];
    return s;
  };
  for (uint64_t r = 0; r < nrows; r++) rn[r] = std::sqrt(nsq(rr.data() + (size_t)r * dim));
  if (nthreads < 1) nthreads = 1;
  std::atomic<uint32_t> next(0);
  auto worker = [&]() {
    std::vector<std::pair<float, uint64_t>> sc(nrows);
    for (;;) {
      uint32_t qi = next.fetch_add(1);
      if (qi >= nq) break;
      const float* q = qq.data() + (size_t)qi * dim;
      const float qn = std::sqrt(nsq(q));
      for (uint64_t r = 0; r < nrows; r++) {
        const float* v = rr.data() + (size_t)r * dim;
        float dot = 0.f;
        for (uint32_t i = 0; i < dim; i++) dot += q[i] * v[i];
        float s = dot;
        if (metric == VO_COSINE) {
          const float eps = std::numeric_limits<float>::epsilon();
          s = (qn < eps || rn[r] < eps) ? 0.0f : dot / (qn * rn[r]);
        }
        sc[r] = {s, r};
      }
      const size_t kk = std::min<size_t>(k, nrows);
      std::partial_sort(sc.begin(), sc.begin() + kk, sc.end(), [](const auto& a, const auto& b) {
        int c = total_cmp(b.first, a.first);
        return c ? c < 0 : a.second < b.second;
      });
      for (size_t i = 0; i < kk; i++) {
        out_rows[(size_t)qi * k + i] = sc[i].second;
        out_scores[(size_t)qi * k + i] = sc[i].first;
      }
    }
  };
  Pool::get().run(nthreads, [&](uint32_t) { worker(); });
}

int vo_cpu_has_avx512f(void) { return __builtin_cpu_supports("avx512f") ? 1 : 0; }
// =============================================================================================
// Storage modes (core/quantization.rs): BinaryQuantizedVector (sign bits) and QuantizedVector (SQ8, per-vector
// min/max) + the asymmetric f32-query x SQ8 distances.  Rust never contracts a*b+c, and this file is built with
// -ffp-contract=off: every operation below is one IEEE rounding, in the order the Rust code performs it.
// (f32 `Iterator::sum` folds left to right; it is restated from +0.0 — a start of -0.0, as newer std versions use,
// differs only for an all-negative-zero sequence.)
// =============================================================================================
// BinaryQuantizedVector::from_f32 — quantization.rs:68-86: bit i = (v[i] >= 0.0), byte i/8, bit i%8
void vo_binary_quantize(const float* v, uint32_t dim, uint8_t* out) {
  const uint32_t nb = (dim + 7) / 8;
  for (uint32_t b = 0; b < nb; b++) out[b] = 0;
  for (uint32_t i = 0; i < dim; i++)
    if (v[i] >= 0.0f) out[i / 8] |= (uint8_t)(1u << (i % 8));
}
// BinaryQuantizedVector::hamming_distance — quantization.rs:123-135
uint32_t vo_binary_hamming(const uint8_t* a, const uint8_t* b, uint32_t nbytes) {
  uint32_t d = 0;
  for (uint32_t i = 0; i < nbytes; i++) d += (uint32_t)__builtin_popcount((unsigned)(a[i] ^ b[i]));
  return d;
}
// QuantizedVector::from_f32 — quantization.rs:229-255
void vo_sq8_quantize(const float* v, uint32_t dim, uint8_t* data, float* out_min, float* out_max) {
  float mn = std::numeric_limits<float>::infinity(), mx = -std::numeric_limits<float>::infinity();
  for (uint32_t i = 0; i < dim; i++) {  // f32::min / f32::max: a NaN operand yields the other one
    mn = std::fmin(mn, v[i]);
    mx = std::fmax(mx, v[i]);
  }
  const float range = mx - mn;
  if (range < std::numeric_limits<float>::epsilon()) {
    for (uint32_t i = 0; i < dim; i++) data[i] = 128;
  } else {
    const float scale = 255.0f / range;
    for (uint32_t i = 0; i < dim; i++) {
      const float normalized = (v[i] - mn) * scale;
      float r = std::round(normalized);  // f32::round: half away from zero
      r = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);  // clamp(0.0, 255.0); NaN `as u8` saturates to 0
      data[i] = std::isnan(r) ? 0 : (uint8_t)r;
    }
  }
  *out_min = mn;
  *out_max = mx;
}
// QuantizedVector::to_f32 — quantization.rs:261-273
void vo_sq8_dequantize(const uint8_t* data, float mn, float mx, uint32_t dim, float* out) {
  const float range = mx - mn;
  if (range < std::numeric_limits<float>::epsilon()) {
    for (uint32_t i = 0; i < dim; i++) out[i] = mn;
  } else {
    const float scale = range / 255.0f;
    for (uint32_t i = 0; i < dim; i++) out[i] = (float)data[i] * scale + mn;
  }
}
// dot_product_quantized (:322-345) and dot_product_quantized_simd (:410-469): the same left-to-right sum
float vo_sq8_dot(const float* q, const uint8_t* data, float mn, float mx, uint32_t dim) {
  const float range = mx - mn;
  if (range < std::numeric_limits<float>::epsilon()) {
    float s = 0.0f;
    for (uint32_t i = 0; i < dim; i++) s += q[i];
    return s * mn;
  }
  const float scale = range / 255.0f, offset = mn;
  float sum = 0.0f;
  for (uint32_t i = 0; i < dim; i++) {
    const float dequant = (float)data[i] * scale + offset;
    sum += q[i] * dequant;
  }
  return sum;
}
// euclidean_squared_quantized (:349-374, simd = 0) / euclidean_squared_quantized_simd (:473-518, simd = 1)
float vo_sq8_l2sq(const float* q, const uint8_t* data, float mn, float mx, uint32_t dim, int simd) {
  const float range = mx - mn;
  if (range < std::numeric_limits<float>::epsilon()) {
    float s = 0.0f;
    for (uint32_t i = 0; i < dim; i++) {
      const float d = q[i] - mn;
      s += d * d;  // powi(2)
    }
    return s;
  }
  const float scale = range / 255.0f, offset = mn;
  float sum = 0.0f;
  if (!simd) {
    for (uint32_t i = 0; i < dim; i++) {
      const float dequantized = (float)data[i] * scale + offset;
      const float d = q[i] - dequantized;
      sum += d * d;
    }
    return sum;
  }
  const uint32_t chunks = dim / 4;
  for (uint32_t c = 0; c < chunks; c++) {
    const uint32_t b = c * 4;
    const float d0 = (float)data[b] * scale + offset, d1 = (float)data[b + 1] * scale + offset;
    const float d2 = (float)data[b + 2] * scale + offset, d3 = (float)data[b + 3] * scale + offset;
    const float f0 = q[b] - d0, f1 = q[b + 1] - d1, f2 = q[b + 2] - d2, f3 = q[b + 3] - d3;
    sum += f0 * f0 + f1 * f1 + f2 * f2 + f3 * f3;  // ((f0^2 + f1^2) + f2^2) + f3^2, then += (:507)
  }
  for (uint32_t i = chunks * 4; i < dim; i++) {
    const float dequant = (float)data[i] * scale + offset;
    const float d = q[i] - dequant;
    sum += d * d;
  }
  return sum;
}
// cosine_similarity_quantized (:380-395, simd = 0) / cosine_similarity_quantized_simd (:524-554, simd = 1)
float vo_sq8_cosine(const float* q, const uint8_t* data, float mn, float mx, uint32_t dim, int simd) {
  const float dot = vo_sq8_dot(q, data, mn, mx, dim);
  const float eps = std::numeric_limits<float>::epsilon();
  if (!simd) {
    float qs = 0.0f;
    for (uint32_t i = 0; i < dim; i++) qs += q[i] * q[i];
    const float query_norm = std::sqrt(qs);
    std::vector<float> rec(dim);
    vo_sq8_dequantize(data, mn, mx, dim, rec.data());
    float vs = 0.0f;
    for (uint32_t i = 0; i < dim; i++) vs += rec[i] * rec[i];
    const float quantized_norm = std::sqrt(vs);
    if (query_norm < eps || quantized_norm < eps) return 0.0f;
    return dot / (query_norm * quantized_norm);
  }
  float query_norm_sq = 0.0f;
  for (uint32_t i = 0; i < dim; i++) query_norm_sq += q[i] * q[i];
  const float range = mx - mn;
  const float scale = range < eps ? 0.0f : range / 255.0f;
  float quantized_norm_sq = 0.0f;
  for (uint32_t i = 0; i < dim; i++) {
    const float dequant = (float)data[i] * scale + mn;
    quantized_norm_sq += dequant * dequant;
  }
  const float denom = std::sqrt(query_norm_sq * quantized_norm_sq);
  if (denom < eps) return 0.0f;
  return dot / denom;
}
// quantized_norm_sq of cosine_similarity_quantized_simd alone (what the GPU keeps per row)
float vo_sq8_norm_sq(const uint8_t* data, float mn, float mx, uint32_t dim) {
  const float eps = std::numeric_limits<float>::epsilon();
  const float range = mx - mn;
  const float scale = range < eps ? 0.0f : range / 255.0f;
  float s = 0.0f;
  for (uint32_t i = 0; i < dim; i++) {
    const float dequant = (float)data[i] * scale + mn;
    s += dequant * dequant;
  }
  return s;
}
// exact top-k of f32 queries over SQ8 rows with the *_simd functions (cosine / dot: best = largest; Euclidean:
// squared distance, best = smallest); ties by row index.  `rows` are quantised here with vo_sq8_quantize.
void vo_scan_topk_sq8(int metric, const float* rows, uint64_t nrows, uint32_t dim, const float* queries, uint32_t nq,
                      uint32_t k, uint32_t nthreads, uint64_t* out_rows, float* out_scores) {
  std::vector<uint8_t> codes((size_t)nrows * dim);
  std::vector<float> mn(nrows), mx(nrows);
  for (uint64_t r = 0; r < nrows; r++)
    vo_sq8_quantize(rows + (size_t)r * dim, dim, codes.data() + (size_t)r * dim, &mn[r], &mx[r]);
  const bool hib = metric != VO_EUCLIDEAN;
  if (nthreads < 1) nthreads = 1;
  std::atomic<uint32_t> next(0);
  auto worker = [&]() {
    std::vector<std::pair<float, uint64_t>> sc(nrows);
    for (;;) {
      const uint32_t qi = next.fetch_add(1);
      if (qi >= nq) break;
      const float* q = queries + (size_t)qi * dim;
      for (uint64_t r = 0; r < nrows; r++) {
        const uint8_t* d = codes.data() + (size_t)r * dim;
        float s;
        if (metric == VO_COSINE)
          s = vo_sq8_cosine(q, d, mn[r], mx[r], dim, 1);
        else if (metric == VO_EUCLIDEAN)
          s = vo_sq8_l2sq(q, d, mn[r], mx[r], dim, 1);
        else
          s = vo_sq8_dot(q, d, mn[r], mx[r], dim);
        sc[r] = {s, r};
      }
      const size_t kk = std::min<size_t>(k, nrows);
      std::partial_sort(sc.begin(), sc.begin() + kk, sc.end(), [hib](const auto& a, const auto& b) {
        int c = hib ? total_cmp(b.first, a.first) : total_cmp(a.first, b.first);
        return c ? c < 0 : a.second < b.second;
      });
      for (size_t i = 0; i < kk; i++) {
        out_rows[(size_t)qi * k + i] = sc[i].second;
        out_scores[(size_t)qi * k + i] = sc[i].first;
      }
    }
  };
  Pool::get().run(nthreads, [&](uint32_t) { worker(); });
}
// exact top-k by BinaryQuantizedVector::hamming_distance between the sign-bit codes (smallest first, ties by row)
void vo_scan_topk_binary(const float* rows, uint64_t nrows, uint32_t dim, const float* queries, uint32_t nq, uint32_t k,
                         uint64_t* out_rows, float* out_scores) {
  const uint32_t nb = (dim + 7) / 8;
  std::vector<uint8_t> codes((size_t)nrows * nb), qc(nb);
  for (uint64_t r = 0; r < nrows; r++) vo_binary_quantize(rows + (size_t)r * dim, dim, codes.data() + (size_t)r * nb);
  std::vector<std::pair<uint32_t, uint64_t>> sc(nrows);
  for (uint32_t qi = 0; qi < nq; qi++) {
    vo_binary_quantize(queries + (size_t)qi * dim, dim, qc.data());
    for (uint64_t r = 0; r < nrows; r++) sc[r] = {vo_binary_hamming(qc.data(), codes.data() + (size_t)r * nb, nb), r};
    const size_t kk = std::min<size_t>(k, nrows);
    std::partial_sort(sc.begin(), sc.begin() + kk, sc.end());
    for (size_t i = 0; i < kk; i++) {
      out_rows[(size_t)qi * k + i] = sc[i].second;
      out_scores[(size_t)qi * k + i] = (float)sc[i].first;
    }
  }
}

const char* vo_build_info(void) {
#if VO_HAVE_AVX2
  return "vdb_oracle: g++ " __VERSION__ " avx2+fma intrinsics, -ffp-contract=off";
#else
  return "vdb_oracle: g++ " __VERSION__ " scalar lanes, -ffp-contract=off";
#endif
}

}  // extern "C"
