// sweep_wide.hip — the selection stage for k BEYOND the candidate buffers of the 256 x 256 selection kernel: exact f32 Cosine /
// DotProduct batches with 10 < k <= kWideMaxK (HnswIndex::search_brute_force over a batch, index/hnsw/index/search.rs:176-219; the
// reference benches k = 10, 50, 100: benches/hnsw_benchmark.rs:152-159).  Rounds 2-5 served such calls from the exact f32 kernels
// (1/16 of the bf16 matrix rate: 80 K q/s at k = 11, the streaming kernels beyond k = 48).
//
// The block-local top-k' of the k <= 10 path does not scale (a block's LDS holds 12 keys for each of its 256 queries), and it is not
// needed: with A_k the k-th best APPROXIMATE score over any set of rows and delta the bound of |approximate - exact|, a row of the
// exact top k has an approximate score >= A_k - 2 delta (sweep_split.hip, seed_scores_bf16).  So the selection kernel's WIDE instance
// (sweep_gemm_bf16.hip) keeps no list at all: under a bound tau = A_k - 2 delta that is constant for a launch, every row that passes
// is appended to the query's GLOBAL list; between two launches of the schedule one block per query finds the k-th best of its list,
// raises tau and drops what fell under it (wide_reseed); behind the last launch the list IS the candidate pool — every row outside it
// has an approximate score below the final tau — and wide_rerank_verify re-scores it with the exact chain of the f32 kernels (oracle
// mode M), ranks it and writes the k best.  The proof that k <= 10 needs per query holds here by construction:
//     exact(outside) <= approx(outside) + delta < tau + delta = A_k - 1.02 delta - ... < A_k - delta <= E_k,
// so a query is unproven only when a list overflowed, the pool is larger than what one block re-scores, or the data is not finite
// — those queries take the exact streaming kernel in gathered mode (the k <= 10 path's own fallback).
// Results: ids, ranks and score bits of the exact kernels, as everywhere else (tests/test_gpu_wide_k.py).
// Since selector level 3 (the library's default, round 6) the same path serves k <= 10 as well — it measured faster at every k
// (select_stage.hip select_level_wide) — and Euclidean batches (their augmented DotProduct form) and the SQ8 storage mode have
// their own seed / re-scoring kernels below.  Behind the LAST launch wide_rerank_verify takes wide_reseed's step itself (final
// bound from the whole list, the pool formed in LDS): one launch less per batch.
#include <algorithm>

#include "vdb_device.hpp"
#include "vdb_kernels.hpp"
#include "vdb_wide.hpp"
#include "vdb_block_select.hpp"

namespace vdb {

// level 2's error bound with measured residuals (sweep_split.hip select_eps_q, restated: that one is file-local)
__device__ __forceinline__ float wide_eps(uint32_t dim, const float* rho_q, const uint32_t* rho_max_bits, uint32_t q, float extra = 0.0f) {
  const float acc = 16.0f * (float)dim * 5.9604645e-8f;
  if (!rho_q || !rho_max_bits) return 2.0f * 3.90625e-3f * 1.002f + 1.6e-5f + acc + extra;
  const float rm = __uint_as_float(*rho_max_bits), rq = rho_q[q];
  return (rm + rq + 3.0f * rm * rq) * 1.002f + acc + 4e-6f + extra;  // (+ 4e-6: the normalised images' division by f32 norms; NaN query: NaN -> no bound)
}

// the bound a k-th best approximate score s_k yields: tau = s_k - 2 delta (both sides approximate), as a key of row 0 — a row passes
// `key < tau` when its score is above it.  No bound (NaN / inf arithmetic): the query is given up (sweep_wide.hip header)
__device__ __forceinline__ uint64_t wide_tau_key(float s_k, float delta, bool* ok, float extra = 0.0f) {
  const float lowered = s_k - 2.0f * delta * 1.01f - extra - fabsf(s_k) * 1e-6f;
  *ok = lowered == lowered && fabsf(lowered) < 3.0e38f;
  return make_key<true>(lowered, 0u);
}
// a bound nothing finite passes: the query is out of the selection (its list has overflowed or no bound exists) and the launches
// that follow must not spend their epilogues on it
__device__ __forceinline__ uint64_t wide_tau_closed() { return make_key<true>(3.0e38f, 0u); }

// ---- seed: the k-th best of the sample keys (one per 16 seed rows: seed_scores_bf16) -> tau, delta; empty list -------------------
template <int METRIC>
__global__ __launch_bounds__(256) void wide_seed_kernel(WideArgs a, const uint64_t* seed_keys, uint32_t ngrp) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t ctl[2];
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  constexpr int NPT = kWideSeedGroups / 256;
  uint64_t keys[NPT];
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < NPT; j++) {
    const uint32_t i = tid + 256u * (uint32_t)j;
    keys[j] = i < ngrp ? seed_keys[(size_t)q * ngrp + i] : kKeyInvalid;
    mine += keys[j] != kKeyInvalid ? 1u : 0u;
  }
  // (the query's words are read beside the keys, not behind the barriers)
  const float eps = wide_eps(a.dim, a.rho_q, a.rho_max_bits, q, a.eps_extra);
  const float d = METRIC == kCosine ? eps * 1.001f + 4e-7f : eps * 1.001f * a.qnorms[q] * __uint_as_float(*a.norm_max_bits) + 1e-30f;
  __shared__ uint32_t total;
  if (tid == 0) total = 0;
  __syncthreads();
  if (mine) atomicAdd(&total, mine);
  __syncthreads();
  const uint32_t valid = total;  // (block-uniform)
  bool ok = valid >= a.k && d == d;
  uint64_t tau = wide_tau_closed();
  if (ok) {  // (block-uniform)
    __shared__ uint64_t s256[256];
    const uint32_t hi = ngrp <= 256 ? block_kth_hi_256(keys[0], a.k, s256, ctl) : block_kth_hi<NPT>(keys, a.k, hist, ctl);
    tau = wide_tau_key(key_score<true>((uint64_t)hi << 32), d, &ok);
    if (!ok) tau = wide_tau_closed();
  }
  if (tid == 0) {
    a.delta[q] = d;
    a.tau[q] = tau;
    a.cnt[q] = 0;
    a.state[q] = ok ? 0u : kWideGivenUp;
  }
}

// ---- between two launches, and behind the last: k-th best of the list -> tau; entries under the new tau leave the list ---------
// the list's next bound: its k-th best entry lowered by 2 delta (+ extra), never below the bound the list was filled under
// (delta, extra, old = the query's words, read by the caller beside its list, not behind the selection).  Block-uniform result; 0 = the bound stays (fewer than k entries), 1 = new bound in *tau_out, 2 = the query is given up (entries were dropped, or no
// finite bound exists).  keys = the list in registers (NPT per thread, kKeyInvalid past `raw`).
template <int NPT>
__device__ int wide_next_tau(const WideArgs& a, uint32_t raw, const uint64_t (&keys)[NPT], float delta, float extra, uint64_t old, uint32_t* hist,
                             uint32_t* ctl, uint64_t* s256, uint64_t* tau_out) {
  if (raw > a.cap) return 2;  // entries were dropped: the list no longer holds every row above the bound
  if (raw < a.k) return 0;    // fewer than k rows passed so far: the bound stays (it is valid for any set of rows)
  const uint32_t hi = raw <= 256 ? block_kth_hi_256(keys[0], a.k, s256, ctl, (raw + 7u) & ~7u) : block_kth_hi<NPT>(keys, a.k, hist, ctl);
  bool ok;
  uint64_t tau = wide_tau_key(key_score<true>((uint64_t)hi << 32), delta, &ok, extra);
  if (!ok) return 2;
  if (tau > old) tau = old;  // (keys: smaller = a higher bound) never lower the bound the list was filled under
  *tau_out = tau;
  return 1;
}

__global__ __launch_bounds__(256) void wide_reseed_kernel(WideArgs a) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t ctl[2];
  __shared__ uint32_t wsum[4];
  __shared__ uint64_t s256[256];
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  const uint32_t state = a.state[q], raw = a.cnt[q];
  const float delta = a.delta[q], extra = a.extra ? a.extra[q] : 0.0f;
  const uint64_t old = a.tau[q];
  if (state & kWideGivenUp) return;  // (block-uniform)
  constexpr int NPT = kWideCap / 256;
  uint64_t* list = a.keys + (size_t)q * a.cap;
  uint64_t keys[NPT];
  keys[0] = list[tid];  // (requested beside the count, not behind it: the buffer holds `cap` entries whatever the count says)
  if (tid >= raw) keys[0] = kKeyInvalid;
#pragma unroll
  for (int j = 1; j < NPT; j++) {
    const uint32_t i = tid + 256u * (uint32_t)j;
    keys[j] = (i < raw && i < a.cap) ? list[i] : kKeyInvalid;
  }
  uint64_t tau = 0;
  const int what = wide_next_tau<NPT>(a, raw, keys, delta, extra, old, hist, ctl, s256, &tau);
  if (what == 0) return;
  if (what == 2) {
    if (tid == 0) {
      a.state[q] |= kWideGivenUp;
      a.tau[q] = wide_tau_closed();
    }
    return;
  }
  // compaction: the entries that still pass, in any order
  uint32_t keep = 0;
#pragma unroll
  for (int j = 0; j < NPT; j++) keep += (keys[j] != kKeyInvalid && keys[j] < tau) ? 1u : 0u;
  uint32_t incl = keep;
  const uint32_t lane = tid & 63u, w = tid >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t up = __shfl_up(incl, o, 64);
    if ((int)lane >= o) incl += up;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();  // (every thread has read its entries: the list may be rewritten)
  uint32_t base = incl - keep;
  for (uint32_t x = 0; x < w; x++) base += wsum[x];
#pragma unroll
  for (int j = 0; j < NPT; j++)
    if (keys[j] != kKeyInvalid && keys[j] < tau) list[base++] = keys[j];
  if (tid == 0) {
    a.cnt[q] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    a.tau[q] = tau;
  }
}

// ---- the pool, re-scored exactly: one block per query ------------------------------------------------------------------------
// Candidates in chunks of 64 through the staging scheme of split_rerank_verify (rows travel through LDS in steps of 64 elements,
// double-buffered; one fmaf chain per candidate in oracle mode M's order k = 128 U + 16 m + 4 kk + c, the vector zero-padded to a
// multiple of 128); then every candidate is ranked by counting among the exact keys and the k best are written in rank order.
template <int METRIC>
__global__ __launch_bounds__(256) void wide_rerank_verify(WideArgs a, WideOutArgs o, bool fuse) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr uint32_t kStep = 64, kStride = kStep + 4, kChunk = 64;
  float* qs = reinterpret_cast<float*>(smem);                               // [dim_pad]
  uint64_t* ekeys = reinterpret_cast<uint64_t*>(qs + o.dim_pad);             // [kWidePoolMax] exact keys
  float* stage = reinterpret_cast<float*>(ekeys + kWidePoolMax);             // [2][kChunk][kStride]
  uint32_t* crow = reinterpret_cast<uint32_t*>(stage + 2 * kChunk * kStride);  // [kChunk]
  uint64_t* kth = reinterpret_cast<uint64_t*>(crow + kChunk);                // [1] exact key of rank k - 1
  __shared__ uint32_t cand[kWidePoolMax];  // the pool: row numbers
  __shared__ uint32_t hist[256];
  __shared__ uint32_t ctl[2];
  __shared__ uint32_t wsum[4];
  __shared__ uint64_t s256[256];
  const uint32_t tid = threadIdx.x, qi = blockIdx.x;
  // (probe builds: where a block's time goes — 100 MHz wall clock, stamps of thread 0)
  auto stamp = [&](int i) {
    if (o.stamps && tid == 0) o.stamps[(size_t)qi * 8 + i] = wall_clock64();
  };
  stamp(0);
  // everything that depends on nothing is requested first: the query (its first 1 024 elements), the list's first 256 entries
  const float* q = o.queries + (size_t)qi * o.q_stride;
  const uint64_t* list = a.keys + (size_t)qi * a.cap;
  float qreg[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t i = tid + 256u * (uint32_t)j;
    qreg[j] = i < a.dim ? q[i] : 0.0f;
  }
  const uint64_t first = list[tid];  // (the buffer holds `cap` entries whatever the count says)
  const float qn = METRIC == kCosine ? a.qnorms[qi] : 0.0f;
  const uint32_t raw = a.cnt[qi];
  bool given_up = (a.state[qi] & kWideGivenUp) != 0 || raw > a.cap;
  uint64_t tau_fin = a.tau[qi];
  const float delta = a.delta[qi], extra = a.extra ? a.extra[qi] : 0.0f;
  uint32_t n = 0;
  if (fuse) {
    // the step wide_reseed_kernel takes behind the last launch, done here: the final bound from the whole list, the pool = the entries
    // that pass it (the list itself is left as it is)
    constexpr int NPT = kWideCap / 256;
    uint64_t keys[NPT];
    keys[0] = (!given_up && tid < raw) ? first : kKeyInvalid;
#pragma unroll
    for (int j = 1; j < NPT; j++) {
      const uint32_t i = tid + 256u * (uint32_t)j;
      keys[j] = (!given_up && i < raw) ? list[i] : kKeyInvalid;
    }
    if (!given_up) {  // (block-uniform)
      uint64_t t = 0;
      const int what = wide_next_tau<NPT>(a, raw, keys, delta, extra, tau_fin, hist, ctl, s256, &t);
      if (what == 2) given_up = true;
      if (what == 1) tau_fin = t;
    }
    uint32_t keep = 0;
#pragma unroll
    for (int j = 0; j < NPT; j++) keep += (keys[j] != kKeyInvalid && keys[j] < tau_fin) ? 1u : 0u;
    uint32_t incl = keep;
    const uint32_t lane = tid & 63u, w = tid >> 6;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
      const uint32_t up = __shfl_up(incl, s, 64);
      if ((int)lane >= s) incl += up;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    uint32_t base = incl - keep;
    for (uint32_t x = 0; x < w; x++) base += wsum[x];
    const uint32_t total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (total > kWidePoolMax) given_up = true;
    if (!given_up) {
      n = total;
#pragma unroll
      for (int j = 0; j < NPT; j++)
        if (keys[j] != kKeyInvalid && keys[j] < tau_fin) cand[base++] = key_row(keys[j]);
    }
  } else {
    if (raw > kWidePoolMax) given_up = true;
    n = given_up ? 0u : raw;
    if (tid < n) cand[tid] = key_row(first);
    for (uint32_t i = tid + 256; i < n; i += 256) cand[i] = key_row(list[i]);
  }
  stamp(1);  // the list read, the final bound found, the pool formed
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t i = tid + 256u * (uint32_t)j;
    if (i < o.dim_pad) qs[i] = qreg[j];
  }
  for (uint32_t i = tid + 1024; i < o.dim_pad; i += 256) qs[i] = i < a.dim ? q[i] : 0.0f;
  if (tid == 0) *kth = kKeyInvalid;
  for (uint32_t c0 = 0; c0 < n; c0 += kChunk) {
    const uint32_t nc = min(kChunk, n - c0);
    __syncthreads();  // (qs and the pool written; the previous chunk's chains are done with the stage and crow)
    if (tid < nc) crow[tid] = cand[c0 + tid];
    __syncthreads();
    // step width: a buffer holds 4 096 values (+ 4 of padding per row) — 64 columns of 64 rows, or more columns of fewer rows (a pool
    // is ~25 rows at k = 10: steps of 128 columns halve the exposed fetch latencies of the chunk)
    uint32_t sh = 0;
    while (sh < 3 && (nc << (sh + 1)) <= kChunk) sh++;
    const uint32_t W = kStep << sh, rowf4 = (kStep / 4) << sh, wstride = W + 4;
    const uint32_t nf4 = nc * rowf4;
    // three steps in flight: while step s multiplies out of its buffer, step s + 1 sits in a register set (requested two steps
    // ago), steps s + 2 and s + 3 are on their way into two more.  Measured with stamped blocks (profiles/r06u_*): one request in
    // flight per block cost 292 us per batch at k = 100 (pools of ~256 rows = 0.8 GB of rows), two 207 us, three the same — the
    // steps then run at the rate HBM delivers 256-byte segments of rows spread over the corpus (3.9 TB/s), not at a latency.
    float4 vr[4][4];
    auto fetch = [&](float4 (&v)[4], uint32_t U) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t f = tid + 256u * (uint32_t)i;
        const uint32_t r = f >> (4 + sh), c4 = f & (rowf4 - 1);
        v[i] = (f < nf4 && U + 4 * c4 < a.dim) ? ld4(o.rows + (size_t)crow[r] * o.row_stride + U + 4 * c4) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      }
    };
    auto park = [&](const float4 (&v)[4], uint32_t buf) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t f = tid + 256u * (uint32_t)i;
        const uint32_t r = f >> (4 + sh), c4 = f & (rowf4 - 1);
        if (f < nf4) *reinterpret_cast<float4*>(stage + (size_t)buf * kChunk * kStride + (size_t)r * wstride + 4 * c4) = v[i];
      }
    };
    float acc = 0.0f;
    auto chain = [&](uint32_t U, uint32_t buf) {
      if (tid < nc) {
        const float* x = stage + (size_t)buf * kChunk * kStride + (size_t)tid * wstride;
        const uint32_t groups = min(W, o.dim_pad - U) / 16;  // (dim_pad is a multiple of 128: the padding the chain sees does not depend on W)
#pragma unroll 4
        for (uint32_t m = 0; m < groups; m++) {
          float xr[16];
#pragma unroll
          for (int e = 0; e < 16; e += 4) {
            const float4 w = *reinterpret_cast<const float4*>(x + 16 * m + e);
            xr[e] = w.x; xr[e + 1] = w.y; xr[e + 2] = w.z; xr[e + 3] = w.w;
          }
          const uint32_t base = U + 16 * m;
#pragma unroll
          for (int c = 0; c < 4; c++)
#pragma unroll
            for (int kk = 0; kk < 4; kk++) acc = __builtin_fmaf(xr[4 * kk + c], qs[base + 4 * kk + c], acc);
        }
      }
    };
    const uint32_t nsteps = (o.dim_pad + W - 1) / W;
    fetch(vr[0], 0);
    if (1 < nsteps) fetch(vr[1], W);
    if (2 < nsteps) fetch(vr[2], 2 * W);
    const float vnorm = (METRIC == kCosine && tid < nc) ? o.norms[crow[tid]] : 1.0f;  // (beside the rows, not behind the chain)
    park(vr[0], 0);
    __syncthreads();
    if (c0 == 0) stamp(2);  // the first chunk's first step has arrived
    for (uint32_t s0 = 0; s0 < nsteps; s0 += 4) {
#pragma unroll
      for (int j = 0; j < 4; j++) {  // (unrolled: the register sets are named at compile time)
        const uint32_t st = s0 + (uint32_t)j;
        if (st < nsteps) {  // (block-uniform)
          if (st + 3 < nsteps) fetch(vr[(j + 3) & 3], (st + 3) * W);
          chain(st * W, st & 1u);
          if (st + 1 < nsteps) park(vr[(j + 1) & 3], (st + 1) & 1u);
          __syncthreads();
        }
      }
    }
    if (c0 == 0) stamp(3);  // the first chunk's chains
    if (tid < nc) {
      const uint32_t row = crow[tid];
      const float score = finish_score<METRIC>(acc, qn, vnorm);
      ekeys[c0 + tid] = make_key<true>(score, row);
    }
  }
  __syncthreads();
  stamp(4);  // every chunk
  // rank by counting (keys are unique: a row appears once in a list — every row is swept by exactly one block of one launch)
  const uint32_t kk = min(a.k, n);
  for (uint32_t i = tid; i < n; i += 256) {
    const uint64_t key = ekeys[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; j++) rank += ekeys[j] < key ? 1u : 0u;
    if (rank < kk) {
      const uint32_t row = key_row(key);
      o.out_ids[(size_t)qi * a.k + rank] = o.ext_ids ? o.ext_ids[row] : (uint64_t)row;
      o.out_scores[(size_t)qi * a.k + rank] = key_score<true>(key);
      if (rank + 1 == a.k) *kth = key;
    }
  }
  for (uint32_t e = kk + tid; e < a.k; e += 256) {
    o.out_ids[(size_t)qi * a.k + e] = ~0ull;
    o.out_scores[(size_t)qi * a.k + e] = __uint_as_float(0x7FC00000u);
  }
  __syncthreads();
  stamp(5);  // ranked and written
  if (tid == 0) {
    // proof (file header): every row outside the list has an approximate score under the final bound
    bool ok = !given_up && n >= a.k;
    if (ok) {
      const float A = key_score<true>(tau_fin);
      const double Ek = (double)key_score<true>(*kth);
      ok = Ek > (double)A + (double)delta;  // false for NaN anywhere
    }
    if (fuse) a.tau[qi] = tau_fin;
    o.flags[qi] = ok ? 0u : 1u;
    if (!ok) {  // the query lists itself for the gathered exact pass (sweep_split.hip list_unproven's rule)
      const uint32_t j = atomicAdd(o.qcount, 1u);
      o.qmap[j] = qi;
      o.qslot[qi] = j;
    }
    o.out_n[qi] = kk;
    if (o.stamps) {
      o.stamps[(size_t)qi * 8 + 6] = wall_clock64();
      o.stamps[(size_t)qi * 8 + 7] = ((unsigned long long)raw << 32) | n;
    }
  }
}

// ---- Euclidean batches: the augmented DotProduct form (sweep_split.hip: s = q.v - |v|^2 / 2 as a dot product of (v, -h_hi, -h_lo) and
// (q, 1, 1)); the nearest rows are the rows with the largest s.  Error of a selection score against the true s: delta = eps |q| max|v|
// (the bf16 roundings, measured residuals) + (2^-16 + acc) max h (l2_seed_kernel's formula).  The exact answer is ranked by the
// CANONICAL squared sums (mode C: lane chains + butterfly), which differ from |q|^2 - 2 s_true by at most
// slack = 8 dim 2^-24 (|q|^2 + max|v|^2): the bound is lowered by slack on top of 2 delta (in s: half of it would do), and the proof
// is CHECKED per query in the squared-distance domain as l2_rerank_verify checks it — (E_k + slack) < |q|^2 - 2 (tau + delta) — instead
// of holding by construction.
__global__ __launch_bounds__(256) void wide_seed_l2_kernel(WideArgs a, const uint64_t* seed_keys, uint32_t ngrp, uint32_t dim_a) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t ctl[2];
  __shared__ uint32_t total;
  const uint32_t q = blockIdx.x, tid = threadIdx.x;
  constexpr int NPT = kWideSeedGroups / 256;
  uint64_t keys[NPT];
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < NPT; j++) {
    const uint32_t i = tid + 256u * (uint32_t)j;
    keys[j] = i < ngrp ? seed_keys[(size_t)q * ngrp + i] : kKeyInvalid;
    mine += keys[j] != kKeyInvalid ? 1u : 0u;
  }
  if (tid == 0) total = 0;
  __syncthreads();
  if (mine) atomicAdd(&total, mine);
  __syncthreads();
  const uint32_t valid = total;
  const float nmax = __uint_as_float(*a.norm_max_bits), hmax = 0.5f * nmax * nmax, qn = a.qnorms[q];
  const float acc = 16.0f * (float)dim_a * 5.9604645e-8f;
  float eps_r = 2.0f * 3.90625e-3f * 1.002f + 1.6e-5f + acc;
  if (a.rho_q && a.rho_max_bits) {
    const float rm = __uint_as_float(*a.rho_max_bits), rq = a.rho_q[q];
    eps_r = (rm + rq + 3.0f * rm * rq) * 1.002f + acc;
  }
  const float d = eps_r * 1.001f * qn * nmax + (1.6e-5f + acc) * hmax + 1e-30f;
  const float slack = 8.0f * (float)a.dim * 5.9604645e-8f * (qn * qn + nmax * nmax) * 1.01f + 1e-30f;
  bool ok = valid >= a.k && d == d && slack == slack;
  uint64_t tau = wide_tau_closed();
  if (ok) {
    const uint32_t hi = block_kth_hi<NPT>(keys, a.k, hist, ctl);
    tau = wide_tau_key(key_score<true>((uint64_t)hi << 32), d, &ok, slack);
    if (!ok) tau = wide_tau_closed();
  }
  if (tid == 0) {
    a.delta[q] = d;
    a.extra[q] = slack;
    a.tau[q] = tau;
    a.cnt[q] = 0;
    a.state[q] = ok ? 0u : kWideGivenUp;
  }
}

// the pool re-scored with the canonical (q - v)^2 chain of the small-batch kernels (l2_rerank_verify's arithmetic: float4 chunk c in
// lane c mod 64, fmaf chains over (q - v)^2, xor butterfly, sqrt): a wave takes four candidates at a time; ranking by (distance, row)
__global__ __launch_bounds__(256) void wide_rerank_l2(WideArgs a, WideOutArgs o) {
  __shared__ uint64_t ekeys[kWidePoolMax];
  __shared__ float esums[kWidePoolMax];
  __shared__ float kth_sum;
  const int lane = lane_id();
  const uint32_t wib = threadIdx.x >> 6, tid = threadIdx.x, qi = blockIdx.x;
  const uint32_t raw = a.cnt[qi];
  const bool given_up = (a.state[qi] & kWideGivenUp) != 0 || raw > kWidePoolMax || raw > a.cap;
  const uint32_t n = given_up ? 0u : raw;
  const float* q = o.queries + (size_t)qi * o.q_stride;
  const uint64_t* list = a.keys + (size_t)qi * a.cap;
  const int d4 = (int)((a.dim + 3) / 4);
  if (tid == 0) kth_sum = __uint_as_float(0x7FC00000u);
  for (uint32_t c0 = wib; c0 < n; c0 += 16) {
    const float* p[4];
    uint32_t rowv[4];
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t c = min(c0 + 4u * (uint32_t)u, n - 1u);  // (a clamped duplicate: computed, not stored)
      rowv[u] = key_row(list[c]);
      p[u] = o.rows + (size_t)rowv[u] * o.row_stride;
    }
    for (int ch = lane; ch < d4; ch += 64) {
      const int nv = (int)a.dim - ch * 4;
      float4 x[4];
#pragma unroll
      for (int u = 0; u < 4; u++) x[u] = ld4(p[u] + ch * 4);
      float4 qq;
      if (nv >= 4) {
        qq = make_float4(q[ch * 4], q[ch * 4 + 1], q[ch * 4 + 2], q[ch * 4 + 3]);
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = chain4<kOpL2>(acc[u], qq, x[u]);
      } else {
        qq = make_float4(q[ch * 4], nv > 1 ? q[ch * 4 + 1] : 0.f, nv > 2 ? q[ch * 4 + 2] : 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 4; u++) acc[u] = chain4_tail<kOpL2>(acc[u], qq, x[u], nv);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t c = c0 + 4u * (uint32_t)u;
      const float sum = butterfly_all(acc[u]);
      if (lane == 0 && c < n) {
        esums[c] = sum;
        ekeys[c] = make_key<false>(finish_score<kEuclidean>(sum, 0.f, 0.f), rowv[u]);
      }
    }
  }
  __syncthreads();
  const uint32_t kk = min(a.k, n);
  for (uint32_t i = tid; i < n; i += 256) {
    const uint64_t key = ekeys[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; j++) rank += ekeys[j] < key ? 1u : 0u;
    if (rank < kk) {
      const uint32_t row = key_row(key);
      o.out_ids[(size_t)qi * a.k + rank] = o.ext_ids ? o.ext_ids[row] : (uint64_t)row;
      o.out_scores[(size_t)qi * a.k + rank] = key_score<false>(key);
      if (rank + 1 == a.k) kth_sum = esums[i];
    }
  }
  for (uint32_t e = kk + tid; e < a.k; e += 256) {
    o.out_ids[(size_t)qi * a.k + e] = ~0ull;
    o.out_scores[(size_t)qi * a.k + e] = __uint_as_float(0x7FC00000u);
  }
  __syncthreads();
  if (tid == 0) {
    bool ok = !given_up && n >= a.k;
    if (ok) {  // every row outside the list has a selection score <= tau, a true s <= tau + delta, a true squared distance >= |q|^2 - 2 (tau + delta)
      const double A = (double)key_score<true>(a.tau[qi]);
      const double qn = (double)a.qnorms[qi];
      const double Ek = (double)kth_sum;
      ok = (Ek + (double)a.extra[qi] + 1e-6 * Ek) < qn * qn - 2.0 * (A + (double)a.delta[qi]);  // false for NaN anywhere
    }
    o.flags[qi] = ok ? 0u : 1u;
    if (!ok) {
      const uint32_t j = atomicAdd(o.qcount, 1u);
      o.qmap[j] = qi;
      o.qslot[qi] = j;
    }
    o.out_n[qi] = kk;
  }
}

// ---- SQ8 storage mode: the pool re-scored with the reference's asymmetric distances over the one-byte codes (split_rerank_verify<SQ8>'s
// chain: core/quantization.rs:410-554 — one left-to-right chain per (query, row), multiplies and adds rounded separately; the bits
// sweep_topk_sq8 computes for the whole corpus).  One thread per candidate; the proof holds by construction as for the f32 rows (the
// error bound carries level 3's extra term for the chain's own distance from the dequantised dot product).
template <int METRIC>
__global__ __launch_bounds__(256) void wide_rerank_sq8(WideArgs a, WideOutArgs o) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* qs = reinterpret_cast<float*>(smem);                          // [dim]
  uint64_t* ekeys = reinterpret_cast<uint64_t*>(qs + ((a.dim + 3) & ~3u));  // [kWidePoolMax]
  float* qred = reinterpret_cast<float*>(ekeys + kWidePoolMax);         // sum(q), sum(q*q), left to right
  uint64_t* kth = reinterpret_cast<uint64_t*>(qred + 2);
  const uint32_t tid = threadIdx.x, qi = blockIdx.x;
  const uint32_t raw = a.cnt[qi];
  const bool given_up = (a.state[qi] & kWideGivenUp) != 0 || raw > kWidePoolMax || raw > a.cap;
  const uint32_t n = given_up ? 0u : raw;
  const float* q = o.queries + (size_t)qi * o.q_stride;
  for (uint32_t i = tid; i < a.dim; i += 256) qs[i] = q[i];
  if (tid == 0) *kth = kKeyInvalid;
  __syncthreads();
  if (tid == 64) {  // :330-334 sum(q); :528 sum(q*q) — two chains, two waves
    float s1 = 0.0f;
    for (uint32_t d = 0; d < a.dim; d++) s1 = __fadd_rn(s1, qs[d]);
    qred[0] = s1;
  }
  if (tid == 128) {
    float s2 = 0.0f;
    for (uint32_t d = 0; d < a.dim; d++) s2 = __fadd_rn(s2, __fmul_rn(qs[d], qs[d]));
    qred[1] = s2;
  }
  __syncthreads();
  const uint64_t* list = a.keys + (size_t)qi * a.cap;
  for (uint32_t c = tid; c < n; c += 256) {
    const uint32_t row = key_row(list[c]);
    const float mn = o.sq8_min[row], range = __fsub_rn(o.sq8_max[row], mn);
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(o.sq8_codes + (size_t)row * o.sq8_stride);
    float acc = 0.0f;
    if (range < 1.1920929e-07f) {
      acc = __fmul_rn(qred[0], mn);
    } else {
      const float scale = __fdiv_rn(range, 255.0f);
      for (uint32_t i = 0; i < a.dim; i += 4) {
        const uint32_t w = cw[i >> 2];
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (i + e < a.dim) {
            const float dq = __fadd_rn(__fmul_rn((float)((w >> (8 * e)) & 0xFFu), scale), mn);
            acc = __fadd_rn(acc, __fmul_rn(qs[i + e], dq));
          }
      }
    }
    float score = acc;
    if (METRIC == kCosine) {
      const float denom = sqrtf(__fmul_rn(qred[1], o.sq8_nsq[row]));
      score = denom < 1.1920929e-07f ? 0.0f : __fdiv_rn(acc, denom);
    }
    ekeys[c] = make_key<true>(score, row);
  }
  __syncthreads();
  const uint32_t kk = min(a.k, n);
  for (uint32_t i = tid; i < n; i += 256) {
    const uint64_t key = ekeys[i];
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n; j++) rank += ekeys[j] < key ? 1u : 0u;
    if (rank < kk) {
      const uint32_t row = key_row(key);
      o.out_ids[(size_t)qi * a.k + rank] = o.ext_ids ? o.ext_ids[row] : (uint64_t)row;
      o.out_scores[(size_t)qi * a.k + rank] = key_score<true>(key);
      if (rank + 1 == a.k) *kth = key;
    }
  }
  for (uint32_t e = kk + tid; e < a.k; e += 256) {
    o.out_ids[(size_t)qi * a.k + e] = ~0ull;
    o.out_scores[(size_t)qi * a.k + e] = __uint_as_float(0x7FC00000u);
  }
  __syncthreads();
  if (tid == 0) {
    bool ok = !given_up && n >= a.k;
    if (ok) {
      const float A = key_score<true>(a.tau[qi]);
      const double Ek = (double)key_score<true>(*kth);
      ok = Ek > (double)A + (double)a.delta[qi];  // false for NaN anywhere
    }
    o.flags[qi] = ok ? 0u : 1u;
    if (!ok) {
      const uint32_t j = atomicAdd(o.qcount, 1u);
      o.qmap[j] = qi;
      o.qslot[qi] = j;
    }
    o.out_n[qi] = kk;
  }
}

// ---- launchers -------------------------------------------------------------------------------------------------------------------
void launch_wide_seed(int metric, const WideArgs& a, const uint64_t* seed_keys, uint32_t ngrp, uint32_t nq, hipStream_t st) {
  if (metric == kCosine)
    hipLaunchKernelGGL((wide_seed_kernel<kCosine>), dim3(nq), dim3(256), 0, st, a, seed_keys, ngrp);
  else
    hipLaunchKernelGGL((wide_seed_kernel<kDot>), dim3(nq), dim3(256), 0, st, a, seed_keys, ngrp);
}
void launch_wide_reseed(const WideArgs& a, uint32_t nq, hipStream_t st) { hipLaunchKernelGGL(wide_reseed_kernel, dim3(nq), dim3(256), 0, st, a); }
void launch_wide_seed_l2(const WideArgs& a, const uint64_t* seed_keys, uint32_t ngrp, uint32_t dim_a, uint32_t nq, hipStream_t st) {
  hipLaunchKernelGGL(wide_seed_l2_kernel, dim3(nq), dim3(256), 0, st, a, seed_keys, ngrp, dim_a);
}
void launch_wide_rerank_l2(const WideArgs& a, const WideOutArgs& o, uint32_t nq, hipStream_t st) {
  hipLaunchKernelGGL(wide_rerank_l2, dim3(nq), dim3(256), 0, st, a, o);
}
void launch_wide_rerank_sq8(int metric, const WideArgs& a, const WideOutArgs& o, uint32_t nq, hipStream_t st) {
  const size_t lds = ((size_t)((a.dim + 3) & ~3u) * 4 + (size_t)kWidePoolMax * 8 + 8 + 8 + 15) & ~(size_t)15;
  if (metric == kCosine)
    hipLaunchKernelGGL((wide_rerank_sq8<kCosine>), dim3(nq), dim3(256), lds, st, a, o);
  else
    hipLaunchKernelGGL((wide_rerank_sq8<kDot>), dim3(nq), dim3(256), lds, st, a, o);
}
size_t wide_rerank_lds_bytes(uint32_t dim_pad) {
  return ((size_t)dim_pad * 4 + (size_t)kWidePoolMax * 8 + 2 * (size_t)64 * 68 * 4 + 64 * 4 + 8 + 15) & ~(size_t)15;
}
void launch_wide_rerank(int metric, const WideArgs& a, const WideOutArgs& o, uint32_t nq, bool fuse_final_reseed, hipStream_t st) {
  const size_t lds = wide_rerank_lds_bytes(o.dim_pad);
  if (metric == kCosine)
    hipLaunchKernelGGL((wide_rerank_verify<kCosine>), dim3(nq), dim3(256), lds, st, a, o, fuse_final_reseed);
  else
    hipLaunchKernelGGL((wide_rerank_verify<kDot>), dim3(nq), dim3(256), lds, st, a, o, fuse_final_reseed);
}

}  // namespace vdb
