// multi_cu_walk_step.hip — "one query, several CUs" (VERDICT r04 item 5), MEASURED before it is built: the cost of ONE layer-0
// expansion step of the graph walk (NativeHnsw::search_layer, native/graph.rs:438-520) when its <= 64 neighbour rows are evaluated
//   (A) by the CU that owns the walk (what hnsw_search_kernel's latency-mode instance does: a 1 024-thread block, 16 waves, 4 rows
//       per wave), against
//   (B) by P CUs: the leader publishes the 64 neighbour ids, P - 1 helper workgroups (persistent, polling) and the leader itself
//       each evaluate 64 / P rows, the helpers publish their distances, the leader gathers them.
// Both variants run the SAME dependent chain — the ids of step s + 1 are a hash of the smallest distance of step s, so nothing can be
// prefetched, exactly as in a walk — over the same corpus (n x 768 f32, random rows).  Hand-offs follow MI355X_MICROARCH.md's recipe
// for small cross-CU payloads: one naturally aligned 8-byte {data, tag} granule per value, written by ONE write-through (sc1-class:
// agent-scope relaxed atomic) store and polled with agent-scope relaxed loads; no fences (a granule carries its own tag).
//
// Not product code: a probe.  Build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/out/multi_cu_walk_step tools/probes/multi_cu_walk_step.hip
// Run:   tools/probes/out/multi_cu_walk_step <rows> <steps>        (prints one line per variant)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                            \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));  \
      exit(2);                                                                           \
    }                                                                                    \
  } while (0)

constexpr int kDim = 768, kNb = 64;

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}
// one row . query, a wave per row (12 floats per lane), as the walk kernels do
__device__ __forceinline__ float row_dot(const float* rows, uint32_t id, const float4 (&q)[3], int lane) {
  const float4* p = reinterpret_cast<const float4*>(rows + (size_t)id * kDim);
  float acc = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float4 x = p[c * 64 + lane];
    acc += x.x * q[c].x + x.y * q[c].y + x.z * q[c].z + x.w * q[c].w;
  }
  return wave_sum(acc);
}

// (A) the whole step on one CU: 1 024 threads, wave w evaluates rows w, w + 16, w + 32, w + 48
__global__ __launch_bounds__(1024) void step_one_cu(const float* rows, uint32_t n, const float* query, uint32_t steps, float* out, long long* ticks) {
  __shared__ uint32_t ids[kNb];
  __shared__ float dist[kNb];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float4 q[3];
#pragma unroll
  for (int c = 0; c < 3; c++) q[c] = reinterpret_cast<const float4*>(query)[c * 64 + lane];
  if (threadIdx.x < kNb) ids[threadIdx.x] = mix(threadIdx.x * 2654435761u) % n;
  __syncthreads();
  float best = 0.0f;
  const long long t0 = wall_clock64();
  for (uint32_t s = 0; s < steps; s++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float d = row_dot(rows, ids[w + 16 * j], q, lane);
      if (lane == 0) dist[w + 16 * j] = d;
    }
    __syncthreads();
    if (w == 0) {  // the leader wave: smallest distance of the 64 -> the next step's ids (a dependent chain)
      float d = dist[lane];
#pragma unroll
      for (int sft = 32; sft >= 1; sft >>= 1) d = fminf(d, __shfl_xor(d, sft, 64));
      best = d;
      ids[lane] = mix(__float_as_uint(d) + lane * 0x9E3779B9u + s) % n;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *ticks = wall_clock64() - t0;
    *out = best;
  }
}

// (B) P workgroups of 256 threads; group g evaluates rows [g * 64 / P, (g + 1) * 64 / P); group 0 is the leader.
// mail_ids[64], mail_d[64]: 8-byte granules {value (low 32), tag = step + 1 (high 32)}
__global__ __launch_bounds__(256) void step_multi_cu(const float* rows, uint32_t n, const float* query, uint32_t steps, int P, int stride,
                                                     unsigned long long* mail_ids, unsigned long long* mail_d, float* out, long long* ticks) {
  if ((int)blockIdx.x % stride != 0) return;           // placement: stride 1 = consecutive blocks (8 XCDs), stride 8 = one XCD
  const int g = (int)blockIdx.x / stride;
  if (g >= P) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int per = kNb / P;                              // rows per group (P in {2, 4, 8, 16})
  __shared__ uint32_t ids[kNb];
  __shared__ float dist[kNb];
  float4 q[3];
#pragma unroll
  for (int c = 0; c < 3; c++) q[c] = reinterpret_cast<const float4*>(query)[c * 64 + lane];
  float best = 0.0f;
  long long t0 = 0;
  if (g == 0) {
    if (threadIdx.x < kNb) ids[threadIdx.x] = mix(threadIdx.x * 2654435761u) % n;
    __syncthreads();
    t0 = wall_clock64();
  }
  for (uint32_t s = 0; s < steps; s++) {
    const unsigned long long tag = (unsigned long long)(s + 1) << 32;
    if (g == 0) {
      // publish the ids the helpers need (one wave, one granule per lane), then do the leader's own share
      if (w == 0 && lane >= per) __hip_atomic_store(&mail_ids[lane], tag | ids[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      // helper: poll its `per` granules (one lane each), stage the ids in LDS
      if (w == 0) {
        uint32_t id = 0;
        if (lane < per) {
          unsigned long long v;
          do {
            v = __hip_atomic_load(&mail_ids[g * per + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 32) != (s + 1)) __builtin_amdgcn_s_sleep(1);
          } while ((v >> 32) != (s + 1));
          id = (uint32_t)v;
          ids[g * per + lane] = id;
        }
      }
      __syncthreads();
    }
    // every group: its rows, a wave per row
    for (int r = w; r < per; r += 4) {
      const float d = row_dot(rows, ids[g * per + r], q, lane);
      if (lane == 0) {
        if (g == 0) dist[r] = d;
        else __hip_atomic_store(&mail_d[g * per + r], tag | __float_as_uint(d), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (g == 0) {
      __syncthreads();
      if (w == 0) {  // gather the helpers' distances (lane l polls granule l), then the same reduction as (A)
        float d;
        if (lane < per) {
          d = dist[lane];
        } else {
          unsigned long long v;
          do {
            v = __hip_atomic_load(&mail_d[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((v >> 32) != (s + 1));
          d = __uint_as_float((uint32_t)v);
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) d = fminf(d, __shfl_xor(d, sft, 64));
        best = d;
        ids[lane] = mix(__float_as_uint(d) + lane * 0x9E3779B9u + s) % n;
      }
      __syncthreads();
    }
  }
  if (g == 0 && threadIdx.x == 0) {
    *ticks = wall_clock64() - t0;
    *out = best;
  }
}

int main(int argc, char** argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 1000000u;
  const uint32_t steps = argc > 2 ? (uint32_t)atoi(argv[2]) : 2000u;
  float *rows, *query, *out;
  long long* ticks;
  unsigned long long *mail_ids, *mail_d;
  CK(hipMalloc(&rows, (size_t)n * kDim * 4));
  CK(hipMalloc(&query, kDim * 4));
  CK(hipMalloc(&out, 4));
  CK(hipMalloc(&ticks, 8));
  CK(hipMalloc(&mail_ids, kNb * 8));
  CK(hipMalloc(&mail_d, kNb * 8));
  {
    std::vector<float> h((size_t)n * kDim);
    uint32_t s = 12345;
    for (auto& x : h) {
      s = s * 1664525u + 1013904223u;
      x = (float)((s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
    }
    CK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(query, h.data() + 7 * kDim, kDim * 4, hipMemcpyHostToDevice));
  }
  int rate_khz = 0;
  CK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
  auto report = [&](const char* name, float ref) {
    long long t = 0;
    float o = 0;
    CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost));
    const double us = (double)t / (double)rate_khz * 1e3 / steps;
    printf("%-44s %8.3f us per step   (chain value %.6f%s)\n", name, us, o, (ref != 0.0f && o != ref) ? "  != variant A: CHAIN DIFFERS" : "");
    return o;
  };
  printf("corpus %u x %d f32 (%.1f MB), %u dependent steps of %d neighbour rows, wall clock %d kHz\n", n, kDim, (double)n * kDim * 4 / 1e6, steps, kNb, rate_khz);
  for (int rep = 0; rep < 2; rep++) {  // (rep 0 warms the clocks and, for a small corpus, the caches)
    hipLaunchKernelGGL(step_one_cu, dim3(1), dim3(1024), 0, 0, rows, n, query, steps, out, ticks);
    CK(hipDeviceSynchronize());
  }
  const float ref = report("(A) one CU, 1024 threads", 0.0f);
  for (int stride : {1, 8}) {
    for (int P : {2, 4, 8, 16}) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(mail_ids, 0, kNb * 8));
        CK(hipMemset(mail_d, 0, kNb * 8));
        hipLaunchKernelGGL(step_multi_cu, dim3(P * stride), dim3(256), 0, 0, rows, n, query, steps, P, stride, mail_ids, mail_d, out, ticks);
        CK(hipDeviceSynchronize());
      }
      char name[96];
      snprintf(name, sizeof name, "(B) %2d CUs, %s", P, stride == 1 ? "consecutive blocks (8 XCDs)" : "blocks 0, 8, 16 .. (one XCD)");
      report(name, ref);
    }
  }
  return 0;
}
