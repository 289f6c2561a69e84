// gemm_schedule_model.cpp — the launch schedule of the 256 x 256 selection kernel (velesdb_amd/csrc/vdb_gemm_schedule.hpp, the
// text the library compiles) checked on the CPU over corpus sizes no GPU test reaches: BASELINE configs[3] (10 M rows),
// configs[4] (50 M rows = 6.25 M per shard on 8 GPUs), and on to the 32-bit row limit.  Test infrastructure
// (tests/test_gemm_schedule_cpu.py); no GPU, no HIP.
//
// What a schedule must guarantee — a violation is a silently wrong top-k (HnswIndex::search_brute_force must look at every
// vector, index/hnsw/index/search.rs:176-219):
//   * the launches' row ranges are consecutive, start at row_first, end at n, and every boundary between two launches is a
//     multiple of the tile height (the kernel derives its first row tile as row_lo / 256);
//   * inside a launch the kernel's block map (restated below from sweep_gemm_bf16.hip: xcd = bid & 7, slot = bid >> 3,
//     qt = slot % nqt, g = (slot / nqt) * 8 + xcd; block (g, qt) walks the row tiles row_lo / 256 + g, + G, ...) reaches every
//     (row tile, query tile) pair exactly once: G a positive multiple of 8, blocks = G * nqt;
//   * every (launch, row group) has its own partial-list slot below `lists`;
//   * the query tiles cover the batch: nqt * qper >= nq, qper <= 256, no empty query tile;
//   * at most kGemmMaxLaunches launches, and no 32-bit intermediate wraps (checked against 64-bit arithmetic).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "vdb_gemm_schedule.hpp"

using namespace vdb;

static int g_fail = 0;
static unsigned long long g_cases = 0, g_launches = 0, g_brute = 0;
#define CHECK(cond, ...)                                                         \
  do {                                                                           \
    if (!(cond)) {                                                               \
      if (g_fail < 20) {                                                         \
        std::fprintf(stderr, "CHECK failed line %d: %s: ", __LINE__, #cond);     \
        std::fprintf(stderr, __VA_ARGS__);                                       \
        std::fprintf(stderr, "\n");                                              \
      }                                                                          \
      g_fail++;                                                                  \
    }                                                                            \
  } while (0)

static void check_case(uint32_t nq, uint32_t row_first, uint32_t n, int n_cus, const uint32_t head[3], uint32_t max_rows) {
  GemmSchedule s;
  gemm_schedule(nq, row_first, n, n_cus, head, max_rows, &s);
  g_cases++;
  const char* fmt = "nq %u row_first %u n %u cus %d head %u/%u/%u max_rows %u";
#define CASE fmt, nq, row_first, n, n_cus, head[0], head[1], head[2], max_rows
  CHECK(s.n_launch >= 1 && s.n_launch <= kGemmMaxLaunches, CASE);
  if (s.n_launch < 1 || s.n_launch > kGemmMaxLaunches) return;
  uint64_t lo = row_first, lists = 0;
  const uint32_t nqt = (nq + 255) / 256;
  for (int j = 0; j < s.n_launch; j++) {
    const Bf16GemmPlan& p = s.bp[j];
    g_launches++;
    CHECK(p.row_lo == lo, CASE);
    CHECK(p.row_lo % kGemmTileRows == 0, CASE);
    CHECK(p.row_hi > p.row_lo && p.row_hi <= n, CASE);
    CHECK(j + 1 == s.n_launch ? p.row_hi == n : p.row_hi % kGemmTileRows == 0, CASE);
    CHECK(p.nqt == nqt && p.qper >= 1 && p.qper <= kGemmTileQueries && (uint64_t)p.nqt * p.qper >= nq, CASE);
    CHECK((uint64_t)(p.nqt - 1) * p.qper < nq, CASE);  // the last query tile is not empty
    CHECK(p.G >= 8 && p.G % 8 == 0, CASE);
    CHECK(p.blocks > 0 && (uint64_t)p.blocks == (uint64_t)p.G * p.nqt, CASE);
    const uint64_t tiles = ((uint64_t)p.row_hi - p.row_lo + 255) / 256;
    CHECK(p.G <= (tiles + 7) / 8 * 8, CASE);  // no row group without a tile beyond the rounding to XCD rounds
    if (n_cus / (int)nqt >= 8) CHECK(p.blocks <= n_cus, CASE);  // one block per CU: never more than the chip holds at once
    // the kernel's block map is a bijection onto (query tile, row group)
    std::vector<uint8_t> seen((size_t)p.G * p.nqt, 0);
    for (uint32_t bid = 0; bid < (uint32_t)p.blocks; bid++) {
      const uint32_t xcd = bid & 7u, slot = bid >> 3, qt = slot % p.nqt, g = (slot / p.nqt) * 8u + xcd;
      CHECK(g < p.G && qt < p.nqt, CASE);
      if (g < p.G && qt < p.nqt) {
        CHECK(!seen[(size_t)g * p.nqt + qt], CASE);
        seen[(size_t)g * p.nqt + qt] = 1;
      }
    }
    // small launches: walk the tiles as the kernel does and count
    if (tiles <= 4200 && (g_cases % 3 == 0 || tiles <= 64)) {
      g_brute++;
      const uint32_t t0 = p.row_lo / 256, t1 = (uint32_t)(((uint64_t)p.row_hi + 255) / 256);
      std::vector<uint8_t> cover(t1 - t0, 0);
      for (uint32_t g = 0; g < p.G; g++)
        for (uint64_t rt = (uint64_t)t0 + g; rt < t1; rt += p.G) cover[rt - t0]++;
      for (uint8_t c : cover) CHECK(c == 1, CASE);
    }
    lists += p.G;
    lo = p.row_hi;
  }
  CHECK(lo == n, CASE);
  CHECK(lists == s.lists, CASE);
#undef CASE
}

int main() {
  const uint32_t nqs[] = {1, 64, 256, 257, 1000, 1024, 4096};
  const int cus[] = {256, 304, 64, 1};
  std::vector<uint32_t> ns = {1,         255,        256,        257,        65535,      65536,      65537,       100000,
                              1000000,   1000001,    6250000,    10000000,   16777216,   50000000,   2147483648u, 4294966784u,
                              4294967295u};
  uint64_t st = 88172645463325252ull;
  auto rnd = [&] {
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    return st;
  };
  for (int i = 0; i < 16; i++) ns.push_back((uint32_t)(rnd() % (i < 8 ? 3000000u : 4294967295u)) + 1u);
  const uint32_t heads[][3] = {{0, 0, 0}, {1, 4, 16}, {1, 0, 0}, {16, 0, 0}, {1, 2, 3}, {16, 64, 0}, {1000, 0, 0}, {65535, 65535, 65535}};
  const uint32_t max_rows[] = {0, 1u << 21, 1u << 18, 256, 1000, 100, 4294967040u};
  for (uint32_t n : ns)
    for (uint32_t nq : nqs)
      for (int c : cus)
        for (const auto& h : heads)
          for (uint32_t mr : max_rows) {
            const uint32_t firsts[] = {0u, 4096u, 16384u, (uint32_t)((rnd() % n) / 256 * 256)};
            for (uint32_t f : firsts)
              if (f < n) check_case(nq, f, n, c, h, mr);
          }
  // the selection stage's own head (select_stage.hip: one head launch of max(2^18, n / 16) rows, in units of the chip's row groups)
  for (uint32_t n : ns)
    for (uint32_t nq : nqs) {
      if (n < 65536) continue;
      const uint32_t G2 = (uint32_t)std::max(8, 256 / (int)((nq + 255) / 256) / 8 * 8);
      const uint32_t h[3] = {(uint32_t)((std::max<uint64_t>(1u << 18, n / 16) + (uint64_t)G2 * 256 - 1) / ((uint64_t)G2 * 256)), 0u, 0u};
      check_case(nq, 4096, n, 256, h, 1u << 21);
      check_case(nq, 0, n, 256, h, 1u << 21);
    }
  std::printf("{\"cases\": %llu, \"launches\": %llu, \"launches_walked_tile_by_tile\": %llu, \"violations\": %d, \"ok\": %s}\n", g_cases, g_launches,
              g_brute, g_fail, g_fail ? "false" : "true");
  return g_fail ? 1 : 0;
}
