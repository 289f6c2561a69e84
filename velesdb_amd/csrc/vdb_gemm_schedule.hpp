// vdb_gemm_schedule.hpp — how a corpus is cut into launches of the 256 x 256 selection kernel (sweep_gemm_bf16.hip): host arithmetic
// only, nothing of HIP in it, so that tests/gemm_schedule_model.cpp can check on the CPU — for corpus sizes no GPU test reaches
// (BASELINE configs[3] / configs[4]: 10 M rows, 6.25 M rows per shard, and up to the 32-bit row limit) — that the launches of a
// schedule cover every row tile exactly once, that every partial list has its own slot, and that no intermediate value wraps.
// A schedule that skips a tile is a silently wrong top-k.
#pragma once
#include <stdint.h>

#include <algorithm>

namespace vdb {

constexpr uint32_t kGemmTileRows = 256, kGemmTileQueries = 256;  // the block tile of the kernel (kG16BM x kG16BN)

// one launch: blocks = G row groups x nqt query tiles; block (g, qt) takes the row tiles row_lo / 256 + g, + G, ... below
// ceil(row_hi / 256) for the queries qt * qper ... of the batch
struct Bf16GemmPlan {
  uint32_t nqt, qper, G;
  uint32_t row_lo, row_hi;  // row range of the launch (row_lo a multiple of 256)
  int blocks;
};

constexpr int kGemmMaxLaunches = 64;
struct GemmSchedule {
  Bf16GemmPlan bp[kGemmMaxLaunches];
  int n_launch = 0;
  uint32_t lists = 0;  // row groups (= partial lists per query) over all launches
};

inline void sweep_gemm_bf16_plan(uint32_t nq, uint32_t row_lo, uint32_t row_hi, int n_cus, Bf16GemmPlan* p) {
  p->nqt = (nq + kGemmTileQueries - 1) / kGemmTileQueries;
  p->qper = (nq + p->nqt - 1) / p->nqt;
  p->row_lo = row_lo;
  p->row_hi = row_hi;
  const uint32_t ntiles = (uint32_t)(((uint64_t)row_hi - row_lo + kGemmTileRows - 1) / kGemmTileRows);  // (64-bit: a range within one tile of 2^32 rows)
  // row groups: whole XCD rounds, never more blocks than the chip holds at once (one block per CU)
  uint32_t G = (uint32_t)std::max(8, n_cus / (int)p->nqt / 8 * 8);
  G = std::min(G, (ntiles + 7) / 8 * 8);
  p->G = G;
  p->blocks = (int)(G * p->nqt);
}

// rows [row_first, n): `head_tiles[i]` (x the row groups the chip holds at once) 256-row tiles per row group for the first launches
// (0 = none; a step is taken only while at least as much again is left), then launches of <= max_launch_rows rows (0 = one launch
// for the rest).  A launch boundary costs one merge + re-seed (~20 us); launches longer than ~2 M rows let the query tiles of a row
// group drift apart in L2 (10 M rows in one launch: 2.1 x the corpus from HBM).
inline void gemm_schedule(uint32_t nq, uint32_t row_first, uint32_t n, int n_cus, const uint32_t head_tiles[3], uint32_t max_launch_rows, GemmSchedule* s) {
  s->n_launch = 0;
  s->lists = 0;
  max_launch_rows -= max_launch_rows % kGemmTileRows;  // a launch starts on a row tile (less than one tile = no limit)
  const uint32_t G2 = (uint32_t)std::max(8, n_cus / (int)((nq + kGemmTileQueries - 1) / kGemmTileQueries) / 8 * 8);  // row groups the chip holds at once
  uint32_t lo = row_first, left = (uint32_t)(((uint64_t)n - row_first + kGemmTileRows - 1) / kGemmTileRows);
  auto push = [&](uint32_t hi) {
    sweep_gemm_bf16_plan(nq, lo, hi, n_cus, &s->bp[s->n_launch]);
    s->lists += s->bp[s->n_launch].G;
    s->n_launch++;
    lo = hi;
  };
  int ns = 0;
  while (ns < 3 && head_tiles[ns]) ns++;
  for (int j = 0; j < ns && left >= 2 * head_tiles[j] * G2; j++) {  // (the rest must be worth at least as much again)
    uint32_t t = head_tiles[j] * G2;
    // the launch behind this one is the last of the head: whole row tiles per row group for the rest
    if (j == ns - 1 || left < 2 * head_tiles[j + 1] * G2) t += (left - t) % G2;
    push(lo + t * (uint32_t)kGemmTileRows);
    left -= t;
  }
  while (lo < n) {
    uint32_t hi = n;
    if (max_launch_rows && (uint64_t)lo + max_launch_rows < n) {
      hi = lo + max_launch_rows;
      if (n - hi < max_launch_rows / 4 || s->n_launch == kGemmMaxLaunches - 1) hi = n;  // (a short tail joins the launch in front of it)
    }
    push(hi);
  }
}

}  // namespace vdb
