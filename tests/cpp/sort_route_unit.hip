// Host-side unit tests of the sort's building blocks (the counterpart of the reference's FOR_HOST_TEST suite,
// tests/test_embedding_ops.cu:121-374, for the pieces of THIS design that are pure integer logic): compiled by hipcc
// with --offload-host-only (no device code, no GPU, no HIP runtime call) and run in the CPU test suite.
//   * PlanPassFrom / RouteArray: whatever subset of radix passes the device skips, every working pass must read what the
//     previous working pass wrote, the first must read the caller's input, the last must write the caller's output,
//     and no pass may read and write the same buffer -- for wide and for narrow (32-bit-in-scratch) arrays;
//   * PlanPass::next: the next WORKING pass (whose histogram a chained scatter pass counts);
//   * ImplicitPayloadDivisor: (i * magic) >> shift == i / d over the whole int range the API allows;
//   * ScatterTileOfBlock: a bijection of [0, tiles) for every tile and XCD count;
//   * SortSegmentLength / RadixSortPlan: whole tiles, workspace regions in order, large enough, not overlapping.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "cuembed/include/radix_sort_kernels.hpp"

using namespace cuembed::detail;

static int g_fail = 0;
#define CHECK(cond, ...)                                \
  do {                                                  \
    if (!(cond)) {                                      \
      std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
      std::fprintf(stderr, __VA_ARGS__);                \
      std::fprintf(stderr, "\n");                       \
      ++g_fail;                                         \
    }                                                   \
  } while (0)

static void Routes() {
  for (int passes = 1; passes <= 8; ++passes) {
    for (unsigned varying_digits = 0; varying_digits < (1u << passes); ++varying_digits) {
      unsigned long long varying = 0;
      for (int q = 0; q < passes; ++q)
        if ((varying_digits >> q) & 1u) varying |= 0x5aull << (8 * q);
      for (int narrow = 0; narrow <= 1; ++narrow) {
        SortMode mode{};
        mode.use_varying = 1;
        mode.narrow_keys = narrow ? kNarrowAlways : kNarrowNever;
        int holds = kBufIn;     // where the array lives right now
        int working = 0;
        for (int p = 0; p < passes; ++p) {
          const PassPlan plan = PlanPassFrom(true, varying, 0ull, 0ull, p, passes, mode);
          const bool expect_active = p == 0 || ((varying >> (8 * p)) & 0xff) != 0;
          CHECK(plan.active == expect_active, "passes %d varying %x pass %d active", passes, varying_digits, p);
          if (!plan.active) continue;
          ++working;
          const ArrayRoute r = RouteArray(plan, narrow != 0);
          CHECK(r.src == holds, "passes %d varying %x narrow %d pass %d reads %d, the array is in %d", passes, varying_digits,
                narrow, p, r.src, holds);
          CHECK(r.src != r.dst, "pass %d reads and writes buffer %d", p, r.dst);
          CHECK(r.dst != kBufIn, "pass %d writes the caller's input", p);
          if (!narrow) CHECK(r.dst == kBufOut || r.dst == kBufTmp0, "wide arrays use out and one scratch buffer");
          int next = -1;
          for (int q = p + 1; q < passes; ++q)
            if ((varying >> (8 * q)) & 0xff) { next = q; break; }
          CHECK(plan.next == next, "passes %d varying %x pass %d next %d want %d", passes, varying_digits, p, plan.next, next);
          CHECK(plan.first == (p == 0), "first");
          holds = r.dst;
        }
        CHECK(holds == kBufOut, "passes %d varying %x narrow %d: the result ends in buffer %d", passes, varying_digits, narrow, holds);
        CHECK(working >= 1, "pass 0 always runs");
      }
    }
    // nothing known on the device: every pass runs, same invariants
    SortMode fixed{};
    int holds = kBufIn;
    for (int p = 0; p < passes; ++p) {
      const PassPlan plan = PlanPassFrom(false, 0ull, 0ull, 0ull, p, passes, fixed);
      const ArrayRoute r = RouteArray(plan, false);
      CHECK(plan.active && r.src == holds && r.src != r.dst, "fixed route, passes %d pass %d", passes, p);
      holds = r.dst;
    }
    CHECK(holds == kBufOut, "fixed route ends in the caller's output");
  }
}

static void NarrowDecisions() {
  SortMode mode{};
  mode.use_varying = 1;
  mode.narrow_keys = kNarrowIfConstantHigh;
  mode.narrow_v1 = kNarrowIfConstantHigh;
  const unsigned long long all = 0x0000000700000000ull;   // AND of all keys: a constant high half of 7
  PassPlan a = PlanPassFrom(true, 0x00ffffffull, all, 0x7fffffffull, 1, 8, mode);
  CHECK(a.narrow_keys && a.key_high == 0x0000000700000000ull && a.narrow_v1, "constant high halves travel narrow");
  PassPlan b = PlanPassFrom(true, 0x100ffffffull, all, 0x1ffffffffull, 1, 8, mode);
  CHECK(!b.narrow_keys && b.key_high == 0 && !b.narrow_v1, "varying high halves travel wide");
}

static void Quotients() {
  std::mt19937_64 rng(3);
  for (int d : {1, 2, 3, 5, 7, 8, 16, 63, 64, 65, 100, 127, 128, 129, 1000, 4096, 65535, 65536, 1000003, 1 << 30, INT32_MAX}) {
    SortMode m{};
    ImplicitPayloadDivisor(d, &m);
    for (int64_t i : {int64_t{0}, int64_t{1}, int64_t{d} - 1, int64_t{d}, int64_t{d} + 1, int64_t{2} * d - 1, int64_t{2} * d,
                      int64_t{INT32_MAX} - 1, int64_t{INT32_MAX}})
      if (i >= 0 && i <= INT32_MAX) CHECK(ImplicitPayload(m, i) == static_cast<unsigned>(i / d), "d %d i %lld", d, (long long)i);
    for (int t = 0; t < 20000; ++t) {
      const int64_t i = static_cast<int64_t>(rng() % (uint64_t{1} << 31));
      CHECK(ImplicitPayload(m, i) == static_cast<unsigned>(i / d), "d %d i %lld", d, (long long)i);
    }
  }
}

static void TileMaps() {
  for (int xcds : {1, 2, 4, 8})
    for (int tiles : {1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 255, 256, 1000, 1024, 1031}) {
      std::vector<int> seen(tiles, 0);
      for (int b = 0; b < tiles; ++b) {
        const int t = ScatterTileOfBlock(b, tiles, xcds);
        CHECK(t >= 0 && t < tiles, "tile %d of %d", t, tiles);
        if (t >= 0 && t < tiles) ++seen[t];
      }
      for (int t = 0; t < tiles; ++t) CHECK(seen[t] == 1, "xcds %d tiles %d: tile %d is taken %d times", xcds, tiles, t, seen[t]);
    }
}

static void Plans() {
  for (size_t n : {size_t{1}, size_t{4096}, size_t{4097}, size_t{16384}, size_t{16385}, size_t{229376}, size_t{229377}, size_t{262145},
                   size_t{1} << 22, size_t{1} << 27}) {
    const RadixSortPlan<uint64_t, int64_t, float> plan(n, 64);
    CHECK(plan.passes == 8, "64 key bits are 8 passes");
    CHECK(plan.keys_tmp < plan.v1_tmp && plan.v1_tmp < plan.v2_tmp && plan.v2_tmp < plan.tile_hist &&
              plan.tile_hist < plan.bin_total && plan.bin_total < plan.tile_bits && plan.tile_bits < plan.varying &&
              plan.varying < plan.total, "regions in order at n = %zu", n);
    CHECK(plan.v1_tmp - plan.keys_tmp >= n * 8 && plan.v2_tmp - plan.v1_tmp >= n * 8 && plan.tile_hist - plan.v2_tmp >= n * 4,
          "scratch arrays hold n elements at n = %zu", n);
    if (plan.chained_tiles > 0)
      CHECK(plan.bin_total - plan.tile_hist >= static_cast<size_t>(plan.passes) * plan.chained_tiles * 256 * 4 &&
                static_cast<size_t>(plan.chained_tiles) * 1024 >= n, "chained counters at n = %zu", n);
    CHECK((n <= kChainedSortMax) == (plan.chained_tiles > 0), "chained range at n = %zu", n);
    CHECK(plan.total >= n * (8 + 8 + 4) && plan.total <= n * (8 + 8 + 4) + n / 2 + (size_t{8} << 20), "workspace is the scratch arrays + the per-tile counters + a few MB");
    CHECK(plan.total >= RunHeadScanWorkBytes(n), "the run-head scan shares the sort's workspace");
    for (int blocks : {1, 2, 3, 8, 64, 100}) {
      const size_t len = SortSegmentLength(n, blocks);
      CHECK(len % kSortTile == 0 && len > 0, "segments are whole tiles");
      const size_t segments = (n + len - 1) / len;
      CHECK(segments <= static_cast<size_t>(kMaxSortSegments) && (n > size_t{kFoldScanTiles} * kSortTile || segments == 1),
            "n %zu blocks %d: %zu segments", n, blocks, segments);
    }
  }
}

int main() {
  Routes();
  NarrowDecisions();
  Quotients();
  TileMaps();
  Plans();
  if (g_fail) {
    std::fprintf(stderr, "%d checks failed\n", g_fail);
    return 1;
  }
  std::printf("sort building blocks: all host-side checks passed\n");
  return 0;
}
