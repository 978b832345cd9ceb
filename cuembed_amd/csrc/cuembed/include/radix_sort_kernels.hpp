// MI355X (gfx950 / CDNA4) stable LSD radix sort of (key, value[, value2]) and the run-head
// scan, hand-written for the index transposition of the embedding backward pass.
//
// The reference calls cub::DeviceRadixSort / DeviceScan here (index_transforms.cuh:108-136,
// :160-199, :287-322).  This is a purpose-built replacement, shaped by what Transpose needs:
//   * 8-bit digits; only ceil(index_bits / 8) passes -- the caller may bound the key range;
//   * up to TWO payload arrays move with the key (sample id and, for weighted lookups, the
//     weight), so no (id, weight) struct has to be packed before and unpacked after the sort;
//   * no per-call state to zero: nothing but kernels is enqueued (a library sort issues several
//     small memsets per call, each a 5 us launch at this problem size);
//   * workspace size is pure host arithmetic (no device query).
//   * passes in which every key has the same digit are skipped on the device, and 64-bit
//     arrays travel as 32 bits between passes when their high halves carry nothing (see PlanPass).
// One pass = three launches (two when there are at most kFoldScanTiles tiles; the whole sort is
// one launch, SingleTileSortKernel, up to one tile):
//   RadixTileHistogramKernel   per 4096-key tile: 256-bin histogram of the current digit
//   RadixScanTilesKernel       per bin: exclusive prefix over the tiles, and the bin total
//   RadixScatterKernel         per tile: stable rank of every key among equal digits, scatter
// The caller may ask for BLOCKS of the input to be sorted on their own (RadixSortPairs `blocks`; Transpose's
// sample_blocks): same launches, the scan runs per (block, bin) and the scatter adds its block's start.
// Histogram and scatter workgroups take their tile from ScatterTileOfBlock(), which keeps runs of neighbouring
// tiles on one XCD: the 4-byte histogram words and the 64-byte runs of neighbouring tiles share 128-byte lines,
// and the XCDs' L2s are not coherent -- what is written from one L2 leaves as one line, not as eight pieces.
// (A single-kernel-per-pass variant with decoupled look-back was built and measured 30 % SLOWER
// here: the per-tile status words have to bypass the XCDs' non-coherent L2s, so every look-back
// hop is a ~1 us memory-side access, and with 1024 tiles starting together the walks are long.)
// Ranking is wave-synchronous: a 64-lane wavefront handles 64 consecutive keys per round, finds
// the lanes holding the same digit with 8 ballots (`match-any`), takes its rank from the
// popcount of the lower peers, and the lowest peer bumps the wave's digit counter in LDS -- no
// LDS atomics, no dependence on digit skew, and input order is preserved by construction.
// What bounds the scatter pass (21 us for 4.19M pairs; 23 before the tile map): a build that stores every tile back
// contiguously takes 21 us, so it is the ~110 VALU instructions per key of the ranking and the
// dependent LDS steps, not the scattered stores.  Round 3 built two rankings WITHOUT ballots (lane-private byte
// counters: one wavefront per tile, and a blocked 256-thread tile ranked in two 4-bit steps); both were exact and
// both were slower (27 vs 23 us per pass, 0.123 vs 0.107 ms per transpose): docs/EXPERIMENTS.md.  Measured and rejected earlier: peers through per-wave
// lane bitmaps in LDS (ds_or, read back, clear: -50 VALU per round, 1-5 % slower); a one-compare
// shortcut for wavefronts whose 64 keys share the digit (2 % slower); 2048-key tiles (10 % slower).
#ifndef CUEMBED_INCLUDE_RADIX_SORT_KERNELS_HPP_
#define CUEMBED_INCLUDE_RADIX_SORT_KERNELS_HPP_

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "cuembed/include/block_sort_kernels.hpp"
#include "cuembed/include/blocked_order.hpp"
#include "cuembed/include/device_shape.hpp"
#include "cuembed/include/sort_common.hpp"

namespace cuembed {
namespace detail {

// Passes that cannot move anything are skipped ON THE DEVICE.  The first pass also reduces the
// keys to `varying` = the bits in which any two keys differ (OR of all keys & ~AND of all keys);
// a later pass whose digit of `varying` is zero would be the identity permutation, so its three
// kernels return at once (an empty launch is ~2 us instead of ~35 us of work), and the remaining
// passes pick their source and destination buffers so that the last one that runs still writes
// the caller's output.  The host enqueues the same launches whatever the data (no read-back, graph
// capturable): int64 indices below 2^24 cost 3 working passes + 5 empty ones, not 8.
// Reading `varying` first costs every kernel ~1 us, so sorts of at most kStaticRoutePasses passes
// (a caller-supplied key bound, where a constant digit is unlikely) pass state == nullptr and
// take the fixed route.
//
// 64-bit arrays travel NARROW between passes when their high halves carry nothing.  An int64
// first payload (sample ids: < nnz <= INT_MAX) is kept as 32 bits in the scratch buffers when
// every value is in [0, 2^32) -- known from the caller's bound, or found out on the device: pass 0
// ORs all payload values next to the key bits (a negative or >= 2^32 value sets a high bit and the
// array travels wide, like cub::DeviceRadixSort::SortPairs carries it in the reference,
// index_transforms.cuh:108-136).  int64 keys are kept as their low 32 bits when their high halves
// do not vary (known from the caller's bound, or from `varying` on the device -- the constant high
// half is put back on the way out).  A scratch buffer of n 64-bit elements then holds TWO 32-bit
// arrays, so all intermediate passes ping-pong inside the scratch and only the last one touches
// the caller's output: a middle pass moves 8 bytes per pair and direction instead of 16.
//
// Keys are SIGNED integers in the API (IndexT): sorting all 8 * sizeof(IndexT) bits orders them as
// signed numbers, as cub does for signed key types -- the digit that holds the sign bit is XOR-ed
// with 0x80 (`sign_pass`).  A caller-supplied bound (index_bits < all bits) promises keys in
// [0, 2^index_bits); no digit is flipped then.
constexpr int kStaticRoutePasses = 3;
constexpr int kSelfSumTiles = 4096; // up to 16.7M keys every run-head scan workgroup sums the earlier tiles itself
enum SortBuffer : int { kBufIn = 0, kBufOut = 1, kBufTmp0 = 2, kBufTmp1 = 3 };
enum NarrowKeys : int { kNarrowNever = 0, kNarrowAlways = 1, kNarrowIfConstantHigh = 2 };
enum : int { kStateVarying = 0, kStateAllBits = 1, kStatePayloadBits = 2, kStateWords = 3 };

//! Counters of the high-word kernels' ticket queue (SortWorkQueue): the next ticket and the number of finished items,
//! 256 bytes apart (an atomic costs ~10 ns on ONE address, tools/ticket_probe.hip).  They sit behind the three state
//! words and are zeroed by whoever writes those.
constexpr int kBarrierGroups = 1;
constexpr int kBarrierStride = 64;     // in counters (unsigned)
constexpr int kBarrierWords = (kBarrierGroups + 1) * kBarrierStride / 2;   // in state words (unsigned long long)
__device__ __forceinline__ unsigned* SortBarrierCounters(unsigned long long* state) {
  return reinterpret_cast<unsigned*>(state + kStateWords + 1);
}
__device__ __forceinline__ void ZeroSortBarrier(unsigned long long* state) {
  unsigned* c = SortBarrierCounters(state);
#pragma unroll
  for (int g = 0; g <= kBarrierGroups; ++g) c[g * kBarrierStride] = 0u;
}


//! What the device knows about the arrays after pass 0: state[kStateVarying] = bits in which keys
//! differ, state[kStateAllBits] = AND of all keys, state[kStatePayloadBits] = OR of all 64-bit first
//! payloads.  state == nullptr: nothing is decided on the device (fixed route, all passes run).
struct PassPlan {
  bool active;        //!< this pass moves data
  bool first;         //!< reads the caller's input
  int later;          //!< working passes after this one
  int next;           //!< the next working pass after this one, or -1
  bool narrow_keys;   //!< 64-bit keys are stored as 32 bits in the scratch buffers
  bool narrow_v1;     //!< a 64-bit first payload is stored as 32 bits in the scratch buffers
  unsigned long long key_high;  //!< constant high half to put back (narrow_keys only)
};

//! ... from the three words themselves (`have_state` false: nothing is decided on the device).  Host-callable: the
//! routing rules are unit-tested on the CPU (tests/cpp/sort_route_unit.hip).
__host__ __device__ __forceinline__ PassPlan PlanPassFrom(const bool have_state, const unsigned long long state_varying,
                                                 const unsigned long long state_all, const unsigned long long state_payload,
                                                 const int pass, const int passes, const SortMode& mode) {
  const bool know_keys = have_state && mode.use_varying;
  const unsigned long long varying = know_keys ? state_varying : ~0ull;
  unsigned active = 1u;  // pass 0 always runs
  for (int q = 1; q < passes; ++q)
    if ((varying >> (8 * q)) & 0xffull) active |= 1u << q;
  PassPlan plan;
  plan.active = (active >> pass) & 1u;
  plan.first = pass == 0;
  plan.later = __builtin_popcount(active >> (pass + 1));
  plan.next = (active >> (pass + 1)) != 0 ? pass + __builtin_ffs(static_cast<int>(active >> (pass + 1))) : -1;
  plan.narrow_keys = mode.narrow_keys == kNarrowAlways ||
                     (mode.narrow_keys == kNarrowIfConstantHigh && know_keys && (varying >> 32) == 0);
  plan.key_high = (mode.narrow_keys == kNarrowIfConstantHigh && plan.narrow_keys)
                      ? (state_all & 0xffffffff00000000ull) : 0ull;
  plan.narrow_v1 = mode.narrow_v1 == kNarrowAlways ||
                   (mode.narrow_v1 == kNarrowIfConstantHigh && have_state && (state_payload >> 32) == 0);
  return plan;
}

__device__ __forceinline__ PassPlan PlanPass(const unsigned long long* __restrict__ state, const int pass,
                                             const int passes, const SortMode& mode) {
  if (state == nullptr) return PlanPassFrom(false, 0ull, 0ull, 0ull, pass, passes, mode);
  return PlanPassFrom(true, state[kStateVarying], state[kStateAllBits], state[kStatePayloadBits], pass, passes, mode);
}

//! Buffers of one array.  Wide: out / scratch alternate backwards from the last working pass.
//! Narrow: the two halves of the scratch alternate, the last working pass writes `out`.
struct ArrayRoute {
  int src, dst;
};
__host__ __device__ __forceinline__ ArrayRoute RouteArray(const PassPlan& plan, const bool narrow) {
  ArrayRoute r;
  if (narrow) {
    r.dst = plan.later == 0 ? kBufOut : (plan.later & 1 ? kBufTmp1 : kBufTmp0);
    r.src = plan.first ? kBufIn : ((plan.later + 1) & 1 ? kBufTmp1 : kBufTmp0);
  } else {
    r.dst = (plan.later % 2 == 0) ? kBufOut : kBufTmp0;
    r.src = plan.first ? kBufIn : (r.dst == kBufOut ? kBufTmp0 : kBufOut);
  }
  return r;
}

//! One array of the sort in its three places; `tmp` holds n elements of T, or two arrays of n
//! 32-bit elements when the array travels narrow.
template <typename T>
struct SortArray {
  const T* in;
  T* out;
  T* tmp;
};

//! Reads ITEMS elements per lane (positions pos0 + r * stride, r = 0..) from wherever the
//! route says.  `high` is OR-ed onto narrow elements.
template <typename T, int ITEMS>
__device__ __forceinline__ void LoadRouted(const SortArray<T>& a, const int where, const bool narrow,
                                           const int64_t n, const int64_t pos0, const int stride, const T high,
                                           T (&item)[ITEMS]) {
  if constexpr (sizeof(T) == 8) {
    if (narrow && where >= kBufTmp0) {
      const unsigned* p = reinterpret_cast<const unsigned*>(a.tmp) + (where == kBufTmp1 ? n : 0);
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) {
        const int64_t i = pos0 + static_cast<int64_t>(r) * stride;
        item[r] = i < n ? static_cast<T>(static_cast<T>(p[i]) | high) : T(0);
      }
      return;
    }
  }
  const T* p = where == kBufIn ? a.in : (where == kBufOut ? a.out : a.tmp);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int64_t i = pos0 + static_cast<int64_t>(r) * stride;
    if constexpr (std::is_empty<T>::value) {
      (void)i;
    } else {
      item[r] = i < n ? p[i] : T();
    }
  }
}

//! Tile of workgroup b in the histogram and scatter passes.  Workgroups are dealt round-robin to the XCDs (b % xcds; 8 on a
//! full MI355X, from the device: device_shape.hpp), so with
//! tile = b neighbouring tiles -- whose 64-byte runs of a bin are neighbours in the output -- land on different,
//! non-coherent L2s and each writes its half of a 128-byte line on its own.  This map keeps runs of tiles on one
//! XCD (tile = (b % 8) * ceil(tiles / 8) + b / 8 while that is a tile, the identity for the ragged rest), so that
//! halves written a few workgroups apart meet in that L2 before the line leaves it (the histogram pass writes ONE
//! 4-byte word per bin and tile: 32 neighbouring tiles share a line).  Any bijection is correct.
__host__ __device__ __forceinline__ int ScatterTileOfBlock(const int b, const int num_tiles, const int xcds) {
  const int per_xcd = num_tiles / xcds;           // tiles of the rectangular part, per XCD
  if (b >= per_xcd * xcds) return b;              // ragged rest (fewer than `xcds` tiles)
  return (b % xcds) * per_xcd + (b / xcds);
}

//! tile_hist[bin * num_tiles + tile] = number of keys of the tile whose digit is `bin` (kChained: [tile][bin], and
//! the same array of every LATER pass is zeroed -- the scatter passes fill those themselves, see RadixScatterKernel).
//! A tile is kSortThreads * ITEMS keys.
//! Plain LDS atomics: order does not matter for counting.  In pass 0 the tile's OR and AND of
//! its keys, and the OR of its 64-bit first payloads (`payload64`, or nullptr), go to
//! tile_bits[kStateWords * tile + ...].
template <typename KeyT, int ITEMS = kSortItems, bool kChained = false>
__device__ __forceinline__ void
RadixTileHistogramBody(const SortArray<KeyT>& keys, const int64_t n, const int pass, const int passes,
                       const SortMode& mode, unsigned* __restrict__ tile_hist, const int num_tiles,
                       unsigned long long* __restrict__ tile_bits,
                       const unsigned long long* __restrict__ state,
                       const unsigned long long* __restrict__ payload64, const int xcds, const int block) {
  __shared__ unsigned count[kSortWaves][kSortBins];  // one sub-histogram per wave: 4x less contention
  __shared__ unsigned long long wave_bits[kSortWaves][kStateWords];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int shift = 8 * pass;
  const unsigned flip = SignFlip(mode, pass);
  const int tile = ScatterTileOfBlock(block, num_tiles, xcds);
  const int64_t base = static_cast<int64_t>(tile) * (kSortThreads * ITEMS) + tid;
  KeyT key[ITEMS];
  int where = kBufIn;
  bool narrow = false;
  KeyT high = 0;
  if (pass > 0) {
    const PassPlan plan = PlanPass(state, pass, passes, mode);
    if (!plan.active) return;
    narrow = plan.narrow_keys;
    high = static_cast<KeyT>(plan.key_high);
    where = RouteArray(plan, narrow).src;
  }
  LoadRouted<KeyT>(keys, where, narrow, n, base, kSortThreads, high, key);  // all loads in flight first
  // (requesting 32-bit keys from both possible sources before the state words arrive, as the chained scatter does, was
  // measured here and in the tiled scatter: nothing at 0.5 M keys, slower from 2 M up -- twice the key reads)
#pragma unroll
  for (int w = 0; w < kSortWaves; ++w) count[w][tid] = 0;
  __syncthreads();
  const bool reduce_bits = pass == 0 && tile_bits != nullptr;
  if (reduce_bits) {
    unsigned long long any = 0ull, all = ~0ull, pay = 0ull;
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      const int64_t i = base + static_cast<int64_t>(r) * kSortThreads;
      if (i < n) {
        any |= static_cast<unsigned long long>(key[r]);
        all &= static_cast<unsigned long long>(key[r]);
        if (payload64 != nullptr) pay |= payload64[i];
      }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      any |= __shfl_xor(any, d);
      all &= __shfl_xor(all, d);
      pay |= __shfl_xor(pay, d);
    }
    if ((tid & 63) == 0) {
      wave_bits[wave][0] = any;
      wave_bits[wave][1] = all;
      wave_bits[wave][2] = pay;
    }
  }
  // From the second pass on, equal keys sit next to each other, and a power-law batch has runs
  // of thousands of them: 64 lanes adding to ONE counter serialise.  So only the first lane of
  // every run of equal digits inside the wavefront adds, and it adds the run's length (the
  // distance to the next run head in the ballot of heads).  Random digits: every lane is a head.
  const int lane = tid & 63;
  if (pass == 0) {
    // the lowest digit of unsorted keys: equal neighbours are rare, one LDS atomic per key is the cheaper way
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      if (base + static_cast<int64_t>(r) * kSortThreads < n)
        atomicAdd(&count[wave][static_cast<unsigned>(key[r] & 0xff) ^ flip], 1u);
    }
  } else {
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      const bool in_range = base + static_cast<int64_t>(r) * kSortThreads < n;
      const unsigned digit = in_range ? (static_cast<unsigned>((key[r] >> shift) & 0xff) ^ flip) : 0xffffffffu;
      const unsigned before = __shfl_up(digit, 1);
      const bool head = lane == 0 || before != digit;
      const unsigned long long heads = __ballot(head);
      const unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);
      const unsigned run = above != 0 ? static_cast<unsigned>(__ffsll(static_cast<long long>(above)))
                                      : static_cast<unsigned>(64 - lane);
      if (head && in_range) atomicAdd(&count[wave][digit], run);
    }
  }
  __syncthreads();
  unsigned total = 0;
#pragma unroll
  for (int w = 0; w < kSortWaves; ++w) total += count[w][tid];
  if constexpr (kChained) {
    tile_hist[static_cast<size_t>(tile) * kSortBins + tid] = total;
    for (int q = 1; q < passes; ++q) tile_hist[(static_cast<size_t>(q) * num_tiles + tile) * kSortBins + tid] = 0u;
  } else {
    tile_hist[static_cast<size_t>(tid) * num_tiles + tile] = total;
  }
  if (reduce_bits && tid == 0) {
    unsigned long long any = 0ull, all = ~0ull, pay = 0ull;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) {
      any |= wave_bits[w][0];
      all &= wave_bits[w][1];
      pay |= wave_bits[w][2];
    }
    tile_bits[kStateWords * tile] = any;
    tile_bits[kStateWords * tile + 1] = all;
    tile_bits[kStateWords * tile + 2] = pay;
  }
}

template <typename KeyT, int ITEMS = kSortItems, bool kChained = false>
__global__ void __launch_bounds__(kSortThreads)
RadixTileHistogramKernel(const SortArray<KeyT> keys, const int64_t n, const int pass, const int passes,
                         const SortMode mode, unsigned* __restrict__ tile_hist, const int num_tiles,
                         unsigned long long* __restrict__ tile_bits,
                         const unsigned long long* __restrict__ state,
                         const unsigned long long* __restrict__ payload64, const int xcds) {
  RadixTileHistogramBody<KeyT, ITEMS, kChained>(keys, n, pass, passes, mode, tile_hist, num_tiles, tile_bits, state,
                                                payload64, xcds, static_cast<int>(blockIdx.x));
}

//! Block-wide exclusive scan helper (256 threads): returns the exclusive prefix of `v` and the
//! block total through `total`.
__device__ __forceinline__ unsigned BlockExclusiveScan(unsigned v, unsigned* total) {
  __shared__ unsigned wave_sum[kSortWaves];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  unsigned incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned up = __shfl_up(incl, d);
    if (lane >= d) incl += up;
  }
  __syncthreads();  // wave_sum may still be read by a previous call
  if (lane == 63) wave_sum[wave] = incl;
  __syncthreads();
  unsigned before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kSortWaves; ++w) {
    const unsigned s = wave_sum[w];
    if (w < wave) before += s;
    all += s;
  }
  *total = all;
  return before + incl - v;
}

//! One workgroup per (segment, bin): tile_hist[bin][tiles of the segment] becomes its exclusive prefix over
//! those tiles; bin_total[segment][bin] receives the sum.  In pass 0 an extra workgroup (the last of the grid) folds
//! the tiles' OR/AND words into `state` (single writer, so nothing has to be zeroed first).
//! Few tiles (<= kFoldScanTiles): the per-bin work is folded into the scatter kernel and only
//! that extra workgroup is launched (grid = 1), in pass 0.
__device__ __forceinline__ void
RadixScanTilesBody(unsigned* __restrict__ tile_hist, const int num_tiles,
                   unsigned* __restrict__ bin_total, const int pass, const int passes, const SortMode& mode,
                   const unsigned long long* __restrict__ tile_bits,
                   unsigned long long* __restrict__ state, const int segment_tiles, const int block, const int grid) {
  if (block == grid - 1 && grid % kSortBins != 0) {  // the extra workgroup of pass 0
    __shared__ unsigned long long wave_bits[kSortWaves][kStateWords];
    unsigned long long any = 0ull, all = ~0ull, pay = 0ull;
    for (int t = threadIdx.x; t < num_tiles; t += kSortThreads) {
      any |= tile_bits[kStateWords * t];
      all &= tile_bits[kStateWords * t + 1];
      pay |= tile_bits[kStateWords * t + 2];
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      any |= __shfl_xor(any, d);
      all &= __shfl_xor(all, d);
      pay |= __shfl_xor(pay, d);
    }
    if ((threadIdx.x & 63) == 0) {
      wave_bits[threadIdx.x >> 6][0] = any;
      wave_bits[threadIdx.x >> 6][1] = all;
      wave_bits[threadIdx.x >> 6][2] = pay;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      any = 0ull;
      all = ~0ull;
      pay = 0ull;
#pragma unroll
      for (int w = 0; w < kSortWaves; ++w) {
        any |= wave_bits[w][0];
        all &= wave_bits[w][1];
        pay |= wave_bits[w][2];
      }
      state[kStateVarying] = any & ~all;
      state[kStateAllBits] = all;
      state[kStatePayloadBits] = pay;
      ZeroSortBarrier(state);         // the arrival counters of RadixHighPassesKernel's grid barrier
    }
    return;
  }
  // workgroup (segment, bin): the tiles of one segment (a block of the input that is sorted on its own;
  // one segment = the whole input unless the caller asked for sample blocks)
  const int bin = block % kSortBins;
  const int segment = block / kSortBins;
  const int first = segment * segment_tiles;
  const int last = first + segment_tiles < num_tiles ? first + segment_tiles : num_tiles;
  unsigned* row = tile_hist + static_cast<size_t>(bin) * num_tiles;
  unsigned carry = 0;
  // four consecutive tiles per thread and round: 1024 tiles (4.19 M keys) are ONE round -- one load, one block scan,
  // one store per thread instead of four dependent rounds (this kernel is all latency: 4.7 -> see EXPERIMENTS)
  constexpr int kPerThread = 4;
  // (the first round is requested BEFORE the state words say whether this pass runs at all: one memory round trip
  // instead of two in a row)
  unsigned ahead[kPerThread];
  {
    const int t0 = first + static_cast<int>(threadIdx.x) * kPerThread;
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) ahead[q] = t0 + q < last ? row[t0 + q] : 0u;
  }
  if (pass > 0 && !PlanPass(state, pass, passes, mode).active) return;
  for (int base = first; base < last; base += kSortThreads * kPerThread) {
    const int t0 = base + static_cast<int>(threadIdx.x) * kPerThread;
    unsigned v[kPerThread];
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) v[q] = base == first ? ahead[q] : (t0 + q < last ? row[t0 + q] : 0u);
    unsigned mine = 0;
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) mine += v[q];
    unsigned total;
    unsigned excl = carry + BlockExclusiveScan(mine, &total);
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
      if (t0 + q < last) row[t0 + q] = excl;
      excl += v[q];
    }
    carry += total;
  }
  if (threadIdx.x == 0) bin_total[block] = carry;   // [segment][bin]
}

__global__ void __launch_bounds__(kSortThreads)
RadixScanTilesKernel(unsigned* __restrict__ tile_hist, const int num_tiles,
                     unsigned* __restrict__ bin_total, const int pass, const int passes, const SortMode mode,
                     const unsigned long long* __restrict__ tile_bits,
                     unsigned long long* __restrict__ state, const int segment_tiles) {
  RadixScanTilesBody(tile_hist, num_tiles, bin_total, pass, passes, mode, tile_bits, state, segment_tiles,
                     static_cast<int>(blockIdx.x), static_cast<int>(gridDim.x));
}

//! Stable rank of the tile's keys by the digit at `shift`.  Lane l of wave w holds the keys at
//! TILE-LOCAL positions first_pos + r * 64 (r = 0..ITEMS-1; first_pos = wave * 64 * ITEMS + lane;
//! 32-bit on purpose: 64-bit position compares cost the int64 scatter kernel 17 VGPRs); slot[r]
//! receives the key's tile-local position in digit order (0xffffffff for positions >= n, the
//! number of keys in the tile).
//! `wave_count` ([kSortWaves][kSortBins], zeroed by the caller, barrier before the call) ends as
//! the exclusive prefix over waves, `tile_start[d]` as the tile-local position of the first key
//! with digit d.  Contains barriers: every thread of the workgroup must call it.
template <typename KeyT, int ITEMS>
__device__ __forceinline__ void RankTile(const KeyT (&key)[ITEMS], const int shift, const unsigned flip,
                                         const int first_pos, const int n,
                                         unsigned (*wave_count)[kSortBins], unsigned* tile_start,
                                         unsigned (&slot)[ITEMS]) {
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const bool valid = first_pos + r * 64 < n;
    const unsigned digit = static_cast<unsigned>((key[r] >> shift) & 0xff) ^ flip;
    const unsigned long long peers = MatchDigit(digit, valid);
    // every peer reads the wave's running count of its digit (one LDS broadcast per digit), then
    // the lowest peer bumps it; a wavefront's LDS operations execute in program order
    const unsigned lower = CountBelow(peers);
    unsigned start = 0;
    if (valid) start = wave_count[wave][digit];
    __builtin_amdgcn_wave_barrier();
    if (valid && lower == 0) wave_count[wave][digit] = start + static_cast<unsigned>(__popcll(peers));
    __builtin_amdgcn_wave_barrier();
    slot[r] = valid ? start + lower : 0xffffffffu;
  }
  __syncthreads();
  {  // per digit: wave counts -> exclusive prefix over waves; tile totals -> tile-local starts
    unsigned run = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) {
      const unsigned c = wave_count[w][tid];
      wave_count[w][tid] = run;
      run += c;
    }
    unsigned total;
    tile_start[tid] = BlockExclusiveScan(run, &total);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    if (slot[r] != 0xffffffffu) {
      const unsigned digit = static_cast<unsigned>((key[r] >> shift) & 0xff) ^ flip;
      slot[r] += tile_start[digit] + wave_count[wave][digit];
    }
  }
}

//! Scatter pass.  Position of a key = (keys with a smaller digit) + (equal-digit keys in
//! earlier tiles) + (equal-digit keys of earlier waves of this tile) + (its rank in its wave).
//! Keys and payloads are first put in digit order INSIDE the tile through LDS, so that the
//! global stores of a wavefront are runs of consecutive addresses (one run per digit present)
//! instead of 64 scattered elements: 2-3x faster on the low, uniformly distributed digits.
template <typename T>
__device__ __forceinline__ void StoreRouted(const SortArray<T>& a, const int where, const bool narrow,
                                            const int64_t n, const unsigned dest, const T value) {
  if constexpr (sizeof(T) == 8) {
    if (narrow && where >= kBufTmp0) {
      unsigned* p = reinterpret_cast<unsigned*>(a.tmp) + (where == kBufTmp1 ? n : 0);
      p[dest] = static_cast<unsigned>(value);
      return;
    }
  }
  T* p = where == kBufOut ? a.out : a.tmp;
  p[dest] = value;
}

template <typename T, int ITEMS>
__device__ __forceinline__ void StageAndStore(unsigned char* stage_raw, const T (&item)[ITEMS],
                                              const unsigned (&slot)[ITEMS],
                                              const unsigned (&dest)[ITEMS], const int count,
                                              const SortArray<T>& a, const int where, const bool narrow,
                                              const int64_t n) {
  T* stage = reinterpret_cast<T*>(stage_raw);
  __syncthreads();  // previous user of the staging buffer is done
#pragma unroll
  for (int r = 0; r < ITEMS; ++r)
    if (slot[r] != 0xffffffffu) stage[slot[r]] = item[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int q = r * kSortThreads + threadIdx.x;
    if (q < count) StoreRouted<T>(a, where, narrow, n, dest[r], stage[q]);
  }
}

// (second launch bound = wavefronts per SIMD: 4 workgroups of 4 waves per CU, i.e. at most 128
// VGPRs -- with 1024 tiles on 256 CUs a fifth of the tiles would otherwise wait for a second round)
//
// kChained (inputs of up to kChainedSortMax keys, tiles of kSortThreads * ITEMS keys): ONE launch per pass.  At these
// sizes a dependent launch costs 3.5-5 us whatever it does (profiles/r05_small_sort_baseline.txt), so the histogram
// launch of every pass but the first is folded into the scatter launch before it: while a workgroup of pass p writes a
// key to its new position it also counts it -- a global atomic, one per run of equal (destination tile, next digit)
// among neighbouring keys, so a hot index costs a few atomics per wavefront, not one per lookup -- into the histogram
// of the pass that runs NEXT ([pass][tile][bin] words in `tile_prefix`, zeroed by pass 0's histogram launch).  Every
// workgroup adds up the per-tile counts it needs itself (no scan launch), and the OR / AND words of the keys are folded
// by every workgroup of pass 0 (no extra workgroup): 12-16 launches become 1 + the number of passes.
template <typename KeyT, typename V1, typename V2, int ITEMS = kSortItems, bool kChained = false>
__device__ __forceinline__ void
RadixScatterBody(const SortArray<KeyT>& keys, const SortArray<V1>& v1, const SortArray<V2>& v2,
                 const int64_t n, const int pass, const int passes, const SortMode& mode,
                 const unsigned* __restrict__ tile_prefix, const unsigned* __restrict__ bin_total,
                 const int num_tiles, const unsigned long long* __restrict__ state, const int segment_tiles,
                 const int xcds, unsigned* next_hist /* kChained: the [pass][tile][bin] words, writable */,
                 const unsigned long long* __restrict__ tile_bits /* kChained, pass 0: the tiles' OR / AND words */,
                 unsigned long long* __restrict__ state_out /* kChained, pass 0: where their fold goes */, const int block) {
  constexpr int kTile = kSortThreads * ITEMS;
  constexpr bool kHasV1 = !std::is_same<V1, NoPayload>::value;
  constexpr bool kHasV2 = !std::is_same<V2, NoPayload>::value;
  constexpr size_t kStageElem = sizeof(KeyT) > sizeof(V1) ? sizeof(KeyT) : sizeof(V1);
  __shared__ __attribute__((aligned(16))) unsigned char stage[kTile * (kStageElem > sizeof(V2) ? kStageElem : sizeof(V2))];
  __shared__ unsigned digit_base[kSortBins];   // global position of the tile's first key with this digit
  __shared__ unsigned tile_start[kSortBins];   // tile-local position of the first key with this digit
  __shared__ unsigned wave_count[kSortWaves][kSortBins];  // becomes the exclusive prefix over waves
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int tile = ScatterTileOfBlock(block, num_tiles, xcds);
  const int64_t tile_base = static_cast<int64_t>(tile) * kTile;
  const int count = static_cast<int>(n - tile_base < kTile ? n - tile_base : kTile);
  const int64_t wave_base = tile_base + wave * (64 * ITEMS);
  KeyT key[ITEMS];
  // which of (caller's input, caller's output, scratch) this pass reads and writes follows from
  // the passes that run at all and from how each array is stored in the scratch
  PassPlan plan;
  bool keys_requested = false, payload_requested = false;
  constexpr int kEarlyItems = (kChained && kHasV1 && sizeof(V1) == 4) ? ITEMS : 1;
  V1 early1_out[kEarlyItems], early1_tmp[kEarlyItems];   // (kChained: a 32-bit first payload from both possible sources)
  if constexpr (kChained) {
    if (pass == 0) {
      // the first pass always reads the caller's input: request it, then find out what the later passes will do
      LoadRouted<KeyT>(keys, kBufIn, false, n, wave_base + lane, 64, KeyT(0), key);
      keys_requested = true;
      if (tile_bits != nullptr) {
        __shared__ unsigned long long fold[kSortWaves][kStateWords];
        unsigned long long any = 0ull, all = ~0ull, pay = 0ull;
        for (int t = tid; t < num_tiles; t += kSortThreads) {
          any |= tile_bits[kStateWords * t];
          all &= tile_bits[kStateWords * t + 1];
          pay |= tile_bits[kStateWords * t + 2];
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
          any |= __shfl_xor(any, d);
          all &= __shfl_xor(all, d);
          pay |= __shfl_xor(pay, d);
        }
        if (lane == 0) {
          fold[wave][0] = any;
          fold[wave][1] = all;
          fold[wave][2] = pay;
        }
        __syncthreads();
        any = 0ull;
        all = ~0ull;
        pay = 0ull;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
          any |= fold[w][0];
          all &= fold[w][1];
          pay |= fold[w][2];
        }
        plan = PlanPassFrom(true, any & ~all, all, pay, 0, passes, mode);
        if (block == 0 && tid == 0) {   // (every workgroup computes the same three words; one writes them)
          state_out[kStateVarying] = any & ~all;
          state_out[kStateAllBits] = all;
          state_out[kStatePayloadBits] = pay;
          ZeroSortBarrier(state_out);         // (RadixHighPassesKernel's arrival counters)
        }
      } else {
        plan = PlanPassFrom(false, 0ull, 0ull, 0ull, 0, passes, mode);
      }
    } else {
      // A later pass reads the caller's output or the scratch, depending on how many passes run after it -- which the
      // three state words say.  Waiting for them before requesting the keys costs a full memory round trip (~2 us: the
      // words were written by another XCD) per pass; 32-bit arrays have only two possible sources, so BOTH are
      // requested together with the state and the right one is kept.
      if constexpr (sizeof(KeyT) == 4) {
        if (state != nullptr) {
          KeyT from_out[ITEMS], from_tmp[ITEMS];
          LoadRouted<KeyT>(keys, kBufOut, false, n, wave_base + lane, 64, KeyT(0), from_out);
          LoadRouted<KeyT>(keys, kBufTmp0, false, n, wave_base + lane, 64, KeyT(0), from_tmp);
          if constexpr (kHasV1 && sizeof(V1) == 4) {
            LoadRouted<V1>(v1, kBufOut, false, n, wave_base + lane, 64, V1(0), early1_out);
            LoadRouted<V1>(v1, kBufTmp0, false, n, wave_base + lane, 64, V1(0), early1_tmp);
          }
          plan = PlanPass(state, pass, passes, mode);
          const bool use_out = RouteArray(plan, false).src == kBufOut;
#pragma unroll
          for (int r = 0; r < ITEMS; ++r) key[r] = use_out ? from_out[r] : from_tmp[r];
          keys_requested = true;
          payload_requested = kHasV1 && sizeof(V1) == 4;
        } else {
          plan = PlanPass(state, pass, passes, mode);
        }
      } else {
        plan = PlanPass(state, pass, passes, mode);
      }
    }
  } else {
    plan = PlanPass(state, pass, passes, mode);
  }
  if (!plan.active) return;
  const int shift = 8 * pass;
  const unsigned flip = SignFlip(mode, pass);
  const bool narrow_v1 = kHasV1 && sizeof(V1) == 8 && plan.narrow_v1;  // all values in [0, 2^32)
  const ArrayRoute key_route = RouteArray(plan, plan.narrow_keys);
  const ArrayRoute v1_route = RouteArray(plan, narrow_v1);
  const ArrayRoute v2_route = RouteArray(plan, false);
  // ---- load first (everything in flight at once): the digit bases below are computed under the loads' latency ----
  if (!keys_requested)
    LoadRouted<KeyT>(keys, key_route.src, plan.narrow_keys, n, wave_base + lane, 64,
                     static_cast<KeyT>(plan.key_high), key);
  // A 32-bit first payload is requested now as well, so that its latency overlaps the ranking.
  // A 64-bit one would push the kernel past 128 VGPRs (3 instead of 4 resident workgroups per CU,
  // i.e. a second round for a quarter of the 1024 tiles); it is loaded after the keys have left.
  constexpr bool kEarlyV1 = kHasV1 && sizeof(V1) <= 4;
  V1 item1[ITEMS];
  const bool implicit_v1 = kHasV1 && plan.first && mode.v1_div > 0;
  if constexpr (kEarlyV1) {
    if (payload_requested) {
      if constexpr (kChained && kHasV1 && sizeof(V1) == 4) {
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) item1[r] = v1_route.src == kBufOut ? early1_out[r] : early1_tmp[r];
      }
    } else if (!implicit_v1) {
      LoadRouted<V1>(v1, v1_route.src, false, n, wave_base + lane, 64, V1(0), item1);
    }
  }
  {
    unsigned before_me, bin_sum;
    unsigned segment_start = 0;   // the tile's segment is sorted on its own: positions start at its first element
    if constexpr (kChained) {
      // thread `bin` adds up this pass's counts of its bin over the tiles ([tile][bin]: a coalesced row per tile, 8
      // rows in flight).  (Four bins per lane with 16-byte loads, the tiles split over the wavefronts and the first
      // batch requested before anything else, was built and measured: the same from 64 tiles up, 10 % slower at 16.)
      const unsigned* counts = tile_prefix + static_cast<size_t>(pass) * num_tiles * kSortBins + tid;
      before_me = 0;
      bin_sum = 0;
#pragma unroll 8
      for (int t = 0; t < num_tiles; ++t) {
        const unsigned c = counts[static_cast<size_t>(t) * kSortBins];
        if (t < tile) before_me += c;
        bin_sum += c;
      }
    } else if (bin_total != nullptr) {
      const int segment = tile / segment_tiles;
      segment_start = static_cast<unsigned>(segment) * static_cast<unsigned>(segment_tiles) * kTile;
      before_me = tile_prefix[static_cast<size_t>(tid) * num_tiles + tile];
      bin_sum = bin_total[segment * kSortBins + tid];
    } else {
      // few tiles: no scan launch -- thread `bin` adds up the raw tile histograms of its bin itself
      const unsigned* row = tile_prefix + static_cast<size_t>(tid) * num_tiles;
      before_me = 0;
      bin_sum = 0;
      for (int t = 0; t < num_tiles; ++t) {
        const unsigned c = row[t];
        if (t < tile) before_me += c;
        bin_sum += c;
      }
    }
    unsigned total;
    const unsigned smaller = BlockExclusiveScan(bin_sum, &total);
    digit_base[tid] = segment_start + smaller + before_me;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) wave_count[w][tid] = 0;
  }
  __syncthreads();

  // ---- rank inside the wave ----
  unsigned slot[ITEMS];  // tile-local position in digit order
  RankTile<KeyT>(key, shift, flip, wave * (64 * ITEMS) + lane, count, wave_count, tile_start, slot);

  // ---- keys: through LDS into digit order, then out in runs ----
  KeyT* stage_keys = reinterpret_cast<KeyT*>(stage);
#pragma unroll
  for (int r = 0; r < ITEMS; ++r)
    if (slot[r] != 0xffffffffu) stage_keys[slot[r]] = key[r];
  __syncthreads();
  unsigned dest[ITEMS];  // global position of the element at tile-local position q
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int q = r * kSortThreads + tid;
    if (q < count) {
      const KeyT k = stage_keys[q];
      const unsigned digit = static_cast<unsigned>((k >> shift) & 0xff) ^ flip;
      dest[r] = digit_base[digit] + (static_cast<unsigned>(q) - tile_start[digit]);
      StoreRouted<KeyT>(keys, key_route.dst, plan.narrow_keys, n, dest[r], k);
    }
  }
  if constexpr (kChained) {
    // ---- the histogram of the NEXT working pass: its tiles are cut from the positions this pass writes ----
    if (plan.next >= 0) {
      unsigned* counts = next_hist + static_cast<size_t>(plan.next) * num_tiles * kSortBins;
      const int next_shift = 8 * plan.next;
      const unsigned next_flip = SignFlip(mode, plan.next);
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) {
        const int q = r * kSortThreads + tid;
        unsigned word = 0xffffffffu;   // (destination tile, next digit) of the key at tile-local position q
        if (q < count) {
          const KeyT k = stage_keys[q];
          word = (dest[r] / kTile) * kSortBins + (static_cast<unsigned>((k >> next_shift) & 0xff) ^ next_flip);
        }
        // neighbours in digit order go to neighbouring positions: only the first lane of a run of equal words adds,
        // and it adds the run's length
        const unsigned before = __shfl_up(word, 1);
        const bool head = lane == 0 || before != word;
        const unsigned long long heads = __ballot(head);
        const unsigned long long above = lane == 63 ? 0ull : heads >> (lane + 1);
        const unsigned run = above != 0 ? static_cast<unsigned>(__ffsll(static_cast<long long>(above)))
                                        : static_cast<unsigned>(64 - lane);
        if (head && word != 0xffffffffu) atomicAdd(&counts[word], run);
      }
    }
  }
  // ---- payloads take the same route ----
  if constexpr (kHasV1) {
    if (implicit_v1) {
      if constexpr (!std::is_same<V1, NoPayload>::value) {
#pragma unroll
        for (int r = 0; r < ITEMS; ++r)
          item1[r] = static_cast<V1>(ImplicitPayload(mode, wave_base + lane + r * 64));
      }
    } else {
      if constexpr (!kEarlyV1) LoadRouted<V1>(v1, v1_route.src, narrow_v1, n, wave_base + lane, 64, V1(0), item1);
    }
    StageAndStore<V1>(stage, item1, slot, dest, count, v1, v1_route.dst, narrow_v1, n);
  }
  if constexpr (kHasV2) {
    V2 item[ITEMS];
    LoadRouted<V2>(v2, v2_route.src, false, n, wave_base + lane, 64, V2(), item);
    StageAndStore<V2>(stage, item, slot, dest, count, v2, v2_route.dst, false, n);
  }
}

template <typename KeyT, typename V1, typename V2, int ITEMS = kSortItems, bool kChained = false>
__global__ void __launch_bounds__(kSortThreads, 4)
RadixScatterKernel(const SortArray<KeyT> keys, const SortArray<V1> v1, const SortArray<V2> v2,
                   const int64_t n, const int pass, const int passes, const SortMode mode,
                   const unsigned* __restrict__ tile_prefix, const unsigned* __restrict__ bin_total,
                   const int num_tiles, const unsigned long long* __restrict__ state, const int segment_tiles,
                   const int xcds, unsigned* next_hist = nullptr, const unsigned long long* __restrict__ tile_bits = nullptr,
                   unsigned long long* __restrict__ state_out = nullptr) {
  RadixScatterBody<KeyT, V1, V2, ITEMS, kChained>(keys, v1, v2, n, pass, passes, mode, tile_prefix, bin_total, num_tiles,
                                                  state, segment_tiles, xcds, next_hist, tile_bits, state_out,
                                                  static_cast<int>(blockIdx.x));
}

//! Inputs of up to this many keys are sorted by the chained kernels (one launch per pass, RadixScatterKernel<...,
//! kChained>), in tiles of kSortThreads * kChainedSortItems keys -- a workgroup then ranks 4 rounds of 64 keys per
//! wavefront instead of 16, and 64-224 compute units work on a pass instead of 16-56.  The limit is where the tiled
//! passes catch up (profiles/r05_chained_sort_limit.txt: 196,608 keys 60.7 against 66.7 us, 262,144 keys 70.8 against
//! 67.3): a pass's atomics grow with the keys (one per key in a pass whose next digit is random), three launches do not.
//! In 4096-key tiles beyond that (fewer, longer sums over the tiles) the same scheme is slower than the tiled passes
//! at every size tried (524,288 keys: 99.6 against 70.5 us; 1,048,576: 147 against 78).
constexpr size_t kChainedSortMax = size_t{224} << 10;
constexpr int kChainedSortItems = 4;
inline size_t ChainedSortLimit() {   // tuning: CUEMBED_CHAINED_SORT_MAX (read once; never above kChainedSortMax)
  static const size_t limit = [] {
    const char* e = std::getenv("CUEMBED_CHAINED_SORT_MAX");
    const long long v = e != nullptr ? std::atoll(e) : static_cast<long long>(kChainedSortMax);
    return v < 0 ? size_t{0} : (static_cast<size_t>(v) > kChainedSortMax ? kChainedSortMax : static_cast<size_t>(v));
  }();
  return limit;
}
inline int ChainedSortTiles(const size_t n) {
  const size_t tile = static_cast<size_t>(kSortThreads) * kChainedSortItems;
  return static_cast<int>((n + tile - 1) / tile);
}

// ---------------------------------------------------------------------------
// The HIGH WORD of 64-bit keys in one launch.  Lookup indices are below 2^31 (the API's row counts are `int`), so
// passes 4..7 of an int64 sort through the reference signature (all 64 bits: index_transforms.cuh:108-136) are
// skipped on the device practically always -- but the host cannot know, and twelve launches that return at once cost
// ~45 us (C4, int64: 0.187 ms against 0.129 with a key bound).  This kernel is ALL of them: one workgroup per
// compute unit, which returns at once when no high digit varies and otherwise runs the working passes itself --
// histogram, scan and scatter of every tile, handed out in execution order from a ticket queue (SortWorkQueue below:
// agent-scope release / acquire, NO grid barrier and no assumption about which workgroups are resident).  The same
// result, but SLOW when it has work (a generic COO transpose whose keys go beyond 2^32; round 5 measured the grid-barrier
// form of this kernel, tools/persistent_sort_probe.hip, 4.2 M keys using all 64 bits: 0.529 ms with one workgroup per
// compute unit against 0.285 as 24 launches: every phase boundary costs a release that walks the XCD's L2, and a phase
// at one workgroup per compute unit runs at a third of the launched kernels' rate).  That measurement is also why the
// WHOLE sort is not one persistent launch.  CUEMBED_SORT_HIGH_WORD_LAUNCHES=1 brings the launches back.
// ---------------------------------------------------------------------------
// Phases inside the launch are ordered WITHOUT a grid barrier and without assuming that the workgroups are resident
// together (an ordinary launch guarantees nothing of the kind: under RCCL's persistent kernels, other streams or other
// processes some of them may not get a slot for as long as the others spin -- a barrier would then hang).  The work is a
// single ordered list of items -- (pass, phase, tile) in execution order -- and a workgroup CLAIMS the next item with a
// ticket (one atomic), waits until every item of the earlier phases is done (a counter of finished items) and runs it.
// Whoever holds a ticket is running, and waits only for lower tickets, which by induction are held by running
// workgroups or finished: the launch completes with ANY number of resident workgroups, one included.  Workgroups that
// get their slot late find the tickets gone and leave.  (tests: a grid many times what the device can hold, and sorts
// racing CU-filling kernels on other streams.)
struct SortWorkQueue {
  unsigned* ticket;   //!< next item to hand out
  unsigned* done;     //!< items finished (their stores released at agent scope)
  unsigned* slot;     //!< one LDS word: the claimed ticket, for the whole workgroup
  __device__ __forceinline__ unsigned Claim() const {
    __syncthreads();                                 // (the previous item is through with LDS, `slot` included)
    if (threadIdx.x == 0) *slot = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return *slot;
  }
  __device__ __forceinline__ void WaitFor(const unsigned finished_items) const {
    if (threadIdx.x == 0) {
      while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < finished_items)
        __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // drop what other XCDs have rewritten
    }
    __syncthreads();
  }
  __device__ __forceinline__ void Finish() const {
    __syncthreads();                                 // every thread's stores of the item are issued
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // write this XCD's dirty lines back
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
};

//! The working passes among [first_pass, passes): at most four (the high word of a 64-bit key).
struct ActivePasses {
  int pass[4];
  int count;
};
__device__ __forceinline__ ActivePasses FindActivePasses(const unsigned long long* state, const int first_pass,
                                                         const int passes, const SortMode mode) {
  ActivePasses a;
  a.count = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) a.pass[k] = 0;
  for (int p = first_pass; p < passes && a.count < 4; ++p)
    if (PlanPass(state, p, passes, mode).active) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k == a.count) a.pass[k] = p;
      ++a.count;
    }
  return a;
}

template <typename KeyT, typename V1, typename V2>
__global__ void __launch_bounds__(kSortThreads)
RadixHighPassesKernel(const SortArray<KeyT> keys, const SortArray<V1> v1, const SortArray<V2> v2, const int64_t n,
                      const int first_pass, const int passes, const SortMode mode, unsigned* tile_hist,
                      unsigned* bin_total, const int num_tiles, unsigned long long* state, const int segment_tiles,
                      const int segments, const int xcds) {
  __shared__ unsigned claimed;
  const ActivePasses act = FindActivePasses(state, first_pass, passes, mode);
  if (act.count == 0) return;                        // (the same three words for every workgroup)
  unsigned* counter = SortBarrierCounters(state);
  const SortWorkQueue queue{counter, counter + kBarrierStride, &claimed};
  const unsigned scan_blocks = static_cast<unsigned>(kSortBins * segments);
  const unsigned tiles = static_cast<unsigned>(num_tiles);
  // A ticket is a CHUNK of consecutive tiles (as many as give every workgroup of the grid one ticket per phase): the
  // release behind an item walks the XCD's L2, and one per tile made the kernel twice as slow as one per chunk.
  const unsigned tile_chunk = (tiles + gridDim.x - 1) / gridDim.x;
  const unsigned tile_tickets = (tiles + tile_chunk - 1) / tile_chunk;
  const unsigned scan_chunk = (scan_blocks + gridDim.x - 1) / gridDim.x;
  const unsigned scan_tickets = (scan_blocks + scan_chunk - 1) / scan_chunk;
  const unsigned per_pass = 2u * tile_tickets + scan_tickets;    // histogram of every tile, scan, scatter of every tile
  const unsigned total = per_pass * static_cast<unsigned>(act.count);
  for (;;) {
    const unsigned t = queue.Claim();
    if (t >= total) return;
    const unsigned a = t / per_pass, r = t - a * per_pass;
    int p = act.pass[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (static_cast<unsigned>(k) == a) p = act.pass[k];
    if (r < tile_tickets) {
      queue.WaitFor(a * per_pass);                               // the previous pass has scattered everything
      const unsigned end = (r + 1u) * tile_chunk < tiles ? (r + 1u) * tile_chunk : tiles;
      for (unsigned b = r * tile_chunk; b < end; ++b) {
        RadixTileHistogramBody<KeyT, kSortItems, false>(keys, n, p, passes, mode, tile_hist, num_tiles, nullptr, state,
                                                        nullptr, xcds, static_cast<int>(b));
        __syncthreads();
      }
    } else if (r < tile_tickets + scan_tickets) {
      queue.WaitFor(a * per_pass + tile_tickets);                // every tile is counted
      const unsigned q = r - tile_tickets;
      const unsigned end = (q + 1u) * scan_chunk < scan_blocks ? (q + 1u) * scan_chunk : scan_blocks;
      for (unsigned b = q * scan_chunk; b < end; ++b) {
        RadixScanTilesBody(tile_hist, num_tiles, bin_total, p, passes, mode, nullptr, state, segment_tiles,
                           static_cast<int>(b), static_cast<int>(scan_blocks));
        __syncthreads();
      }
    } else {
      queue.WaitFor(a * per_pass + tile_tickets + scan_tickets); // every bin is scanned
      const unsigned q = r - tile_tickets - scan_tickets;
      const unsigned end = (q + 1u) * tile_chunk < tiles ? (q + 1u) * tile_chunk : tiles;
      for (unsigned b = q * tile_chunk; b < end; ++b) {
        RadixScatterBody<KeyT, V1, V2, kSortItems, false>(keys, v1, v2, n, p, passes, mode, tile_hist, bin_total, num_tiles,
                                                          state, segment_tiles, xcds, nullptr, nullptr, nullptr,
                                                          static_cast<int>(b));
        __syncthreads();
      }
    }
    queue.Finish();
  }
}

//! ... and the same for the chained passes (inputs of up to kChainedSortMax keys): one scatter phase per working pass.
template <typename KeyT, typename V1, typename V2>
__global__ void __launch_bounds__(kSortThreads)
RadixHighPassesChainedKernel(const SortArray<KeyT> keys, const SortArray<V1> v1, const SortArray<V2> v2, const int64_t n,
                             const int first_pass, const int passes, const SortMode mode, unsigned* tile_hist,
                             const int num_tiles, unsigned long long* state, const int xcds) {
  __shared__ unsigned claimed;
  const ActivePasses act = FindActivePasses(state, first_pass, passes, mode);
  if (act.count == 0) return;
  unsigned* counter = SortBarrierCounters(state);
  const SortWorkQueue queue{counter, counter + kBarrierStride, &claimed};
  const unsigned tiles = static_cast<unsigned>(num_tiles);
  const unsigned chunk = (tiles + gridDim.x - 1) / gridDim.x;     // tiles per ticket (see RadixHighPassesKernel)
  const unsigned tickets = (tiles + chunk - 1) / chunk;
  const unsigned total = tickets * static_cast<unsigned>(act.count);
  for (;;) {
    const unsigned t = queue.Claim();
    if (t >= total) return;
    const unsigned a = t / tickets, r = t - a * tickets;
    int p = act.pass[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (static_cast<unsigned>(k) == a) p = act.pass[k];
    queue.WaitFor(a * tickets);          // the pass adds up the counts the previous one left with atomics
    const unsigned end = (r + 1u) * chunk < tiles ? (r + 1u) * chunk : tiles;
    for (unsigned b = r * chunk; b < end; ++b) {
      RadixScatterBody<KeyT, V1, V2, kChainedSortItems, true>(keys, v1, v2, n, p, passes, mode, tile_hist, nullptr, num_tiles,
                                                              state, num_tiles, xcds, tile_hist, nullptr, nullptr,
                                                              static_cast<int>(b));
      __syncthreads();
    }
    queue.Finish();
  }
}

//! Grid of the high-word kernels: one workgroup per compute unit (a grid for throughput -- the kernels are correct with
//! any number of resident workgroups, see SortWorkQueue).  Kernels that find no work (every lookup index) return at once.
//! CUEMBED_SORT_HIGH_WORD_WORKGROUPS (read once) overrides it: tests run grids far beyond what the device holds.
inline int HighWordWorkgroups() {
  static const int forced = [] {
    const char* e = std::getenv("CUEMBED_SORT_HIGH_WORD_WORKGROUPS");
    return e != nullptr ? std::atoi(e) : 0;
  }();
  if (forced > 0) return forced;
  const int units = CurrentDeviceShape().compute_units;
  return units < 1 ? 1 : units;
}

//! (no more workgroups than tiles -- unless a test forces the grid)
inline int HighWordGrid(const int tiles, const int units) {
  static const bool forced = std::getenv("CUEMBED_SORT_HIGH_WORD_WORKGROUPS") != nullptr;
  return (forced || tiles > units) ? units : tiles;
}

//! CUEMBED_SORT_HIGH_WORD_LAUNCHES=1 (read once): the passes over the high word of 64-bit keys as launches of their own
//! again -- for a process whose keys really use more than 32 bits (see RadixHighPassesKernel: 1.07 against 0.28 ms).
inline bool HighWordInOneLaunch() {
  static const bool v = [] {
    const char* e = std::getenv("CUEMBED_SORT_HIGH_WORD_LAUNCHES");
    return !(e != nullptr && std::atoi(e) != 0);
  }();
  return v;
}

//! Largest input the one-workgroup sort takes (tuning: CUEMBED_BLOCK_SORT_MAX, read once; never above kBlockSortMax).
inline int BlockSortLimit() {
  static const int limit = [] {
    const char* e = std::getenv("CUEMBED_BLOCK_SORT_MAX");
    const int v = e != nullptr ? std::atoi(e) : kBlockSortMax;
    return v < 1 ? 1 : (v > kBlockSortMax ? kBlockSortMax : v);
  }();
  return limit;
}

template <typename KeyT, typename V1, typename V2>
struct RadixSortPlan {
  int passes;
  int num_tiles;
  int chained_tiles;   //!< > 0: n <= kChainedSortMax, tiles of the chained kernels
  size_t keys_tmp, v1_tmp, v2_tmp, tile_hist, bin_total, tile_bits, varying, total;  // byte offsets
  RadixSortPlan(const size_t n, const int key_bits) {
    passes = (key_bits + 7) / 8;
    if (passes < 1) passes = 1;
    num_tiles = static_cast<int>((n + kSortTile - 1) / kSortTile);
    if (num_tiles < 1) num_tiles = 1;
    chained_tiles = (n > 0 && n <= ChainedSortLimit()) ? ChainedSortTiles(n) : 0;
    size_t off = 0;
    keys_tmp = off;
    off += SortAlign(n * sizeof(KeyT));
    v1_tmp = off;
    if (!std::is_same<V1, NoPayload>::value) off += SortAlign(n * sizeof(V1));
    v2_tmp = off;
    if (!std::is_same<V2, NoPayload>::value) off += SortAlign(n * sizeof(V2));
    tile_hist = off;   // [bin][tile] of the current pass; chained: [pass][tile][bin] of every pass
    const size_t hist_words = chained_tiles > 0 ? static_cast<size_t>(passes) * chained_tiles * kSortBins
                                                : static_cast<size_t>(kSortBins) * num_tiles;
    off += SortAlign((hist_words > static_cast<size_t>(kSortBins) * num_tiles ? hist_words
                                                                             : static_cast<size_t>(kSortBins) * num_tiles) *
                     sizeof(unsigned));
    bin_total = off;
    off += SortAlign(static_cast<size_t>(kMaxSortSegments) * kSortBins * sizeof(unsigned));
    tile_bits = off;
    off += SortAlign(static_cast<size_t>(kStateWords) * (chained_tiles > num_tiles ? chained_tiles : num_tiles) *
                     sizeof(unsigned long long));
    varying = off;   // the three state words + the arrival counter of RadixHighPassesKernel
    off += SortAlign((kStateWords + 1 + kBarrierWords) * sizeof(unsigned long long));
    total = off;
  }
};

template <typename IndexT>
inline void RunHeadScan(const IndexT* indices, const size_t n, IndexT* remapped, char* work, hipStream_t stream);

//! Stable sort of n (key, v1[, v2]) by the low `key_bits` bits of the key.  Inputs are not
//! modified; outputs and `work` (at least RadixSortPlan::total bytes) must not overlap the inputs.
//!   signed_keys: the keys are two's-complement numbers; with key_bits = all bits they are put in
//!                signed order (negative keys first).  With key_bits < all bits the keys must lie in
//!                [0, 2^key_bits).
//!   v1_bits    : a 64-bit v1 whose values are known to lie in [0, 2^v1_bits), v1_bits <= 32, is
//!                kept as 32 bits between passes without looking; 0 = unknown, decided on the
//!                device from the values themselves (pass 0 reads them once more for that).
//!   v1_div     : > 0: v1_in is not read; the first payload of element i is i / v1_div.
//!   blocks     : > 1: the input is cut into (at most) this many consecutive blocks of equal length -- a whole
//!                number of 4096-element tiles each, the last one takes what is left -- and every block is sorted
//!                ON ITS OWN: the output is the concatenation of the sorted blocks (same kernels, same launches;
//!                only the tile scan and the digit bases are per block).  At most kMaxSortSegments; ignored for
//!                inputs of up to kFoldScanTiles tiles.
//!   remapped   : not null: also receives the run-head ids of the SORTED keys (RunHeadScan below: remapped[i] = number
//!                of k in (0, i] with keys_out[k] != keys_out[k - 1]) -- inside the one launch of a small sort, by the
//!                run-head scan's own launches after a large one (`work` is re-used: the sort is done with it by then).
template <typename KeyT, typename V1, typename V2>
inline void RadixSortPairs(const KeyT* keys_in, KeyT* keys_out, const V1* v1_in, V1* v1_out,
                           const V2* v2_in, V2* v2_out, const size_t n, const int key_bits,
                           char* work, hipStream_t stream, const bool signed_keys = false,
                           const int v1_bits = 0, const int v1_div = 0, const int blocks = 1,
                           typename std::make_signed<KeyT>::type* remapped = nullptr) {
  if (n == 0) return;
  const RadixSortPlan<KeyT, V1, V2> plan(n, key_bits);
  const int sign_pass = (signed_keys && key_bits >= static_cast<int>(8 * sizeof(KeyT))) ? plan.passes - 1 : -1;
  if (n <= static_cast<size_t>(BlockSortLimit())) {   // one workgroup, one launch: block_sort_kernels.hpp
    SortMode small{};
    if (v1_div > 0) ImplicitPayloadDivisor(v1_div, &small);
    BlockSortLaunch<KeyT, V1, V2>(keys_in, keys_out, v1_in, v1_out, v2_in, v2_out, static_cast<int>(n), plan.passes,
                                  sign_pass, small, remapped, stream);
    return;
  }
  const SortArray<KeyT> keys{keys_in, keys_out, reinterpret_cast<KeyT*>(work + plan.keys_tmp)};
  const SortArray<V1> v1{v1_in, v1_out, reinterpret_cast<V1*>(work + plan.v1_tmp)};
  const SortArray<V2> v2{v2_in, v2_out, reinterpret_cast<V2*>(work + plan.v2_tmp)};
  unsigned* tile_hist = reinterpret_cast<unsigned*>(work + plan.tile_hist);
  unsigned* bin_total = reinterpret_cast<unsigned*>(work + plan.bin_total);
  constexpr bool kWideV1 = !std::is_same<V1, NoPayload>::value && sizeof(V1) == 8;
  SortMode mode{};
  if (v1_div > 0) ImplicitPayloadDivisor(v1_div, &mode);
  mode.use_varying = plan.passes > kStaticRoutePasses;
  mode.sign_pass = sign_pass;
  mode.narrow_keys = kNarrowNever;
  if (sizeof(KeyT) == 8) {
    if (key_bits <= 32) mode.narrow_keys = kNarrowAlways;                 // the caller's bound says so
    else if (mode.use_varying) mode.narrow_keys = kNarrowIfConstantHigh;  // decided from the keys themselves
  }
  mode.narrow_v1 = !kWideV1 ? kNarrowNever
                            : ((v1_bits > 0 && v1_bits <= 32) || v1_div > 0 ? kNarrowAlways : kNarrowIfConstantHigh);
  const bool device_state = mode.use_varying || mode.narrow_v1 == kNarrowIfConstantHigh;
  unsigned long long* tile_bits =
      device_state ? reinterpret_cast<unsigned long long*>(work + plan.tile_bits) : nullptr;
  unsigned long long* state =
      device_state ? reinterpret_cast<unsigned long long*>(work + plan.varying) : nullptr;
  const unsigned long long* payload64 =
      mode.narrow_v1 == kNarrowIfConstantHigh ? reinterpret_cast<const unsigned long long*>(v1_in) : nullptr;
  const int64_t count = static_cast<int64_t>(n);
  const bool fold_scan = plan.num_tiles <= kFoldScanTiles;  // launch-bound sizes: one launch less per pass
  // segments (blocks sorted on their own): whole tiles each; one segment unless the caller asked for more
  const int segment_tiles = static_cast<int>(SortSegmentLength(n, blocks) / kSortTile);
  const int segments = (plan.num_tiles + segment_tiles - 1) / segment_tiles;
  const int xcds = CurrentDeviceShape().xcds;   // tile maps keep runs of tiles on one XCD
  if (plan.chained_tiles > 0 && segments == 1) {
    // one histogram launch, then ONE launch per pass (see RadixScatterKernel, kChained)
    const int tiles = plan.chained_tiles;
    RadixTileHistogramKernel<KeyT, kChainedSortItems, true><<<tiles, kSortThreads, 0, stream>>>(
        keys, count, 0, plan.passes, mode, tile_hist, tiles, tile_bits, nullptr, payload64, xcds);
    // (64-bit keys through the reference signature: the passes over the high word are one launch, see above)
    const int launched_passes =
        (sizeof(KeyT) == 8 && plan.passes > 4 && mode.use_varying && HighWordInOneLaunch()) ? 4 : plan.passes;
    for (int p = 0; p < launched_passes; ++p)
      RadixScatterKernel<KeyT, V1, V2, kChainedSortItems, true><<<tiles, kSortThreads, 0, stream>>>(
          keys, v1, v2, count, p, plan.passes, mode, tile_hist, nullptr, tiles, state, tiles, xcds, tile_hist,
          p == 0 ? tile_bits : nullptr, state);
    if constexpr (sizeof(KeyT) == 8) {
      if (launched_passes < plan.passes) {
        const int units = HighWordWorkgroups();
        RadixHighPassesChainedKernel<KeyT, V1, V2><<<HighWordGrid(tiles, units), kSortThreads, 0, stream>>>(
            keys, v1, v2, count, launched_passes, plan.passes, mode, tile_hist, tiles, state, xcds);
      }
    }
    if (remapped != nullptr)
      RunHeadScan<typename std::make_signed<KeyT>::type>(
          reinterpret_cast<const typename std::make_signed<KeyT>::type*>(keys_out), n, remapped, work, stream);
    return;
  }
  // 64-bit keys through the reference signature: the four passes over the high word are ONE launch (see
  // RadixHighPassesKernel); it needs the three state words, which such a sort always has
  const int tiled_passes =
      (sizeof(KeyT) == 8 && plan.passes > 4 && mode.use_varying && !fold_scan && HighWordInOneLaunch()) ? 4 : plan.passes;
  for (int p = 0; p < tiled_passes; ++p) {
    RadixTileHistogramKernel<KeyT><<<plan.num_tiles, kSortThreads, 0, stream>>>(
        keys, count, p, plan.passes, mode, tile_hist, plan.num_tiles, tile_bits, state, payload64, xcds);
    const int scan_blocks = (fold_scan ? 0 : kSortBins * segments) + (p == 0 && device_state ? 1 : 0);
    if (scan_blocks > 0)
      RadixScanTilesKernel<<<scan_blocks, kSortThreads, 0, stream>>>(
          tile_hist, plan.num_tiles, bin_total, p, plan.passes, mode, tile_bits, state, segment_tiles);
    RadixScatterKernel<KeyT, V1, V2><<<plan.num_tiles, kSortThreads, 0, stream>>>(
        keys, v1, v2, count, p, plan.passes, mode, tile_hist, fold_scan ? nullptr : bin_total,
        plan.num_tiles, state, segment_tiles, xcds);
  }
  if constexpr (sizeof(KeyT) == 8) {
    if (tiled_passes < plan.passes) {
      const int units = HighWordWorkgroups();
      RadixHighPassesKernel<KeyT, V1, V2><<<HighWordGrid(plan.num_tiles, units), kSortThreads, 0, stream>>>(
          keys, v1, v2, count, tiled_passes, plan.passes, mode, tile_hist, bin_total, plan.num_tiles, state, segment_tiles,
          segments, xcds);
    }
  }
  if (remapped != nullptr)
    RunHeadScan<typename std::make_signed<KeyT>::type>(reinterpret_cast<const typename std::make_signed<KeyT>::type*>(keys_out),
                                                       n, remapped, work, stream);
}

// ---------------------------------------------------------------------------
// Run-head scan: remapped[i] = number of positions k in (0, i] with indices[k] != indices[k-1].
// Two launches over 4096-element tiles: count the run heads per tile, then scan inside every tile
// on top of the sum of the earlier tiles' counts (which every workgroup adds up itself).
// For a sample-blocked array (blocked_order.hpp) `block_tiles` > 0 makes the first element of every block a
// run head whatever its neighbour holds (blocks are whole tiles, so only a tile's first element can be one).
// ---------------------------------------------------------------------------
//! Run-head ballots of the kSortItems x 64 consecutive elements a wavefront owns (element r * 64 + lane of the wave's
//! range is lane `lane` of round r): every element is loaded ONCE (all rounds in flight together); a lane's
//! predecessor is its neighbour's element (cross-lane read), lane 0's is the last element of the previous round, and
//! only the wave's very first element reads one word more.  `cur` returns the elements (the compaction needs them).
template <typename IndexT>
__device__ __forceinline__ void WaveRunHeads(const IndexT* __restrict__ indices, const int64_t wave_base, const int64_t n,
                                             const bool forced_first, IndexT (&cur)[kSortItems],
                                             unsigned long long (&heads)[kSortItems]) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = wave_base + r * 64 + lane;
    cur[r] = i < n ? indices[i] : IndexT(0);
  }
  IndexT edge = IndexT(0);   // the element before the wave's first one
  if (lane == 0 && wave_base > 0 && wave_base <= n) edge = indices[wave_base - 1];
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = wave_base + r * 64 + lane;
    IndexT prev = __shfl_up(cur[r], 1);
    const IndexT last_of_previous_round = r > 0 ? __shfl(cur[r > 0 ? r - 1 : 0], 63) : edge;
    if (lane == 0) prev = last_of_previous_round;
    heads[r] = __ballot((i > 0 && i < n && cur[r] != prev) || (forced_first && r == 0));
  }
}

//! The tile of this workgroup starts a block of a sample-blocked array.
__device__ __forceinline__ bool TileStartsBlock(const int block_tiles) {
  return block_tiles > 0 && blockIdx.x > 0 && static_cast<int>(blockIdx.x) % block_tiles == 0;
}

//! Flags are 0/1, so a wavefront counts and scans 64 of them with one ballot and a popcount.
template <typename IndexT>
__global__ void __launch_bounds__(kSortThreads)
RunHeadCountKernel(const IndexT* __restrict__ indices, const int64_t n, unsigned* __restrict__ tile_sum,
                   const int block_tiles) {
  __shared__ unsigned wave_sum[kSortWaves];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t wave_base = static_cast<int64_t>(blockIdx.x) * kSortTile + wave * (64 * kSortItems);
  const bool forced = TileStartsBlock(block_tiles) && threadIdx.x == 0;   // (the tile's first element is in range)
  IndexT cur[kSortItems];
  unsigned long long heads[kSortItems];
  WaveRunHeads<IndexT>(indices, wave_base, n, forced, cur, heads);
  unsigned c = 0;
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) c += static_cast<unsigned>(__popcll(heads[r]));
  if (lane == 0) wave_sum[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned t = 0;
#pragma unroll
    for (int w = 0; w < kSortWaves; ++w) t += wave_sum[w];
    tile_sum[blockIdx.x] = t;
  }
}

//! Many tiles (> kSelfSumTiles): tile_count[t] becomes the number of run heads in tiles 0..t-1
//! (one workgroup, in place), so that RunHeadScanKernel does not re-add O(tiles^2) words.
__global__ void __launch_bounds__(kSortThreads)
RunHeadTilePrefixKernel(unsigned* __restrict__ tile_count, const int num_tiles) {
  unsigned carry = 0;
  for (int base = 0; base < num_tiles; base += kSortThreads) {
    const int t = base + threadIdx.x;
    const unsigned v = t < num_tiles ? tile_count[t] : 0u;
    unsigned total;
    const unsigned excl = BlockExclusiveScan(v, &total);
    if (t < num_tiles) tile_count[t] = carry + excl;
    carry += total;
  }
}

//! Up to this many tiles (16,384 elements) the run-head scan is ONE launch: every workgroup counts the run heads of
//! the elements before its tile itself -- ONE batch of 16-byte loads per lane -- instead of waiting for a count launch
//! (two dependent launches of ~4.5 us each).  Not beyond: the array was written by other XCDs and comes from memory,
//! ~2 us per dependent batch (measured: 6.2 us at 4 tiles against 9.0 for two launches, but 7.8 at 8 and 12.2 at 16).
constexpr int kSelfCountTiles = 4;

//! Run heads among elements [1, upto) of `indices` (element i is one when it differs from element i - 1), counted by the
//! whole workgroup; every thread returns its share (the caller adds them up).  16 bytes per lane and load when the array
//! is 16-byte aligned, all loads of a batch requested before the first compare.
template <typename IndexT, int kThreads = kSortThreads>
__device__ __forceinline__ unsigned CountRunHeadsBefore(const IndexT* __restrict__ indices, const int64_t upto) {
  constexpr int kVec = 16 / static_cast<int>(sizeof(IndexT));
  typedef IndexT __attribute__((ext_vector_type(kVec))) vec_t;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  unsigned heads = 0;
  if ((reinterpret_cast<uintptr_t>(indices) & 15) != 0) {   // (a caller's oddly aligned view: element by element)
    for (int64_t i = 1 + tid; i < upto; i += kThreads) heads += indices[i] != indices[i - 1] ? 1u : 0u;
    return heads;
  }
  constexpr int kBatch = kThreads > kSortThreads ? 12 : 16;   // (1024-thread workgroups have 128 registers per lane)
  const int64_t groups = upto / kVec;                        // whole vectors; the ragged end is handled below
  for (int64_t g0 = 0; g0 < groups; g0 += static_cast<int64_t>(kBatch) * kThreads) {
    vec_t v[kBatch];
    IndexT edge[kBatch];
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      const int64_t g = g0 + static_cast<int64_t>(b) * kThreads + tid;
      v[b] = g < groups ? *reinterpret_cast<const vec_t*>(indices + g * kVec) : vec_t(0);
      // the element before a wavefront's first vector comes from memory; the other lanes get it from their neighbour
      edge[b] = (lane == 0 && g > 0 && g < groups) ? indices[g * kVec - 1] : IndexT(0);
    }
#pragma unroll
    for (int b = 0; b < kBatch; ++b) {
      const int64_t g = g0 + static_cast<int64_t>(b) * kThreads + tid;
      IndexT prev = __shfl_up(v[b][kVec - 1], 1);
      if (lane == 0) prev = edge[b];
      if (g < groups) {
        if (g > 0) heads += v[b][0] != prev ? 1u : 0u;
#pragma unroll
        for (int e = 1; e < kVec; ++e) heads += v[b][e] != v[b][e - 1] ? 1u : 0u;
      }
    }
  }
  for (int64_t i = groups * kVec + tid; i < upto; i += kThreads)
    if (i > 0) heads += indices[i] != indices[i - 1] ? 1u : 0u;
  return heads;
}

//! What RunHeadScanKernel does with u[i] = the number of run heads in (0, i]:
//!   kIds       remapped[i] = u[i]                       (ComputeCompressedGradIndices)
//!   kCompact   the same, and unique_keys[u[i]] = indices[i] at every run head (and i = 0); block_start[b] =
//!              u[first element of block b], block_start[number of blocks] = u[n - 1] + 1.
enum class RunHeadOutput { kIds, kCompact };
//! kCompact also writes every kFenceStride-th distinct key to a compact array: the coarse index that
//! BlockedRankSearchKernel searches in LDS before it touches the full lists.
constexpr unsigned kFenceStride = 256;

template <typename IndexT, RunHeadOutput kOut, bool kSelfCount = false>
__global__ void __launch_bounds__(kSortThreads)
RunHeadScanKernel(const IndexT* __restrict__ indices, const int64_t n,
                  const unsigned* __restrict__ tile_count /* run heads per tile; null: no count launch ran (few tiles) */,
                  const bool tile_count_is_prefix,
                  IndexT* __restrict__ remapped,
                  const int block_tiles,
                  IndexT* __restrict__ unique_keys, unsigned* __restrict__ block_start,
                  IndexT* __restrict__ fence_keys) {
  __shared__ unsigned wave_sum[kSortWaves];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t wave_base = static_cast<int64_t>(blockIdx.x) * kSortTile + wave * (64 * kSortItems);
  const bool starts_block = TileStartsBlock(block_tiles);
  const bool forced = starts_block && threadIdx.x == 0;
  unsigned long long heads[kSortItems];
  IndexT cur[kSortItems];
  WaveRunHeads<IndexT>(indices, wave_base, n, forced, cur, heads);
  unsigned c = 0;
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) c += static_cast<unsigned>(__popcll(heads[r]));
  // run heads in all earlier tiles: every workgroup adds up the raw per-tile counts itself (at
  // most a few thousand words from L2) -- cheaper than a separate single-workgroup scan launch
  __shared__ unsigned wave_before[kSortWaves];
  unsigned before = 0;
  if (tile_count != nullptr) {
    if (tile_count_is_prefix) {
      if (threadIdx.x == 0) before = tile_count[blockIdx.x];
    } else {
      for (int t = threadIdx.x; t < static_cast<int>(blockIdx.x); t += kSortThreads) before += tile_count[t];
    }
  } else if (blockIdx.x > 0) {
    // no count launch (few tiles): the run heads of everything before this tile are counted here.  The tile's own
    // first element is a head of THIS tile (WaveRunHeads); blocks (block_tiles > 0) never come this way.  (A variant
    // of its own: the batch of loads costs 52 registers that the scan of a large array must not pay.)
    if constexpr (kSelfCount) before = CountRunHeadsBefore<IndexT>(indices, static_cast<int64_t>(blockIdx.x) * kSortTile);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
  if (lane == 0) {
    wave_sum[wave] = c;
    wave_before[wave] = before;
  }
  __syncthreads();
  unsigned running = 0;
#pragma unroll
  for (int w = 0; w < kSortWaves; ++w) running += wave_before[w];
  for (int w = 0; w < wave; ++w) running += wave_sum[w];
  const unsigned long long upto = LanesBelow(lane) | (1ull << lane);  // lanes <= this one
#pragma unroll
  for (int r = 0; r < kSortItems; ++r) {
    const int64_t i = wave_base + r * 64 + lane;
    const unsigned u = running + static_cast<unsigned>(__popcll(heads[r] & upto));
    if (i < n) {
      remapped[i] = static_cast<IndexT>(u);
      if constexpr (kOut == RunHeadOutput::kCompact) {
        if (((heads[r] >> lane) & 1ull) != 0 || i == 0) {
          const IndexT key = cur[r];
          unique_keys[u] = key;
          if ((u & (kFenceStride - 1)) == 0) fence_keys[u / kFenceStride] = key;   // every kFenceStride-th distinct key
        }
        if (r == 0 && threadIdx.x == 0 && (starts_block || i == 0))
          block_start[block_tiles > 0 ? static_cast<int>(blockIdx.x) / block_tiles : 0] = u;
        if (i == n - 1)
          block_start[block_tiles > 0 ? (static_cast<int>(blockIdx.x) / block_tiles) + 1 : 1] = u + 1;
      }
    }
    running += static_cast<unsigned>(__popcll(heads[r]));
  }
}

//! The run-head scan of 5 .. kWideSelfCountTiles tiles (up to 65,536 elements) in ONE launch: 1024-thread workgroups,
//! one per 4096-element tile; every workgroup counts the run heads before its tile itself, four times as many lanes
//! sharing that work as in RunHeadScanKernel (measured: 4.7 us at 8 tiles and 7.1 at 16 against 8.8 for the count
//! launch + the scan launch; 9.2 at 32 tiles, hence the limit).  kIds output only.
constexpr int kWideSelfCountTiles = 16;
constexpr int kWideScanThreads = 1024;

template <typename IndexT>
__global__ void __launch_bounds__(kWideScanThreads)
RunHeadScanWideKernel(const IndexT* __restrict__ indices, const int64_t n, IndexT* __restrict__ remapped) {
  constexpr int kWaves = kWideScanThreads / 64;
  constexpr int kRounds = kSortTile / kWideScanThreads;      // 4 rounds of 64 consecutive elements per wavefront
  __shared__ unsigned wave_sum[kWaves];
  __shared__ unsigned wave_before[kWaves];
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int64_t wave_base = static_cast<int64_t>(blockIdx.x) * kSortTile + wave * (64 * kRounds);
  IndexT cur[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int64_t i = wave_base + r * 64 + lane;
    cur[r] = i < n ? indices[i] : IndexT(0);
  }
  IndexT edge = IndexT(0);   // the element before the wavefront's first one
  if (lane == 0 && wave_base > 0 && wave_base <= n) edge = indices[wave_base - 1];
  // the elements before this tile: requested now, counted below
  unsigned before = blockIdx.x > 0
                        ? CountRunHeadsBefore<IndexT, kWideScanThreads>(indices, static_cast<int64_t>(blockIdx.x) * kSortTile)
                        : 0u;
  unsigned long long heads[kRounds];
  unsigned c = 0;
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int64_t i = wave_base + r * 64 + lane;
    IndexT prev = __shfl_up(cur[r], 1);
    const IndexT last_of_previous_round = r > 0 ? __shfl(cur[r > 0 ? r - 1 : 0], 63) : edge;
    if (lane == 0) prev = last_of_previous_round;
    heads[r] = __ballot(i > 0 && i < n && cur[r] != prev);
    c += static_cast<unsigned>(__popcll(heads[r]));
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) before += __shfl_xor(before, d);
  if (lane == 0) {
    wave_sum[wave] = c;
    wave_before[wave] = before;
  }
  __syncthreads();
  unsigned running = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    running += wave_before[w];
    if (w < wave) running += wave_sum[w];
  }
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int64_t i = wave_base + r * 64 + lane;
    const unsigned u = running + CountBelow(heads[r]) + static_cast<unsigned>((heads[r] >> lane) & 1ull);
    if (i < n) remapped[i] = static_cast<IndexT>(u);
    running += static_cast<unsigned>(__popcll(heads[r]));
  }
}

inline size_t RunHeadScanWorkBytes(const size_t n) {
  const size_t tiles = (n + kSortTile - 1) / kSortTile;
  return SortAlign((tiles ? tiles : 1) * sizeof(unsigned));
}

//! The launches of one run-head scan; `tile_sum` holds one word per tile.
template <typename IndexT, RunHeadOutput kOut>
inline void RunHeadScanLaunch(const IndexT* indices, const size_t n, IndexT* remapped, unsigned* tile_sum,
                              const int block_tiles, IndexT* unique_keys, unsigned* block_start,
                              IndexT* fence_keys, hipStream_t stream) {
  if (n == 0) return;
  const int tiles = static_cast<int>((n + kSortTile - 1) / kSortTile);
  if (tiles == 1 || (tiles <= kSelfCountTiles && block_tiles == 0)) {  // one launch instead of two
    RunHeadScanKernel<IndexT, kOut, true><<<tiles, kSortThreads, 0, stream>>>(
        indices, static_cast<int64_t>(n), nullptr, false, remapped, block_tiles, unique_keys, block_start, fence_keys);
    return;
  }
  if constexpr (kOut == RunHeadOutput::kIds) {
    // still one launch, on four times the lanes (64-bit elements: half as many per 16-byte load, half the range)
    if (tiles <= kWideSelfCountTiles / static_cast<int>(sizeof(IndexT) / 4) && block_tiles == 0) {
      RunHeadScanWideKernel<IndexT><<<tiles, kWideScanThreads, 0, stream>>>(indices, static_cast<int64_t>(n), remapped);
      return;
    }
  }
  // every workgroup of the scan adds up the counts of the earlier tiles itself: tiles^2 / 2 words
  // in total -- 2 MB at 1024 tiles, but 5e11 bytes at the API's limit of 2^31 lookups; beyond
  // kSelfSumTiles a single-workgroup prefix pass (one more launch) replaces it
  const bool prefix = tiles > kSelfSumTiles;
  RunHeadCountKernel<IndexT><<<tiles, kSortThreads, 0, stream>>>(indices, static_cast<int64_t>(n), tile_sum,
                                                                  block_tiles);
  if (prefix) RunHeadTilePrefixKernel<<<1, kSortThreads, 0, stream>>>(tile_sum, tiles);
  RunHeadScanKernel<IndexT, kOut><<<tiles, kSortThreads, 0, stream>>>(
      indices, static_cast<int64_t>(n), tile_sum, prefix, remapped, block_tiles, unique_keys, block_start, fence_keys);
}

template <typename IndexT>
inline void RunHeadScan(const IndexT* indices, const size_t n, IndexT* remapped, char* work,
                        hipStream_t stream) {
  RunHeadScanLaunch<IndexT, RunHeadOutput::kIds>(indices, n, remapped, reinterpret_cast<unsigned*>(work),
                                                 /*block_tiles=*/0, nullptr, nullptr, nullptr, stream);
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_RADIX_SORT_KERNELS_HPP_
