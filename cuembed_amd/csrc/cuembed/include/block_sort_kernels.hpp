// MI355X (gfx950 / CDNA4) -- the WHOLE index transposition of a small batch in ONE launch of ONE workgroup.
//
// Up to kBlockSortMax (4,096) lookups: row ids (fixed hotness: i / num_hots, never materialised), the stable LSD radix
// sort of (index, sample id[, weight]) and the run-head remap of ComputeCompressedGradIndices -- everything the
// reference does with ~10 cub launches (index_transforms.cuh:95-137, :278-323) -- without a single intermediate array.
// Why one workgroup: at these sizes a dependent launch is not bound by its work but by its cold start, ~3.5-5 us each
// on MI355X (profiles/r05_small_sort_baseline.txt: the 12-14 launches of the tiled path are 55-65 us for 16 k pairs
// whatever the kernels do), while 1024 threads of one CU rank 4 k keys per pass in about 2 us.
//   * 1024 threads = 16 wavefronts, four per SIMD; wavefront w owns `chunk` = 64 * ceil(n / 1024) CONSECUTIVE positions,
//     64 per round, so the number of ranking rounds follows n (1,024 pairs: ONE round per wavefront and pass; the
//     256-thread kernel this replaces gave all 1,024 keys to wavefront 0: 19.7 us).
//   * keys and payloads stay in registers over all passes; a pass ranks them (ballot match, see sort_common.hpp) and
//     permutes them through LDS -- a 32-bit key and a 32-bit payload as ONE 64-bit element (one conflict-ridden
//     scattered LDS write instead of two) -- with four barriers per pass (the digit counters are double-buffered).
//   * why not more than 4 keys per lane: with 16 (16,384 pairs, built and measured) a pass takes 13 us -- 16 k scattered
//     LDS writes per array on ONE compute unit -- and the whole sort 42 us, against 32 us for the chained kernels of
//     radix_sort_kernels.hpp on 16 compute units (profiles/r05_small_sort_*.txt); at 8 k the two meet.
//   * digits that no two keys differ in are skipped (block-local OR / AND), so int64 keys below 2^24 cost 3 passes.
//   * after the last pass every wavefront holds a sorted run of consecutive positions: run heads are one neighbour
//     compare + ballot, the remapped ids a popcount prefix -- written with the sorted arrays, no extra launch.
#ifndef CUEMBED_INCLUDE_BLOCK_SORT_KERNELS_HPP_
#define CUEMBED_INCLUDE_BLOCK_SORT_KERNELS_HPP_

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <type_traits>

#include "cuembed/include/sort_common.hpp"

namespace cuembed {
namespace detail {

constexpr int kBlockSortThreads = 1024;
constexpr int kBlockSortWaves = kBlockSortThreads / 64;
constexpr int kBlockSortMaxItems = 4;
constexpr int kBlockSortMax = kBlockSortThreads * kBlockSortMaxItems;   // 4,096 pairs

//! A 32-bit key and a 32-bit first payload go through LDS as one 64-bit element.
template <typename A, typename B>
constexpr bool BlockSortPacksKeyAndPayload() {
  return !std::is_empty<B>::value && sizeof(A) == 4 && sizeof(B) == 4;
}

template <typename A, typename B, typename C>
constexpr size_t BlockSortStageElem() {
  size_t e = sizeof(A);
  if (!std::is_empty<B>::value && sizeof(B) > e) e = sizeof(B);
  if (!std::is_empty<C>::value && sizeof(C) > e) e = sizeof(C);
  if (BlockSortPacksKeyAndPayload<A, B>() && e < 8) e = 8;
  return e;
}

//! One array through LDS: the element of rank slot[r] goes to tile position slot[r]; afterwards round r of the lane holds
//! the element at the lane's own position again (padding elements included: they rank last and stay last).
//! `first` = nothing has read the staging area since the last barrier.
template <typename T, int ITEMS>
__device__ __forceinline__ void BlockSortPermute(unsigned char* stage_raw, T (&item)[ITEMS], const unsigned (&slot)[ITEMS],
                                                 const int rounds, const int first_pos, const bool first) {
  T* stage = reinterpret_cast<T*>(stage_raw);
  if (!first) __syncthreads();   // the previous array has been read back
#pragma unroll
  for (int r = 0; r < ITEMS; ++r)
    if (r < rounds) stage[slot[r]] = item[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r)
    if (r < rounds) item[r] = stage[first_pos + r * 64];
}

//! keys_out / v1_out / v2_out = the n (key, v1[, v2]) sorted stably by the low 8 * passes bits of the key (`sign_pass`:
//! the pass whose digit carries the sign bit of two's-complement keys, or -1).  mode.v1_div > 0: v1 of element i is
//! i / v1_div, v1_in is not read.  remapped != nullptr: remapped[i] = number of positions k in (0, i] of the SORTED keys
//! with key[k] != key[k - 1].  One workgroup of kBlockSortThreads threads, n <= kBlockSortThreads * ITEMS.
template <typename KeyT, typename V1, typename V2, int ITEMS>
__global__ void __launch_bounds__(kBlockSortThreads)
BlockSortKernel(const KeyT* __restrict__ keys_in, KeyT* __restrict__ keys_out,
                const V1* __restrict__ v1_in, V1* __restrict__ v1_out,
                const V2* __restrict__ v2_in, V2* __restrict__ v2_out, const int n, const int passes,
                const int sign_pass, const SortMode mode,
                typename std::make_signed<KeyT>::type* __restrict__ remapped) {
  constexpr bool kHasV1 = !std::is_same<V1, NoPayload>::value;
  constexpr bool kHasV2 = !std::is_same<V2, NoPayload>::value;
  constexpr int kWaves = kBlockSortWaves;
  constexpr int kGroups = 4;                       // the cross-wave prefix is taken in 4 groups of 4 wavefronts
  constexpr int kGroupWaves = kWaves / kGroups;
  __shared__ __attribute__((aligned(16))) unsigned char stage[kBlockSortThreads * ITEMS * BlockSortStageElem<KeyT, V1, V2>()];
  // per wavefront and digit; becomes the exclusive prefix over wavefronts.  Two copies: while a pass works on one, the
  // other is zeroed for the next pass (no barrier of its own)
  __shared__ unsigned wave_count_buffers[2][kWaves][kSortBins];
  __shared__ __attribute__((aligned(16))) unsigned group_total[kGroups][kSortBins];
  __shared__ unsigned tile_start[kSortBins];
  __shared__ unsigned long long wave_bits[kWaves][2];
  __shared__ KeyT wave_last_key[kWaves];
  __shared__ unsigned wave_heads[kWaves];
  const int tid = threadIdx.x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  const int rounds = (n + kBlockSortThreads - 1) / kBlockSortThreads;   // <= ITEMS
  const int first_pos = wave * rounds * 64 + lane;                      // the lane's position in round 0
  // Positions >= n (the last wavefronts' tail) hold PADDING: the key that sorts last -- every digit 0xff as the passes
  // see it.  A stable sort leaves padding behind every real element whatever the real keys are, so no pass needs a
  // per-lane "valid" mask (16 rounds of them would not fit the scalar registers); only the final stores look at n.
  const KeyT padding = sign_pass >= 0 ? static_cast<KeyT>(~KeyT(0) ^ (KeyT(0x80) << (8 * sign_pass))) : static_cast<KeyT>(~KeyT(0));
  KeyT key[ITEMS];
  V1 item1[ITEMS];
  V2 item2[ITEMS];
  unsigned long long any = 0ull, all = ~0ull;
  // (loads are unconditional on a clamped position and the padding is selected in afterwards: a predicated load per
  // element makes the compiler merge the whole register array at every branch, which at 16 elements per lane spills)
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int i = first_pos + r * 64;
    const bool real = r < rounds && i < n;
    const int at = real ? i : 0;
    const KeyT k = keys_in[at];
    key[r] = real ? k : padding;
    if constexpr (kHasV1) {
      if (mode.v1_div > 0) {
        item1[r] = static_cast<V1>(ImplicitPayload(mode, at));
      } else {
        item1[r] = v1_in[at];
      }
    }
    if constexpr (kHasV2) item2[r] = v2_in[at];
    any |= real ? static_cast<unsigned long long>(k) : 0ull;
    all &= real ? static_cast<unsigned long long>(k) : ~0ull;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    any |= __shfl_xor(any, d);
    all &= __shfl_xor(all, d);
  }
  if (lane == 0) {
    wave_bits[wave][0] = any;
    wave_bits[wave][1] = all;
  }
  constexpr int kCountWordsPerThread = kWaves * kSortBins / kBlockSortThreads;
#pragma unroll
  for (int q = 0; q < kCountWordsPerThread; ++q) (&wave_count_buffers[0][0][0])[q * kBlockSortThreads + tid] = 0;
  __syncthreads();
  any = 0ull;
  all = ~0ull;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) {
    any |= wave_bits[w][0];
    all &= wave_bits[w][1];
  }
  const unsigned long long varying = any & ~all;
  int buffer = 0;
  for (int pass = 0; pass < passes; ++pass) {
    const int shift = 8 * pass;
    if (((varying >> shift) & 0xffull) == 0) continue;   // every key has the same digit here
    const unsigned flip = pass == sign_pass ? 0x80u : 0u;
    unsigned (*wave_count)[kSortBins] = wave_count_buffers[buffer];   // zeroed during the previous working pass / above
    // ---- rank inside the wavefront: 64 consecutive keys per round ----
    unsigned slot[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      slot[r] = 0u;
      if (r < rounds) {                                  // (wave-uniform)
        const unsigned digit = static_cast<unsigned>((key[r] >> shift) & 0xff) ^ flip;
        const unsigned long long peers = MatchDigit(digit, true);
        const unsigned lower = CountBelow(peers);
        // every peer reads the wavefront's running count of its digit, then the lowest peer bumps it (a wavefront's LDS
        // operations execute in program order)
        const unsigned start = wave_count[wave][digit];
        __builtin_amdgcn_wave_barrier();
        if (lower == 0) wave_count[wave][digit] = start + static_cast<unsigned>(__popcll(peers));
        __builtin_amdgcn_wave_barrier();
        slot[r] = start + lower;
      }
    }
    __syncthreads();
    // ---- per digit: counts -> exclusive prefix over the 16 wavefronts, in 4 groups of 4 ----
    const int bin = tid & (kSortBins - 1);
    const int group = tid >> 8;
    unsigned c[kGroupWaves];
    unsigned sum = 0;
#pragma unroll
    for (int k = 0; k < kGroupWaves; ++k) {
      c[k] = wave_count[group * kGroupWaves + k][bin];
      sum += c[k];
    }
    group_total[group][bin] = sum;
    // the other copy of the counters was last read before the previous pass's permutation barriers: zero it now
#pragma unroll
    for (int q = 0; q < kCountWordsPerThread; ++q)
      (&wave_count_buffers[buffer ^ 1][0][0])[q * kBlockSortThreads + tid] = 0;
    __syncthreads();
    {
      unsigned before = 0;
#pragma unroll
      for (int g = 0; g < kGroups; ++g)
        if (g < group) before += group_total[g][bin];
#pragma unroll
      for (int k = 0; k < kGroupWaves; ++k) {
        wave_count[group * kGroupWaves + k][bin] = before;
        before += c[k];
      }
    }
    if (wave == 0) {   // digit totals -> tile-local starts: lane l owns digits 4 l .. 4 l + 3
      typedef unsigned __attribute__((ext_vector_type(4))) word4_t;
      word4_t t = word4_t{0u, 0u, 0u, 0u};
#pragma unroll
      for (int g = 0; g < kGroups; ++g) t += *reinterpret_cast<const word4_t*>(&group_total[g][4 * lane]);
      const unsigned mine = t.x + t.y + t.z + t.w;
      unsigned incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
      }
      const unsigned excl = incl - mine;
      tile_start[4 * lane + 0] = excl;
      tile_start[4 * lane + 1] = excl + t.x;
      tile_start[4 * lane + 2] = excl + t.x + t.y;
      tile_start[4 * lane + 3] = excl + t.x + t.y + t.z;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      if (r < rounds) {
        const unsigned digit = static_cast<unsigned>((key[r] >> shift) & 0xff) ^ flip;
        slot[r] += tile_start[digit] + wave_count[wave][digit];
      }
    }
    // ---- every array through LDS into the new order ----
    // (the staging area was last read in the previous pass; the barriers above separate that from these writes)
    if constexpr (BlockSortPacksKeyAndPayload<KeyT, V1>()) {
      unsigned long long both[ITEMS];
#pragma unroll
      for (int r = 0; r < ITEMS; ++r)
        both[r] = (static_cast<unsigned long long>(__builtin_bit_cast(unsigned, item1[r])) << 32) | key[r];
      BlockSortPermute<unsigned long long, ITEMS>(stage, both, slot, rounds, first_pos, /*first=*/true);
#pragma unroll
      for (int r = 0; r < ITEMS; ++r) {
        key[r] = static_cast<KeyT>(both[r]);
        item1[r] = __builtin_bit_cast(V1, static_cast<unsigned>(both[r] >> 32));
      }
    } else {
      BlockSortPermute<KeyT, ITEMS>(stage, key, slot, rounds, first_pos, /*first=*/true);
      if constexpr (kHasV1) BlockSortPermute<V1, ITEMS>(stage, item1, slot, rounds, first_pos, false);
    }
    if constexpr (kHasV2) BlockSortPermute<V2, ITEMS>(stage, item2, slot, rounds, first_pos, false);
    buffer ^= 1;
  }
  // (the same "position < n" masks as at the top, but the compiler must not keep 16 of them alive across the passes:
  // that is 32 scalar registers the ballots of the ranking need -- so the bound is laundered and compared afresh)
  int n_late = n;
  asm volatile("" : "+s"(n_late));
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const int i = first_pos + r * 64;
    if (r < rounds && i < n_late) {
      keys_out[i] = key[r];
      if constexpr (kHasV1) v1_out[i] = item1[r];
      if constexpr (kHasV2) v2_out[i] = item2[r];
    }
  }
  if (remapped == nullptr) return;
  // ---- run heads of the sorted keys -> dense ids (ComputeCompressedGradIndices) ----
#pragma unroll
  for (int r = 0; r < ITEMS; ++r)
    if (r == rounds - 1 && lane == 63) wave_last_key[wave] = key[r];   // (valid whenever the next wavefront has a key)
  __syncthreads();
  const KeyT edge = wave > 0 ? wave_last_key[wave - 1] : KeyT(0);
  // (the ballots are taken twice -- once to count, once to number -- rather than kept: 16 of them are 32 SGPRs)
  auto heads_of = [&](const int r) {
    const int i = first_pos + r * 64;
    KeyT prev = __shfl_up(key[r], 1);
    const KeyT last_of_previous_round = r > 0 ? __shfl(key[r > 0 ? r - 1 : 0], 63) : edge;
    if (lane == 0) prev = last_of_previous_round;
    return __ballot(i > 0 && i < n_late && key[r] != prev);
  };
  unsigned count = 0;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r)
    if (r < rounds) count += static_cast<unsigned>(__popcll(heads_of(r)));
  if (lane == 0) wave_heads[wave] = count;
  __syncthreads();
  unsigned running = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w)
    if (w < wave) running += wave_heads[w];
  using RemapT = typename std::make_signed<KeyT>::type;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    if (r < rounds) {
      const int i = first_pos + r * 64;
      const unsigned long long heads = heads_of(r);
      const unsigned at_or_below = CountBelow(heads) + static_cast<unsigned>((heads >> lane) & 1ull);
      if (i < n_late) remapped[i] = static_cast<RemapT>(running + at_or_below);
      running += static_cast<unsigned>(__popcll(heads));
    }
  }
}

//! The launch (n <= kBlockSortMax).
template <typename KeyT, typename V1, typename V2>
inline void BlockSortLaunch(const KeyT* keys_in, KeyT* keys_out, const V1* v1_in, V1* v1_out, const V2* v2_in,
                            V2* v2_out, const int n, const int passes, const int sign_pass, const SortMode& mode,
                            typename std::make_signed<KeyT>::type* remapped, hipStream_t stream) {
  BlockSortKernel<KeyT, V1, V2, kBlockSortMaxItems><<<1, kBlockSortThreads, 0, stream>>>(
      keys_in, keys_out, v1_in, v1_out, v2_in, v2_out, n, passes, sign_pass, mode, remapped);
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_BLOCK_SORT_KERNELS_HPP_
