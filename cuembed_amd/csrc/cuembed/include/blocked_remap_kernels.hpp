// MI355X (gfx950 / CDNA4): dense gradient-row ids for a SAMPLE-BLOCKED sorted COO (blocked_order.hpp).
//
// Input: `indices`, the concatenation of P blocks, each sorted ascending on its own (Transpose with
// sample_blocks = P).  Output: remapped[i] = the number of the (block, table row) pair of lookup i -- what
// ComputeCompressedGradIndices gives when every block starts a new id -- and, per pair, row_ids[pair] = the id
// ComputeCompressedGradIndices gives that table row in the reference's FULLY sorted order (its rank among all
// distinct values of the array), plus kSharedRowBit when the row also occurs in an earlier block.  The
// scatter-add kernel translates its staged lookups through row_ids (M <= nnz entries; 680 k pairs at C4 for
// 4.19 M lookups: L2-resident).  No sort, no merge, no second pass over the lookups:
//   1. run-head count + scan with compaction (RunHeadScanKernel<kCompact>): remapped[], and the distinct keys of
//      every block, one after the other, in `unique_keys`, block b's at [block_start[b], block_start[b + 1]);
//   2. BlockedRankSearchKernel: every entry finds its lower bound in each OTHER block's list (binary search in an
//      L2-resident array) and whether an EARLIER block holds the key ("first" = no);
//   3. BlockedRankFinishKernel: rank = sum over the blocks of (first-flagged entries below the lower bound): the
//      prefix counts come from the ballot words of step 2 (word mask + word prefix inside a 1024-entry tile + the
//      tile counts, which every workgroup scans for itself in LDS).
// Everything is integer and deterministic; nothing is read back by the host.
#ifndef CUEMBED_INCLUDE_BLOCKED_REMAP_KERNELS_HPP_
#define CUEMBED_INCLUDE_BLOCKED_REMAP_KERNELS_HPP_

#include "cuembed/include/radix_sort_kernels.hpp"

namespace cuembed {
namespace detail {

constexpr int kRankThreads = 256;
constexpr int kRankItems = 4;                             // entries per thread (independent searches in flight)
constexpr int kRankTile = kRankThreads * kRankItems;      // 1024 entries per workgroup = 16 ballot words
constexpr int kRankWords = kRankTile / 64;
constexpr int kRankSelfScanTiles = 4096;                  // tile counts a finishing workgroup scans itself (LDS)

//! Workspace of ComputeCompressedGradIndicesBlocked for n lookups in `blocks` blocks (byte offsets).
template <typename IndexT>
struct BlockedRemapPlan {
  size_t tile_sum, unique_keys, fence_keys, block_start, lower_bounds, masks, word_prefix, tile_count, num_unique, total;
  size_t cap;       //!< entries the per-entry arrays hold (>= the number of distinct (block, key) pairs)
  int rank_tiles;   //!< grid of the two rank kernels (most workgroups leave at once: M is only known on the device)
  BlockedRemapPlan(const size_t n, const int blocks) {
    cap = (n + kRankTile - 1) / kRankTile * kRankTile;
    rank_tiles = static_cast<int>(cap / kRankTile);
    size_t off = 0;
    tile_sum = off;
    off += RunHeadScanWorkBytes(n);
    unique_keys = off;
    off += SortAlign((cap + 64) * sizeof(IndexT));
    fence_keys = off;
    off += SortAlign((cap / kFenceStride + 2) * sizeof(IndexT));
    block_start = off;
    off += SortAlign((kMaxCoalescedBlocks + 1) * sizeof(unsigned));
    lower_bounds = off;
    off += SortAlign(static_cast<size_t>(blocks > 1 ? blocks - 1 : 1) * cap * sizeof(unsigned));
    masks = off;
    off += SortAlign((cap / 64 + 1) * sizeof(unsigned long long));
    word_prefix = off;
    off += SortAlign((cap / 64 + 1) * sizeof(unsigned));
    tile_count = off;
    off += SortAlign((static_cast<size_t>(rank_tiles) + 1) * sizeof(unsigned));
    num_unique = off;
    off += SortAlign(sizeof(unsigned));
    total = off;
  }
};

//! Step 2.  Entry e (tile-local word w = r * 4 + wave, lane l: e = tile * 1024 + w * 64 + l, so that the 64 lanes
//! of a wavefront hold 64 consecutive entries and one ballot is one word of the first-flag bitmap) belongs to block
//! b = the last block with block_start[b] <= e.  For every other block b' it finds lb = the number of keys of b'
//! below its own (kRankItems independent binary searches per thread in flight; the lists are L2-resident) and stores
//! it at lower_bounds[(b' < b ? b' : b' - 1) * cap + e].
//! The search is two-level: every kFenceStride-th key of the other block's list (`fence_keys`, written by the
//! compaction) is loaded into LDS once per workgroup and searched there; only the last log2(kFenceStride) = 8 steps
//! read the full list (19 dependent L2 reads per entry before: 17.7 -> see docs/EXPERIMENTS.md).  Lists with more
//! than kMaxLdsFences fences (> 1 M distinct keys per block) skip the LDS level.
constexpr int kMaxLdsFences = 4096;

template <typename IndexT>
__global__ void __launch_bounds__(kRankThreads)
BlockedRankSearchKernel(const IndexT* __restrict__ unique_keys, const IndexT* __restrict__ fence_keys,
                        const unsigned* __restrict__ block_start,
                        const int blocks, const size_t cap, unsigned* __restrict__ lower_bounds,
                        unsigned long long* __restrict__ masks, unsigned* __restrict__ word_prefix,
                        unsigned* __restrict__ tile_count) {
  __shared__ unsigned s_start[kMaxCoalescedBlocks + 1];
  __shared__ unsigned s_word[kRankWords];
  __shared__ IndexT s_fence[kMaxLdsFences];
  if (static_cast<int>(threadIdx.x) <= blocks) s_start[threadIdx.x] = block_start[threadIdx.x];
  __syncthreads();
  const unsigned total = s_start[blocks];                       // M: distinct (block, key) pairs
  const unsigned tile_base = blockIdx.x * kRankTile;
  if (tile_base >= total) return;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  unsigned e[kRankItems];
  IndexT key[kRankItems];
  int mine[kRankItems];
  bool valid[kRankItems], first[kRankItems];
#pragma unroll
  for (int r = 0; r < kRankItems; ++r) {
    e[r] = tile_base + static_cast<unsigned>((r * kSortWaves + wave) * 64 + lane);
    valid[r] = e[r] < total;
    key[r] = valid[r] ? unique_keys[e[r]] : IndexT(0);
    first[r] = true;
    mine[r] = 0;
    for (int q = 1; q < blocks; ++q)
      if (e[r] >= s_start[q]) mine[r] = q;
  }
  for (int other = 0; other < blocks; ++other) {
    const unsigned base = s_start[other];
    const unsigned len = s_start[other + 1] - base;
    const IndexT* list = unique_keys + base;
    unsigned lo[kRankItems], hi[kRankItems];
#pragma unroll
    for (int r = 0; r < kRankItems; ++r) {
      lo[r] = 0;
      hi[r] = (valid[r] && mine[r] != other) ? len : 0u;
    }
    // ---- level 1: the fences of this list that are whole multiples of kFenceStride inside it, in LDS.
    // Fence f holds the key at entry f * kFenceStride of the concatenated lists: f_first .. f_last lie in this list.
    const unsigned f_first = (base + kFenceStride - 1) / kFenceStride;
    const unsigned f_end = len == 0 ? f_first : (base + len - 1) / kFenceStride + 1;   // one past the last fence
    const unsigned fences = f_end > f_first ? f_end - f_first : 0u;
    const bool two_level = fences > 0 && fences <= static_cast<unsigned>(kMaxLdsFences);
    __syncthreads();                                    // (the previous list's fences are no longer read)
    if (two_level) {
      for (unsigned f = threadIdx.x; f < fences; f += kRankThreads) s_fence[f] = fence_keys[f_first + f];
      __syncthreads();
      // number of fences whose key is below mine (the kRankItems searches of a thread in lockstep: their LDS reads
      // overlap); the lower bound then lies in the stride that ends at that fence
      unsigned fa[kRankItems], fb[kRankItems];
#pragma unroll
      for (int r = 0; r < kRankItems; ++r) {
        fa[r] = 0;
        fb[r] = lo[r] < hi[r] ? fences : 0u;
      }
      const int fence_steps = 32 - __clz(static_cast<int>(fences));
      for (int q = 0; q < fence_steps; ++q) {
        unsigned fm[kRankItems];
        IndexT fv[kRankItems];
#pragma unroll
        for (int r = 0; r < kRankItems; ++r) {
          fm[r] = (fa[r] + fb[r]) >> 1;
          fv[r] = s_fence[fa[r] < fb[r] ? fm[r] : 0u];
        }
#pragma unroll
        for (int r = 0; r < kRankItems; ++r) {
          if (fa[r] < fb[r]) {
            if (fv[r] < key[r]) fa[r] = fm[r] + 1;
            else fb[r] = fm[r];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < kRankItems; ++r) {
        if (lo[r] < hi[r]) {
          // fences [0, a) are below the key, fence a (if any) is not: entry positions relative to `base`
          const unsigned a = fa[r];
          lo[r] = a == 0 ? 0u : (f_first + a - 1) * kFenceStride - base + 1;   // all entries up to fence a-1 are below
          hi[r] = a == fences ? len : (f_first + a) * kFenceStride - base;      // fence a itself is >= key
        }
      }
    }
    // lower bound among `len` keys, or -- after level 1 -- inside one stride of at most kFenceStride keys
    const int steps = len == 0 ? 0 : 32 - __clz(static_cast<int>(two_level && len > kFenceStride ? kFenceStride : len));
    for (int s = 0; s < steps; ++s) {
      unsigned mid[kRankItems];
      IndexT v[kRankItems];
#pragma unroll
      for (int r = 0; r < kRankItems; ++r) {
        mid[r] = (lo[r] + hi[r]) >> 1;
        v[r] = list[lo[r] < hi[r] ? mid[r] : 0u];               // all requests of a step go out together
      }
#pragma unroll
      for (int r = 0; r < kRankItems; ++r) {
        if (lo[r] < hi[r]) {
          if (v[r] < key[r]) lo[r] = mid[r] + 1;
          else hi[r] = mid[r];
        }
      }
    }
    IndexT at[kRankItems];   // the key at the lower bound (all four requested together)
#pragma unroll
    for (int r = 0; r < kRankItems; ++r) at[r] = list[(valid[r] && other < mine[r] && lo[r] < len) ? lo[r] : 0u];
#pragma unroll
    for (int r = 0; r < kRankItems; ++r) {
      if (valid[r] && mine[r] != other) {
        lower_bounds[static_cast<size_t>(other < mine[r] ? other : other - 1) * cap + e[r]] = lo[r];
        if (other < mine[r] && lo[r] < len && at[r] == key[r]) first[r] = false;
      }
    }
  }
  // the first-flag bitmap of the tile: one word per (r, wave), its popcount, and the words' exclusive prefix
#pragma unroll
  for (int r = 0; r < kRankItems; ++r) {
    const unsigned long long m = __ballot(valid[r] && first[r]);
    if (lane == 0) {
      masks[tile_base / 64 + r * kSortWaves + wave] = m;
      s_word[r * kSortWaves + wave] = static_cast<unsigned>(__popcll(m));
    }
  }
  __syncthreads();
  if (threadIdx.x < kRankWords) {
    const unsigned c = s_word[threadIdx.x];
    unsigned incl = c;
#pragma unroll
    for (int d = 1; d < kRankWords; d <<= 1) {
      const unsigned up = __shfl_up(incl, d, kRankWords);
      if (static_cast<int>(threadIdx.x) >= d) incl += up;
    }
    word_prefix[tile_base / 64 + threadIdx.x] = incl - c;
    if (threadIdx.x == kRankWords - 1) tile_count[blockIdx.x] = incl;
  }
}

//! More rank tiles than a finishing workgroup scans itself: tile_count[t] becomes the exclusive prefix over the
//! ACTIVE tiles (the others never wrote their word), tile_count[num_tiles] the total.  One workgroup.
__global__ void __launch_bounds__(kSortThreads)
BlockedRankTilePrefixKernel(unsigned* __restrict__ tile_count, const unsigned* __restrict__ block_start,
                            const int blocks, const int num_tiles) {
  const unsigned total_entries = block_start[blocks];
  const int active = static_cast<int>((total_entries + kRankTile - 1) / kRankTile);
  unsigned carry = 0;
  for (int base = 0; base < active; base += kSortThreads) {
    const int t = base + threadIdx.x;
    const unsigned v = t < active ? tile_count[t] : 0u;
    unsigned total;
    const unsigned excl = BlockExclusiveScan(v, &total);
    if (t < active) tile_count[t] = carry + excl;
    carry += total;
  }
  if (threadIdx.x == 0) tile_count[num_tiles] = carry;
}

//! Step 3.  table[e] = (number of distinct keys of the whole array below key(e)) | (kSharedRowBit unless e is the
//! first occurrence of its key); *num_unique = the number of distinct keys.
__global__ void __launch_bounds__(kRankThreads)
BlockedRankFinishKernel(const unsigned* __restrict__ block_start, const int blocks, const size_t cap,
                        const unsigned* __restrict__ lower_bounds, const unsigned long long* __restrict__ masks,
                        const unsigned* __restrict__ word_prefix, const unsigned* __restrict__ tile_count,
                        const bool tile_count_is_prefix, const int num_tiles, unsigned* __restrict__ table,
                        unsigned* __restrict__ num_unique, unsigned* __restrict__ num_unique_user) {
  __shared__ unsigned s_start[kMaxCoalescedBlocks + 1];
  __shared__ unsigned s_below_start[kMaxCoalescedBlocks + 1];   // first-flagged entries below block_start[b]
  __shared__ unsigned s_tile_prefix[kRankSelfScanTiles + 1];
  __shared__ unsigned s_total;
  if (static_cast<int>(threadIdx.x) <= blocks) s_start[threadIdx.x] = block_start[threadIdx.x];
  __syncthreads();
  const unsigned total = s_start[blocks];
  const unsigned tile_base = blockIdx.x * kRankTile;
  if (tile_base >= total) return;
  const int active = static_cast<int>((total + kRankTile - 1) / kRankTile);
  if (!tile_count_is_prefix) {
    unsigned carry = 0;
    for (int base = 0; base < active; base += kRankThreads) {
      const int t = base + threadIdx.x;
      const unsigned v = t < active ? tile_count[t] : 0u;
      unsigned sum;
      const unsigned excl = BlockExclusiveScan(v, &sum);
      if (t < active) s_tile_prefix[t] = carry + excl;
      carry += sum;
    }
    if (threadIdx.x == 0) s_total = carry;
  } else if (threadIdx.x == 0) {
    s_total = tile_count[num_tiles];
  }
  __syncthreads();
  const unsigned all_first = s_total;
  // first-flagged entries among [0, x)
  auto below = [&](const unsigned x) -> unsigned {
    if (x >= total) return all_first;
    const unsigned w = x >> 6;
    const unsigned t = x / kRankTile;
    const unsigned tp = tile_count_is_prefix ? tile_count[t] : s_tile_prefix[t];
    return tp + word_prefix[w] + static_cast<unsigned>(__popcll(masks[w] & ((1ull << (x & 63)) - 1ull)));
  };
  if (static_cast<int>(threadIdx.x) <= blocks) s_below_start[threadIdx.x] = below(s_start[threadIdx.x]);
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  // the kRankItems entries of a thread side by side: their loads (lower bound, then prefix word and mask) overlap
  unsigned e[kRankItems], rank[kRankItems];
  int mine[kRankItems];
#pragma unroll
  for (int r = 0; r < kRankItems; ++r) {
    e[r] = tile_base + static_cast<unsigned>((r * kSortWaves + wave) * 64 + lane);
    rank[r] = 0;
    mine[r] = 0;
    for (int q = 1; q < blocks; ++q)
      if (e[r] >= s_start[q]) mine[r] = q;
  }
  for (int other = 0; other < blocks; ++other) {
    unsigned rel[kRankItems];
#pragma unroll
    for (int r = 0; r < kRankItems; ++r) {
      rel[r] = 0;
      if (e[r] < total)
        rel[r] = other == mine[r] ? e[r] - s_start[other]
                                  : lower_bounds[static_cast<size_t>(other < mine[r] ? other : other - 1) * cap + e[r]];
    }
    unsigned upto[kRankItems];
#pragma unroll
    for (int r = 0; r < kRankItems; ++r) upto[r] = e[r] < total ? below(s_start[other] + rel[r]) : 0u;
#pragma unroll
    for (int r = 0; r < kRankItems; ++r) rank[r] += upto[r] - s_below_start[other];
  }
#pragma unroll
  for (int r = 0; r < kRankItems; ++r) {
    if (e[r] >= total) continue;
    const bool first = ((masks[e[r] >> 6] >> (e[r] & 63)) & 1ull) != 0;
    table[e[r]] = rank[r] | (first ? 0u : kSharedRowBit);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    *num_unique = all_first;
    if (num_unique_user != nullptr) *num_unique_user = all_first;
  }
}

//! num_unique of a fully sorted array = remapped[n - 1] + 1 (the one-block case of the blocked call).
template <typename IndexT>
__global__ void LastIdPlusOneKernel(const IndexT* __restrict__ remapped, const int64_t n, unsigned* __restrict__ out) {
  *out = static_cast<unsigned>(remapped[n - 1]) + 1u;
}

template <typename IndexT>
inline void BlockedRunHeadRemap(const IndexT* indices, const size_t n, const int blocks, const size_t block_len,
                                IndexT* remapped, unsigned* row_ids, unsigned* num_unique_user, char* work,
                                hipStream_t stream) {
  if (n == 0) return;
  const BlockedRemapPlan<IndexT> plan(n, blocks);
  unsigned* tile_sum = reinterpret_cast<unsigned*>(work + plan.tile_sum);
  IndexT* unique_keys = reinterpret_cast<IndexT*>(work + plan.unique_keys);
  unsigned* block_start = reinterpret_cast<unsigned*>(work + plan.block_start);
  unsigned* lower_bounds = reinterpret_cast<unsigned*>(work + plan.lower_bounds);
  unsigned long long* masks = reinterpret_cast<unsigned long long*>(work + plan.masks);
  unsigned* word_prefix = reinterpret_cast<unsigned*>(work + plan.word_prefix);
  unsigned* tile_count = reinterpret_cast<unsigned*>(work + plan.tile_count);
  unsigned* num_unique = reinterpret_cast<unsigned*>(work + plan.num_unique);
  const int block_tiles = static_cast<int>(block_len / kSortTile);
  IndexT* fence_keys = reinterpret_cast<IndexT*>(work + plan.fence_keys);
  RunHeadScanLaunch<IndexT, RunHeadOutput::kCompact>(indices, n, remapped, tile_sum, block_tiles, unique_keys,
                                                     block_start, fence_keys, stream);
  BlockedRankSearchKernel<IndexT><<<plan.rank_tiles, kRankThreads, 0, stream>>>(
      unique_keys, fence_keys, block_start, blocks, plan.cap, lower_bounds, masks, word_prefix, tile_count);
  const bool prefix = plan.rank_tiles > kRankSelfScanTiles;
  if (prefix)
    BlockedRankTilePrefixKernel<<<1, kSortThreads, 0, stream>>>(tile_count, block_start, blocks, plan.rank_tiles);
  BlockedRankFinishKernel<<<plan.rank_tiles, kRankThreads, 0, stream>>>(
      block_start, blocks, plan.cap, lower_bounds, masks, word_prefix, tile_count, prefix, plan.rank_tiles, row_ids,
      num_unique, num_unique_user);
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_BLOCKED_REMAP_KERNELS_HPP_
