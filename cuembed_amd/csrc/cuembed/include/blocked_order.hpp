// The "sample-blocked" COO order (extension; shared by Transpose, ComputeCompressedGradIndicesBlocked
// and EmbeddingBackward).
//
// Transpose(..., sample_blocks = P) cuts the sample-major input into consecutive blocks of
// SampleBlockLength() lookups and sorts every block on its own.  While EmbeddingBackward scatters a
// block, each L2 gathers from 1 / P of grad_y only (C4: 8.4 MB -> 4.2 MB per 4 MiB L2, the cliff of
// docs/EXPERIMENTS.md).  A table row that is looked up from several blocks has one run per block.  Two
// ways to turn that into a compressed gradient:
//   * uncoalesced (round 3): ComputeCompressedGradIndices over the blocked array -> one gradient row
//     per (block, table row);
//   * coalesced (this file's constants): ComputeCompressedGradIndicesBlocked numbers the (block, table
//     row) pairs -- that is the remapped array -- and gives every pair, in a table of its own, the id the
//     REFERENCE's fully sorted order would give its table row (the rank of the row among all distinct rows
//     of the batch), plus kSharedRowBit when the same table row already occurred in an EARLIER block.
//     EmbeddingBackward(..., sample_blocks = P, block_row_ids) then scatters block after block
//     (stream-ordered launches; the staging translates pair -> row): a run without the bit ends in a plain
//     store as always, a run with it is ADDED to what the earlier blocks stored (read-modify-write inside a
//     workgroup, float atomics across workgroups).  The result has the reference's layout: num_unique
//     ascending rows, the same inverse_mapping.
#ifndef CUEMBED_INCLUDE_BLOCKED_ORDER_HPP_
#define CUEMBED_INCLUDE_BLOCKED_ORDER_HPP_

#include <cstddef>
#include <cstdint>

namespace cuembed {
namespace detail {

constexpr int kSortTile = 4096;        // keys per workgroup of the radix sort and of the run-head scan
constexpr int kFoldScanTiles = 32;     // up to 131072 keys the tile scan is done inside the scatter kernel
constexpr int kMaxSortSegments = 64;   // input blocks that can be sorted on their own in one call
//! Blocks that the COALESCED compressed gradient supports (ComputeCompressedGradIndicesBlocked keeps one
//! lower bound per (unique row of a block, other block)).
constexpr int kMaxCoalescedBlocks = 8;
//! Bit of a remapped id from ComputeCompressedGradIndicesBlocked: "this table row also occurs in an earlier block".
constexpr uint32_t kSharedRowBit = 0x40000000u;

//! Elements per block when n elements are sorted in `blocks` blocks (RadixSortPairs): a whole number of tiles,
//! ceil(tiles / blocks) of them; the last block takes what is left.  Inputs of up to kFoldScanTiles tiles are
//! always ONE block (the result is then the full sort).
inline size_t SortSegmentLength(const size_t n, const int blocks) {
  const size_t tiles = n == 0 ? 1 : (n + kSortTile - 1) / kSortTile;
  size_t want = blocks < 1 ? 1 : (blocks > kMaxSortSegments ? kMaxSortSegments : blocks);
  if (tiles <= static_cast<size_t>(kFoldScanTiles)) want = 1;
  return (tiles + want - 1) / want * kSortTile;
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_BLOCKED_ORDER_HPP_
