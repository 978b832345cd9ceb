// MI355X (gfx950 / CDNA4) embedding lookup -- index transformations (host API).
//
// Drop-in for the reference's cuembed/include/index_transforms.cuh: same
// function templates, parameter order, two-phase workspace query
// (`work == nullptr` => `*lwork` receives the bytes needed, nothing runs) and
// stream-ordered asynchronous execution; `hipStream_t` replaces `cudaStream_t`.
//   ExtractRowIdsFromFixed       <-> index_transforms.cuh:45-55
//   ExtractRowIdsFromCSR         <-> index_transforms.cuh:66-74
//   ExtractRowIdsForConcat       <-> index_transforms.cuh:85-93
//   Transpose                    <-> index_transforms.cuh:224-250
//   ComputeCompressedGradIndices <-> index_transforms.cuh:278-323
//   ComputeCompressedGradIndicesBlocked: extension (the same ids for a sample-blocked order, blocked_order.hpp)
//   BagOrderByLength: extension (ForwardOptions::sample_order for ragged CSR batches)
// Every result is integer (or a permutation of the inputs) and bit-exact.
#ifndef CUEMBED_INCLUDE_INDEX_TRANSFORMS_HPP_
#define CUEMBED_INCLUDE_INDEX_TRANSFORMS_HPP_

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstring>
#include <type_traits>

#include "cuembed/include/blocked_remap_kernels.hpp"
#include "cuembed/include/cuembed_assert.hpp"
#include "cuembed/include/hint_kernels.hpp"
#include "cuembed/include/index_kernels.hpp"
#include "cuembed/include/radix_sort_kernels.hpp"

namespace cuembed {

namespace detail {
inline size_t AlignUp(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr int kIndexBlockThreads = 256;
}  // namespace detail

/*!
 * \brief row_ids[t] = t / num_hots for t in [0, batch_size * num_hots):
 * num_hots = 3 -> [0, 0, 0, 1, 1, 1, 2, 2, 2, ...]
 */
template <typename IndexT>
void ExtractRowIdsFromFixed(const int batch_size,
                            const int num_hots,
                            IndexT* row_ids,
                            const hipStream_t stream = 0) {
  const int64_t nnz = static_cast<int64_t>(batch_size) * num_hots;
  if (nnz <= 0) return;
  const int threads = detail::kIndexBlockThreads;
  const int64_t per_block = static_cast<int64_t>(threads) * detail::kSequenceItemsPerThread;
  const detail::QuotientMagic q(num_hots);
  detail::FillQuotientKernel<IndexT>
      <<<static_cast<unsigned>((nnz + per_block - 1) / per_block), threads, 0, stream>>>(
          nnz, num_hots, q.magic, q.shift, row_ids);
}

/*!
 * \brief Expand CSR offsets to one sample id per lookup:
 * offsets = [0, 2, 3, 5] -> row_ids = [0, 0, 1, 2, 2].
 * row_ids[i] = b for i in [offsets[b], offsets[b + 1]).
 */
template <typename IndexT, typename OffsetT>
void ExtractRowIdsFromCSR(const OffsetT* offsets,
                          const int batch_size,
                          IndexT* row_ids,
                          const hipStream_t stream = 0) {
  if (batch_size <= 0) return;
  const int threads = detail::kIndexBlockThreads;
  const int per_block = detail::CsrSamplesPerBlock(batch_size);
  const int blocks = (batch_size + per_block - 1) / per_block;
  detail::ExpandCsrKernel<OffsetT, IndexT>
      <<<blocks, threads, 0, stream>>>(offsets, batch_size, row_ids, per_block);
}

/*!
 * \brief row_ids = [0, 1, 2, ..., nnz - 1] (every lookup is its own output row).
 */
template <typename IndexT>
void ExtractRowIdsForConcat(const int nnz, IndexT* row_ids, const hipStream_t stream = 0) {
  if (nnz <= 0) return;
  const int threads = detail::kIndexBlockThreads;
  const int64_t per_block = static_cast<int64_t>(threads) * detail::kSequenceItemsPerThread;
  detail::FillQuotientKernel<IndexT>
      <<<static_cast<unsigned>((nnz + per_block - 1) / per_block), threads, 0, stream>>>(
          nnz, 1, 0u, 0, row_ids);
}

/**
 * @brief Table-row caching hook (extension; the reference lists an embedding cache as future work,
 * README.md:111-112, embedding_lookup_kernels.cuh:114-115) for tables that do NOT live in this
 * GPU's HBM -- pinned host memory read over PCIe, or a peer's memory -- with the most frequently
 * used rows copied into a device buffer.
 *
 * EmbeddingForward addresses a row as `params + int64(index) * embed_width`; nothing requires
 * the row to lie inside the table's own allocation.  With
 *     cache_row_offset = (cache_rows - params) / embed_width     (elements; the two buffers must
 *                                                                  differ by a whole number of rows)
 * the index `cache_row_offset + slot` reaches row `slot` of the device cache through the SAME
 * base pointer.  This helper rewrites the indices of a batch accordingly:
 *     translated[i] = slot_of_row[indices[i]] >= 0 ? cache_row_offset + slot_of_row[indices[i]]
 *                                                   : indices[i]                        (int64)
 * and EmbeddingForward<..., int64_t, ...>(params, ..., translated, ...) then reads cached rows
 * from HBM and only the others from wherever `params` lives -- unmodified kernel, bit-identical
 * results as long as the cached copies equal the table rows.  slot_of_row has one int32 per table
 * row (-1: not cached); which rows to cache (e.g. the most frequent ones of recent batches) is the
 * caller's policy -- cuembed_amd/row_cache.py has a simple one.  `num_rows` is the length of
 * slot_of_row: indices outside [0, num_rows) are copied through unchanged (never used to index it).
 */
template <typename IndexT>
void TranslateIndicesForRowCache(const IndexT* indices,
                                 const int64_t nnz,
                                 const int32_t* slot_of_row,
                                 const int64_t num_rows,
                                 const int64_t cache_row_offset,
                                 int64_t* translated,
                                 const hipStream_t stream = 0) {
  if (nnz <= 0) return;
  const int threads = detail::kIndexBlockThreads;
  const int64_t per_block = static_cast<int64_t>(threads) * detail::kSequenceItemsPerThread;
  detail::TranslateForRowCacheKernel<IndexT>
      <<<static_cast<unsigned>((nnz + per_block - 1) / per_block), threads, 0, stream>>>(
          indices, nnz, slot_of_row, num_rows, cache_row_offset, translated);
}

/**
 * @brief Reorder sample-major COO lookups into index-major order: a STABLE sort
 * of (rows[i] [, weights[i]]) by the key cols[i].
 *
 * Naming follows the reference's code, not its README (index_transforms.cuh
 * :95-137; callers pass rows = sample ids, cols = lookup indices):
 *   transpose_rows    <- cols sorted ascending               (table row ids)
 *   transpose_cols    <- rows carried along, input order kept inside a run
 *   transpose_weights <- weights carried along (only when weights != nullptr)
 *
 * Two-phase: call with work == nullptr to get *lwork, then with a buffer of at
 * least that many bytes.
 *
 * Like the reference (cub::DeviceRadixSort::SortPairs over all bits of a signed key type) this
 * is a generic COO transpose: `cols` may hold any IndexT values -- negative keys sort first -- and
 * `rows` any IndexT values (int64 rows beyond 2^32 or negative are carried at full width).
 *
 * Extensions over the reference (results identical, fewer bytes moved):
 *  - `index_bits` (default: all bits of IndexT, the reference's behaviour).  A caller that knows
 *    every lookup index lies in [0, 2^index_bits) -- e.g. index_bits = ceil(log2(num_categories))
 *    -- lets the radix sort skip the always-zero high digits: 3 passes instead of 8 for int64 ids
 *    of a 10M-row table.  Keys outside that range are a contract violation (undefined order).
 *  - `row_bits` (default 0 = unknown).  int64 `rows` known to lie in [0, 2^row_bits), row_bits <=
 *    32 (sample ids always do: they are < nnz <= INT_MAX), travel as 32 bits between the passes
 *    without the library having to look at them first.
 *  - `sample_blocks` (default 1 = the reference's result).  > 1 CHANGES the result: the input (sample-major
 *    lookups) is cut into that many consecutive blocks of equal length and each block is transposed on its
 *    own; the output is the concatenation of the sorted blocks.  For the COMPRESSED gradient path only:
 *    ComputeCompressedGradIndices and EmbeddingBackward then produce one gradient row per (block, table row)
 *    -- an uncoalesced compressed gradient: a table row looked up from several blocks appears once per
 *    block in inverse_mapping -- and while a block is being scattered every L2 gathers from 1 / sample_blocks
 *    of grad_y only (C4: EmbeddingBackward 0.258 -> 0.191 ms with 2 blocks, 572 k -> 679 k gradient rows).
 *    RecommendedSampleBlocks() picks the count.  Never use it with a dense gradient.
 *  - `transpose_remapped_indices` (default nullptr).  Not null: also receives ComputeCompressedGradIndices' output for
 *    the sorted indices (nnz entries) -- the same values, one call.  Up to 4,096 lookups the whole thing is ONE launch
 *    of one workgroup (block_sort_kernels.hpp: a dependent launch costs 3.5-5 us at these sizes, the reference's
 *    sequence is ~10 of them); beyond, the run-head scan's launches follow the sort's on the stream, sharing `work`.
 */
template <typename IndexT, typename WeightT>
void Transpose(const IndexT* rows,
               const IndexT* cols,
               const WeightT* weights,
               const int nnz,
               IndexT* transpose_rows,
               IndexT* transpose_cols,
               WeightT* transpose_weights,
               char* work,
               size_t* lwork,
               const hipStream_t stream = 0,
               const int index_bits = static_cast<int>(sizeof(IndexT) * 8),
               const int row_bits = 0,
               const int sample_blocks = 1,
               IndexT* transpose_remapped_indices = nullptr) {
  using KeyT = typename std::make_unsigned<IndexT>::type;  // bit pattern; signed order via the top digit
  const int key_bits = (index_bits > 0 && index_bits < static_cast<int>(sizeof(IndexT) * 8))
                           ? index_bits
                           : static_cast<int>(sizeof(IndexT) * 8);
  const size_t n = static_cast<size_t>(nnz > 0 ? nnz : 0);
  const KeyT* keys_in = reinterpret_cast<const KeyT*>(cols);
  KeyT* keys_out = reinterpret_cast<KeyT*>(transpose_rows);

  if (weights == nullptr) {
    // keys = lookup index, payload = sample id
    const detail::RadixSortPlan<KeyT, IndexT, detail::NoPayload> plan(n, key_bits);
    if (work == nullptr) {
      *lwork = plan.total;
      return;
    }
    assert(*lwork >= plan.total);
    detail::RadixSortPairs<KeyT, IndexT, detail::NoPayload>(
        keys_in, keys_out, rows, transpose_cols, nullptr, nullptr, n, key_bits, work, stream,
        /*signed_keys=*/true, row_bits, /*v1_div=*/0, sample_blocks, transpose_remapped_indices);
    return;
  }
  // weighted: sample id AND weight move with the key as two payload arrays (the reference
  // packs them into a struct before the sort and unpacks it afterwards,
  // index_transforms.cuh:139-200)
  const detail::RadixSortPlan<KeyT, IndexT, WeightT> plan(n, key_bits);
  if (work == nullptr) {
    *lwork = plan.total;
    return;
  }
  assert(*lwork >= plan.total);
  detail::RadixSortPairs<KeyT, IndexT, WeightT>(keys_in, keys_out, rows, transpose_cols, weights,
                                                transpose_weights, n, key_bits, work, stream,
                                                /*signed_keys=*/true, row_bits, /*v1_div=*/0, sample_blocks,
                                                transpose_remapped_indices);
}

//! Where Transpose(..., sample_blocks) cuts: lookups [k * L, (k + 1) * L) form block k, L = this value
//! (a multiple of 4096; inputs of up to 131,072 lookups are one block whatever was asked for).
inline int64_t TransposeSampleBlockLength(const int64_t nnz, const int sample_blocks) {
  return static_cast<int64_t>(detail::SortSegmentLength(static_cast<size_t>(nnz > 0 ? nnz : 0), sample_blocks));
}

/**
 * @brief Transpose() of a FIXED-HOTNESS batch without materialising the sample ids (extension):
 * the same result as ExtractRowIdsFromFixed(batch_size, num_hots, row_ids) followed by
 * Transpose(row_ids, indices, weights, batch_size * num_hots, ...), but the first radix pass
 * derives the sample id of lookup i as i / num_hots instead of reading an array that a kernel
 * would have had to write first (16.8 MB written and read back at the north-star shape, and one
 * launch).  num_hots = 1 gives the concat layout (ExtractRowIdsForConcat: row id = position).
 * Two-phase workspace query, `index_bits` and `sample_blocks` as for Transpose().
 */
template <typename IndexT, typename WeightT>
void TransposeFixedHotness(const IndexT* indices,
                           const WeightT* weights,
                           const int batch_size,
                           const int num_hots,
                           IndexT* transpose_indices,
                           IndexT* transpose_sample_ids,
                           WeightT* transpose_weights,
                           char* work,
                           size_t* lwork,
                           const hipStream_t stream = 0,
                           const int index_bits = static_cast<int>(sizeof(IndexT) * 8),
                           const int sample_blocks = 1,
                           IndexT* transpose_remapped_indices = nullptr) {
  using KeyT = typename std::make_unsigned<IndexT>::type;
  const int key_bits = (index_bits > 0 && index_bits < static_cast<int>(sizeof(IndexT) * 8))
                           ? index_bits
                           : static_cast<int>(sizeof(IndexT) * 8);
  const int64_t nnz = static_cast<int64_t>(batch_size) * num_hots;
  const size_t n = static_cast<size_t>(nnz > 0 ? nnz : 0);
  const KeyT* keys_in = reinterpret_cast<const KeyT*>(indices);
  KeyT* keys_out = reinterpret_cast<KeyT*>(transpose_indices);
  if (weights == nullptr) {
    const detail::RadixSortPlan<KeyT, IndexT, detail::NoPayload> plan(n, key_bits);
    if (work == nullptr) {
      *lwork = plan.total;
      return;
    }
    assert(*lwork >= plan.total && num_hots > 0 && nnz <= INT32_MAX);
    detail::RadixSortPairs<KeyT, IndexT, detail::NoPayload>(
        keys_in, keys_out, nullptr, transpose_sample_ids, nullptr, nullptr, n, key_bits, work, stream,
        /*signed_keys=*/true, /*v1_bits=*/31, /*v1_div=*/num_hots, sample_blocks, transpose_remapped_indices);
    return;
  }
  const detail::RadixSortPlan<KeyT, IndexT, WeightT> plan(n, key_bits);
  if (work == nullptr) {
    *lwork = plan.total;
    return;
  }
  assert(*lwork >= plan.total && num_hots > 0 && nnz <= INT32_MAX);
  detail::RadixSortPairs<KeyT, IndexT, WeightT>(keys_in, keys_out, nullptr, transpose_sample_ids, weights,
                                                transpose_weights, n, key_bits, work, stream,
                                                /*signed_keys=*/true, /*v1_bits=*/31, /*v1_div=*/num_hots, sample_blocks,
                                                transpose_remapped_indices);
}

namespace detail {
//! keys[s] = bound - min(bag length of sample s, bound): ascending keys = descending lengths
template <typename OffsetT>
__global__ void __launch_bounds__(kIndexBlockThreads)
BagLengthKeysKernel(const OffsetT* __restrict__ offsets, const int batch_size, const int bound,
                    int32_t* __restrict__ keys) {
  const int s = blockIdx.x * kIndexBlockThreads + threadIdx.x;
  if (s >= batch_size) return;
  int64_t len = static_cast<int64_t>(offsets[s + 1]) - static_cast<int64_t>(offsets[s]);
  len = len < 0 ? 0 : (len > bound ? bound : len);
  keys[s] = bound - static_cast<int32_t>(len);
}
}  // namespace detail

/**
 * @brief sample_order for ForwardOptions (extension): the samples of a CSR batch by DESCENDING bag length, ties in
 * input order -- a permutation of [0, batch_size).  A key kernel and the library's own stable sort over the keys
 * (TransposeFixedHotness with hotness 1: the payload is the position); `max_length` > 0, a bound on the bag length,
 * keeps the sort to the key bits that exist (longer bags are ranked as max_length: that costs balance, never
 * correctness); 0 = unknown; < 0 = "rank bags of 255 lookups and more alike".  With a bound of at most 255 and up to
 * 131,072 samples the order comes out of TWO small launches (hint_kernels.hpp: a stable counting sort in chunks of
 * 1,024 samples, the same permutation as the general sort gives) -- cheap enough to compute for every fresh offsets
 * array of a C3-like batch.  It only depends on the offsets.  Two-phase workspace query as for Transpose().
 */
template <typename OffsetT>
void BagOrderByLength(const OffsetT* offsets,
                      const int batch_size,
                      const int max_length,
                      int32_t* sample_order,
                      char* work,
                      size_t* lwork,
                      const hipStream_t stream = 0) {
  const int bound = max_length > 0 ? max_length : (max_length < 0 ? 255 : INT32_MAX);
  if (bound <= 255 && batch_size > 0 && batch_size <= detail::kBagOrderMaxBatch) {
    const int chunks = (batch_size + detail::kBagOrderThreads - 1) / detail::kBagOrderThreads;
    const size_t need = static_cast<size_t>(chunks) * 256 * sizeof(unsigned);     // the chunks' key counts
    if (work == nullptr) {
      *lwork = need;
      return;
    }
    assert(*lwork >= need);
    unsigned* chunk_hist = reinterpret_cast<unsigned*>(work);
    detail::BagChunkHistogramKernel<OffsetT><<<chunks, detail::kBagOrderThreads, 0, stream>>>(offsets, batch_size, bound,
                                                                                            chunk_hist);
    detail::BagOrderScatterKernel<OffsetT><<<chunks, detail::kBagOrderThreads, 0, stream>>>(offsets, batch_size, bound,
                                                                                          chunk_hist, sample_order);
    return;
  }
  int bits = 0;
  while (bits < 31 && (int64_t{1} << bits) <= bound) ++bits;   // keys lie in [0, bound]
  const int batch = batch_size > 0 ? batch_size : 0;
  const size_t keys_bytes = detail::AlignUp(static_cast<size_t>(batch) * sizeof(int32_t), 256);
  size_t sort_bytes = 0;
  TransposeFixedHotness<int32_t, float>(nullptr, nullptr, batch, 1, nullptr, nullptr, nullptr, nullptr, &sort_bytes,
                                        stream, bits);
  const size_t need = 2 * keys_bytes + sort_bytes;
  if (work == nullptr) {
    *lwork = need;
    return;
  }
  assert(*lwork >= need);
  if (batch == 0) return;
  int32_t* keys = reinterpret_cast<int32_t*>(work);
  int32_t* sorted_keys = reinterpret_cast<int32_t*>(work + keys_bytes);
  detail::BagLengthKeysKernel<OffsetT>
      <<<(batch + detail::kIndexBlockThreads - 1) / detail::kIndexBlockThreads, detail::kIndexBlockThreads, 0, stream>>>(
          offsets, batch, bound, keys);
  TransposeFixedHotness<int32_t, float>(keys, nullptr, batch, 1, sorted_keys, sample_order, nullptr,
                                        work + 2 * keys_bytes, &sort_bytes, stream, bits);
}

/**
 * @brief Map sorted lookup indices to dense ids 0..num_unique-1:
 * indices = [4, 4, 7, 8, 8, 8, 18] -> remapped_indices = [0, 0, 1, 2, 2, 2, 3].
 * (num_unique = remapped_indices[nnz - 1] + 1, read back by the caller.)
 * Run-head flags are generated on the fly and scanned over 4096-element tiles (count, scan
 * of the tile counts, scan inside the tiles); the reference runs cub adjacent-difference, a
 * memset and an inclusive scan (index_transforms.cuh:285-322).
 * Two-phase workspace query as for Transpose().
 */
template <typename IndexT>
void ComputeCompressedGradIndices(const IndexT* indices,
                                  const int nnz,
                                  IndexT* remapped_indices,
                                  char* work,
                                  size_t* lwork,
                                  const hipStream_t stream = 0) {
  const size_t n = static_cast<size_t>(nnz > 0 ? nnz : 0);
  const size_t need = detail::RunHeadScanWorkBytes(n);
  if (work == nullptr) {
    *lwork = need;
    return;
  }
  assert(*lwork >= need);
  detail::RunHeadScan<IndexT>(indices, n, remapped_indices, work, stream);
}

/**
 * @brief ComputeCompressedGradIndices for the output of Transpose(..., sample_blocks) (extension).
 *
 * `indices` holds `sample_blocks` blocks of TransposeSampleBlockLength(nnz, sample_blocks) lookups, each block
 * sorted on its own.  Two outputs that EmbeddingBackward(..., sample_blocks, block_row_ids) consumes together:
 *   remapped_indices[i] = the number of the (block, table row) pair of lookup i: ids count up through the array, a
 *                         new one wherever the index changes or a block begins (M <= nnz pairs in all);
 *   block_row_ids[pair] = the dense id the REFERENCE's fully sorted order assigns to that table row -- its rank
 *                         among all distinct rows of the batch -- with bit 30 (detail::kSharedRowBit) set when the
 *                         same row also occurs in an EARLIER block.  Room for nnz entries always suffices.
 * so that the gradient row of lookup i is block_row_ids[remapped_indices[i]] & 0x3fffffff, exactly the reference's
 * transpose_remapped_indices value for that lookup, and EmbeddingBackward produces the reference's compressed gradient
 * (num_unique ascending rows, the same inverse_mapping) while gathering grad_y block by block.
 * `num_unique` (device pointer, may be null) receives the number of distinct rows.
 * At most 8 blocks (detail::kMaxCoalescedBlocks), nnz < 2^30.  One block (sample_blocks <= 1, or an input of up
 * to 131,072 lookups): remapped_indices is exactly ComputeCompressedGradIndices' (the gradient rows themselves);
 * block_row_ids is neither written here nor read by EmbeddingBackward.
 * Two-phase workspace query as for Transpose().
 */
template <typename IndexT>
void ComputeCompressedGradIndicesBlocked(const IndexT* indices,
                                         const int nnz,
                                         const int sample_blocks,
                                         IndexT* remapped_indices,
                                         uint32_t* block_row_ids,
                                         uint32_t* num_unique,
                                         char* work,
                                         size_t* lwork,
                                         const hipStream_t stream = 0) {
  const size_t n = static_cast<size_t>(nnz > 0 ? nnz : 0);
  const size_t block_len = detail::SortSegmentLength(n, sample_blocks);
  const int blocks = n == 0 ? 1 : static_cast<int>((n + block_len - 1) / block_len);
  CUEMBED_ASSERT(blocks <= detail::kMaxCoalescedBlocks);
  CUEMBED_ASSERT(blocks == 1 || n < (size_t{1} << 30));
  const size_t need = blocks == 1 ? detail::RunHeadScanWorkBytes(n) : detail::BlockedRemapPlan<IndexT>(n, blocks).total;
  if (work == nullptr) {
    *lwork = need;
    return;
  }
  assert(*lwork >= need);
  if (n == 0) return;
  if (blocks == 1) {
    detail::RunHeadScan<IndexT>(indices, n, remapped_indices, work, stream);
    if (num_unique != nullptr)
      detail::LastIdPlusOneKernel<IndexT><<<1, 1, 0, stream>>>(remapped_indices, static_cast<int64_t>(n), num_unique);
    return;
  }
  detail::BlockedRunHeadRemap<IndexT>(indices, n, blocks, block_len, remapped_indices, block_row_ids, num_unique, work,
                                      stream);
}

}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_INDEX_TRANSFORMS_HPP_
