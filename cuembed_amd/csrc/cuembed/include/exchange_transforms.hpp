// Device-side halves of the owner-partitioned sparse gradient exchange (extensions; the reference is single-GPU:
// README.md:108-119 lists multi-device as future work).
//
// A data-parallel step sums the ranks' COMPRESSED gradients (EmbeddingBackward's num_unique rows + inverse_mapping,
// ids ascending) without a host read-back: the row-id space is cut into one range per rank, every rank sends the rows
// of range r to rank r in a FIXED-size slot (an equal-split all-to-all), the owner merges what it got -- Transpose +
// ComputeCompressedGradIndices + EmbeddingBackward with a device-side row count, pad_to_capacity -- and the merged
// pieces, fixed-size again, are all-gathered (cuembed_amd/distributed.py: SparseGradExchange).
// The collectives are the caller's (RCCL); what is here is the index work around them, which as a chain of tensor
// operations cost more host time than the whole forward + backward step:
//
//   PackRowsByOwner   where every owner's range starts in the ascending ids (one small launch), then ids and rows
//                     into the slots of the send buffers, padding id behind each slot's rows (one copy launch);
//   FinishOwnerPiece  after the owner's merge: the device-side row count without the padding ids' run, that run's
//                     row zeroed, valid ids behind the count, the count and the overflow word in-band behind the ids.
//
// Nothing here reads anything back; counts, range starts and the overflow word stay in device memory.
#ifndef CUEMBED_INCLUDE_EXCHANGE_TRANSFORMS_HPP_
#define CUEMBED_INCLUDE_EXCHANGE_TRANSFORMS_HPP_

#include <hip/hip_runtime.h>

#include <cstdint>

#include "cuembed/include/cuembed_assert.hpp"

namespace cuembed {
namespace detail {

constexpr int kExchangeThreads = 256;
constexpr int kExchangeRowsPerThread = 4;
constexpr int kExchangeMaxWorld = 1024;

// cuts[0 .. world] ascend (cuts[0] = 0, cuts[world] = num_categories): how many of them are <= x.
__device__ __forceinline__ int CutsNotAbove(const int64_t* __restrict__ cuts, const int world, const int64_t x) {
  int lo = 0, hi = world + 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cuts[mid] <= x) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// starts[r] = number of valid ids below cuts[r], r = 0 .. world (range r of the ascending ids = [starts[r],
// starts[r + 1])).  Thread i looks at the neighbours ids[i - 1], ids[i] and writes the starts that fall between them:
// one pass over the ids instead of world + 1 dependent binary searches through global memory.
template <typename IndexT>
__global__ void __launch_bounds__(kExchangeThreads)
OwnerRangeStartsKernel(const IndexT* __restrict__ ids, const int64_t num_rows, const IndexT* __restrict__ count,
                       const int64_t input_capacity, const int64_t* __restrict__ cuts, const int world,
                       int64_t* __restrict__ starts, unsigned long long* __restrict__ flag) {
  __shared__ int64_t s_cuts[kExchangeMaxWorld + 1];
  for (int r = threadIdx.x; r <= world; r += kExchangeThreads) s_cuts[r] = cuts[r];
  __syncthreads();
  const int64_t given = count != nullptr ? static_cast<int64_t>(*count) : num_rows;
  const int64_t valid = given < 0 ? 0 : (given < num_rows ? given : num_rows);
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kExchangeThreads + threadIdx.x;
  if (i == 0 && input_capacity > 0 && given > input_capacity) atomicOr(flag, 1ull);
  if (i > valid) return;
  // (both neighbours are loaded at clamped positions, whatever i is: a load under a condition costs a branch)
  const int64_t last = num_rows > 0 ? num_rows - 1 : 0;
  int64_t below = 0, here = 0;
  if (num_rows > 0) {
    below = static_cast<int64_t>(ids[i > 0 ? i - 1 : 0]);
    here = static_cast<int64_t>(ids[i < last ? i : last]);
  }
  const int from = i == 0 ? 0 : CutsNotAbove(s_cuts, world, below);
  const int to = i == valid ? world + 1 : CutsNotAbove(s_cuts, world, here);
  for (int r = from; r < to; ++r) starts[r] = i;
}

// Slot r of the send buffers <- rows [starts[r], starts[r + 1]) of the rank's gradient, at most slot_capacity of them;
// entries behind them get the padding id (their rows are left as they are: the id drops them at the owner).
// grid = (slot blocks, world); block = (lanes per row, rows per pass); a thread moves one 16-byte (or narrower) piece of
// kExchangeRowsPerThread rows, all loads before the first store.
template <typename IndexT, typename VecT>
__global__ void __launch_bounds__(kExchangeThreads)
PackRowsByOwnerKernel(const IndexT* __restrict__ ids, const VecT* __restrict__ rows, const int vecs_per_row,
                      const int64_t* __restrict__ starts, const int64_t slot_capacity, const int64_t padding_id,
                      int64_t* __restrict__ send_ids, VecT* __restrict__ send_rows,
                      unsigned long long* __restrict__ flag) {
  const int r = static_cast<int>(blockIdx.y);
  const int64_t first = starts[r];
  const int64_t span = starts[r + 1] - first;
  const int64_t have = span < 0 ? 0 : span;
  const int rows_per_pass = static_cast<int>(blockDim.y);
  const int64_t j0 = static_cast<int64_t>(blockIdx.x) * (rows_per_pass * kExchangeRowsPerThread);
  if (blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0 && have > slot_capacity) atomicOr(flag, 1ull);
  int64_t* slot_ids = send_ids + static_cast<int64_t>(r) * slot_capacity;
  VecT* slot_rows = send_rows + static_cast<int64_t>(r) * slot_capacity * vecs_per_row;
  if (j0 >= have) {   // (the same for the whole workgroup) nothing but padding here
    if (threadIdx.x == 0) {
#pragma unroll
      for (int u = 0; u < kExchangeRowsPerThread; ++u) {
        const int64_t j = j0 + u * rows_per_pass + threadIdx.y;
        if (j < slot_capacity) slot_ids[j] = padding_id;
      }
    }
    return;
  }
  int64_t src[kExchangeRowsPerThread];
  bool real[kExchangeRowsPerThread];
#pragma unroll
  for (int u = 0; u < kExchangeRowsPerThread; ++u) {
    const int64_t j = j0 + u * rows_per_pass + threadIdx.y;
    real[u] = j < have && j < slot_capacity;
    src[u] = first + (j < have ? j : have - 1);       // (have >= 1 here) an existing row, whatever j is
  }
  if (threadIdx.x == 0) {
    IndexT id[kExchangeRowsPerThread];
#pragma unroll
    for (int u = 0; u < kExchangeRowsPerThread; ++u) id[u] = ids[src[u]];
#pragma unroll
    for (int u = 0; u < kExchangeRowsPerThread; ++u) {
      const int64_t j = j0 + u * rows_per_pass + threadIdx.y;
      if (j < slot_capacity) slot_ids[j] = real[u] ? static_cast<int64_t>(id[u]) : padding_id;
    }
  }
  for (int v = threadIdx.x; v < vecs_per_row; v += blockDim.x) {
    VecT piece[kExchangeRowsPerThread];
#pragma unroll
    for (int u = 0; u < kExchangeRowsPerThread; ++u) piece[u] = rows[src[u] * vecs_per_row + v];
#pragma unroll
    for (int u = 0; u < kExchangeRowsPerThread; ++u) {
      const int64_t j = j0 + u * rows_per_pass + threadIdx.y;
      if (real[u]) slot_rows[j * vecs_per_row + v] = piece[u];
    }
  }
}

// After the owner's merge (sorted ids of everything it received, their compressed-gradient ids, EmbeddingBackward with
// pad_to_capacity into ids[capacity + 1] / rows[capacity + 1]): see FinishOwnerPiece.
template <typename ElemT>
__global__ void __launch_bounds__(kExchangeThreads)
FinishOwnerPieceKernel(const int64_t* __restrict__ sorted_ids, const int64_t* __restrict__ remapped, const int64_t nnz,
                       const int64_t capacity, const int64_t num_categories, const int64_t pad_lo,
                       const int64_t pad_len, int64_t* __restrict__ ids, ElemT* __restrict__ rows, const int embed_width,
                       int64_t* __restrict__ tail, unsigned long long* __restrict__ flag,
                       int64_t* __restrict__ count_out) {
  const int64_t has_pad = sorted_ids[nnz - 1] >= num_categories ? 1 : 0;
  const int64_t count = remapped[nnz - 1] + 1 - has_pad;
  const bool overflow = count > capacity;
  const int64_t i = static_cast<int64_t>(blockIdx.x) * kExchangeThreads + threadIdx.x;
  if (i <= capacity) {
    // (an overflowing merge wrote nothing: count > capacity >= i keeps every id as it was)
    const int64_t kept = ids[i];
    const int64_t id = i < count ? kept : pad_lo + i % pad_len;
    ids[i] = id;
    if (tail != nullptr && i < capacity) tail[i] = id;
  }
  if (blockIdx.x == 0) {
    // the padding ids' run was summed into the row right behind the real ones: zero it (without such a run, the spare row)
    const int64_t spare = has_pad != 0 ? count : capacity;
    ElemT* zero_row = rows + (spare < capacity ? spare : capacity) * embed_width;
    for (int c = threadIdx.x; c < embed_width; c += kExchangeThreads) zero_row[c] = ElemT(0.0f);
    if (threadIdx.x == 0) {
      const unsigned long long word = *flag | (overflow ? 1ull : 0ull);
      *flag = word;
      if (tail != nullptr) {
        tail[capacity] = count < capacity ? count : capacity;
        tail[capacity + 1] = static_cast<int64_t>(word);
      }
      if (count_out != nullptr) *count_out = count;
    }
  }
}

template <typename IndexT, typename VecT>
void LaunchPackRows(const IndexT* ids, const void* rows, const int64_t row_bytes, const int64_t* starts, const int world,
                    const int64_t slot_capacity, const int64_t padding_id, int64_t* send_ids, void* send_rows,
                    unsigned long long* flag, hipStream_t stream) {
  const int vecs = static_cast<int>(row_bytes / static_cast<int64_t>(sizeof(VecT)));
  int lanes = 1;
  while (lanes < vecs && lanes < 64) lanes <<= 1;
  const int rows_per_pass = kExchangeThreads / lanes;
  const int64_t per_block = static_cast<int64_t>(rows_per_pass) * kExchangeRowsPerThread;
  const dim3 grid(static_cast<unsigned>((slot_capacity + per_block - 1) / per_block), static_cast<unsigned>(world));
  hipLaunchKernelGGL((PackRowsByOwnerKernel<IndexT, VecT>), grid, dim3(lanes, rows_per_pass), 0, stream, ids,
                     static_cast<const VecT*>(rows), vecs, starts, slot_capacity, padding_id, send_ids,
                     static_cast<VecT*>(send_rows), flag);
}

}  // namespace detail

// ids[num_rows] ascending in their first *count entries (count == nullptr: all of them), rows[num_rows, embed_width];
// cuts[world + 1] on the device (cuts[r] = first row id of owner r, cuts[world] = num_categories).
// -> send_ids[world * slot_capacity], send_rows[world * slot_capacity, embed_width]: slot r holds the ids / rows of
//    owner r's range in order, then the padding id num_categories (rows behind the ids are not written).
// range_starts[world + 1]: device scratch (afterwards: where each range starts).  *flag |= 1 when a range has more than
// slot_capacity rows (the slot then holds the first slot_capacity of them) or, with input_capacity > 0, when *count
// exceeds input_capacity (callers that cut worst-case buffers at input_capacity rows pass the cut length as num_rows).
template <typename IndexT, typename ElemT>
void PackRowsByOwner(const IndexT* ids, const ElemT* rows, const int64_t num_rows, const int embed_width,
                     const IndexT* count, const int64_t* cuts, const int world, const int64_t slot_capacity,
                     const int64_t input_capacity, const int64_t num_categories, int64_t* send_ids, ElemT* send_rows,
                     int64_t* range_starts, int64_t* flag, const hipStream_t stream = 0) {
  CUEMBED_ASSERT(num_rows >= 0 && embed_width > 0 && world >= 1 && world <= detail::kExchangeMaxWorld);
  CUEMBED_ASSERT(slot_capacity >= 1 && slot_capacity * world <= (int64_t{1} << 40));
  CUEMBED_ASSERT(cuts != nullptr && send_ids != nullptr && send_rows != nullptr && range_starts != nullptr &&
                 flag != nullptr);
  CUEMBED_ASSERT(num_rows == 0 || (ids != nullptr && rows != nullptr));
  auto* word = reinterpret_cast<unsigned long long*>(flag);
  const int64_t positions = num_rows + 1;
  const unsigned blocks = static_cast<unsigned>((positions + detail::kExchangeThreads - 1) / detail::kExchangeThreads);
  hipLaunchKernelGGL((detail::OwnerRangeStartsKernel<IndexT>), dim3(blocks), dim3(detail::kExchangeThreads), 0, stream,
                     ids, num_rows, count, input_capacity, cuts, world, range_starts, word);
  const int64_t row_bytes = static_cast<int64_t>(embed_width) * static_cast<int64_t>(sizeof(ElemT));
  const uintptr_t both = reinterpret_cast<uintptr_t>(rows) | reinterpret_cast<uintptr_t>(send_rows) |
                         static_cast<uintptr_t>(row_bytes);
  if (both % 16 == 0)
    detail::LaunchPackRows<IndexT, uint4>(ids, rows, row_bytes, range_starts, world, slot_capacity, num_categories,
                                          send_ids, send_rows, word, stream);
  else if (both % 8 == 0)
    detail::LaunchPackRows<IndexT, uint2>(ids, rows, row_bytes, range_starts, world, slot_capacity, num_categories,
                                          send_ids, send_rows, word, stream);
  else if (both % 4 == 0)
    detail::LaunchPackRows<IndexT, uint32_t>(ids, rows, row_bytes, range_starts, world, slot_capacity, num_categories,
                                             send_ids, send_rows, word, stream);
  else
    detail::LaunchPackRows<IndexT, uint16_t>(ids, rows, row_bytes, range_starts, world, slot_capacity, num_categories,
                                             send_ids, send_rows, word, stream);
}

// The owner's merge is the library's own compressed backward over what it received:
//   TransposeFixedHotness(recv_ids as nnz samples of hotness 1, index bits of num_categories + 1, remapped)
//   EmbeddingBackward(recv_rows, ..., capacity_rows = capacity + 1, pad_to_capacity) into ids / rows
// (the padding id num_categories sorts behind every real id and becomes ONE extra run at the end).  This call finishes
// the piece on the device: count = distinct real ids (*count_out, may be nullptr); the padding run's row -- or the spare
// last row -- zeroed; ids[i] for i >= count = pad_lo + i % pad_len (valid row ids, different ones in turn: one id for
// the whole tail would serialise whoever coalesces the result); and, if `tail` (capacity + 2 words) is given, the
// all-gather's id buffer: tail[0 .. capacity) = ids, tail[capacity] = min(count, capacity), tail[capacity + 1] = *flag.
// *flag |= 1 when count > capacity (the merge then wrote nothing and ids / rows hold what they held before).
template <typename ElemT>
void FinishOwnerPiece(const int64_t* sorted_ids, const int64_t* remapped_ids, const int64_t nnz, const int64_t capacity,
                      const int64_t num_categories, const int64_t pad_lo, const int64_t pad_len, int64_t* ids,
                      ElemT* rows, const int embed_width, int64_t* tail, int64_t* flag, int64_t* count_out,
                      const hipStream_t stream = 0) {
  CUEMBED_ASSERT(nnz >= 1 && capacity >= 1 && embed_width > 0 && pad_len >= 1);
  CUEMBED_ASSERT(sorted_ids != nullptr && remapped_ids != nullptr && ids != nullptr && rows != nullptr &&
                 flag != nullptr);
  const unsigned blocks =
      static_cast<unsigned>((capacity + 1 + detail::kExchangeThreads - 1) / detail::kExchangeThreads);
  hipLaunchKernelGGL((detail::FinishOwnerPieceKernel<ElemT>), dim3(blocks), dim3(detail::kExchangeThreads), 0, stream,
                     sorted_ids, remapped_ids, nnz, capacity, num_categories, pad_lo, pad_len, ids, rows, embed_width,
                     tail, reinterpret_cast<unsigned long long*>(flag), count_out);
}

}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_EXCHANGE_TRANSFORMS_HPP_
