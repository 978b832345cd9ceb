// MI355X (gfx950 / CDNA4) embedding lookup -- EmbeddingBackward host API.
// Included by embedding_lookup.hpp; see that file for the conventions.
#ifndef CUEMBED_INCLUDE_EMBEDDING_BACKWARD_HPP_
#define CUEMBED_INCLUDE_EMBEDDING_BACKWARD_HPP_

#include "cuembed/include/embedding_lookup.hpp"

namespace cuembed {

namespace detail {

constexpr int kMaxSegmentLen = 128;
constexpr int kMinSegmentLen = 8;
constexpr int kMaxScatterStageBytes = 48 * 1024;
//! Lanes wanted in flight on the whole chip before segments are shortened:
//! 256 CUs x 2048 lanes x 0.4 (the reference's 40 % target,
//! embedding_lookup.cuh:312, :365-375, evaluated for MI355X without a device query).
constexpr int64_t kBackwardTargetLanes = static_cast<int64_t>(256) * 2048 * 4 / 10;

inline int ChooseSegmentLen(const int64_t nnz, const int lanes_per_row) {
  int len = kMaxSegmentLen;
  if (const char* env = std::getenv("CUEMBED_BWD_SEGMENT_LEN")) {  // tuning knob
    const int v = std::atoi(env);
    if (v >= kMinSegmentLen && v <= 4096) return v;
  }
  while (len > kMinSegmentLen && (nnz / len) * lanes_per_row < kBackwardTargetLanes) len /= 2;
  return len;
}

//! Column slices for the backward gather (see SegmentedScatterAddKernel 2b): slices of at
//! least 128 bytes (one L2 line), at most 4.  CUEMBED_BWD_SLICES overrides (tuning knob).
//! Only for >= 1M lookups: EmbeddingBackward is not told the batch size, and with few lookups
//! grad_y fits the L2s anyway (measured: 0.340 -> 0.290 ms at C4, but 25 -> 28 us at nnz = 262k).
inline int ChooseColumnSlices(const size_t row_bytes, const int lanes_per_row, const int64_t nnz) {
  int slices = 1;
  while (nnz >= (int64_t{1} << 20) && slices < 4 && row_bytes / (slices * 2) >= 128 &&
         lanes_per_row % (slices * 2) == 0)
    slices *= 2;
  if (const char* env = std::getenv("CUEMBED_BWD_SLICES")) {
    const int v = std::atoi(env);
    if ((v == 1 || v == 2 || v == 4 || v == 8) && lanes_per_row % v == 0) slices = v;
  }
  return slices;
}

template <typename GradT, typename IndexT, int N>
inline void LaunchScatterAdd(const GradT* grad_y, int width, const IndexT* rows,
                             const IndexT* sample_ids, const GradT* weights, int64_t nnz,
                             GradT* grad_out, RowSplit split, hipStream_t stream,
                             const int64_t zero_rows /* > 0: zero only what needs it, see below */,
                             const IndexT* run_ids, IndexT* inverse_mapping /* compressed gradient only */) {
  const int slices = ChooseColumnSlices(static_cast<size_t>(width) * sizeof(GradT), split.lanes_per_row, nnz);
  const int lanes = split.lanes_per_row / slices;  // lanes of one column slice
  int segments_per_block = lanes >= kDefaultBlockThreads ? 1 : kDefaultBlockThreads / lanes;
  int segment_len = ChooseSegmentLen(nnz, split.lanes_per_row);
  // Keep the staged COO triples of one workgroup within the LDS budget: shorten the
  // segments first (down to 32 lookups), then put fewer segments in a workgroup.
  while (ScatterStageBytes<GradT, IndexT>(segments_per_block, segment_len, lanes, N, weights != nullptr) >
         static_cast<size_t>(kMaxScatterStageBytes)) {
    if (segment_len > 32) segment_len /= 2;
    else if (segments_per_block > 1) segments_per_block /= 2;
    else if (segment_len > kMinSegmentLen) segment_len /= 2;
    else break;
  }
  const int64_t num_segments = (nnz + segment_len - 1) / segment_len;
  const int64_t nz_blocks = (num_segments + segments_per_block - 1) / segments_per_block;
  // one workgroup per (nz block, slice); with slices > 1 the 8 / slices XCDs that share a
  // slice split the nz blocks, so the grid is a whole number of rounds of 8 workgroups
  const int per_slice = slices > 1 ? 8 / slices : 1;
  const int64_t grid_blocks = slices > 1 ? (nz_blocks + per_slice - 1) / per_slice * 8 : nz_blocks;
  const dim3 block(lanes, segments_per_block, 1);
  const dim3 grid(static_cast<unsigned>(grid_blocks), 1, 1);
  if (zero_rows > 0) {
    const int64_t tail_blocks = (zero_rows + kZeroTailRowsPerBlock - 1) / kZeroTailRowsPerBlock;
    ZeroSharedAndTailRowsKernel<GradT, IndexT><<<static_cast<unsigned>(nz_blocks + tail_blocks), 256, 0, stream>>>(
        rows, nnz, segments_per_block * segment_len, nz_blocks, width, zero_rows, grad_out);
  }
  const size_t lds =
      ScatterStageBytes<GradT, IndexT>(segments_per_block, segment_len, lanes, N, weights != nullptr);
  if (weights != nullptr)
    SegmentedScatterAddKernel<GradT, IndexT, N, true><<<grid, block, lds, stream>>>(
        grad_y, width, rows, sample_ids, weights, nnz, segment_len, grad_out, slices, run_ids, inverse_mapping);
  else
    SegmentedScatterAddKernel<GradT, IndexT, N, false><<<grid, block, lds, stream>>>(
        grad_y, width, rows, sample_ids, weights, nnz, segment_len, grad_out, slices, run_ids, inverse_mapping);
}

}  // namespace detail

/**
 * @brief Embedding backward: scatter-add `grad_y` rows into the gradient of the
 * table, from index-sorted COO lookups (the output of Transpose()).  Full
 * gradient (`transpose_remapped_indices == nullptr`, `grad_embedding` has one row
 * per table row) or compressed gradient (`transpose_remapped_indices` from
 * ComputeCompressedGradIndices(), `grad_embedding` has `num_unique` rows and
 * `inverse_mapping[num_unique]` receives the table row of each).  Same contract
 * as the reference (embedding_lookup.cuh:397-483): the output must be zero
 * before the scatter; `skip_grad_init` means the caller already zeroed it.
 */
template <typename GradT, typename IndexT>
void EmbeddingBackward(const GradT* grad_y,
                       const int embed_width,
                       const int num_grad_embedding_rows,
                       const int nnz,
                       const IndexT* transpose_indices,
                       const IndexT* transpose_sample_ids,
                       const IndexT* transpose_remapped_indices,
                       const GradT* transpose_weights,
                       const bool skip_grad_init,
                       GradT* grad_embedding,
                       IndexT* inverse_mapping,
                       const hipStream_t stream = 0) {
  static_assert(std::is_same<GradT, float>::value || std::is_same<GradT, __half>::value ||
                    std::is_same<GradT, __hip_bfloat16>::value,
                "EmbeddingBackward: gradients must be float, __half or __hip_bfloat16");
  using ElemT = detail::DeviceElemT<GradT>;

  const IndexT* rows =
      transpose_remapped_indices != nullptr ? transpose_remapped_indices : transpose_indices;

  if (transpose_remapped_indices != nullptr && nnz > 0) CUEMBED_ASSERT(inverse_mapping != nullptr);
  // Zero-initialisation.  Dense gradient: rows without lookups must read zero -> memset.
  // Compressed gradient: every row is produced by the scatter itself, so only the rows that can
  // receive atomics (and an over-allocated tail) are zeroed, by a small kernel (LaunchScatterAdd).
  const bool compressed = transpose_remapped_indices != nullptr;
  if (!skip_grad_init && (!compressed || nnz <= 0)) {
    (void)hipMemsetAsync(grad_embedding, 0,
                         static_cast<size_t>(num_grad_embedding_rows) *
                             static_cast<size_t>(embed_width) * sizeof(GradT),
                         stream);
  }
  if (nnz <= 0) return;
  const int64_t zero_rows = (!skip_grad_init && compressed) ? num_grad_embedding_rows : 0;

  const IndexT* run_ids = compressed ? transpose_indices : nullptr;  // inverse mapping is written by the scatter
  const ElemT* gy = reinterpret_cast<const ElemT*>(grad_y);
  const ElemT* w = reinterpret_cast<const ElemT*>(transpose_weights);
  ElemT* out = reinterpret_cast<ElemT*>(grad_embedding);
  const detail::RowSplit split = detail::SplitRow<ElemT>(embed_width, grad_y, grad_embedding);
  constexpr int kMaxN = 16 / static_cast<int>(sizeof(ElemT));
  if (split.elems_per_lane == kMaxN)
    detail::LaunchScatterAdd<ElemT, IndexT, kMaxN>(gy, embed_width, rows, transpose_sample_ids, w,
                                                   nnz, out, split, stream, zero_rows, run_ids, inverse_mapping);
  else if (split.elems_per_lane == kMaxN / 2)
    detail::LaunchScatterAdd<ElemT, IndexT, kMaxN / 2>(gy, embed_width, rows,
                                                       transpose_sample_ids, w, nnz, out, split,
                                                       stream, zero_rows, run_ids, inverse_mapping);
  else
    detail::LaunchScatterAdd<ElemT, IndexT, kMaxN / 4>(gy, embed_width, rows,
                                                       transpose_sample_ids, w, nnz, out, split,
                                                       stream, zero_rows, run_ids, inverse_mapping);
}

}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_EMBEDDING_BACKWARD_HPP_
