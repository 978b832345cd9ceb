// MI355X (gfx950 / CDNA4) embedding lookup -- EmbeddingBackward host API.
// Included by embedding_lookup.hpp; see that file for the conventions.
#ifndef CUEMBED_INCLUDE_EMBEDDING_BACKWARD_HPP_
#define CUEMBED_INCLUDE_EMBEDDING_BACKWARD_HPP_

#include "cuembed/include/device_shape.hpp"
#include "cuembed/include/embedding_lookup.hpp"

namespace cuembed {

//! Launch-shape overrides of the backward kernels (tuning / tests; 0 = the built-in heuristic).
//! Process-wide like SetForwardReductionOrder; the initial values come from the environment
//! (CUEMBED_BWD_SEGMENT_LEN, CUEMBED_BWD_SLICES), read ONCE at first use.
struct BackwardTuning {
  int segment_len;    //!< lookups per nz-segment (8 .. 4096)
  int column_slices;  //!< XCD column slices of the gather: 1, 2, 4 or 8
};

namespace detail {
inline std::atomic<int>& BackwardTuningCell(const int which) {
  static std::atomic<int> cell[2] = {
      {[] { const char* e = std::getenv("CUEMBED_BWD_SEGMENT_LEN"); return e ? std::atoi(e) : 0; }()},
      {[] { const char* e = std::getenv("CUEMBED_BWD_SLICES"); return e ? std::atoi(e) : 0; }()}};
  return cell[which];
}
}  // namespace detail

inline void SetBackwardTuning(const BackwardTuning& t) {
  detail::BackwardTuningCell(0).store(t.segment_len, std::memory_order_relaxed);
  detail::BackwardTuningCell(1).store(t.column_slices, std::memory_order_relaxed);
}
inline BackwardTuning GetBackwardTuning() {
  return BackwardTuning{detail::BackwardTuningCell(0).load(std::memory_order_relaxed),
                        detail::BackwardTuningCell(1).load(std::memory_order_relaxed)};
}

namespace detail {

constexpr int kMaxSegmentLen = 128;
constexpr int kMinSegmentLen = 8;
constexpr int kMaxScatterStageBytes = 32 * 1024;   // five workgroups per CU (160 KiB of LDS)
//! Lanes wanted in flight on the device before segments are shortened: CUs x resident lanes per CU x 0.4 (the
//! reference's 40 % target, embedding_lookup.cuh:312, :365-375; 256 x 2048 x 0.4 on a full MI355X).
inline int64_t BackwardTargetLanes(const DeviceShape& dev) {
  return static_cast<int64_t>(dev.compute_units) * dev.lanes_per_cu * 4 / 10;
}

inline int ChooseSegmentLen(const int64_t nnz, const int lanes_per_row, const DeviceShape& dev) {
  int len = kMaxSegmentLen;
  const int forced = BackwardTuningCell(0).load(std::memory_order_relaxed);
  if (forced >= kMinSegmentLen && forced <= 4096) return forced & ~7;  // the walk is unrolled by 8
  while (len > kMinSegmentLen && (nnz / len) * lanes_per_row < BackwardTargetLanes(dev)) len /= 2;
  return len;
}

//! Column slices for the backward gather (see SegmentedScatterAddKernel 2b): at most one per XCD and a divisor of the
//! XCD count; a single-XCD partition never slices.  EmbeddingBackward is not told the batch size (how much of grad_y an
//! L2 fronts), so the rule goes by the number of lookups:
//!   >= 2^20 lookups: slices of at least 128 bytes (one L2 line) -- 4 for 512-byte rows, 8 for rows of 1 KiB and more
//!                    (measured at the C4 index set: 0.340 -> 0.290 ms; fp32 W = 256 0.579 -> 0.540 ms against 4 slices);
//!   >= 2^17 lookups: slices of at least 256 bytes -- 2 for 512-byte rows, 4 for 1 KiB.  Measured over 60 shapes of
//!                    131 k - 786 k lookups (profiles/r05_backward_mid_size_slices.txt): B = 16,384 x H = 16 26.5 ->
//!                    19.1 us, 32,768 x 16 44.8 -> 36.2, 16,384 x 32 (1 KiB rows) 92.7 -> 52.1; small batches with long
//!                    bags (grad_y fits the L2s anyway) -3 ... +3 %; the one shape that loses is hotness 1 with a
//!                    quarter of a million samples (1 KiB rows, uniform indices: +7 %).  128-byte slices there: up to
//!                    +13 % on small batches, hence the wider slices;
//!   fewer: no slices (the launch floor).
inline int ChooseColumnSlices(const size_t row_bytes, const int lanes_per_row, const int64_t nnz,
                              const DeviceShape& dev) {
  int slices = 1;
  const size_t min_slice_bytes = nnz >= (int64_t{1} << 20) ? 128 : 256;
  while (nnz >= (int64_t{1} << 17) && slices * 2 <= dev.xcds && dev.xcds % (slices * 2) == 0 &&
         row_bytes / (slices * 2) >= min_slice_bytes && lanes_per_row % (slices * 2) == 0)
    slices *= 2;
  const int v = BackwardTuningCell(1).load(std::memory_order_relaxed);
  if (v >= 1 && v <= dev.xcds && dev.xcds % v == 0 && lanes_per_row % v == 0) slices = v;
  return slices;
}

//! Launch shape of SegmentedScatterAddKernel (host arithmetic only).
struct ScatterShape {
  int slices, lanes, segments_per_block, segment_len, xcds;
  int64_t nz_blocks, grid_blocks;
  size_t lds;
  //! Workgroups that walk `count` consecutive lookups (one launch), and the grid that holds them.
  int64_t NzBlocks(const int64_t count) const {
    const int64_t num_segments = (count + segment_len - 1) / segment_len;
    return (num_segments + segments_per_block - 1) / segments_per_block;
  }
  int64_t GridBlocks(const int64_t blocks) const {
    // one workgroup per (nz block, slice); with slices > 1 the xcds / slices XCDs that share a
    // slice split the nz blocks, so the grid is a whole number of rounds of `xcds` workgroups
    const int per_slice = slices > 1 ? xcds / slices : 1;
    return slices > 1 ? (blocks + per_slice - 1) / per_slice * xcds : blocks;
  }
};

template <typename GradT, typename IndexT, int N>
inline ScatterShape PlanScatter(const int width, const int64_t nnz, const RowSplit split, const bool weighted,
                                const DeviceShape& dev) {
  ScatterShape s;
  s.xcds = dev.xcds;
  s.slices = ChooseColumnSlices(static_cast<size_t>(width) * sizeof(GradT), split.lanes_per_row, nnz, dev);
  s.lanes = split.lanes_per_row / s.slices;  // lanes of one column slice
  s.segments_per_block = s.lanes >= kDefaultBlockThreads ? 1 : kDefaultBlockThreads / s.lanes;
  s.segment_len = ChooseSegmentLen(nnz, split.lanes_per_row, dev);
  // Keep the staged COO triples of one workgroup within the LDS budget: shorten the
  // segments first (down to 32 lookups), then put fewer segments in a workgroup.
  while (ScatterStageBytes<GradT, IndexT>(s.segments_per_block, s.segment_len, s.lanes, N, weighted) >
         static_cast<size_t>(kMaxScatterStageBytes)) {
    if (s.segment_len > 32) s.segment_len /= 2;
    else if (s.segments_per_block > 1) s.segments_per_block /= 2;
    else if (s.segment_len > kMinSegmentLen) s.segment_len /= 2;
    else break;
  }
  s.segment_len = s.segment_len < 8 ? 8 : s.segment_len & ~7;
  s.nz_blocks = s.NzBlocks(nnz);
  s.grid_blocks = s.GridBlocks(s.nz_blocks);
  s.lds = ScatterStageBytes<GradT, IndexT>(s.segments_per_block, s.segment_len, s.lanes, N, weighted);
  return s;
}

//! `sample_block_len` > 0: the COO is a sample-blocked order (blocked_order.hpp) -- consecutive blocks of that many
//! lookups, each sorted on its own, row ids from ComputeCompressedGradIndicesBlocked -- and every block gets its own
//! stream-ordered launch (a run whose row an earlier block stored is added to it).  0: one launch over everything.
template <typename GradT, typename IndexT, int N>
inline void LaunchScatterAdd(const GradT* grad_y, int width, const IndexT* rows,
                             const IndexT* sample_ids, const GradT* weights, int64_t nnz,
                             GradT* grad_out, RowSplit split, hipStream_t stream,
                             const bool zero_shared /* compressed gradient: zero only what needs it, see below */,
                             const int64_t zero_rows /* ... and the rows from the last id up to here (<= 0: none) */,
                             const IndexT* run_ids, IndexT* inverse_mapping /* compressed gradient only */,
                             const int64_t sample_block_len = 0, const uint32_t* block_row_ids = nullptr,
                             const int64_t capacity_rows = 0, uint32_t* capacity_overflow = nullptr,
                             const bool pad_to_capacity = false) {
  const DeviceShape dev = CurrentDeviceShape();
  const ScatterShape s = PlanScatter<GradT, IndexT, N>(width, nnz, split, weights != nullptr, dev);
  const dim3 block(s.lanes, s.segments_per_block, 1);
  const int block_len = s.segments_per_block * s.segment_len;
  const bool blocked = sample_block_len > 0 && sample_block_len < nnz;
  const int64_t launch_len = blocked ? sample_block_len : nnz;
  const int launches = static_cast<int>((nnz + launch_len - 1) / launch_len);
  const uint32_t* pair_rows = blocked ? block_row_ids : nullptr;
  if (zero_shared) {
    const int64_t tail_blocks = ZeroTailBlocks(zero_rows, dev);
    const int64_t per_launch = s.NzBlocks(launch_len);
    ZeroSharedAndTailRowsKernel<GradT, IndexT>
        <<<static_cast<unsigned>(per_launch * launches + tail_blocks), 256, 0, stream>>>(
            rows, nnz, block_len, per_launch, launch_len, launches, width, zero_rows, grad_out, pair_rows,
            capacity_rows, pad_to_capacity);
  }
  int seg_shift = -1;
  if ((s.segment_len & (s.segment_len - 1)) == 0)
    for (seg_shift = 0; (1 << seg_shift) < s.segment_len; ++seg_shift) {}
  for (int p = 0; p < launches; ++p) {
    const int64_t first = p * launch_len;
    const int64_t count = first + launch_len < nnz ? launch_len : nnz - first;
    const dim3 grid(static_cast<unsigned>(s.GridBlocks(s.NzBlocks(count))), 1, 1);
    const IndexT* run_ids_p = run_ids != nullptr ? run_ids + first : nullptr;
#define CUEMBED_LAUNCH_SCATTER(W, BLK, WIN)                                                                       \
  SegmentedScatterAddKernel<GradT, IndexT, N, W, BLK, WIN><<<grid, block, s.lds, stream>>>(                         \
      grad_y, width, rows + first, sample_ids + first, (W) ? weights + first : weights, count, s.segment_len,        \
      seg_shift, grad_out, s.slices, s.xcds, run_ids_p, inverse_mapping, pair_rows, capacity_rows, capacity_overflow)
    const bool adds_to_rows = blocked && p > 0;   // (the first block finds nothing stored yet: plain kernel)
    // window depth: column-sliced launches gather mostly from L2 (short window, more wavefronts); unsliced ones miss
    const bool short_window = s.slices > 1 || adds_to_rows;
    if (weights != nullptr) {
      if (adds_to_rows) CUEMBED_LAUNCH_SCATTER(true, true, kBackwardWindowHits);
      else if (short_window) CUEMBED_LAUNCH_SCATTER(true, false, kBackwardWindowHits);
      else CUEMBED_LAUNCH_SCATTER(true, false, kBackwardWindowMisses);
    } else {
      if (adds_to_rows) CUEMBED_LAUNCH_SCATTER(false, true, kBackwardWindowHits);
      else if (short_window) CUEMBED_LAUNCH_SCATTER(false, false, kBackwardWindowHits);
      else CUEMBED_LAUNCH_SCATTER(false, false, kBackwardWindowMisses);
    }
#undef CUEMBED_LAUNCH_SCATTER
  }
  if (pad_to_capacity && zero_rows > 0) {   // the zeroed rows past the count get their row ids (see NamePaddedRowsKernel)
    const int64_t blocks = (zero_rows + 255) / 256;
    NamePaddedRowsKernel<IndexT><<<static_cast<unsigned>(blocks < 1024 ? blocks : 1024), 256, 0, stream>>>(
        rows, nnz, inverse_mapping, zero_rows);
  }
}

}  // namespace detail

/**
 * @brief How many sample blocks Transpose(..., sample_blocks) should cut a batch into so that, while one
 * block is being scattered, the part of grad_y an L2 gathers from fits that L2 (extension; see Transpose()).
 * grad_y is `batch_size` rows of `embed_width` GradT; EmbeddingBackward gives every XCD a column slice of
 * the row (ChooseColumnSlices: >= 128 bytes from 2^20 lookups up), so an L2 fronts
 * batch_size x slice bytes -- 8.4 MB at C4 for 4 MiB of L2.  Returns 1 when nothing is to be gained.
 */
template <typename GradT>
inline int RecommendedSampleBlocks(const int embed_width, const int batch_size, const int64_t nnz,
                                   const detail::DeviceShape& dev) {
  const size_t row_bytes = static_cast<size_t>(embed_width > 0 ? embed_width : 0) * sizeof(GradT);
  if (row_bytes == 0 || row_bytes % 4 != 0 || batch_size <= 0) return 1;   // nothing EmbeddingBackward would slice
  const int lane_bytes = row_bytes % 16 == 0 ? 16 : (row_bytes % 8 == 0 ? 8 : 4);   // as SplitRow for aligned buffers
  if (nnz < (int64_t{1} << 20)) return 1;   // small problems: grad_y fits the L2s anyway
  const int slices = detail::ChooseColumnSlices(row_bytes, static_cast<int>(row_bytes / lane_bytes), nnz, dev);
  if (slices <= 1 && dev.xcds > 1) return 1;   // rows too narrow to slice: not measured, nothing recommended
  const size_t per_l2 = static_cast<size_t>(batch_size) * (row_bytes / slices);
  const size_t budget = dev.l2_bytes_per_xcd;   // one XCD's L2 (4 MiB on MI355X)
  size_t blocks = (per_l2 + budget - 1) / budget;
  if (blocks < 1) blocks = 1;
  if (blocks > 64) blocks = 64;   // what one Transpose call sorts separately (detail::kMaxSortSegments)
  return static_cast<int>(blocks);
}
//! ... for the current device.
template <typename GradT>
inline int RecommendedSampleBlocks(const int embed_width, const int batch_size, const int64_t nnz) {
  return RecommendedSampleBlocks<GradT>(embed_width, batch_size, nnz, detail::CurrentDeviceShape());
}

/**
 * @brief Embedding backward: scatter-add `grad_y` rows into the gradient of the
 * table, from index-sorted COO lookups (the output of Transpose()).  Full
 * gradient (`transpose_remapped_indices == nullptr`, `grad_embedding` has one row
 * per table row) or compressed gradient (`transpose_remapped_indices` from
 * ComputeCompressedGradIndices(), `grad_embedding` has `num_unique` rows and
 * `inverse_mapping[num_unique]` receives the table row of each).  Same contract
 * as the reference (embedding_lookup.cuh:397-483): the output must be zero
 * before the scatter; `skip_grad_init` means the caller already zeroed it.
 *
 * Extension: a compressed call may pass `num_grad_embedding_rows < 0` = "the number of unique rows
 * is only known on the device" (it is transpose_remapped_indices[nnz - 1] + 1).  `grad_embedding`
 * and `inverse_mapping` must then hold at least that many rows (nnz always suffices); rows past
 * the last id are left untouched.  This takes the host read-back of num_unique out of a training
 * step (the reference's benchmark reads it back between Transpose and EmbeddingBackward,
 * manual_benchmark.cu:392-394).  `capacity_rows` > 0 (compressed gradient only) states how many rows `grad_embedding`
 * and `inverse_mapping` really hold: if the device-side row count exceeds it, NOTHING is written (per launch: a
 * sample-blocked call may have written its earlier blocks) and `*capacity_overflow` (a device word the caller
 * zeroed once; may be null) is OR-ed with 1 -- a flag to read back whenever convenient instead of a silent overrun.
 * 0 = unchecked, the reference's contract.  `pad_to_capacity` (with capacity_rows > 0, a device-side count and
 * skip_grad_init = false): the rows from the count up to the capacity are ZEROED and their inverse_mapping entries set
 * to rows of the batch (different ones, in turn), so that (inverse_mapping, grad_embedding) over all capacity_rows entries is a
 * valid uncoalesced COO gradient -- coalescing it gives the reference's -- that a caller can hand on without ever
 * reading the count back (the torch op does, for small batches).
 *
 * Extension: `sample_blocks` > 1 (compressed gradient only) says that the COO comes from
 * Transpose(..., sample_blocks) and transpose_remapped_indices + block_row_ids from
 * ComputeCompressedGradIndicesBlocked(..., sample_blocks) (pair numbers and the pair -> gradient row table).
 * The blocks are scattered one after the other (stream-ordered launches), so that every L2
 * gathers from 1 / sample_blocks of grad_y at a time; the result has the REFERENCE's layout -- num_unique
 * ascending rows, the same inverse_mapping as the fully sorted order gives -- with the sum of a table row
 * taken block by block (fp32 partial sums per block, one GradT rounding per block: within the bound stated in
 * include/cuembed_amd.h; exact on exactly representable data).  C4: 0.257 -> 0.232 ms.
 * Precondition of this order: nnz < 2^30 and sample ids < 2^30 (bit 30 of a staged sample id carries a flag).
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
                       const hipStream_t stream = 0,
                       const int sample_blocks = 1,
                       const uint32_t* block_row_ids = nullptr,
                       const int capacity_rows = 0,
                       uint32_t* capacity_overflow = nullptr,
                       const bool pad_to_capacity = false) {
  static_assert(std::is_same<GradT, float>::value || std::is_same<GradT, __half>::value ||
                    std::is_same<GradT, __hip_bfloat16>::value,
                "EmbeddingBackward: gradients must be float, __half or __hip_bfloat16");
  using ElemT = detail::DeviceElemT<GradT>;
  const detail::RowSplit split = detail::SplitRow<ElemT>(embed_width, grad_y, grad_embedding);
  const IndexT* rows =
      transpose_remapped_indices != nullptr ? transpose_remapped_indices : transpose_indices;

  if (transpose_remapped_indices != nullptr && nnz > 0) CUEMBED_ASSERT(inverse_mapping != nullptr);
  if (transpose_remapped_indices == nullptr) CUEMBED_ASSERT(num_grad_embedding_rows >= 0);  // "unknown" is a compressed-only extension
  if (capacity_rows > 0) CUEMBED_ASSERT(transpose_remapped_indices != nullptr);            // ... and so is its capacity
  if (pad_to_capacity) CUEMBED_ASSERT(capacity_rows > 0 && num_grad_embedding_rows < 0 && !skip_grad_init && sample_blocks <= 1);
  // Zero-initialisation.  Dense gradient: rows without lookups must read zero -> memset.
  // Compressed gradient: every row is produced by the scatter itself, so only the rows that can
  // receive atomics (and an over-allocated tail) are zeroed, by a small kernel (LaunchScatterAdd).
  const bool compressed = transpose_remapped_indices != nullptr;
  if (!skip_grad_init && (!compressed || nnz <= 0) && num_grad_embedding_rows > 0) {
    (void)hipMemsetAsync(grad_embedding, 0,
                         static_cast<size_t>(num_grad_embedding_rows) *
                             static_cast<size_t>(embed_width) * sizeof(GradT),
                         stream);
  }
  if (nnz <= 0) return;
  const bool zero_shared = !skip_grad_init && compressed;
  // (padded: every row from the device-side count up to the capacity is zeroed and named, see below)
  const int64_t zero_rows = pad_to_capacity ? capacity_rows : num_grad_embedding_rows;
  // sample-blocked order: one launch per block (the length Transpose cut the input at)
  int64_t sample_block_len = 0;
  if (sample_blocks > 1) {
    CUEMBED_ASSERT(compressed);   // a dense gradient has nothing to gain and the ids carry no flags
    sample_block_len = static_cast<int64_t>(detail::SortSegmentLength(static_cast<size_t>(nnz), sample_blocks));
    CUEMBED_ASSERT((nnz + sample_block_len - 1) / sample_block_len <= detail::kMaxCoalescedBlocks);
    if (sample_block_len < nnz) {
      CUEMBED_ASSERT(block_row_ids != nullptr);
      CUEMBED_ASSERT(nnz < (1 << 30));   // bit 30 of a staged sample id is a flag in this order (sample ids < 2^30)
    }
  }

  const IndexT* run_ids = compressed ? transpose_indices : nullptr;  // inverse mapping is written by the scatter
  const ElemT* gy = reinterpret_cast<const ElemT*>(grad_y);
  const ElemT* w = reinterpret_cast<const ElemT*>(transpose_weights);
  ElemT* out = reinterpret_cast<ElemT*>(grad_embedding);
  constexpr int kMaxN = 16 / static_cast<int>(sizeof(ElemT));
  if (split.elems_per_lane == kMaxN)
    detail::LaunchScatterAdd<ElemT, IndexT, kMaxN>(gy, embed_width, rows, transpose_sample_ids, w, nnz, out, split,
                                                   stream, zero_shared, zero_rows, run_ids, inverse_mapping,
                                                   sample_block_len, block_row_ids, capacity_rows, capacity_overflow,
                                                   pad_to_capacity);
  else if (split.elems_per_lane == kMaxN / 2)
    detail::LaunchScatterAdd<ElemT, IndexT, kMaxN / 2>(gy, embed_width, rows, transpose_sample_ids, w, nnz, out,
                                                       split, stream, zero_shared, zero_rows, run_ids, inverse_mapping,
                                                       sample_block_len, block_row_ids, capacity_rows,
                                                       capacity_overflow, pad_to_capacity);
  else
    detail::LaunchScatterAdd<ElemT, IndexT, kMaxN / 4>(gy, embed_width, rows, transpose_sample_ids, w, nnz, out,
                                                       split, stream, zero_shared, zero_rows, run_ids, inverse_mapping,
                                                       sample_block_len, block_row_ids, capacity_rows,
                                                       capacity_overflow, pad_to_capacity);
}

/**
 * @brief EmbeddingBackward in the REFERENCE's arithmetic (extension, opt-in; for verification and for callers that need
 * the reference's bits): same arguments and outputs as EmbeddingBackward, but every `grad += grad_y * weight` is done
 * in GradT -- product and running sum rounded to GradT at every lookup, in nz order -- exactly like the CPU reference
 * (utils/include/embedding_lookup_cpu.hpp:131-143).  Bit-identical to it for ANY data: fp16 / bf16 gradients that
 * are not exactly representable, runs of any length.  (EmbeddingBackward keeps fp32 partial sums and rounds once per
 * flush: identical on exactly representable data, closer to the true sum otherwise.)  A rounding chain cannot be cut
 * into partial sums, so one run is ONE chain of dependent additions (the hottest row of the C4 batch: 65,528 of them);
 * short runs are walked by one lane group each, runs of more than 256 lookups by a whole workgroup whose gather groups
 * stage the rows in LDS while one wavefront runs the chain out of it.  C4: 1.2 ms in fp16 (4.5 x the default path's 0.26 ms
 * on the same data; 26.4 ms before the long-run path), 1.8 ms in fp32 (3.4 x).  skip_grad_init = true ADDS to what grad_embedding holds, like the
 * reference's loop on a buffer the caller did not zero.
 */
template <typename GradT, typename IndexT>
void EmbeddingBackwardReferenceSums(const GradT* grad_y,
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
  using ElemT = detail::DeviceElemT<GradT>;
  const detail::RowSplit split = detail::SplitRow<ElemT>(embed_width, grad_y, grad_embedding);
  const bool compressed = transpose_remapped_indices != nullptr;
  if (compressed && nnz > 0) CUEMBED_ASSERT(inverse_mapping != nullptr);
  CUEMBED_ASSERT(num_grad_embedding_rows >= 0);
  // rows without lookups must read zero; rows with lookups are stored whole by the group that walks their run
  if (!skip_grad_init && num_grad_embedding_rows > 0)
    (void)hipMemsetAsync(grad_embedding, 0,
                         static_cast<size_t>(num_grad_embedding_rows) * static_cast<size_t>(embed_width) * sizeof(GradT), stream);
  if (nnz <= 0) return;
  const IndexT* rows = compressed ? transpose_remapped_indices : transpose_indices;
  const IndexT* run_ids = compressed ? transpose_indices : nullptr;
  const ElemT* gy = reinterpret_cast<const ElemT*>(grad_y);
  const ElemT* w = reinterpret_cast<const ElemT*>(transpose_weights);
  ElemT* out = reinterpret_cast<ElemT*>(grad_embedding);
  const int lanes = split.lanes_per_row;
  const int groups = lanes >= detail::kReferenceBlockThreads ? 1 : detail::kReferenceBlockThreads / lanes;
  const dim3 block(lanes, groups, 1);
  const int64_t spans = (static_cast<int64_t>(nnz) + detail::kReferenceSpan - 1) / detail::kReferenceSpan;
  const dim3 plain_grid(static_cast<unsigned>((spans + groups - 1) / groups), 1, 1);
  constexpr int kMaxN = 16 / static_cast<int>(sizeof(ElemT));
  // runs of 256 lookups and more are walked by the whole workgroup out of LDS (see the kernel); 0: shape without that path
  const size_t row_bytes = static_cast<size_t>(embed_width) * sizeof(ElemT);
  const int chunk = detail::ReferenceLongRun(row_bytes, lanes, groups).chunk_rows;
  const size_t lds = chunk > 0 ? 2 * static_cast<size_t>(chunk) * (row_bytes + sizeof(ElemT)) : 0;
  const dim3 grid(chunk > 0 ? 2 * plain_grid.x : plain_grid.x, 1, 1);      // (long-run half first, then the short runs)
#define CUEMBED_LAUNCH_REFERENCE(NN)                                                                               \
  do {                                                                                                             \
    if (w != nullptr)                                                                                              \
      detail::ReferenceSumsScatterKernel<ElemT, IndexT, NN, true><<<grid, block, lds, stream>>>(                    \
          gy, embed_width, rows, transpose_sample_ids, w, nnz, out, skip_grad_init, run_ids, inverse_mapping, chunk); \
    else                                                                                                           \
      detail::ReferenceSumsScatterKernel<ElemT, IndexT, NN, false><<<grid, block, lds, stream>>>(                   \
          gy, embed_width, rows, transpose_sample_ids, w, nnz, out, skip_grad_init, run_ids, inverse_mapping, chunk); \
  } while (0)
  if (split.elems_per_lane == kMaxN) CUEMBED_LAUNCH_REFERENCE(kMaxN);
  else if (split.elems_per_lane == kMaxN / 2) CUEMBED_LAUNCH_REFERENCE(kMaxN / 2);
  else CUEMBED_LAUNCH_REFERENCE(kMaxN / 4);
#undef CUEMBED_LAUNCH_REFERENCE
}

}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_EMBEDDING_BACKWARD_HPP_
