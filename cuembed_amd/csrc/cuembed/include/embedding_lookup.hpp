// MI355X (gfx950 / CDNA4) embedding lookup -- header-only host API.
//
// Drop-in for the reference's cuembed/include/embedding_lookup.cuh: the same
// function templates in the same namespace with the same parameter order and
// the same abort-on-misuse contract; `hipStream_t` replaces `cudaStream_t`.
//   EmbeddingForward  <-> embedding_lookup.cuh:245-308
//   EmbeddingBackward <-> embedding_lookup.cuh:423-483
// The launch heuristics are this library's own (64-lane wavefronts, 256 CUs);
// they do not change results.  Every call is asynchronous on `stream`, allocates
// nothing and keeps no state; all pointers are device (or managed) pointers
// owned by the caller.
#ifndef CUEMBED_INCLUDE_EMBEDDING_LOOKUP_HPP_
#define CUEMBED_INCLUDE_EMBEDDING_LOOKUP_HPP_

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <type_traits>

#include "cuembed/include/cuembed_assert.hpp"
#include "cuembed/include/embedding_types.hpp"
#include "cuembed/include/gather_reduce_kernels.hpp"
#include "cuembed/include/hint_kernels.hpp"
#include "cuembed/include/scatter_add_kernels.hpp"

namespace cuembed {

//! In which order EmbeddingForward may add the rows of one sample.
//!   kSequential (default): lookup order, one rounding per add -- bit-identical to the
//!     reference's host loop and to its GPU kernel, for every batch size.
//!   kAllowSplit: when the batch is too small to fill the chip, the hotness loop of a
//!     sample may be split over several wavefronts and the partial rows combined through
//!     LDS (GatherReduceSplitKernel).  Same result up to fp rounding (<= 1e-3 relative in
//!     fp32, 1e-2 in fp16), several times faster for batches of a few thousand samples.
enum class ReductionOrder { kSequential = 0, kAllowSplit = 1 };

//! How EmbeddingForward (sum / mean) loads table rows.  Never changes a result.
//!   kDefault: ordinary loads -- rows stay in L2 and the Infinity Cache as long as the hardware
//!     keeps them; right whenever some rows are looked up more than once (any skewed index
//!     distribution: at C2, alpha = 1.15, 86 % of the lookups are repeats).
//!   kStreaming: non-temporal loads -- for batches in which (nearly) every lookup hits a different
//!     row (uniform indices over a table far larger than the caches): the kernel is then bound by
//!     HBM and rows that will not be re-used no longer pass through the L2's replacement
//!     (C2 shape, alpha = 0: 0.379 -> 0.355 ms).  With re-use it is much SLOWER (alpha = 1.15:
//!     0.136 -> 0.222 ms); the launcher cannot see the distribution, so it is the caller's choice.
//! (Extension: the reference has no such knob, embedding_lookup_kernels.cuh:34-77.)
enum class RowLoadPolicy { kDefault = 0, kStreaming = 1 };

//! Per-call options of EmbeddingForward (the overload that takes them keeps NO state anywhere).
struct ForwardOptions {
  ReductionOrder reduction_order = ReductionOrder::kSequential;
  RowLoadPolicy row_loads = RowLoadPolicy::kDefault;
  //! CSR only (extension): a permutation of [0, batch_size) on the device -- the order in which the samples are
  //! handed to the wavefronts.  Results do not depend on it (every sample is still pooled in lookup order into its own
  //! output row); time does when bag lengths vary: the two bags of a wavefront run in lockstep and wavefronts with
  //! unequal bags end at different times.  With the bags in descending order of length (BagOrderByLength,
  //! index_transforms.hpp; it only depends on the offsets) C3 takes 0.156 instead of 0.169 ms.
  const int32_t* sample_order = nullptr;
  //! (extension) the row-load decision left in device memory by DecideRowLoads (below): when non-null the kernels read
  //! word 0 of it (0: ordinary loads, 1: non-temporal) INSTEAD of `row_loads` -- the host never learns the answer, so
  //! the decision can be taken from the batch's own indices inside a stream-ordered step or a HIP graph.
  const uint32_t* row_loads_device = nullptr;
};

namespace detail {
//! Process-wide DEFAULTS for callers of the reference signature, which has no room for options
//! (initial values from CUEMBED_FORWARD_ORDER=split / CUEMBED_FORWARD_ROW_LOADS=streaming, read once).
//! Code that shares a process with other users of the library should pass ForwardOptions instead.
inline std::atomic<int>& ForwardOrderCell() {
  static std::atomic<int> cell{[] {
    const char* env = std::getenv("CUEMBED_FORWARD_ORDER");
    return (env != nullptr && std::strcmp(env, "split") == 0) ? 1 : 0;
  }()};
  return cell;
}
inline std::atomic<int>& ForwardRowLoadCell() {
  static std::atomic<int> cell{[] {
    const char* env = std::getenv("CUEMBED_FORWARD_ROW_LOADS");
    return (env != nullptr && std::strcmp(env, "streaming") == 0) ? 1 : 0;
  }()};
  return cell;
}
}  // namespace detail

namespace detail {
//! Tuning / tests only (results never depend on it): 0 = the launcher decides, 1 = never the wide-load kernel, 2 = whenever
//! its shape allows it with one sample per workgroup, 3 / 4 / 5 / ... = with 2 / 4 / 8 / ... samples per workgroup (as far as
//! the row allows).
inline std::atomic<int>& ForwardWideLoadCell() {
  static std::atomic<int> cell{0};
  return cell;
}
}  // namespace detail
inline void SetForwardWideLoad(const int mode) { detail::ForwardWideLoadCell().store(mode, std::memory_order_relaxed); }

inline void SetForwardReductionOrder(ReductionOrder order) {
  detail::ForwardOrderCell().store(static_cast<int>(order), std::memory_order_relaxed);
}
inline ReductionOrder GetForwardReductionOrder() {
  return static_cast<ReductionOrder>(detail::ForwardOrderCell().load(std::memory_order_relaxed));
}
inline void SetForwardRowLoadPolicy(RowLoadPolicy policy) {
  detail::ForwardRowLoadCell().store(static_cast<int>(policy), std::memory_order_relaxed);
}
inline RowLoadPolicy GetForwardRowLoadPolicy() {
  return static_cast<RowLoadPolicy>(detail::ForwardRowLoadCell().load(std::memory_order_relaxed));
}
//! The process-wide defaults as per-call options.
inline ForwardOptions DefaultForwardOptions() {
  ForwardOptions o;
  o.reduction_order = GetForwardReductionOrder();
  o.row_loads = GetForwardRowLoadPolicy();
  return o;
}

namespace detail {

constexpr int kDefaultBlockThreads = 256;
//! Below this many wavefronts the sequential mapping cannot cover the chip's 1024 SIMDs
//! twice; the split kernel (one sample per 256-thread workgroup) is then considered.
constexpr int64_t kSplitBelowWaves = 2048;
//! ... and when the bit-exact wide-load kernel is taken (ForwardWideLoadPays below).
constexpr int64_t kWideLoadMaxWaves = 256;
constexpr int kWideLoadMinHotness = 32;
constexpr int kWideLoadCsrBatch = 1024;
//! Index (+weight) staging budget per workgroup.  Small enough that eight
//! 256-thread workgroups still fit a CU's 160 KiB LDS.
constexpr int kMaxStageBytes = 16 * 1024;

//! How a row is split over lanes and how many samples share a workgroup.
struct RowSplit {
  int elems_per_lane;   //!< N: 16, 8 or 4 bytes worth of elements
  int lanes_per_row;    //!< width / N  (<= 1024)
  int rows_per_block;   //!< samples (forward) or nz-segments (backward) per workgroup
};

//! Widest naturally aligned access that the row stride AND every base pointer
//! allow (the reference's DivideRowIntoVectors, embedding_lookup.cuh:160-181,
//! looks at the row size only and would fault on an unaligned view).
template <typename ElemT>
inline RowSplit SplitRow(const int embed_width, const void* p0, const void* p1) {
  const size_t row_bytes = static_cast<size_t>(embed_width) * sizeof(ElemT);
  CUEMBED_ASSERT(embed_width > 0);
  CUEMBED_ASSERT(row_bytes % 4 == 0);
  const uintptr_t bits = reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) |
                         static_cast<uintptr_t>(row_bytes);
  int bytes_per_lane = 4;
  if (bits % 16 == 0) bytes_per_lane = 16;
  else if (bits % 8 == 0) bytes_per_lane = 8;
  CUEMBED_ASSERT(reinterpret_cast<uintptr_t>(p0) % 4 == 0 &&
                 reinterpret_cast<uintptr_t>(p1) % 4 == 0);
  RowSplit s;
  s.elems_per_lane = bytes_per_lane / static_cast<int>(sizeof(ElemT));
  s.lanes_per_row = embed_width / s.elems_per_lane;
  CUEMBED_ASSERT(s.lanes_per_row <= kMaxBlockThreads);
  s.rows_per_block =
      s.lanes_per_row >= kDefaultBlockThreads ? 1 : kDefaultBlockThreads / s.lanes_per_row;
  return s;
}

//! Complete launch description of one forward call.
struct ForwardLaunch {
  RowSplit split;
  size_t stage_bytes;  //!< dynamic LDS: staged indices (+ weights)
  bool staged;         //!< fixed-hotness indices staged in LDS
  unsigned grid;
};

//! Forward launch heuristics (counterpart of GetKernelLaunchParams,
//! embedding_lookup.cuh:186-208): 256-thread workgroups of
//! `rows_per_block` samples; fixed-hotness indices are staged in LDS when the
//! workgroup's share fits kMaxStageBytes (halving the samples per workgroup
//! first), otherwise -- and always for CSR and concat -- they are read from
//! global memory.
template <typename ElemT, typename IndexT>
inline ForwardLaunch PlanForward(const int embed_width, const void* params, const void* ret,
                                 const int batch, const int num_hots, const bool is_csr,
                                 const bool weighted, const bool concat) {
  ForwardLaunch f;
  f.split = SplitRow<ElemT>(embed_width, params, ret);
  f.stage_bytes = 0;
  f.staged = false;
  if (!is_csr && !concat) {
    const size_t per_sample =
        static_cast<size_t>(num_hots) * (sizeof(IndexT) + (weighted ? sizeof(ElemT) : 0));
    int rows = f.split.rows_per_block;
    while (rows > 1 && rows * per_sample > static_cast<size_t>(kMaxStageBytes)) rows /= 2;
    if (rows * per_sample <= static_cast<size_t>(kMaxStageBytes)) {
      f.staged = true;
      f.split.rows_per_block = rows;
      f.stage_bytes = rows * per_sample;
    }
  }
  if (is_csr && !concat && f.split.lanes_per_row <= 64 && 64 % f.split.lanes_per_row == 0) {
    // CSR bags differ in length and nothing is shared inside a workgroup (no LDS, no barrier: the lanes of a bag hand
    // indices to each other across lanes): ONE wavefront per workgroup, so that a wavefront's slot is free as soon as
    // ITS bags are pooled instead of when the longest of a 256-thread workgroup's 8 bags is (C3: 0.187 -> 0.175 ms;
    // two wavefronts measure the same, four are the old shape).  Giving a wavefront SEVERAL bags per lane group and
    // pooling them in order of length (ranked in registers) was measured and loses: 4 / 8 / 16 / 32 bags per
    // wavefront 0.180 / 0.198 / 0.199 / 0.292 ms -- fewer, longer-lived wavefronts and one exposed offset + index
    // latency per bag in sequence cost more than the lockstep of two unequal bags (docs/EXPERIMENTS.md).
    f.split.rows_per_block = 64 / f.split.lanes_per_row;
  }
  f.grid = static_cast<unsigned>((batch + f.split.rows_per_block - 1) / f.split.rows_per_block);
  return f;
}

//! Whether GatherReduceWideLoadKernel (one sample per workgroup; bit-exact) is taken instead of the sequential kernel.
//! Shape: the row splits over a power-of-two number of lanes and is at most 1 KiB.  Size, from the measured crossover
//! (tools/small_batch_forward_probe.py, profiles/r05_small_batch_forward_probe.csv; device times of graph replays):
//!   CSR bags: up to 1,024 samples (one round of workgroups on the chip) -- the sequential kernel walks the two bags of a
//!             wavefront in lockstep: 256 samples x U[0,128] lookups of 128-byte rows 12.2 -> 4.5 us, 1,024 samples of
//!             512-byte rows 12.6 -> 11.5; from 2,048 samples on it loses (512-byte rows: 16 -> 21 us) except for rows of
//!             up to 128 bytes, which still gain there (13.2 -> 9.3 us) and lose from 4,096;
//!   fixed hotness: from 32 lookups per sample, while the sequential mapping has at most min(256, 2 x hotness)
//!             wavefronts (1,024 x 64 lookups of 128-byte rows 4.6 -> 4.2 us, 256 x 64 of 512-byte rows 5.0 -> 4.1,
//!             1,024 x 256 of 128-byte rows 15.2 -> 8.2); below 32 lookups per sample the sequential kernel is at the
//!             launch floor and the wide one 0.5-1 us above it.
//! Returns the samples per workgroup (a power of two; 1 = the shapes above), or 0 when the sequential kernel is taken.
//!   CSR, larger batches of narrow rows: several samples per workgroup, lanes x slices threads and a part of the LDS
//!             each (a sample needs one thread per 32-bit word of the row for the pooling), as many as keep the grid at
//!             kWideLoadCsrBatch workgroups, while samples x row bytes <= 1024 (rows of up to 128 bytes) or 512.
inline int ForwardWideLoadSamples(const int lanes, const size_t row_bytes, const int batch, const int num_hots,
                                  const bool is_csr) {
  const int mode = ForwardWideLoadCell().load(std::memory_order_relaxed);
  const bool shape_ok = lanes >= 1 && lanes <= kWideLoadThreads && kWideLoadThreads % lanes == 0 && batch > 0 &&
                        row_bytes >= 4 && row_bytes <= static_cast<size_t>(kWideLoadMaxRowBytes);
  if (!shape_ok || mode == 1) return 0;
  int most = 1;   // samples a workgroup can hold: one pooling thread per word of the row, for every sample
  while (most * 2 * (row_bytes / 4) <= static_cast<size_t>(kWideLoadThreads) && most * 2 * lanes <= kWideLoadThreads) most *= 2;
  if (mode >= 2) {                       // forced: 2 -> 1 sample per workgroup, 3 -> 2, 4 -> 4, ...
    const int want = 1 << (mode - 2);
    return want < most ? want : most;
  }
  if (is_csr) {
    // measured (profiles/r05_csr_mid_batch_forward_probe.csv): several samples per workgroup pay while samples x row
    // bytes <= 1024 for rows of up to 128 bytes -- 4,096 bags of 128-byte rows 13.6 -> 8.9 us (4 per workgroup), 8,192
    // bags 19.2 -> 18.5 (fp16: 24.8 -> 19.1; 8), 16,384 bags of 64-byte rows 32 -> 20 (16) -- and <= 512 beyond: 256-byte
    // rows gain with 2 (2,048 bags: 10.6 -> 9.2) and lose with 4 (15.0 -> 17.5), 512-byte rows lose with 2 (13 -> 19)
    const int budget = row_bytes <= 128 ? 1024 : 512;
    const int allowed = static_cast<int>(budget / row_bytes) > 1 ? static_cast<int>(budget / row_bytes) : 1;
    int samples = 1;
    while (samples < most && samples < allowed && (batch + samples - 1) / samples > kWideLoadCsrBatch) samples *= 2;
    const int64_t limit = row_bytes <= 128 && samples == 1 ? 2 * kWideLoadCsrBatch : kWideLoadCsrBatch;
    return (batch + samples - 1) / samples <= limit ? samples : 0;
  }
  const int64_t waves = static_cast<int64_t>(batch) * lanes / 64;
  const int64_t limit = 2 * static_cast<int64_t>(num_hots) < kWideLoadMaxWaves ? 2 * static_cast<int64_t>(num_hots)
                                                                              : kWideLoadMaxWaves;
  return num_hots >= kWideLoadMinHotness && waves <= limit ? 1 : 0;
}
inline bool ForwardWideLoadPays(const int lanes, const size_t row_bytes, const int batch, const int num_hots,
                                const bool is_csr) {
  return ForwardWideLoadSamples(lanes, row_bytes, batch, num_hots, is_csr) > 0;
}

template <typename ElemT, typename AccT, typename IndexT, typename OffsetT, int N>
inline void LaunchGatherReduce(const ElemT* table, int width, const IndexT* indices,
                               const OffsetT* offsets, const ElemT* weights, int batch,
                               int num_hots, bool is_mean, ElemT* out, const ForwardLaunch& f,
                               hipStream_t stream, const ForwardOptions& options) {
  const bool weighted = weights != nullptr;
  const bool stream_rows = options.row_loads == RowLoadPolicy::kStreaming;
  const int lanes = f.split.lanes_per_row;
  const bool lanes_fit_waves = (lanes <= 64 && 64 % lanes == 0) || lanes == 128;
  if (options.reduction_order == ReductionOrder::kAllowSplit && lanes_fit_waves &&
      static_cast<int64_t>(batch) * lanes / 64 < kSplitBelowWaves &&
      (offsets != nullptr || num_hots >= 8)) {
    const dim3 sblock(lanes, kSplitBlockThreads / lanes, 1);
    const dim3 sgrid(batch, 1, 1);
    if (weighted)
      GatherReduceSplitKernel<ElemT, AccT, IndexT, OffsetT, N, true><<<sgrid, sblock, 0, stream>>>(
          table, width, batch, indices, offsets, num_hots, weights, is_mean, out);
    else
      GatherReduceSplitKernel<ElemT, AccT, IndexT, OffsetT, N, false><<<sgrid, sblock, 0, stream>>>(
          table, width, batch, indices, offsets, num_hots, weights, is_mean, out);
    return;
  }
  // small batches, bit-exact: one sample per workgroup, a bag's loads spread over all of its threads, the adds in order
  if (const int samples = ForwardWideLoadSamples(lanes, static_cast<size_t>(width) * sizeof(ElemT), batch, num_hots,
                                                 offsets != nullptr)) {
    const int slices = kWideLoadThreads / (lanes * samples);
    const dim3 wblock(lanes, slices, samples);
    const dim3 wgrid(static_cast<unsigned>((batch + samples - 1) / samples), 1, 1);
    const size_t chunk = static_cast<size_t>(kForwardUnroll) * slices;
    const size_t lds = samples * (chunk * lanes * sizeof(Pack<ElemT, N>) + (weighted ? chunk * sizeof(ElemT) : 0));
    if (weighted)
      GatherReduceWideLoadKernel<ElemT, AccT, IndexT, OffsetT, N, true><<<wgrid, wblock, lds, stream>>>(
          table, width, batch, indices, offsets, num_hots, weights, is_mean, out, stream_rows, options.row_loads_device);
    else
      GatherReduceWideLoadKernel<ElemT, AccT, IndexT, OffsetT, N, false><<<wgrid, wblock, lds, stream>>>(
          table, width, batch, indices, offsets, num_hots, weights, is_mean, out, stream_rows, options.row_loads_device);
    return;
  }
  const dim3 block(f.split.lanes_per_row, f.split.rows_per_block, 1);
  const dim3 grid(f.grid, 1, 1);
#define CUEMBED_LAUNCH_GR(W, SRC)                                                         \
  GatherReduceKernel<ElemT, AccT, IndexT, OffsetT, N, W, SRC>                             \
      <<<grid, block, f.stage_bytes, stream>>>(table, width, batch, indices, offsets,     \
                                               num_hots, weights, is_mean, out, 1, stream_rows, \
                                               offsets != nullptr ? options.sample_order : nullptr, \
                                               options.row_loads_device)
  if (f.staged) {
    if (weighted) CUEMBED_LAUNCH_GR(true, IndexSource::kLdsStaged);
    else CUEMBED_LAUNCH_GR(false, IndexSource::kLdsStaged);
  } else if (64 % f.split.lanes_per_row == 0) {
    if (weighted) CUEMBED_LAUNCH_GR(true, IndexSource::kWaveShuffle);
    else CUEMBED_LAUNCH_GR(false, IndexSource::kWaveShuffle);
  } else {
    if (weighted) CUEMBED_LAUNCH_GR(true, IndexSource::kGlobal);
    else CUEMBED_LAUNCH_GR(false, IndexSource::kGlobal);
  }
#undef CUEMBED_LAUNCH_GR
}

template <typename ElemT, typename IndexT, int N>
inline void LaunchGatherConcat(const ElemT* table, int width, const IndexT* indices, int batch,
                               int num_hots, ElemT* out, const ForwardLaunch& f,
                               hipStream_t stream) {
  const dim3 block(f.split.lanes_per_row, f.split.rows_per_block, 1);
  const dim3 grid(f.grid, 1, 1);
  GatherConcatKernel<ElemT, IndexT, N>
      <<<grid, block, 0, stream>>>(table, width, batch, indices, num_hots, out);
}

}  // namespace detail

/**
 * @brief Embedding forward: gather the rows named by `indices` and combine them
 * per sample.  Fixed hotness (`offsets == nullptr`, `num_hots > 0`) or CSR
 * (`offsets[batch_size + 1]`, `num_hots == 0`).  Same contract as the reference
 * (embedding_lookup.cuh:210-308).
 *
 * @tparam InputT   table element type (float or __half)
 * @tparam OutputT  result element type (must equal InputT)
 * @tparam IndexT   int32_t or int64_t
 * @tparam OffsetT  CSR offset type (int32_t or int64_t)
 * @tparam fp16_math accumulate __half tables in fp16 instead of fp32
 *
 * @param params      table, row-major [rows x embed_width]
 * @param embed_width elements per row (row bytes must be a multiple of 4)
 * @param indices     lookup indices (fixed: [batch x num_hots]; CSR: [nnz])
 * @param offsets     CSR offsets or nullptr
 * @param weights     per-lookup weights (same layout as indices) or nullptr
 * @param batch_size  number of samples
 * @param num_hots    fixed hotness, 0 for CSR
 * @param mode        kSum, kMean or kConcat (concat: fixed hotness, unweighted)
 * @param ret         output: [batch x width] (sum/mean), [batch x num_hots x width] (concat)
 * @param stream      HIP stream
 * @param options     (extension, second overload) per-call ForwardOptions; the reference signature uses
 *                    DefaultForwardOptions(), i.e. the process-wide defaults
 */
template <typename InputT,
          typename OutputT,
          typename IndexT,
          typename OffsetT,
          bool fp16_math = false>
void EmbeddingForward(const InputT* params,
                      const int embed_width,
                      const IndexT* indices,
                      const OffsetT* offsets,
                      const GetElemT<InputT>* weights,
                      const int batch_size,
                      const int num_hots,
                      const CombineMode mode,
                      OutputT* ret,
                      const hipStream_t stream,
                      const ForwardOptions& options) {
  static_assert(std::is_same<InputT, OutputT>::value,
                "EmbeddingForward: OutputT must equal InputT");
  using HostElemT = GetElemT<InputT>;
  static_assert(std::is_same<HostElemT, float>::value || std::is_same<HostElemT, __half>::value ||
                    std::is_same<HostElemT, __hip_bfloat16>::value,
                "EmbeddingForward: table elements must be float, __half or __hip_bfloat16");
  using ElemT = detail::DeviceElemT<HostElemT>;
  using AccT = typename std::conditional<fp16_math && detail::IsHalf<ElemT>::value, ElemT,
                                         float>::type;

  // Same argument contract as the reference (embedding_lookup.cuh:261-267).
  CUEMBED_ASSERT(weights == nullptr || mode != CombineMode::kConcat);
  CUEMBED_ASSERT((offsets != nullptr && num_hots == 0) || (offsets == nullptr && num_hots > 0));
  CUEMBED_ASSERT(offsets == nullptr || mode != CombineMode::kConcat);
  CUEMBED_ASSERT(options.sample_order == nullptr || offsets != nullptr);   // a scheduling hint for ragged bags only
  if (batch_size <= 0) return;

  const ElemT* table = reinterpret_cast<const ElemT*>(params);
  const ElemT* w = reinterpret_cast<const ElemT*>(weights);
  ElemT* out = reinterpret_cast<ElemT*>(ret);
  const detail::ForwardLaunch split = detail::PlanForward<ElemT, IndexT>(
      embed_width, params, ret, batch_size, num_hots, offsets != nullptr, weights != nullptr,
      mode == CombineMode::kConcat);
  constexpr int kMaxN = 16 / static_cast<int>(sizeof(ElemT));
  const int elems_per_lane = split.split.elems_per_lane;

  if (mode == CombineMode::kConcat) {
    if (elems_per_lane == kMaxN)
      detail::LaunchGatherConcat<ElemT, IndexT, kMaxN>(table, embed_width, indices, batch_size,
                                                       num_hots, out, split, stream);
    else if (elems_per_lane == kMaxN / 2)
      detail::LaunchGatherConcat<ElemT, IndexT, kMaxN / 2>(table, embed_width, indices,
                                                           batch_size, num_hots, out, split, stream);
    else
      detail::LaunchGatherConcat<ElemT, IndexT, kMaxN / 4>(table, embed_width, indices,
                                                           batch_size, num_hots, out, split, stream);
    return;
  }
  const bool is_mean = mode == CombineMode::kMean;
  if (elems_per_lane == kMaxN)
    detail::LaunchGatherReduce<ElemT, AccT, IndexT, OffsetT, kMaxN>(
        table, embed_width, indices, offsets, w, batch_size, num_hots, is_mean, out, split, stream, options);
  else if (elems_per_lane == kMaxN / 2)
    detail::LaunchGatherReduce<ElemT, AccT, IndexT, OffsetT, kMaxN / 2>(
        table, embed_width, indices, offsets, w, batch_size, num_hots, is_mean, out, split, stream, options);
  else
    detail::LaunchGatherReduce<ElemT, AccT, IndexT, OffsetT, kMaxN / 4>(
        table, embed_width, indices, offsets, w, batch_size, num_hots, is_mean, out, split, stream, options);
}

//! The reference's signature (embedding_lookup.cuh:245-259): options = the process-wide defaults.
template <typename InputT,
          typename OutputT,
          typename IndexT,
          typename OffsetT,
          bool fp16_math = false>
void EmbeddingForward(const InputT* params,
                      const int embed_width,
                      const IndexT* indices,
                      const OffsetT* offsets,
                      const GetElemT<InputT>* weights,
                      const int batch_size,
                      const int num_hots,
                      const CombineMode mode,
                      OutputT* ret,
                      const hipStream_t stream = 0) {
  EmbeddingForward<InputT, OutputT, IndexT, OffsetT, fp16_math>(params, embed_width, indices, offsets, weights,
                                                                batch_size, num_hots, mode, ret, stream,
                                                                DefaultForwardOptions());
}

//! Tables below this size, and batches of fewer lookups, are never streamed (the batch is latency-bound or the table
//! lives in the caches whatever the loads are): DecideRowLoads answers "default" for them without looking.
constexpr int64_t kStreamingMinTableBytes = int64_t{1} << 30;
constexpr int64_t kStreamingMinLookups = int64_t{1} << 18;
//! ... and at least this share of a strided sample's rows must be distinct INSIDE their group of 4,096 (x / 65,536).
//! Measured crossover at the C2 shape (tools/row_loads_crossover_probe.py, profiles/r06_row_loads_crossover.jsonl):
//! power-law exponent 0.5 -- 99.93 % distinct inside a group, 3 repeats per 4,096 -- is where streaming stops
//! paying (- 1 %); at exponent 0.75 (96.9 %) it already costs 20 %.  A few repeats per group are what separates the two:
//! the threshold sits at 8 repeats per 4,096.
constexpr unsigned kStreamingDistinctPer65536 = 65408;    // 0.998

/**
 * @brief RowLoadPolicy decided ON THE DEVICE from the batch's own indices (extension): an evenly strided sample of up
 * to 65,536 lookups is cut into groups of 4,096, every group's DISTINCT rows are counted exactly (one workgroup and
 * one LDS hash set per group), and `decision[0]` becomes 1 (kStreaming) when at least `distinct_per_65536` / 65536 of
 * the sample is distinct, the table has `table_bytes` >= 1 GiB and the batch >= 2^18 lookups -- else 0.  One launch,
 * no read-back: pass `decision` as ForwardOptions::row_loads_device to the EmbeddingForward calls that follow (this
 * batch and, for a stationary index distribution, the next few hundred).
 * `decision`: FOUR 32-bit words of device memory, zeroed once by the caller (word 0 = the decision, the others are the
 * kernel's arrival counters and are left at zero); calls that share them must be stream-ordered.
 */
template <typename IndexT>
void DecideRowLoads(const IndexT* indices,
                    const int64_t nnz,
                    const int64_t table_bytes,
                    uint32_t* decision,
                    const hipStream_t stream = 0,
                    const unsigned distinct_per_65536 = kStreamingDistinctPer65536) {
  CUEMBED_ASSERT(decision != nullptr);
  if (nnz < kStreamingMinLookups || table_bytes < kStreamingMinTableBytes || indices == nullptr) {
    detail::ClearRowLoadsDecisionKernel<0><<<1, 64, 0, stream>>>(decision);
    return;
  }
  int64_t groups = nnz / detail::kDecideGroupSample;
  groups = groups > detail::kDecideMaxGroups ? detail::kDecideMaxGroups : groups;
  detail::DecideRowLoadsKernel<IndexT><<<static_cast<unsigned>(groups), detail::kDecideThreads, 0, stream>>>(
      indices, nnz, detail::kDecideGroupSample, distinct_per_65536, decision);
}

/**
 * @brief Gradient of a weighted sum-forward with respect to the per-lookup weights (extension;
 * nn.EmbeddingBag's per_sample_weights gradient):
 * grad_weights[s, j] = dot(params[indices[s, j], :], grad_y[s, :]).  Same index layouts as
 * EmbeddingForward (fixed hotness or CSR); grad_weights has one entry per lookup.
 */
template <typename InputT, typename IndexT, typename OffsetT>
void EmbeddingWeightGrad(const InputT* params,
                         const int embed_width,
                         const IndexT* indices,
                         const OffsetT* offsets,
                         const InputT* grad_y,
                         const int batch_size,
                         const int num_hots,
                         InputT* grad_weights,
                         const hipStream_t stream = 0) {
  using ElemT = detail::DeviceElemT<GetElemT<InputT>>;
  CUEMBED_ASSERT((offsets != nullptr && num_hots == 0) || (offsets == nullptr && num_hots > 0));
  if (batch_size <= 0) return;
  const detail::RowSplit split = detail::SplitRow<ElemT>(embed_width, params, grad_y);
  int group = 1;
  while (group < split.lanes_per_row && group < 64) group *= 2;
  const int samples_per_block = detail::kDefaultBlockThreads / group;
  const dim3 block(group, samples_per_block, 1);
  const dim3 grid((batch_size + samples_per_block - 1) / samples_per_block, 1, 1);
  const ElemT* table = reinterpret_cast<const ElemT*>(params);
  const ElemT* gy = reinterpret_cast<const ElemT*>(grad_y);
  ElemT* gw = reinterpret_cast<ElemT*>(grad_weights);
  constexpr int kMaxN = 16 / static_cast<int>(sizeof(ElemT));
#define CUEMBED_LAUNCH_WG(NN)                                                                   \
  detail::WeightGradKernel<ElemT, IndexT, OffsetT, NN><<<grid, block, 0, stream>>>(             \
      table, embed_width, batch_size, indices, offsets, num_hots, gy, gw)
  if (split.elems_per_lane == kMaxN) CUEMBED_LAUNCH_WG(kMaxN);
  else if (split.elems_per_lane == kMaxN / 2) CUEMBED_LAUNCH_WG(kMaxN / 2);
  else CUEMBED_LAUNCH_WG(kMaxN / 4);
#undef CUEMBED_LAUNCH_WG
}

}  // namespace cuembed

#include "cuembed/include/embedding_backward.hpp"

#endif  // CUEMBED_INCLUDE_EMBEDDING_LOOKUP_HPP_
