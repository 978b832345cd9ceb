// MI355X (gfx950 / CDNA4) gather-reduce kernels for EmbeddingForward.
//
// What they compute is what the reference's EmbeddingLookUpKernel computes
// (cuembed/include/embedding_lookup_kernels.cuh:34-170 with the ops of
// embedding_lookup_ops.cuh:72-495): for sample s,
//     sum    : out[s,:]   = sum_j  w_j * table[idx[s,j],:]        (j ascending)
//     mean   : out[s,:]   = sum(...) * (1 / sum_j w_j)   (zeros when the sum is 0)
//     concat : out[s,j,:] = table[idx[s,j],:]
// How they do it is CDNA4-specific:
//   * one lane owns 16 (or 8/4) bytes of the row; a 512-byte row is 32 lanes, so a
//     64-lane wavefront streams TWO samples and every global_load_dwordx4 moves
//     1 KiB as two fully coalesced 512-B row segments;
//   * the hotness loop is batched: kUnroll independent row loads are issued
//     back-to-back (kUnroll KiB in flight per wave) before the first one is
//     consumed -- with 6..8 waves per SIMD that is far more than the ~10-20 KiB
//     per CU needed to cover HBM/L2 latency;
//   * accumulation is strictly in lookup order, one unfused add (and one unfused
//     multiply when weighted) per element, so results are bit-identical to the
//     sequential host loop of the reference (embedding_lookup_cpu.hpp:57-93);
//   * fixed-hotness indices (and weights) of the workgroup's samples are staged
//     once into LDS with coalesced loads; CSR bags read their index straight
//     from global memory (a wave-broadcast load that hits L1 31 times out of 32);
//   * all row addressing is 64-bit (a 10M x 256 table is 2.56 G elements).
// No MFMA: the op is a bandwidth-bound gather, there is no contraction to feed.
#ifndef CUEMBED_INCLUDE_GATHER_REDUCE_KERNELS_HPP_
#define CUEMBED_INCLUDE_GATHER_REDUCE_KERNELS_HPP_

#include "cuembed/include/embedding_types.hpp"

namespace cuembed {
namespace detail {

constexpr int kForwardUnroll = 8;
constexpr int kMaxBlockThreads = 1024;

//! Where a lane finds the lookup indices of its sample.
enum class IndexSource {
  kLdsStaged,  //!< fixed hotness, indices of the whole workgroup staged in LDS
  kGlobal      //!< CSR offsets (or fixed hotness too large to stage)
};

template <typename ElemT, int N>
__device__ __forceinline__ Pack<ElemT, N> LoadPack(const ElemT* p) {
  return *reinterpret_cast<const Pack<ElemT, N>*>(p);
}

template <typename ElemT, int N>
__device__ __forceinline__ void StorePack(ElemT* p, const Pack<ElemT, N>& v) {
  *reinterpret_cast<Pack<ElemT, N>*>(p) = v;
}

// ---------------------------------------------------------------------------
// Sum / mean.
//   block = (lanes_per_row, samples_per_block); grid = ceil(batch / samples_per_block)
//   dynamic LDS (kLdsStaged only) = samples_per_block * num_hots *
//                                   (sizeof(IndexT) [+ sizeof(ElemT) if weighted])
// ---------------------------------------------------------------------------
template <typename ElemT,    // table / output element: float or _Float16
          typename AccT,     // accumulator: float, or _Float16 ("fp16_math")
          typename IndexT,   // int32_t / int64_t
          typename OffsetT,  // CSR offset type (unused for kLdsStaged)
          int N,             // elements per lane
          bool kWeighted,
          IndexSource kSource>
__global__ void __launch_bounds__(kMaxBlockThreads)
GatherReduceKernel(const ElemT* __restrict__ table,
                   const int width,
                   const int batch,
                   const IndexT* __restrict__ indices,
                   const OffsetT* __restrict__ offsets,  // null => fixed hotness
                   const int num_hots,
                   const ElemT* __restrict__ weights,
                   const bool is_mean,
                   ElemT* __restrict__ out) {
  using A = Arith<AccT>;
  const int lane_x = threadIdx.x;
  const int slot = threadIdx.y;
  const int samples_per_block = blockDim.y;
  const int64_t sample = static_cast<int64_t>(blockIdx.x) * samples_per_block + slot;

  // ---- locate this sample's indices -----------------------------------
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const IndexT* lds_idx = nullptr;
  const ElemT* lds_w = nullptr;
  const IndexT* g_idx = nullptr;
  const ElemT* g_w = nullptr;
  int hot = num_hots;

  if constexpr (kSource == IndexSource::kLdsStaged) {
    IndexT* stage_idx = reinterpret_cast<IndexT*>(lds_raw);
    ElemT* stage_w = reinterpret_cast<ElemT*>(stage_idx + samples_per_block * num_hots);
    const int64_t first = static_cast<int64_t>(blockIdx.x) * samples_per_block * num_hots;
    const int64_t remaining = static_cast<int64_t>(batch) * num_hots - first;
    const int count = static_cast<int>(
        remaining < static_cast<int64_t>(samples_per_block) * num_hots
            ? remaining
            : static_cast<int64_t>(samples_per_block) * num_hots);
    const int tid = slot * blockDim.x + lane_x;
    const int nthreads = blockDim.x * samples_per_block;
    for (int i = tid; i < count; i += nthreads) {
      stage_idx[i] = indices[first + i];
      if constexpr (kWeighted) stage_w[i] = weights[first + i];
    }
    __syncthreads();
    if (sample >= batch) return;
    lds_idx = stage_idx + slot * num_hots;
    lds_w = stage_w + slot * num_hots;
  } else {
    if (sample >= batch) return;
    int64_t begin;
    if (offsets != nullptr) {
      begin = static_cast<int64_t>(offsets[sample]);
      hot = static_cast<int>(static_cast<int64_t>(offsets[sample + 1]) - begin);
    } else {
      begin = sample * num_hots;
    }
    g_idx = indices + begin;
    g_w = weights + begin;
  }

  auto index_at = [&](int j) -> int64_t {
    if constexpr (kSource == IndexSource::kLdsStaged) return static_cast<int64_t>(lds_idx[j]);
    else return static_cast<int64_t>(g_idx[j]);
  };
  auto weight_at = [&](int j) -> ElemT {
    if constexpr (kSource == IndexSource::kLdsStaged) return lds_w[j];
    else return g_w[j];
  };

  // ---- gather + reduce --------------------------------------------------
  const ElemT* lane_base = table + static_cast<int64_t>(lane_x) * N;
  AccT acc[N];
#pragma unroll
  for (int e = 0; e < N; ++e) acc[e] = static_cast<AccT>(0);
  float weight_sum = 0.f;

  int j = 0;
  for (; j + kForwardUnroll <= hot; j += kForwardUnroll) {
    Pack<ElemT, N> row[kForwardUnroll];
    ElemT w[kForwardUnroll];
#pragma unroll
    for (int u = 0; u < kForwardUnroll; ++u) {
      const int64_t r = index_at(j + u);
      if constexpr (kWeighted) w[u] = weight_at(j + u);
      row[u] = LoadPack<ElemT, N>(lane_base + r * width);
    }
#pragma unroll
    for (int u = 0; u < kForwardUnroll; ++u) {
      if constexpr (kWeighted) {
        const AccT wa = A::widen(w[u]);
        weight_sum += static_cast<float>(w[u]);
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], A::mul(A::widen(row[u].v[e]), wa));
      } else {
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], A::widen(row[u].v[e]));
      }
    }
  }
  for (; j < hot; ++j) {
    const int64_t r = index_at(j);
    const Pack<ElemT, N> row = LoadPack<ElemT, N>(lane_base + r * width);
    if constexpr (kWeighted) {
      const ElemT w = weight_at(j);
      const AccT wa = A::widen(w);
      weight_sum += static_cast<float>(w);
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], A::mul(A::widen(row.v[e]), wa));
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], A::widen(row.v[e]));
    }
  }

  // ---- epilogue -----------------------------------------------------------
  if (is_mean) {
    // Reference combiner (embedding_lookup_ops.cuh:273-285): scale by the
    // reciprocal of the accumulated weight; zeros when that is 0.
    if constexpr (!kWeighted) weight_sum = static_cast<float>(hot);
    const float inv = (weight_sum == 0.f) ? 0.f : 1.0f / weight_sum;
    const AccT scale = static_cast<AccT>(inv);
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = A::mul(acc[e], scale);
  }
  Pack<ElemT, N> result;
#pragma unroll
  for (int e = 0; e < N; ++e) result.v[e] = static_cast<ElemT>(acc[e]);
  StorePack<ElemT, N>(out + sample * width + static_cast<int64_t>(lane_x) * N, result);
}

// ---------------------------------------------------------------------------
// Concat (fixed hotness only): out[s, j, :] = table[idx[s, j], :].
// Same block/grid shape; indices always read from global (each is used once).
// ---------------------------------------------------------------------------
template <typename ElemT, typename IndexT, int N>
__global__ void __launch_bounds__(kMaxBlockThreads)
GatherConcatKernel(const ElemT* __restrict__ table,
                   const int width,
                   const int batch,
                   const IndexT* __restrict__ indices,
                   const int num_hots,
                   ElemT* __restrict__ out) {
  const int lane_x = threadIdx.x;
  const int64_t sample = static_cast<int64_t>(blockIdx.x) * blockDim.y + threadIdx.y;
  if (sample >= batch) return;
  const IndexT* my_idx = indices + sample * num_hots;
  const ElemT* lane_base = table + static_cast<int64_t>(lane_x) * N;
  ElemT* dst = out + sample * num_hots * width + static_cast<int64_t>(lane_x) * N;
  int j = 0;
  for (; j + kForwardUnroll <= num_hots; j += kForwardUnroll) {
    Pack<ElemT, N> row[kForwardUnroll];
#pragma unroll
    for (int u = 0; u < kForwardUnroll; ++u) {
      row[u] = LoadPack<ElemT, N>(lane_base + static_cast<int64_t>(my_idx[j + u]) * width);
    }
#pragma unroll
    for (int u = 0; u < kForwardUnroll; ++u) {
      StorePack<ElemT, N>(dst + static_cast<int64_t>(j + u) * width, row[u]);
    }
  }
  for (; j < num_hots; ++j) {
    StorePack<ElemT, N>(dst + static_cast<int64_t>(j) * width,
                        LoadPack<ElemT, N>(lane_base + static_cast<int64_t>(my_idx[j]) * width));
  }
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_GATHER_REDUCE_KERNELS_HPP_
