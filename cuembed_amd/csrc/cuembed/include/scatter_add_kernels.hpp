// MI355X (gfx950 / CDNA4) kernels for EmbeddingBackward.
//
// Input is the transposed (index-sorted) COO produced by Transpose(): runs of
// equal row ids are contiguous.  The work is cut into fixed-length nz-segments so
// that it is balanced no matter how skewed the run lengths are (at the north-star
// shape one row owns a run of 65,528 lookups); `lanes_per_row` lanes walk one
// segment in nz order, gathering grad_y[sample_id] row slices with kBackwardUnroll
// loads in flight and keeping the running sum in fp32 registers.  When a run ends:
//   * the run lies entirely inside this segment  -> one plain vector store;
//   * the run continues from / into a neighbour segment -> hardware float atomics
//     (global_atomic_add_f32 / global_atomic_pk_add_f16) into the zeroed output.
// This is the reference's scheme (embedding_lookup_kernels.cuh:175-220,
// embedding_lookup_ops.cuh:518-564, :647-662) with three differences: shared-run
// detection looks at the real neighbours instead of treating every first/last
// run as shared (fewer atomics), partial sums are fp32 for fp16 gradients too
// (one rounding per flush instead of one per lookup), and all row addressing is
// 64-bit (the reference's int32 `row * embed_width`, ops.cuh:610-618, overflows
// beyond 2^31 elements -- a dense 10M x 256 gradient).
#ifndef CUEMBED_INCLUDE_SCATTER_ADD_KERNELS_HPP_
#define CUEMBED_INCLUDE_SCATTER_ADD_KERNELS_HPP_

#include "cuembed/include/embedding_types.hpp"
#include "cuembed/include/gather_reduce_kernels.hpp"

namespace cuembed {
namespace detail {

constexpr int kBackwardUnroll = 8;

template <int N>
__device__ __forceinline__ void FlushAtomic(float* dst, const float (&acc)[N]) {
#pragma unroll
  for (int e = 0; e < N; ++e) unsafeAtomicAdd(dst + e, acc[e]);
}

template <int N>
__device__ __forceinline__ void FlushAtomic(_Float16* dst, const float (&acc)[N]) {
  static_assert(N % 2 == 0, "fp16 rows are split in multiples of 4 bytes");
  typedef _Float16 __attribute__((ext_vector_type(2))) half2_t;
#pragma unroll
  for (int e = 0; e < N; e += 2) {
    half2_t v;
    v.x = static_cast<_Float16>(acc[e]);
    v.y = static_cast<_Float16>(acc[e + 1]);
    unsafeAtomicAdd(reinterpret_cast<__half2*>(dst + e), *reinterpret_cast<__half2*>(&v));
  }
}

template <typename GradT, int N>
__device__ __forceinline__ void FlushStore(GradT* dst, const float (&acc)[N]) {
  Pack<GradT, N> p;
#pragma unroll
  for (int e = 0; e < N; ++e) p.v[e] = static_cast<GradT>(acc[e]);
  StorePack<GradT, N>(dst, p);
}

//! block = (lanes_per_row, segments_per_block); grid = ceil(num_segments / segments_per_block)
template <typename GradT, typename IndexT, int N, bool kWeighted>
__global__ void __launch_bounds__(kMaxBlockThreads)
SegmentedScatterAddKernel(const GradT* __restrict__ grad_y,
                          const int width,
                          const IndexT* __restrict__ rows,        // sorted; remapped or raw ids
                          const IndexT* __restrict__ sample_ids,
                          const GradT* __restrict__ weights,
                          const int64_t nnz,
                          const int segment_len,
                          GradT* __restrict__ grad_out) {
  using A = Arith<float>;
  const int lane_x = threadIdx.x;
  const int64_t segment = static_cast<int64_t>(blockIdx.x) * blockDim.y + threadIdx.y;
  const int64_t begin = segment * segment_len;
  if (begin >= nnz) return;
  const int64_t end = (begin + segment_len < nnz) ? begin + segment_len : nnz;

  // A run is "shared" when it also has lookups in a neighbouring segment.
  bool run_shared = begin > 0 && rows[begin - 1] == rows[begin];
  const bool tail_shared = end < nnz && rows[end] == rows[end - 1];

  const GradT* lane_src = grad_y + static_cast<int64_t>(lane_x) * N;
  GradT* lane_dst = grad_out + static_cast<int64_t>(lane_x) * N;

  float acc[N];
#pragma unroll
  for (int e = 0; e < N; ++e) acc[e] = 0.f;

  auto end_of_run = [&](int64_t row, bool is_last_of_segment) {
    GradT* dst = lane_dst + row * width;
    if (run_shared || (is_last_of_segment && tail_shared)) FlushAtomic<N>(dst, acc);
    else FlushStore<GradT, N>(dst, acc);
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = 0.f;
    run_shared = false;
  };

  int64_t i = begin;
  int64_t row_cur = static_cast<int64_t>(rows[i]);
  for (; i + kBackwardUnroll <= end; i += kBackwardUnroll) {
    Pack<GradT, N> g[kBackwardUnroll];
    GradT w[kBackwardUnroll];
    int64_t row_next[kBackwardUnroll];
#pragma unroll
    for (int u = 0; u < kBackwardUnroll; ++u) {
      const int64_t sid = static_cast<int64_t>(sample_ids[i + u]);
      if constexpr (kWeighted) w[u] = weights[i + u];
      // row id of the FOLLOWING lookup (clamped at the end of the array)
      row_next[u] = (i + u + 1 < nnz) ? static_cast<int64_t>(rows[i + u + 1]) : -1;
      g[u] = LoadPack<GradT, N>(lane_src + sid * width);
    }
#pragma unroll
    for (int u = 0; u < kBackwardUnroll; ++u) {
      if constexpr (kWeighted) {
        const float wf = static_cast<float>(w[u]);
#pragma unroll
        for (int e = 0; e < N; ++e)
          acc[e] = A::add(acc[e], A::mul(static_cast<float>(g[u].v[e]), wf));
      } else {
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], static_cast<float>(g[u].v[e]));
      }
      const bool last = (i + u + 1 == end);
      if (last || row_next[u] != row_cur) end_of_run(row_cur, last);
      row_cur = row_next[u];
    }
  }
  for (; i < end; ++i) {
    const int64_t sid = static_cast<int64_t>(sample_ids[i]);
    const Pack<GradT, N> g = LoadPack<GradT, N>(lane_src + sid * width);
    const int64_t row_next = (i + 1 < nnz) ? static_cast<int64_t>(rows[i + 1]) : -1;
    if constexpr (kWeighted) {
      const float wf = static_cast<float>(weights[i]);
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], A::mul(static_cast<float>(g.v[e]), wf));
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], static_cast<float>(g.v[e]));
    }
    const bool last = (i + 1 == end);
    if (last || row_next != row_cur) end_of_run(row_cur, last);
    row_cur = row_next;
  }
}

//! inverse_mapping[remapped[i]] = indices[i] at the first lookup of every run
//! (reference: CompactSparseIndicesKernel, embedding_lookup_kernels.cuh:289-302).
template <typename IndexT>
__global__ void CompactRunHeadsKernel(const IndexT* __restrict__ indices,
                                      const IndexT* __restrict__ remapped,
                                      IndexT* __restrict__ inverse_mapping,
                                      const int64_t nnz) {
  const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= nnz) return;
  const IndexT v = indices[i];
  if (i == 0 || indices[i - 1] != v) inverse_mapping[remapped[i]] = v;
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_SCATTER_ADD_KERNELS_HPP_
