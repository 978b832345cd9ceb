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

//! LDS needed by SegmentedScatterAddKernel for `segments_per_block` segments of
//! `segment_len` lookups: row ids for [first - 1, last + 1] (two sentinels),
//! sample ids, and weights when weighted.
template <typename GradT, typename IndexT>
__host__ __device__ inline size_t ScatterStageBytes(int segments_per_block, int segment_len,
                                                    bool weighted) {
  const size_t n = static_cast<size_t>(segments_per_block) * segment_len;
  size_t bytes = (n + 2) * sizeof(IndexT) + n * sizeof(IndexT);
  bytes = (bytes + 15) / 16 * 16;
  if (weighted) bytes += n * sizeof(GradT);
  return (bytes + 15) / 16 * 16;
}

//! block = (lanes_per_row, segments_per_block); grid = ceil(num_segments / segments_per_block)
//! dynamic LDS = ScatterStageBytes(...).
//!
//! The workgroup's segments are consecutive, so their COO triples form ONE
//! contiguous range of the sorted arrays: it is copied into LDS with coalesced
//! loads once, and the walk then reads ids from LDS -- only the grad_y row
//! gathers remain on the global-memory critical path (one latency per batch of
//! kBackwardUnroll lookups instead of two).
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
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane_x = threadIdx.x;
  const int segments_per_block = blockDim.y;
  const int block_len = segments_per_block * segment_len;
  const int64_t block_begin = static_cast<int64_t>(blockIdx.x) * block_len;

  // ---- stage rows[block_begin-1 .. block_begin+block_len], sample ids, weights ----
  IndexT* st_rows = reinterpret_cast<IndexT*>(lds_raw);            // [block_len + 2], index 0 = element before
  IndexT* st_sids = st_rows + block_len + 2;                       // [block_len]
  GradT* st_w = reinterpret_cast<GradT*>(
      lds_raw + (((static_cast<size_t>(block_len) * 2 + 2) * sizeof(IndexT) + 15) / 16 * 16));
  {
    const int tid = threadIdx.y * blockDim.x + lane_x;
    const int nthreads = blockDim.x * blockDim.y;
    for (int k = tid; k < block_len + 2; k += nthreads) {
      const int64_t g = block_begin - 1 + k;
      st_rows[k] = (g >= 0 && g < nnz) ? rows[g] : static_cast<IndexT>(-1);
    }
    for (int k = tid; k < block_len; k += nthreads) {
      const int64_t g = block_begin + k;
      if (g < nnz) {
        st_sids[k] = sample_ids[g];
        if constexpr (kWeighted) st_w[k] = weights[g];
      }
    }
  }
  __syncthreads();

  const int seg_off = threadIdx.y * segment_len;  // offset of this segment inside the block
  const int64_t begin = block_begin + seg_off;
  if (begin >= nnz) return;
  const int count = static_cast<int>((begin + segment_len < nnz) ? segment_len : nnz - begin);
  const IndexT* my_rows = st_rows + 1 + seg_off;   // my_rows[-1] = lookup before the segment
  const IndexT* my_sids = st_sids + seg_off;
  const GradT* my_w = st_w + seg_off;

  // A run is "shared" when it also has lookups in a neighbouring segment.
  bool run_shared = my_rows[-1] == my_rows[0];                    // sentinel -1 never matches
  const bool tail_shared = my_rows[count] == my_rows[count - 1];  // sentinel past the end of nnz

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

  int i = 0;
  int64_t row_cur = static_cast<int64_t>(my_rows[0]);
  for (; i + kBackwardUnroll <= count; i += kBackwardUnroll) {
    Pack<GradT, N> g[kBackwardUnroll];
    GradT w[kBackwardUnroll];
    int64_t row_next[kBackwardUnroll];
#pragma unroll
    for (int u = 0; u < kBackwardUnroll; ++u) {
      const int64_t sid = static_cast<int64_t>(my_sids[i + u]);
      if constexpr (kWeighted) w[u] = my_w[i + u];
      row_next[u] = static_cast<int64_t>(my_rows[i + u + 1]);
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
      const bool last = (i + u + 1 == count);
      if (last || row_next[u] != row_cur) end_of_run(row_cur, last);
      row_cur = row_next[u];
    }
  }
  for (; i < count; ++i) {
    const int64_t sid = static_cast<int64_t>(my_sids[i]);
    const Pack<GradT, N> g = LoadPack<GradT, N>(lane_src + sid * width);
    const int64_t row_next = static_cast<int64_t>(my_rows[i + 1]);
    if constexpr (kWeighted) {
      const float wf = static_cast<float>(my_w[i]);
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], A::mul(static_cast<float>(g.v[e]), wf));
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], static_cast<float>(g.v[e]));
    }
    const bool last = (i + 1 == count);
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
