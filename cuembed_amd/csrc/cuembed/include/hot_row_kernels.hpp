// MI355X (gfx950 / CDNA4) kernels for the RUN-AWARE embedding backward (an extension next to
// EmbeddingBackward, see EmbeddingBackwardRunAware in embedding_backward.hpp).
//
// Why.  At the north-star shape (10M x 256 fp16, batch 65536, hotness 64, alpha 1.15) a third of
// the 4.19M sorted lookups sit in ~60 runs longer than 8192 -- rows that almost every sample looks
// up.  SegmentedScatterAddKernel walks such a run like any other: every workgroup on it gathers
// `grad_y[sample]` rows from L2 / the fabric, so each of those runs streams most of grad_y
// (33.5 MB) through the fabric again: ~0.7 GB of the 1.65 GB the kernel moves for 0.36 GB of
// algorithmic bytes.  (The reference has no counterpart: embedding_lookup_kernels.cuh:175-220
// treats all runs alike and pays for the hot ones with serialised atomics.)
//
// What.  Inside a run the lookups are ordered by sample id (Transpose is stable).  So the hot
// runs are processed SAMPLE-CHUNK-major instead of run-major:
//   HotRunDetectKernel    finds the runs that contain two consecutive multiples of `stride`
//                         positions (>= stride + 1 lookups, certainly every run of 2 * stride - 1)
//                         -- at most kHotMaxRuns of them -- and records for each the part that
//                         covers whole nz-blocks of the segmented kernel;
//   HotRowChunkSumKernel  one workgroup per chunk of consecutive samples: the chunk's grad_y rows
//                         (128 KiB) are read ONCE into LDS, then for every hot run the lookups whose
//                         sample lies in the chunk (a contiguous piece of the run, found by binary
//                         search) are summed out of LDS into an fp32 partial row in the workspace;
//   SegmentedScatterAddKernel (scatter_add_kernels.hpp) skips the nz-blocks that lie inside a hot
//                         run, and kHotMaxRuns extra workgroups at the front of its grid add up
//                         the partial rows of one hot run each (in chunk order) and add the total
//                         to the output row with ONE float atomic per element.
// grad_y is then read from the fabric once for all hot runs together.
#ifndef CUEMBED_INCLUDE_HOT_ROW_KERNELS_HPP_
#define CUEMBED_INCLUDE_HOT_ROW_KERNELS_HPP_

#include "cuembed/include/embedding_types.hpp"
#include "cuembed/include/gather_reduce_kernels.hpp"

namespace cuembed {
namespace detail {

constexpr int kHotMaxRuns = 64;            //!< hot runs handled per call; further ones stay in the segmented kernel
constexpr int kHotMinStride = 4096;        //!< a hot run contains positions k * stride and (k + 1) * stride
constexpr int kHotMaxMultiples = 2048;     //!< stride grows with nnz so that nnz / stride stays below this
constexpr int kHotPieces = 2;              //!< partial rows per (chunk, run): long pieces are cut in two for balance
constexpr int kHotSplitLookups = 96;       //!< ... when the run has more than this many lookups per chunk on average
constexpr int kHotChunkBytes = 128 * 1024; //!< grad_y bytes staged in LDS per workgroup (of 160 KiB per CU)
constexpr int kHotThreads = 1024;
constexpr int kHotWaves = kHotThreads / 64;
constexpr int kHotMaxChunks = 1024;        //!< bounds the workspace; more samples -> several chunks per workgroup
constexpr int kHotDetectThreads = 256;

struct HotRun {
  int row;          //!< output row (dense id in a compressed gradient, table row otherwise)
  int begin;        //!< position of the run's first lookup
  int first_block;  //!< nz-blocks [first_block, end_block) of the segmented kernel lie inside the run ...
  int end_block;    //!< ... and are summed by HotRowChunkSumKernel instead
};

struct HotRunTable {
  int count;      //!< valid entries of run[]
  int block_len;  //!< lookups per nz-block of the segmented kernel
  int pad[2];
  HotRun run[kHotMaxRuns];
};

inline int HotStride(const int64_t nnz) {
  int64_t s = kHotMinStride;
  while ((nnz + s - 1) / s > kHotMaxMultiples) s *= 2;
  return static_cast<int>(s);
}

//! Pieces a hot run's per-chunk work is cut into: a pure function of the table entry, so that
//! the chunk kernel and the combining workgroups agree without talking.
__host__ __device__ inline int HotPiecesOf(const HotRun& r, const int block_len, const int samples_per_chunk,
                                           const int num_samples) {
  const int64_t lookups = static_cast<int64_t>(r.end_block - r.first_block) * block_len;
  return lookups * samples_per_chunk > static_cast<int64_t>(kHotSplitLookups) * num_samples ? kHotPieces : 1;
}

// ---------------------------------------------------------------------------------------------
// Detection.  grid = number of multiples of `stride` below nnz; block = kHotDetectThreads.
// Workgroup k looks at the row ids at ALL multiples (a few KB, L2-resident after the first
// workgroup), decides whether a hot run STARTS at multiple k, and if so finds the run's first
// and last position with two cooperative 256-ary searches (2 dependent loads each).  The slot of
// a run is its rank among the hot runs (counted from the same multiples), so nothing has to be
// zeroed or allocated atomically and the table is the same whatever the scheduling.
// ---------------------------------------------------------------------------------------------

//! Smallest p in [lo, hi) with pred(p), else hi; pred is monotone (false ... false true ... true).
//! Cooperative over the workgroup's kHotDetectThreads threads; contains barriers.
template <typename Pred>
__device__ __forceinline__ int CoopFirstTrue(int lo, int hi, int* scratch, Pred pred) {
  // invariant: pred is false below lo; hi is the end of the range or a position known to satisfy pred
  const int tid = threadIdx.x;
  while (true) {
    const int n = hi - lo;
    if (n <= 0) return hi;
    const int step = (n + kHotDetectThreads - 1) / kHotDetectThreads;
    const int p = lo + tid * step;
    if (tid == 0) *scratch = kHotDetectThreads;
    __syncthreads();
    if (p < hi && pred(p)) atomicMin(scratch, tid);
    __syncthreads();
    const int f = *scratch;  // first probe that is true, kHotDetectThreads if none
    __syncthreads();
    const int probes = (n - 1) / step + 1;  // probes that lie below hi
    const int last_false = (f < kHotDetectThreads ? f : probes) - 1;
    if (f < kHotDetectThreads) hi = lo + f * step;
    if (last_false >= 0) lo = lo + last_false * step + 1;
    if (step == 1) return hi;  // every position of the range was probed
  }
}

template <typename IndexT>
__global__ void __launch_bounds__(kHotDetectThreads)
HotRunDetectKernel(const IndexT* __restrict__ rows, const int nnz, const int stride, const int block_len,
                   HotRunTable* __restrict__ table) {
  __shared__ int mult[kHotMaxMultiples + 1];
  __shared__ int scratch;
  __shared__ int red[kHotDetectThreads / 64];
  const int tid = threadIdx.x;
  const int k = blockIdx.x;
  const int num_mult = (nnz - 1) / stride + 1;  // multiples k * stride < nnz
  for (int j = tid; j < num_mult; j += kHotDetectThreads)
    mult[j] = static_cast<int>(rows[static_cast<int64_t>(j) * stride]);
  __syncthreads();
  auto is_start = [&](int j) {
    return j + 1 < num_mult && mult[j] == mult[j + 1] && (j == 0 || mult[j - 1] != mult[j]);
  };
  auto block_sum = [&](int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int w = 0; w < kHotDetectThreads / 64; ++w) s += red[w];
    return s;
  };
  if (k == 0) {  // the table header is written by one workgroup, whatever it finds
    int c = 0;
    for (int j = tid; j < num_mult; j += kHotDetectThreads) c += is_start(j) ? 1 : 0;
    const int total = block_sum(c);
    if (tid == 0) {
      table->count = total < kHotMaxRuns ? total : kHotMaxRuns;
      table->block_len = block_len;
    }
  }
  if (!is_start(k)) return;
  int c = 0;
  for (int j = tid; j < k; j += kHotDetectThreads) c += is_start(j) ? 1 : 0;
  const int slot = block_sum(c);
  if (slot >= kHotMaxRuns) return;
  const int v = mult[k];
  // last multiple that still holds v (equal multiples are contiguous: the ids are sorted)
  int m_local = k;
  for (int j = k + tid; j < num_mult; j += kHotDetectThreads)
    if (mult[j] == v && j > m_local) m_local = j;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const int o = __shfl_xor(m_local, d);
    m_local = o > m_local ? o : m_local;
  }
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = m_local;
  __syncthreads();
  int m = k;
#pragma unroll
  for (int w = 0; w < kHotDetectThreads / 64; ++w) m = red[w] > m ? red[w] : m;
  // first lookup of the run: in ((k - 1) * stride, k * stride]; first lookup after it: in
  // (m * stride, (m + 1) * stride] (or the end of the arrays)
  const int lo_b = k == 0 ? 0 : (k - 1) * stride + 1;
  const int begin = CoopFirstTrue(lo_b, k * stride, &scratch,
                                  [&](int p) { return static_cast<int>(rows[p]) == v; });
  const int64_t next_mult = static_cast<int64_t>(m + 1) * stride;
  const int hi_e = next_mult < nnz ? static_cast<int>(next_mult) : nnz;
  const int end = CoopFirstTrue(m * stride + 1, hi_e, &scratch,
                                [&](int p) { return static_cast<int>(rows[p]) != v; });
  if (tid == 0) {
    HotRun r;
    r.row = v;
    r.begin = begin;
    r.first_block = (begin + block_len - 1) / block_len;
    r.end_block = end / block_len;
    if (r.end_block < r.first_block) r.end_block = r.first_block;
    table->run[slot] = r;
  }
}

// ---------------------------------------------------------------------------------------------
// Chunk sums.  grid = number of sample chunks (<= kHotMaxChunks; a workgroup loops over several
// LDS fills when there are more samples), block = kHotThreads, dynamic LDS = the staged rows.
//   partial[((chunk * kHotMaxRuns + slot) * kHotPieces + piece) * width + column]   (fp32)
// ---------------------------------------------------------------------------------------------
template <typename GradT, typename IndexT, int N, bool kWeighted>
__global__ void __launch_bounds__(kHotThreads)
HotRowChunkSumKernel(const GradT* __restrict__ grad_y, const int width, const int num_samples,
                     const IndexT* __restrict__ sample_ids, const GradT* __restrict__ weights,
                     const HotRunTable* __restrict__ table, float* __restrict__ partial,
                     const int samples_per_fill, const int fills_per_chunk, const int lanes_per_row) {
  using A = Arith<float>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  GradT* staged = reinterpret_cast<GradT*>(lds_raw);
  __shared__ int piece_lo[kHotMaxRuns * kHotPieces];   // per item: lookups [lo, hi) of this fill
  __shared__ int piece_hi[kHotMaxRuns * kHotPieces];
  __shared__ int order[kHotMaxRuns * kHotPieces];      // items by decreasing length
  __shared__ int bound[kHotMaxRuns][2];
  __shared__ HotRun run[kHotMaxRuns];
  __shared__ int next_item;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int count = table->count;
  if (count == 0) return;
  const int block_len = table->block_len;
  if (tid < count) run[tid] = table->run[tid];
  const int samples_per_chunk = samples_per_fill * fills_per_chunk;
  const int groups = 64 / lanes_per_row;   // lookups a wavefront handles at a time
  const int g = lane / lanes_per_row;      // this lane's sub-group and position in the row
  const int l = lane - g * lanes_per_row;
  float acc[N];

  for (int fill = 0; fill < fills_per_chunk; ++fill) {
    const int64_t s0 = static_cast<int64_t>(blockIdx.x) * samples_per_chunk + static_cast<int64_t>(fill) * samples_per_fill;
    if (s0 >= num_samples) break;
    const int64_t s1 = s0 + samples_per_fill < num_samples ? s0 + samples_per_fill : num_samples;
    __syncthreads();  // the previous fill's rows and item lists are no longer in use (and run[] is visible)
    {  // ---- the chunk's grad_y rows: one contiguous range, 16 bytes per lane ----
      typedef unsigned __attribute__((ext_vector_type(4))) raw16_t;
      const raw16_t* src = reinterpret_cast<const raw16_t*>(grad_y + s0 * width);
      raw16_t* dst = reinterpret_cast<raw16_t*>(staged);
      const int64_t n16 = (s1 - s0) * width * static_cast<int64_t>(sizeof(GradT)) / 16;
      for (int64_t i = tid; i < n16; i += kHotThreads) dst[i] = __builtin_nontemporal_load(src + i);
    }
    // ---- where the chunk's samples sit inside every hot run (ids ascend inside a run) ----
    if (tid < 2 * count) {
      const HotRun r = run[tid >> 1];
      const int64_t target = (tid & 1) ? s1 : s0;
      int lo = r.first_block * block_len, hi = r.end_block * block_len;
      while (lo < hi) {
        const int mid = lo + ((hi - lo) >> 1);
        if (static_cast<int64_t>(sample_ids[mid]) < target) lo = mid + 1;
        else hi = mid;
      }
      bound[tid >> 1][tid & 1] = lo;
    }
    if (tid == 0) next_item = 0;
    __syncthreads();
    // ---- work items: (run, piece); longest first, taken by the wavefronts as they finish ----
    const int items = count * kHotPieces;
    if (tid < items) {
      const int h = tid / kHotPieces, p = tid - h * kHotPieces;
      const int pieces = HotPiecesOf(run[h], block_len, samples_per_chunk, num_samples);
      const int lo = bound[h][0], hi = bound[h][1];
      const int cut = pieces == 2 ? lo + ((hi - lo + 1) >> 1) : hi;
      piece_lo[tid] = p == 0 ? lo : cut;
      piece_hi[tid] = p == 0 ? cut : (pieces == 2 ? hi : cut);   // an unused second piece is empty
    }
    __syncthreads();
    if (tid < items) {
      const int len = piece_hi[tid] - piece_lo[tid];
      int rank = 0;
      for (int u = 0; u < items; ++u) {
        const int lu = piece_hi[u] - piece_lo[u];
        rank += (lu > len || (lu == len && u < tid)) ? 1 : 0;
      }
      order[rank] = tid;
    }
    __syncthreads();
    while (true) {
      int it = 0;
      if (lane == 0) it = atomicAdd(&next_item, 1);
      it = __builtin_amdgcn_readfirstlane(it);
      if (it >= items) break;
      const int item = order[it];
      const int h = item / kHotPieces, p = item - h * kHotPieces;
      const int pieces = HotPiecesOf(run[h], block_len, samples_per_chunk, num_samples);
      if (p >= pieces) continue;   // never written, never read
      const int a = piece_lo[item], b = piece_hi[item];
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = 0.f;
      for (int base = a; base < b; base += 64) {
        const int pos = base + lane;
        int sid_l = 0;
        GradT w_l = static_cast<GradT>(0);
        if (pos < b) {
          sid_l = static_cast<int>(static_cast<int64_t>(sample_ids[pos]) - s0);
          if constexpr (kWeighted) w_l = weights[pos];
        }
        const int cnt = b - base < 64 ? b - base : 64;
#pragma unroll 4
        for (int u = 0; u < cnt; u += groups) {
          const int j = u + g;
          const int sid = __shfl(sid_l, j < 64 ? j : 63);
          const Pack<GradT, N> row = *reinterpret_cast<const Pack<GradT, N>*>(
              staged + static_cast<size_t>(sid) * width + l * N);
          if constexpr (kWeighted) {
            const float wf = static_cast<float>(ShuffleElem(w_l, j < 64 ? j : 63, 64));
            if (j < cnt) {
#pragma unroll
              for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], A::mul(static_cast<float>(row.v[e]), wf));
            }
          } else {
            if (j < cnt) {
#pragma unroll
              for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], static_cast<float>(row.v[e]));
            }
          }
        }
      }
      // fold the sub-groups of the wavefront, then one of them writes the partial row
      for (int off = lanes_per_row; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], __shfl_xor(acc[e], off));
      }
      if (g == 0) {
        float* dst = partial + ((static_cast<size_t>(blockIdx.x) * kHotMaxRuns + h) * kHotPieces + p) * width + l * N;
        if (fill > 0) {  // several LDS fills per chunk: the chunk's partial accumulates over them
#pragma unroll
          for (int e = 0; e < N; ++e) acc[e] = A::add(dst[e], acc[e]);
        }
#pragma unroll
        for (int e = 0; e < N; ++e) dst[e] = acc[e];
      }
    }
  }
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_HOT_ROW_KERNELS_HPP_
