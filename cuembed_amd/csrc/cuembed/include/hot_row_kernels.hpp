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
//                         covers whole nz-blocks of the segmented kernel, and where every sample
//                         chunk begins inside it;
//   HotRowChunkSumKernel  one workgroup per chunk of consecutive samples: the chunk's grad_y rows
//                         (128 KiB) are read ONCE into LDS, then for every hot run the lookups whose
//                         sample lies in the chunk (a contiguous piece of the run; the detection
//                         pass recorded where) are summed out of LDS into an fp32 partial row;
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
constexpr int kHotMaxChunks = 1024;        //!< bounds the workspace; more samples -> several chunks per workgroup
constexpr int kHotDetectThreads = 1024;    //!< only the few workgroups that find a hot run use them all
constexpr int kHotBatch = 8;               //!< LDS row reads a wavefront keeps in flight

struct HotRun {
  int row;          //!< output row (dense id in a compressed gradient, table row otherwise)
  int begin;        //!< position of a lookup of the run (= first_block * block_len): where its table row id is read
  int first_block;  //!< nz-blocks [first_block, end_block) of the segmented kernel lie inside the run ...
  int end_block;    //!< ... and are summed by HotRowChunkSumKernel instead
};

struct HotRunTable {
  int count;      //!< valid entries of run[]
  int block_len;  //!< lookups per nz-block of the segmented kernel
  int pad[2];
  HotRun run[kHotMaxRuns];
};

//! Detection stride: about nnz / 512 (a run has to hold ~0.2-0.4 % of all lookups to be worth a
//! slot: at the north-star shape 8192, which selects 58 runs = 32 % of the lookups; 4096 would
//! select 109 and fill the 64 slots with the first ones in row order rather than the longest).
inline int HotStride(const int64_t nnz) {
  int64_t s = kHotMinStride;
  while ((nnz + s - 1) / s > 512) s *= 2;
  return static_cast<int>(s);
}

//! magic / shift with (uint64(i) * magic) >> shift == i / d for 0 <= i < 2^31, d >= 1
//! (s = ceil(log2 d), magic = ceil(2^(31+s) / d) < 2^32; the error stays below 1 / d).
inline void HotFillDivisor(const int d, unsigned* magic, int* shift) {
  int s = 0;
  while ((int64_t{1} << s) < d) ++s;
  const unsigned __int128 one = static_cast<unsigned __int128>(1) << (31 + s);
  *magic = static_cast<unsigned>((one + d - 1) / d);
  *shift = 31 + s;
}

//! Pieces a hot run's per-chunk work is cut into: a pure function of the table entry, so that
//! the chunk kernel and the combining workgroups agree without talking.
__host__ __device__ inline int HotPiecesOf(const HotRun& r, const int block_len, const int samples_per_chunk,
                                           const int num_samples) {
  const int64_t lookups = static_cast<int64_t>(r.end_block - r.first_block) * block_len;
  return lookups * samples_per_chunk > static_cast<int64_t>(kHotSplitLookups) * num_samples ? kHotPieces : 1;
}

// ---------------------------------------------------------------------------------------------
// Detection.  grid = number of multiples of `stride` below nnz; block = kHotDetectThreads;
// `stride` is a multiple of `block_len`.
// Workgroup k reads the row ids at multiples k - 1, k, k + 1 and leaves at once unless a hot run
// STARTS at multiple k (workgroup 0 stays to write the table header).  A starting workgroup then
//   1. reads the ids at all multiples (a few KB): its slot is its rank among the starting
//      multiples -- nothing is zeroed or allocated atomically, the table does not depend on
//      scheduling -- and the run's last multiple follows from the same array;
//   2. finds the nz-blocks that lie inside the run: the first one starts in the stride before
//      multiple k, the last one ends in the stride after the run's last multiple -- stride /
//      block_len candidates each, probed in one step;
//   3. walks the sample ids of those blocks once and records, for every LDS fill of
//      HotRowChunkSumKernel (samples [f * samples_per_fill, (f + 1) * samples_per_fill)), where the
//      fill's lookups begin: ids ascend inside a run, so that is where sid / samples_per_fill
//      changes.  bounds[slot * (num_fills + 1) + f] = first position of fill f (.. + num_fills = end).
// ---------------------------------------------------------------------------------------------
template <typename IndexT>
__global__ void __launch_bounds__(kHotDetectThreads)
HotRunDetectKernel(const IndexT* __restrict__ rows, const IndexT* __restrict__ sample_ids, const int nnz,
                   const int stride, const int block_len, const unsigned fill_magic,
                   const int fill_shift, const int num_fills, HotRunTable* __restrict__ table,
                   int* __restrict__ bounds) {
  __shared__ int mult[kHotMaxMultiples + 1];
  __shared__ int red[kHotDetectThreads / 64];
  __shared__ int first_last[2];
  const int tid = threadIdx.x;
  const int k = blockIdx.x;
  const int num_mult = (nnz - 1) / stride + 1;  // multiples k * stride < nnz
  {
    const int v = static_cast<int>(rows[static_cast<int64_t>(k) * stride]);
    const bool same_next = k + 1 < num_mult && static_cast<int>(rows[static_cast<int64_t>(k + 1) * stride]) == v;
    const bool same_prev = k > 0 && static_cast<int>(rows[static_cast<int64_t>(k - 1) * stride]) == v;
    if (k != 0 && !(same_next && !same_prev)) return;
  }
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
    if (!is_start(0)) return;
  }
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
  if (tid == 0) {
    first_last[0] = k * (stride / block_len);  // the block that starts AT multiple k is inside the run ...
    first_last[1] = m * (stride / block_len);  // ... and so is the one that ends just before multiple m
  }
  __syncthreads();
  // nz-block j covers [j * block_len, (j + 1) * block_len); it lies inside the run when it starts
  // and ends on v.  Blocks between multiples k and m do; the candidates are the blocks of the
  // stride before multiple k and of the stride after multiple m.
  const int ratio = stride / block_len;
  for (int i = tid; i < ratio; i += kHotDetectThreads) {
    if (k > 0) {
      const int j = (k - 1) * ratio + 1 + i;                     // starts in ((k-1) stride, k stride]
      if (static_cast<int>(rows[static_cast<int64_t>(j) * block_len]) == v) atomicMin(&first_last[0], j);
    }
    const int64_t j = static_cast<int64_t>(m) * ratio + i;       // ends in (m stride, (m+1) stride]
    const int64_t last = (j + 1) * block_len - 1;
    if (last < nnz && static_cast<int>(rows[last]) == v) atomicMax(&first_last[1], static_cast<int>(j + 1));
  }
  __syncthreads();
  const int first_block = first_last[0];
  const int end_block = first_last[1] > first_block ? first_last[1] : first_block;
  if (tid == 0) {
    HotRun r;
    r.row = v;
    r.begin = first_block * block_len;  // some lookup of the run; its first one when that sits in a skipped block
    r.first_block = first_block;
    r.end_block = end_block;
    table->run[slot] = r;
  }
  // where every LDS fill's samples begin inside [lo, hi)
  const int lo = first_block * block_len, hi = end_block * block_len;
  int* my_bounds = bounds + static_cast<size_t>(slot) * (num_fills + 1);
  if (hi <= lo) {
    for (int f = tid; f <= num_fills; f += kHotDetectThreads) my_bounds[f] = lo;
    return;
  }
  // sid / samples_per_fill as a multiply-shift (sample ids are below 2^31): see HotFillDivisor
  auto fill_of = [&](const IndexT sid) {
    return static_cast<int>((static_cast<unsigned long long>(static_cast<unsigned>(sid)) * fill_magic) >> fill_shift);
  };
  // kScanBatch loads per thread are issued before the first is used: the walk is a dependent
  // chain of memory round trips otherwise (the stores below keep the compiler from overlapping them)
  // (8, not more: with 32 the detection itself is no faster, and the SEGMENTED kernel of the same
  // translation unit measured 0.194 -> 0.220 ms -- same source, different code placement)
  constexpr int kScanBatch = 8;
  for (int base = lo + tid; base < hi; base += kScanBatch * kHotDetectThreads) {
    IndexT cur[kScanBatch], prev[kScanBatch];
#pragma unroll
    for (int u = 0; u < kScanBatch; ++u) {
      const int p = base + u * kHotDetectThreads;
      cur[u] = p < hi ? sample_ids[p] : IndexT(0);
      prev[u] = (p < hi && p > lo) ? sample_ids[p - 1] : IndexT(0);
    }
#pragma unroll
    for (int u = 0; u < kScanBatch; ++u) {
      const int p = base + u * kHotDetectThreads;
      if (p >= hi) break;
      const int f1 = fill_of(cur[u]);
      const int f0 = p == lo ? -1 : fill_of(prev[u]);
      for (int f = f0 + 1; f <= f1; ++f) my_bounds[f] = p;
      if (p == hi - 1)
        for (int f = f1 + 1; f <= num_fills; ++f) my_bounds[f] = hi;
    }
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
                     const HotRunTable* __restrict__ table, const int* __restrict__ bounds,
                     float* __restrict__ partial, const int samples_per_fill, const int fills_per_chunk,
                     const int num_fills, const int lanes_per_row) {
  using A = Arith<float>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ int piece_lo[kHotMaxRuns * kHotPieces];   // per item: lookups [lo, hi) of this fill
  __shared__ int piece_hi[kHotMaxRuns * kHotPieces];
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
  const int row_bytes = width * static_cast<int>(sizeof(GradT));
  const unsigned char* lane_src = lds_raw + l * N * static_cast<int>(sizeof(GradT));
  const int items = count * kHotPieces;
  float acc[N];

  for (int fill = 0; fill < fills_per_chunk; ++fill) {
    const int f = blockIdx.x * fills_per_chunk + fill;
    const int64_t s0 = static_cast<int64_t>(f) * samples_per_fill;
    if (s0 >= num_samples) break;
    const int64_t s1 = s0 + samples_per_fill < num_samples ? s0 + samples_per_fill : num_samples;
    __syncthreads();  // the previous fill's rows and item lists are no longer in use (and run[] is visible)
    // ---- work items of this fill: (run, piece), from the positions the detection pass recorded ----
    if (tid < items) {
      const int h = tid / kHotPieces, p = tid - h * kHotPieces;
      const int pieces = HotPiecesOf(run[h], block_len, samples_per_chunk, num_samples);
      const int* b = bounds + static_cast<size_t>(h) * (num_fills + 1) + f;
      const int lo = b[0], hi = b[1];
      const int cut = pieces == 2 ? lo + ((hi - lo + 1) >> 1) : hi;
      piece_lo[tid] = p == 0 ? lo : cut;
      piece_hi[tid] = p == 0 ? cut : (pieces == 2 ? hi : cut);   // an unused second piece is empty
    }
    if (tid == 0) next_item = 0;
    {  // ---- the fill's grad_y rows: one contiguous range, 16 bytes per lane ----
      typedef unsigned __attribute__((ext_vector_type(4))) raw16_t;
      const raw16_t* src = reinterpret_cast<const raw16_t*>(grad_y + s0 * width);
      raw16_t* dst = reinterpret_cast<raw16_t*>(lds_raw);
      const int64_t n16 = (s1 - s0) * width * static_cast<int64_t>(sizeof(GradT)) / 16;
      for (int64_t i = tid; i < n16; i += kHotThreads) dst[i] = __builtin_nontemporal_load(src + i);
    }
    __syncthreads();
    while (true) {
      int it = 0;
      if (lane == 0) it = atomicAdd(&next_item, 1);
      it = __builtin_amdgcn_readfirstlane(it);
      if (it >= items) break;
      const int item = it;   // table order: ~7 items per wavefront even out without sorting them by length
      const int h = item / kHotPieces, p = item - h * kHotPieces;
      const int pieces = HotPiecesOf(run[h], block_len, samples_per_chunk, num_samples);
      if (p >= pieces) continue;   // never written, never read
      const int a = piece_lo[item], b = piece_hi[item];
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = 0.f;
      for (int base = a; base < b; base += 64) {
        // 64 lookups per round: every lane fetches one (sample offset, weight) pair, coalesced;
        // the sub-groups then take them `groups` at a time through cross-lane reads
        const int pos = base + lane;
        int off_l = 0;
        float wf_l = 0.f;
        if (pos < b) {
          off_l = static_cast<int>(static_cast<int64_t>(sample_ids[pos]) - s0) * row_bytes;
          if constexpr (kWeighted) wf_l = static_cast<float>(weights[pos]);
        }
        const int cnt = b - base < 64 ? b - base : 64;
        const int full = cnt / groups * groups;  // steps in which every sub-group has a lookup
        int u = 0;
        // kHotBatch steps at a time: all cross-lane reads, then all LDS row reads, then the adds --
        // so that a wavefront has several LDS reads in flight instead of one dependent chain per step
        for (; u + kHotBatch * groups <= full; u += kHotBatch * groups) {
          int off[kHotBatch];
          float wf[kHotBatch];
          Pack<GradT, N> row[kHotBatch];
#pragma unroll
          for (int t = 0; t < kHotBatch; ++t) {
            off[t] = __shfl(off_l, u + t * groups + g);
            wf[t] = kWeighted ? __shfl(wf_l, u + t * groups + g) : 1.f;
          }
#pragma unroll
          for (int t = 0; t < kHotBatch; ++t) row[t] = *reinterpret_cast<const Pack<GradT, N>*>(lane_src + off[t]);
#pragma unroll
          for (int t = 0; t < kHotBatch; ++t) AccumulateRow<GradT, N, kWeighted>(acc, row[t], wf[t]);
        }
        for (; u < full; u += groups) {
          const int off = __shfl(off_l, u + g);
          const float wf = kWeighted ? __shfl(wf_l, u + g) : 1.f;
          const Pack<GradT, N> row = *reinterpret_cast<const Pack<GradT, N>*>(lane_src + off);
          AccumulateRow<GradT, N, kWeighted>(acc, row, wf);
        }
        if (u < cnt) {
          const int j = u + g;
          const int off = __shfl(off_l, j < 64 ? j : 63);
          const float wf = kWeighted ? __shfl(wf_l, j < 64 ? j : 63) : 1.f;
          if (j < cnt) {
            const Pack<GradT, N> row = *reinterpret_cast<const Pack<GradT, N>*>(lane_src + off);
            AccumulateRow<GradT, N, kWeighted>(acc, row, wf);
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
