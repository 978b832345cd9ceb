// MI355X (gfx950 / CDNA4) small integer kernels behind index_transforms.hpp.
// Counterparts of the reference's index_transforms_kernels.cuh:28-81; all
// results are integers and bit-exact.
#ifndef CUEMBED_INCLUDE_INDEX_KERNELS_HPP_
#define CUEMBED_INCLUDE_INDEX_KERNELS_HPP_

#include <hip/hip_runtime.h>

#include <cstdint>

namespace cuembed {
namespace detail {

constexpr int kSequenceItemsPerThread = 4;
//! Samples whose offsets one workgroup stages in LDS for ExpandCsrKernel, at most.  Small enough that
//! a 65,536-sample batch already gives 512 workgroups (2 per CU); large enough that the
//! workgroup's slice of row_ids (~samples x hotness entries) amortises the staging.  Small batches take fewer
//! samples per workgroup (CsrSamplesPerBlock): 1,024 samples in 8 workgroups took 18.7 us, each lane walking 32
//! positions with a binary search apiece.
constexpr int kCsrSamplesPerBlock = 128;
inline int CsrSamplesPerBlock(const int batch) {
  int per_block = kCsrSamplesPerBlock;
  while (per_block > 4 && batch / per_block < 512) per_block /= 2;   // aim at >= 512 workgroups
  return per_block;
}

//! magic / shift with (uint64(i) * magic) >> shift == i / d for every 0 <= i < 2^31, d >= 1
//! (s = ceil(log2 d), magic = ceil(2^(31+s) / d) < 2^32): a 64-bit integer division per element costs more than the
//! store it feeds (FillQuotientKernel at C4: 7.0 -> see docs/EXPERIMENTS.md).
struct QuotientMagic {
  unsigned magic;
  int shift;
  explicit QuotientMagic(const int d) {
    int s = 0;
    while ((int64_t{1} << s) < d) ++s;
    const unsigned __int128 one = static_cast<unsigned __int128>(1) << (31 + s);
    magic = static_cast<unsigned>((one + d - 1) / d);
    shift = 31 + s;
  }
};

//! out[t] = t / divisor.  Every thread writes kSequenceItemsPerThread CONSECUTIVE items (one 16-byte store for
//! 32-bit ids); positions below 2^31 divide by multiply-shift, larger ones (never reached through the int-sized
//! API) by a real division.
template <typename OutT>
__global__ void FillQuotientKernel(const int64_t count, const int divisor, const unsigned magic, const int shift,
                                   OutT* __restrict__ out) {
  const int64_t t0 = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * kSequenceItemsPerThread;
  if (t0 >= count) return;
  OutT v[kSequenceItemsPerThread];
#pragma unroll
  for (int k = 0; k < kSequenceItemsPerThread; ++k) {
    const int64_t t = t0 + k;
    v[k] = static_cast<OutT>(divisor == 1 ? t
                             : (t < (int64_t{1} << 31)
                                    ? static_cast<int64_t>((static_cast<unsigned long long>(static_cast<unsigned>(t)) * magic) >> shift)
                                    : t / divisor));
  }
  if (t0 + kSequenceItemsPerThread <= count) {
    typedef OutT __attribute__((ext_vector_type(kSequenceItemsPerThread))) vec_t;
    vec_t w;
#pragma unroll
    for (int k = 0; k < kSequenceItemsPerThread; ++k) w[k] = v[k];
    *reinterpret_cast<vec_t*>(out + t0) = w;      // (t0 is a multiple of 4 items: 16- / 32-byte aligned)
  } else {
#pragma unroll
    for (int k = 0; k < kSequenceItemsPerThread; ++k)
      if (t0 + k < count) out[t0 + k] = v[k];
  }
}

//! row_ids[i] = b for i in [offsets[b], offsets[b+1]).
//! The reference launches one 256-thread block per sample for ~hotness elements
//! (index_transforms_kernels.cuh:28-37).  Here a workgroup owns
//! kCsrSamplesPerBlock consecutive samples: their offsets are staged in LDS, and
//! the workgroup then writes its contiguous slice of row_ids with coalesced
//! stores, each lane locating its sample by a binary search in LDS.
template <typename OffsetT, typename IndexT>
__global__ void __launch_bounds__(256)
ExpandCsrKernel(const OffsetT* __restrict__ offsets, const int batch, IndexT* __restrict__ row_ids,
                const int samples_per_block /* <= kCsrSamplesPerBlock */) {
  __shared__ int64_t bounds[kCsrSamplesPerBlock + 1];
  const int first_sample = blockIdx.x * samples_per_block;
  const int nsamples =
      (batch - first_sample < samples_per_block) ? batch - first_sample : samples_per_block;
  for (int s = threadIdx.x; s <= nsamples; s += blockDim.x) {
    bounds[s] = static_cast<int64_t>(offsets[first_sample + s]);
  }
  __syncthreads();
  const int64_t lo_pos = bounds[0];
  const int64_t hi_pos = bounds[nsamples];
  for (int64_t pos = lo_pos + threadIdx.x; pos < hi_pos; pos += blockDim.x) {
    // largest s with bounds[s] <= pos  (empty bags have bounds[s] == bounds[s+1])
    int lo = 0, hi = nsamples;  // invariant: bounds[lo] <= pos < bounds[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (bounds[mid] <= pos) lo = mid;
      else hi = mid;
    }
    row_ids[pos] = static_cast<IndexT>(first_sample + lo);
  }
}

//! Row-cache translation (see TranslateIndicesForRowCache in index_transforms.hpp):
//! out[i] = slot_of_row[idx[i]] >= 0 ? cache_row_offset + slot : idx[i]; ids outside [0, num_rows)
//! are passed through untouched (slot_of_row has num_rows entries).
template <typename IndexT>
__global__ void __launch_bounds__(256)
TranslateForRowCacheKernel(const IndexT* __restrict__ indices, const int64_t count,
                           const int32_t* __restrict__ slot_of_row, const int64_t num_rows,
                           const int64_t cache_row_offset, int64_t* __restrict__ out) {
  const int64_t base =
      static_cast<int64_t>(blockIdx.x) * blockDim.x * kSequenceItemsPerThread + threadIdx.x;
  int64_t row[kSequenceItemsPerThread];
  int32_t slot[kSequenceItemsPerThread];
#pragma unroll
  for (int k = 0; k < kSequenceItemsPerThread; ++k) {   // all index loads, then all table loads, in flight together
    const int64_t t = base + static_cast<int64_t>(k) * blockDim.x;
    row[k] = t < count ? static_cast<int64_t>(indices[t]) : 0;
  }
#pragma unroll
  for (int k = 0; k < kSequenceItemsPerThread; ++k) {
    const int64_t t = base + static_cast<int64_t>(k) * blockDim.x;
    slot[k] = (t < count && row[k] >= 0 && row[k] < num_rows) ? slot_of_row[row[k]] : -1;
  }
#pragma unroll
  for (int k = 0; k < kSequenceItemsPerThread; ++k) {
    const int64_t t = base + static_cast<int64_t>(k) * blockDim.x;
    if (t < count) out[t] = slot[k] >= 0 ? cache_row_offset + slot[k] : row[k];
  }
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_INDEX_KERNELS_HPP_
