// Device-side decisions for EmbeddingForward's scheduling hints (extensions; a hint never changes a result).
//
// The reference picks its launch parameters from the shape alone (embedding_lookup.cuh:186-208: one rule for every
// index distribution).  Two of this library's options depend on the DATA -- non-temporal row loads pay only when the
// rows of a batch are (nearly) all distinct, the bag order only for ragged bags -- and a caller of the C++ / C API
// cannot be asked to read index statistics back to the host in the middle of a step.  Both decisions are taken on the
// device here, in one or two small launches each, without a read-back, so they can sit inside a stream-ordered step and a HIP graph:
//   DecideRowLoadsKernel    counts the distinct rows of an evenly strided sample of the batch exactly (LDS hash
//                           sets) and leaves a flag word in device memory that the forward kernels read;
//   BagChunkHistogramKernel + BagOrderScatterKernel
//                           the samples of a CSR batch by descending bag length -- a stable counting sort on lengths
//                           clamped to 255, two small launches -- for ForwardOptions::sample_order.
#ifndef CUEMBED_INCLUDE_HINT_KERNELS_HPP_
#define CUEMBED_INCLUDE_HINT_KERNELS_HPP_

#include <hip/hip_runtime.h>

#include <cstdint>

#include "cuembed/include/sort_common.hpp"

namespace cuembed {
namespace detail {

// ---------------------------------------------------------------------------
// Row-load decision.  `groups` workgroups; workgroup g takes the sample elements j * groups + g (j < per_group) of an
// evenly strided sample of the lookups and counts how many DIFFERENT rows they name: every key goes into an
// open-addressing hash set in LDS (2 x per_group slots, full keys compared: the count is exact, not an estimate).  The
// last workgroup to arrive adds the counts up and writes the decision -- nobody waits for anybody, so residency does
// not matter.  words[0] = decision (1: streaming), words[1] = arrivals, words[2] = distinct rows (both left at zero).
// ---------------------------------------------------------------------------
constexpr int kDecideThreads = 1024;
constexpr int kDecideGroupSample = 4096;   // lookups per workgroup
constexpr int kDecideSlots = 8192;         // hash slots per workgroup (load factor <= 0.5)
constexpr int kDecideMaxGroups = 16;       // 65,536 sampled lookups at most

template <typename IndexT>
__global__ void __launch_bounds__(kDecideThreads)
DecideRowLoadsKernel(const IndexT* __restrict__ indices, const int64_t nnz, const int per_group,
                     const unsigned threshold_per_65536, uint32_t* __restrict__ words) {
  using Key = typename std::conditional<sizeof(IndexT) == 8, unsigned long long, unsigned>::type;
  __shared__ Key slots[kDecideSlots];
  __shared__ unsigned distinct;
  constexpr Key kEmpty = ~Key(0);
  for (int i = threadIdx.x; i < kDecideSlots; i += kDecideThreads) slots[i] = kEmpty;
  if (threadIdx.x == 0) distinct = 0u;
  __syncthreads();
  const int groups = static_cast<int>(gridDim.x);
  const int64_t sampled = static_cast<int64_t>(per_group) * groups;
  const int64_t stride = nnz / sampled > 0 ? nnz / sampled : 1;
  unsigned mine = 0;
  // (a thread's keys are requested together -- unconditional loads at clamped positions -- before the first is inserted)
  constexpr int kKeysPerThread = kDecideGroupSample / kDecideThreads;
  Key mine_keys[kKeysPerThread];
#pragma unroll
  for (int u = 0; u < kKeysPerThread; ++u) {
    const int64_t at = (static_cast<int64_t>(threadIdx.x + u * kDecideThreads) * groups + blockIdx.x) * stride;
    mine_keys[u] = static_cast<Key>(indices[at < nnz ? at : nnz - 1]);
  }
#pragma unroll
  for (int u = 0; u < kKeysPerThread; ++u) {
    const int j = threadIdx.x + u * kDecideThreads;
    const int64_t at = (static_cast<int64_t>(j) * groups + blockIdx.x) * stride;
    if (j >= per_group || at >= nnz) continue;
    const Key key = mine_keys[u];
    if (key == kEmpty) continue;                                   // (-1 is not a row)
    unsigned h = (static_cast<unsigned>(key) ^ static_cast<unsigned>(static_cast<unsigned long long>(key) >> 32)) * 0x9E3779B1u;
    h >>= 32 - 13;                                                  // kDecideSlots = 2^13
    for (;;) {
      const Key seen = atomicCAS(&slots[h], kEmpty, key);
      if (seen == kEmpty) {
        ++mine;
        break;
      }
      if (seen == key) break;
      h = (h + 1) & (kDecideSlots - 1);
    }
  }
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
  if ((threadIdx.x & 63) == 0 && mine != 0) atomicAdd(&distinct, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(words + 2, distinct, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (release: my count is out before my arrival; acquire: the last one sees every count)
    const unsigned before = __hip_atomic_fetch_add(words + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (before + 1u == static_cast<unsigned>(groups)) {
      const unsigned total = __hip_atomic_load(words + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int64_t taken = sampled < nnz ? sampled : nnz;
      const bool streaming = static_cast<uint64_t>(total) * 65536u >= static_cast<uint64_t>(threshold_per_65536) * static_cast<uint64_t>(taken);
      __hip_atomic_store(words + 0, streaming ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(words + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next call
      __hip_atomic_store(words + 1, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

//! (small batches and small tables: the decision is "default" without looking)
template <int kUnused = 0>   // (a template so that the header can be included from several translation units)
__global__ void ClearRowLoadsDecisionKernel(uint32_t* __restrict__ words) {
  if (threadIdx.x == 0 && blockIdx.x == 0) words[0] = 0u;
}

// ---------------------------------------------------------------------------
// Bag order: sample_order = the samples by DESCENDING min(bag length, 255), ties in input order -- a stable counting sort
// over 256 keys in TWO small launches, chunks of 1,024 samples:
//   BagChunkHistogramKernel  workgroup w counts the keys of ITS chunk (one LDS atomic per group of equal keys in a
//                            wavefront: the sort's MatchDigit) and stores the 256 counts;
//   BagOrderScatterKernel    workgroup w adds up the chunks' counts -- all of them for the key totals, the chunks before
//                            its own for its bases -- and ranks its samples with the wave-synchronous match.
// (A ONE-launch form in which every workgroup counts the whole batch itself, so that nobody depends on anybody, was built
// first: 64 rounds of 1,024 LDS atomics per workgroup at 65,536 samples -- 34 us with one round's offsets in flight, 17 us
// with eight and a 32-fold replicated histogram against same-word queues: LDS atomics retire ~2 lanes per clock.  Two
// dependent launches of a few microseconds each beat it.)  Up to 2^17 samples; BagOrderByLength falls back to the general
// sort above that and for length bounds beyond 255.
// ---------------------------------------------------------------------------
constexpr int kBagOrderThreads = 1024;
constexpr int kBagOrderWaves = kBagOrderThreads / 64;
constexpr int kBagOrderMaxBatch = 1 << 17;

template <typename OffsetT>
__device__ __forceinline__ unsigned BagKeyOf(const OffsetT lo, const OffsetT hi, const int bound) {
  int64_t len = static_cast<int64_t>(hi) - static_cast<int64_t>(lo);
  len = len < 0 ? 0 : (len > bound ? bound : len);
  return static_cast<unsigned>(bound - static_cast<int>(len));     // ascending keys = descending lengths
}

template <typename OffsetT>
__global__ void __launch_bounds__(kBagOrderThreads)
BagChunkHistogramKernel(const OffsetT* __restrict__ offsets, const int batch, const int bound /* 1..255 */,
                        unsigned* __restrict__ chunk_hist /* [chunks][256] */) {
  __shared__ unsigned hist[256];
  const int tid = threadIdx.x;
  if (tid < 256) hist[tid] = 0u;
  __syncthreads();
  const int s = static_cast<int>(blockIdx.x) * kBagOrderThreads + tid;
  const bool valid = s < batch;
  const unsigned key = valid ? BagKeyOf(offsets[s], offsets[s + 1], bound) : 0u;
  const unsigned long long peers = MatchDigit(key, valid);
  if (valid && CountBelow(peers) == 0u) atomicAdd(&hist[key], static_cast<unsigned>(__popcll(peers)));
  __syncthreads();
  if (tid < 256) chunk_hist[static_cast<size_t>(blockIdx.x) * 256 + tid] = hist[tid];
}

template <typename OffsetT>
__global__ void __launch_bounds__(kBagOrderThreads)
BagOrderScatterKernel(const OffsetT* __restrict__ offsets, const int batch, const int bound,
                      const unsigned* __restrict__ chunk_hist, int32_t* __restrict__ sample_order) {
  __shared__ unsigned part_before[4][256], part_total[4][256];
  __shared__ unsigned base_of[256];                       // first output position of this workgroup's samples with key k
  __shared__ unsigned wave_count[kBagOrderWaves][256];
  __shared__ unsigned scan_carry[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunks = static_cast<int>(gridDim.x), me = static_cast<int>(blockIdx.x);
  for (int i = tid; i < kBagOrderWaves * 256; i += kBagOrderThreads) (&wave_count[0][0])[i] = 0u;
  // my sample's key first: its two loads are in flight while the chunk counts are added up
  const int s = me * kBagOrderThreads + tid;
  const bool valid = s < batch;
  OffsetT lo = OffsetT(0), hi = OffsetT(0);
  if (valid) {
    lo = offsets[s];
    hi = offsets[s + 1];
  }
  {  // thread (k, quarter q) adds up the counts of key k over chunks q, q + 4, q + 8, ...: independent loads
    const int k = tid & 255, q = tid >> 8;
    unsigned before = 0u, total = 0u;
    // (eight unconditional loads at clamped chunk numbers in flight, then the masked adds: one load per trip of a loop
    // with a runtime bound is one exposed L2 latency per chunk)
    for (int c0 = q; c0 < chunks; c0 += 4 * 8) {
      unsigned h[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + 4 * u;
        h[u] = chunk_hist[static_cast<size_t>(c < chunks ? c : chunks - 1) * 256 + k];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int c = c0 + 4 * u;
        total += c < chunks ? h[u] : 0u;
        before += c < me ? h[u] : 0u;       // (me < chunks)
      }
    }
    part_before[q][k] = before;
    part_total[q][k] = total;
  }
  __syncthreads();
  // base of key k = samples with a smaller key (whole batch) + samples with key k before this workgroup
  unsigned mine = 0u, incl = 0u, before = 0u;
  if (tid < 256) {
    mine = part_total[0][tid] + part_total[1][tid] + part_total[2][tid] + part_total[3][tid];
    before = part_before[0][tid] + part_before[1][tid] + part_before[2][tid] + part_before[3][tid];
    incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned up = __shfl_up(incl, off);
      if (lane >= off) incl += up;
    }
    if (lane == 63) scan_carry[wave] = incl;
  }
  __syncthreads();
  if (tid < 256) {
    unsigned carry = 0u;
    for (int w = 0; w < wave; ++w) carry += scan_carry[w];
    base_of[tid] = carry + incl - mine + before;
  }
  // rank of my sample among the workgroup's samples with the same key: lanes below me in my wavefront + earlier wavefronts
  const unsigned key = valid ? BagKeyOf(lo, hi, bound) : 0u;
  const unsigned long long peers = MatchDigit(key, valid);
  const unsigned below = CountBelow(peers);
  if (valid && below == 0u) wave_count[wave][key] = static_cast<unsigned>(__popcll(peers));
  __syncthreads();
  if (!valid) return;
  unsigned pos = base_of[key] + below;
  for (int w = 0; w < wave; ++w) pos += wave_count[w][key];
  sample_order[pos] = s;
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_HINT_KERNELS_HPP_
