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
//   * the hotness loop is batched: kForwardUnroll independent row loads are issued
//     back-to-back (8 KiB in flight per wave) before the first one is consumed, the
//     tail of a bag included (one predicated batch, not one latency per leftover row);
//   * accumulation is strictly in lookup order, one unfused add (and one unfused
//     multiply when weighted) per element, so results are bit-identical to the
//     sequential host loop of the reference (embedding_lookup_cpu.hpp:57-93);
//   * indices reach the lanes in one of three ways (IndexSource): staged once per
//     workgroup in LDS (fixed hotness), fetched 64-wide by the sample's own lanes and
//     handed around with cross-lane reads (CSR, lanes_per_row | 64), or read per lookup
//     as a wave-broadcast load (any other row split);
//   * small batches can opt into GatherReduceSplitKernel: the hotness loop of one sample
//     split over the waves of a workgroup, partial pooled rows combined through LDS;
//   * pooled rows leave with non-temporal stores; all row addressing is 64-bit
//     (a 10M x 256 table is 2.56 G elements).
// Measured on MI355X the kernel sits on the memory system's own limits in every regime
// (L2-resident table 25-28 TB/s, Infinity-Cache-resident 8.4 TB/s, HBM 6.1 TB/s for random
// 512-byte rows); unroll depth, workgroup shape and occupancy do not move it (docs/EXPERIMENTS.md).
// No MFMA: the op is a bandwidth-bound gather, there is no contraction to feed.
#ifndef CUEMBED_INCLUDE_GATHER_REDUCE_KERNELS_HPP_
#define CUEMBED_INCLUDE_GATHER_REDUCE_KERNELS_HPP_

#include "cuembed/include/embedding_types.hpp"

namespace cuembed {
namespace detail {

constexpr int kForwardUnroll = 8;
constexpr bool kForwardPipelined = false;
constexpr int kMaxBlockThreads = 1024;

//! Where a lane finds the lookup indices of its sample.
enum class IndexSource {
  kLdsStaged,    //!< fixed hotness, indices of the whole workgroup staged in LDS
  kWaveShuffle,  //!< CSR / large hotness, lanes_per_row divides 64: register-staged, cross-lane reads
  kGlobal        //!< CSR / large hotness, any row split: per-lookup broadcast loads
};

__device__ __forceinline__ float ShuffleElem(float v, int src, int width) {
  return __shfl(v, src, width);
}
__device__ __forceinline__ _Float16 ShuffleElem(_Float16 v, int src, int width) {
  const int bits = __shfl(static_cast<int>(__builtin_bit_cast(unsigned short, v)), src, width);
  return __builtin_bit_cast(_Float16, static_cast<unsigned short>(bits));
}

//! XCD-aware work placement.  An MI355X has 8 XCDs with private 4 MiB L2s and deals workgroup b of a 1-D grid to
//! XCD b % xcds (observed, not contractual: only speed depends on it; `xcds` comes from the device,
//! device_shape.hpp).  With `slices` > 1 (a divisor of xcds) a row is cut into `slices` column slices and
//! workgroup b works on slice (b % xcds) % slices of its samples, so that every L2 only ever caches 1/slices of
//! each row: the set of rows that fits an L2 grows `slices`-fold and a row that is needed by samples on different
//! XCDs is fetched from the fabric once per slice instead of once per XCD.  The xcds / slices XCDs that share a
//! slice split the samples.
struct ColumnSlice {
  int slice;      //!< which column slice of the row this workgroup owns
  int64_t block;  //!< index of the workgroup's group of samples
  static __device__ __forceinline__ ColumnSlice Of(unsigned block_idx, int slices, int xcds = 8) {
    ColumnSlice c;
    if (slices <= 1) {
      c.slice = 0;
      c.block = block_idx;
    } else {
      const int xcd = static_cast<int>(block_idx % static_cast<unsigned>(xcds));
      const int per_slice = xcds / slices;  // XCDs sharing one slice
      c.slice = xcd % slices;
      c.block = static_cast<int64_t>(block_idx / static_cast<unsigned>(xcds)) * per_slice + xcd / slices;
    }
    return c;
  }
};

__device__ __forceinline__ __bf16 ShuffleElem(__bf16 v, int src, int width) {
  const int bits = __shfl(static_cast<int>(__builtin_bit_cast(unsigned short, v)), src, width);
  return __builtin_bit_cast(__bf16, static_cast<unsigned short>(bits));
}

//! Row ids and row offsets as the gather kernels compute them.  A row offset is id * width:
//! written naively with a sign-extended 32-bit id and an `int` width that is a full 64 x 64-bit
//! multiply -- three integer multiplies (quarter rate) and three more instructions PER LOOKUP,
//! more issue time than the eight converts and four packed adds that pool the row.  Ids are
//! non-negative, so a 32-bit id is ZERO-extended and the width enters as an unsigned 32-bit value:
//! the compiler then emits one v_mad_u64_u32 (32 x 32 -> 64 plus the base).  64-bit ids keep
//! the general multiply (they may address outside the table on purpose: TranslateIndicesForRowCache).
template <typename IndexT>
__device__ __forceinline__ int64_t WidenIndex(const IndexT r) {
  if constexpr (sizeof(IndexT) == 4) return static_cast<int64_t>(static_cast<uint32_t>(r));
  else return static_cast<int64_t>(r);
}
__device__ __forceinline__ int64_t RowElems(const int64_t r, const int width) {
  return r * static_cast<int64_t>(static_cast<uint32_t>(width));
}
//! base + r * width elements, computed in BYTES so that scaling by the element size and adding the
//! base fold into the same v_mad_u64_u32 (id x row_bytes + base: one instruction per lookup).
template <typename ElemT>
__device__ __forceinline__ const ElemT* RowPtr(const ElemT* base, const int64_t r, const int width) {
  const int64_t row_bytes = static_cast<int64_t>(static_cast<uint32_t>(width) * static_cast<uint32_t>(sizeof(ElemT)));
  return reinterpret_cast<const ElemT*>(reinterpret_cast<const char*>(base) + r * row_bytes);
}

//! Table-row load for RowLoadPolicy::kStreaming: non-temporal (`global_load_dwordx4 ... nt`), i.e. the row does not
//! stay in L2.  Right when nothing is looked up twice (uniform indices at the C2 shape: 0.379 -> 0.355 ms, the
//! rows no longer evict each other for nothing); wrong when rows are re-used (alpha = 1.15: 0.136 -> 0.222 ms,
//! the hot rows lose their residency) -- which the launcher cannot know, hence a caller's option.
template <typename ElemT, int N>
__device__ __forceinline__ Pack<ElemT, N> LoadPackStreaming(const ElemT* p) {
  typedef unsigned __attribute__((ext_vector_type(sizeof(Pack<ElemT, N>) / 4))) raw_t;
  const raw_t raw = __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(p));
  return *reinterpret_cast<const Pack<ElemT, N>*>(&raw);
}

template <typename ElemT, int N>
__device__ __forceinline__ Pack<ElemT, N> LoadPack(const ElemT* p) {
#if defined(CUEMBED_TUNE_ROW_LOAD_ASM)   // tools/tune_forward.py --policies: cache-policy bits by inline asm
  static_assert(sizeof(Pack<ElemT, N>) == 16, "tuning build: 16-byte lanes only");
  typedef unsigned __attribute__((ext_vector_type(4))) raw4_t;
  raw4_t raw4;
  asm volatile("global_load_dwordx4 %0, %1, off " CUEMBED_TUNE_ROW_LOAD_ASM : "=v"(raw4) : "v"(p) : "memory");
  return *reinterpret_cast<const Pack<ElemT, N>*>(&raw4);
#elif defined(CUEMBED_TUNE_ROW_LOAD_NT)   // tools/tune_forward.py --policies: rejected, see docs/EXPERIMENTS.md
  typedef unsigned __attribute__((ext_vector_type(sizeof(Pack<ElemT, N>) / 4))) raw_t;
  const raw_t raw = __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(p));
  return *reinterpret_cast<const Pack<ElemT, N>*>(&raw);
#else
  return *reinterpret_cast<const Pack<ElemT, N>*>(p);
#endif
}

//! (tuning builds with inline-asm loads: the compiler does not see them, so wait by hand)
__device__ __forceinline__ void TuneWaitRowLoads() {
#if defined(CUEMBED_TUNE_ROW_LOAD_ASM)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

template <typename ElemT, int N>
__device__ __forceinline__ void StorePack(ElemT* p, const Pack<ElemT, N>& v) {
  *reinterpret_cast<Pack<ElemT, N>*>(p) = v;
}

//! Store for data this kernel never reads back (pooled output rows): non-temporal, so that it
//! does not displace table rows in L2.
template <typename ElemT, int N>
__device__ __forceinline__ void StorePackStreaming(ElemT* p, const Pack<ElemT, N>& v) {
  typedef unsigned __attribute__((ext_vector_type(sizeof(Pack<ElemT, N>) / 4))) raw_t;
  __builtin_nontemporal_store(*reinterpret_cast<const raw_t*>(&v), reinterpret_cast<raw_t*>(p));
}

//! acc[e] += float(row.v[e]) [* wf] -- one IEEE fp32 add (and one multiply) per element, as
//! everywhere in this library.
//! (Measured and rejected for fp16 rows: v_fma_mix_f32, which converts, multiplies and adds in one
//! instruction and is bit-identical because the product of two fp16 values is exact in fp32.  It
//! halves the instruction count but not the time: C2 forward 0.139 -> 0.185 ms, i.e. the mixed-
//! precision FMA issues well below the rate of v_cvt_f32_f16 + v_add_f32 on gfx950.)
template <typename GradT, int N, bool kWeighted>
__device__ __forceinline__ void AccumulateRow(float (&acc)[N], const Pack<GradT, N>& row, const float wf) {
  using A = Arith<float>;
  if constexpr (kWeighted) {
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], A::mul(static_cast<float>(row.v[e]), wf));
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], static_cast<float>(row.v[e]));
  }
}

//! Running state of one lane while it pools the rows of its sample.
template <typename ElemT, typename AccT, int N, bool kWeighted>
struct RowPool {
  using A = Arith<AccT>;
  AccT acc[N];
  float weight_sum;

  __device__ __forceinline__ RowPool() : weight_sum(0.f) {
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = static_cast<AccT>(0);
  }

  __device__ __forceinline__ void Add(const Pack<ElemT, N>& row, ElemT w) {
    if constexpr (kWeighted) {
      const AccT wa = A::widen(w);
      weight_sum += static_cast<float>(w);
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], A::mul(A::widen(row.v[e]), wa));
    } else {
#pragma unroll
      for (int e = 0; e < N; ++e) acc[e] = A::add(acc[e], A::widen(row.v[e]));
    }
  }

  //! Pools `count` lookups in order.  index_at(j) / weight_at(j) give lookup j's row id
  //! and weight.  kUnroll row loads are issued back-to-back before the first is consumed;
  //! with kPipelined the loads of batch k+1 are issued BEFORE batch k is consumed, so a
  //! wave always has kUnroll..2*kUnroll loads in flight instead of draining to zero
  //! between batches.  `sched_barrier` pins "all loads first, then the adds": without it
  //! the compiler splits a batch (e.g. 5 + 3) to save registers.
  //! kStream: non-temporal row loads (RowLoadPolicy::kStreaming).
  template <int kUnroll, bool kPipelined, bool kStream = false, typename IndexFn, typename WeightFn>
  __device__ __forceinline__ void Gather(const ElemT* lane_base, const int width, const int count,
                                         IndexFn index_at, WeightFn weight_at) {
    auto load_row = [](const ElemT* p) {
      if constexpr (kStream) return LoadPackStreaming<ElemT, N>(p);
      else return LoadPack<ElemT, N>(p);
    };
    int j = 0;
    if constexpr (kPipelined) {
      if (count >= 2 * kUnroll) {
        Pack<ElemT, N> a[kUnroll], b[kUnroll];
        ElemT wa[kUnroll], wb[kUnroll];
        auto issue = [&](Pack<ElemT, N>(&row)[kUnroll], ElemT(&w)[kUnroll], int base) {
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) {
            const int64_t r = index_at(base + u);
            if constexpr (kWeighted) w[u] = weight_at(base + u);
            row[u] = load_row(RowPtr(lane_base, r, width));
          }
        };
        auto consume = [&](Pack<ElemT, N>(&row)[kUnroll], ElemT(&w)[kUnroll]) {
#pragma unroll
          for (int u = 0; u < kUnroll; ++u) Add(row[u], w[u]);
        };
        issue(a, wa, 0);
        // invariant at loop top: batch [j, j+kUnroll) is in flight in `a`
        for (; j + 3 * kUnroll <= count; j += 2 * kUnroll) {
          issue(b, wb, j + kUnroll);
          __builtin_amdgcn_sched_barrier(0);
          consume(a, wa);
          issue(a, wa, j + 2 * kUnroll);
          __builtin_amdgcn_sched_barrier(0);
          consume(b, wb);
        }
        // here j + kUnroll <= count and `a` holds [j, j+kUnroll)
        if (j + 2 * kUnroll <= count) {
          issue(b, wb, j + kUnroll);
          __builtin_amdgcn_sched_barrier(0);
          consume(a, wa);
          consume(b, wb);
          j += 2 * kUnroll;
        } else {
          consume(a, wa);
          j += kUnroll;
        }
      }
    }
    for (; j + kUnroll <= count; j += kUnroll) {
      Pack<ElemT, N> row[kUnroll];
      ElemT w[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int64_t r = index_at(j + u);
        if constexpr (kWeighted) w[u] = weight_at(j + u);
        row[u] = load_row(RowPtr(lane_base, r, width));
      }
      TuneWaitRowLoads();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) Add(row[u], w[u]);
    }
    // Tail (< kUnroll lookups): still ONE batch -- all remaining loads are issued together
    // under a predicate instead of one exposed memory latency per leftover row (bags whose
    // length is not a multiple of kUnroll, i.e. almost every CSR bag, would otherwise spend a
    // third of their time there).
    const int rem = count - j;
    if (rem > 0) {
      Pack<ElemT, N> row[kUnroll];
      ElemT w[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll - 1; ++u) {
        if (u < rem) {
          const int64_t r = index_at(j + u);
          if constexpr (kWeighted) w[u] = weight_at(j + u);
          row[u] = load_row(RowPtr(lane_base, r, width));
        }
      }
      TuneWaitRowLoads();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kUnroll - 1; ++u) {
        if (u < rem) Add(row[u], w[u]);
      }
    }
  }
};

//! Epilogue of one pooled row: mean scaling (reference combiner, embedding_lookup_ops.cuh:273-285: scale by the
//! reciprocal of the accumulated weight; zeros when that is 0), conversion, non-temporal store.
template <typename ElemT, typename AccT, int N, bool kWeighted>
__device__ __forceinline__ void FinishPooledRow(RowPool<ElemT, AccT, N, kWeighted>& pool, const int hot,
                                                const bool is_mean, ElemT* dst) {
  using A = Arith<AccT>;
  if (is_mean) {
    float weight_sum = pool.weight_sum;
    if constexpr (!kWeighted) weight_sum = static_cast<float>(hot);
    const float inv = (weight_sum == 0.f) ? 0.f : 1.0f / weight_sum;
    const AccT scale = static_cast<AccT>(inv);
#pragma unroll
    for (int e = 0; e < N; ++e) pool.acc[e] = A::mul(pool.acc[e], scale);
  }
  Pack<ElemT, N> result;
#pragma unroll
  for (int e = 0; e < N; ++e) result.v[e] = static_cast<ElemT>(pool.acc[e]);
  StorePackStreaming<ElemT, N>(dst, result);
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
          IndexSource kSource,
          int kUnroll = kForwardUnroll,    // row loads issued back-to-back
          bool kPipelined = kForwardPipelined,
          int kBlockThreads = kMaxBlockThreads>
__global__ void __launch_bounds__(kBlockThreads)
GatherReduceKernel(const ElemT* __restrict__ table,
                   const int width,
                   const int batch,
                   const IndexT* __restrict__ indices,
                   const OffsetT* __restrict__ offsets,  // null => fixed hotness
                   const int num_hots,
                   const ElemT* __restrict__ weights,
                   const bool is_mean,
                   ElemT* __restrict__ out,
                   const int column_slices,    // 1, 2, 4 or 8 (see ColumnSlice)
                   const bool stream_rows_host, // RowLoadPolicy::kStreaming: table rows are not kept in L2
                   const int32_t* __restrict__ sample_order = nullptr,    // ForwardOptions::sample_order (CSR only)
                   const uint32_t* __restrict__ row_loads_device = nullptr) {   // ForwardOptions::row_loads_device
  using A = Arith<AccT>;
  // (the decision taken on the device, DecideRowLoads: one scalar load, the same value for every wavefront)
  const bool stream_rows = row_loads_device != nullptr ? (*row_loads_device != 0u) : stream_rows_host;
  const int lane_x = threadIdx.x;
  const int slot = threadIdx.y;
  const int samples_per_block = blockDim.y;
  const ColumnSlice cs = ColumnSlice::Of(blockIdx.x, column_slices);
  const int64_t block_id = cs.block;
  int64_t sample = block_id * samples_per_block + slot;
  const int64_t column0 = (static_cast<int64_t>(cs.slice) * blockDim.x + lane_x) * N;
  const ElemT* lane_base = table + column0;
  RowPool<ElemT, AccT, N, kWeighted> pool;
  int hot = num_hots;

  if constexpr (kSource == IndexSource::kLdsStaged) {
    // ---- fixed hotness: the workgroup's indices (+weights) go through LDS once ----
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    IndexT* stage_idx = reinterpret_cast<IndexT*>(lds_raw);
    ElemT* stage_w = reinterpret_cast<ElemT*>(stage_idx + samples_per_block * num_hots);
    const int64_t first = block_id * samples_per_block * num_hots;
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
    const IndexT* my_idx = stage_idx + slot * num_hots;
    const ElemT* my_w = stage_w + slot * num_hots;
    const auto idx_at = [&](int j) { return WidenIndex(my_idx[j]); };
    const auto w_at = [&](int j) { return my_w[j]; };
    if (stream_rows) pool.template Gather<kUnroll, kPipelined, true>(lane_base, width, hot, idx_at, w_at);
    else pool.template Gather<kUnroll, kPipelined, false>(lane_base, width, hot, idx_at, w_at);
  } else {
    if (sample >= batch) return;
    // scheduling hint: position p of the grid pools sample sample_order[p] (a permutation of the batch; e.g. bags by
    // descending length, so that the bags of a wavefront and of neighbouring wavefronts are alike).  Only WHICH lanes
    // pool a sample changes: its sum and its output row do not.
    if (sample_order != nullptr) sample = sample_order[sample];
    int64_t begin;
    if (offsets != nullptr) {
      begin = static_cast<int64_t>(offsets[sample]);
      hot = static_cast<int>(static_cast<int64_t>(offsets[sample + 1]) - begin);
    } else {
      begin = sample * num_hots;
    }
    const IndexT* my_idx = indices + begin;
    const ElemT* my_w = weights + begin;
    if constexpr (kSource == IndexSource::kWaveShuffle) {
      // ---- lanes_per_row divides 64: the lanes of a sample are consecutive lanes of ONE
      // wavefront.  They fetch lanes_per_row indices (+weights) with one coalesced load
      // and hand them to each other with cross-lane reads (ds_bpermute: no LDS storage,
      // no barrier); the next chunk is fetched while the current one is being pooled.
      const int group = blockDim.x;
      IndexT cur_i = static_cast<IndexT>(0);
      ElemT cur_w = static_cast<ElemT>(0);
      if (lane_x < hot) {
        cur_i = my_idx[lane_x];
        if constexpr (kWeighted) cur_w = my_w[lane_x];
      }
      for (int c = 0; c < hot; c += group) {
        IndexT next_i = static_cast<IndexT>(0);
        ElemT next_w = static_cast<ElemT>(0);
        if (c + group + lane_x < hot) {
          next_i = my_idx[c + group + lane_x];
          if constexpr (kWeighted) next_w = my_w[c + group + lane_x];
        }
        const int n = (hot - c < group) ? hot - c : group;
        const auto idx_at = [&](int j) { return WidenIndex(__shfl(cur_i, j, group)); };
        const auto w_at = [&](int j) { return ShuffleElem(cur_w, j, group); };
        if (stream_rows) pool.template Gather<kUnroll, kPipelined, true>(lane_base, width, n, idx_at, w_at);
        else pool.template Gather<kUnroll, kPipelined, false>(lane_base, width, n, idx_at, w_at);
        cur_i = next_i;
        cur_w = next_w;
      }
    } else {
      // ---- any row split: every lane reads its sample's index straight from global
      // memory (one address per sample: a broadcast load that mostly hits L1).
      const auto idx_at = [&](int j) { return WidenIndex(my_idx[j]); };
      const auto w_at = [&](int j) { return my_w[j]; };
      if (stream_rows) pool.template Gather<kUnroll, kPipelined, true>(lane_base, width, hot, idx_at, w_at);
      else pool.template Gather<kUnroll, kPipelined, false>(lane_base, width, hot, idx_at, w_at);
    }
  }

  // ---- epilogue -----------------------------------------------------------
  FinishPooledRow<ElemT, AccT, N, kWeighted>(pool, hot, is_mean, out + sample * width + column0);
}

// ---------------------------------------------------------------------------
// Sum / mean for SMALL batches: one sample per workgroup, the hotness dimension split
// over the workgroup's waves.
//   block = (lanes_per_row, slices) with lanes_per_row * slices = 256; grid = batch.
//   static LDS: one partial pooled row per wave.
// With few samples the sequential kernel above leaves most of the chip idle (1024
// samples x 32 lanes = 512 wavefronts for 1024 SIMDs) and each wave walks its whole
// bag alone.  Here slice k of a sample pools lookups [k*chunk, (k+1)*chunk) with the
// same batched loads; slices that share a wavefront are folded with cross-lane reads
// (DPP/ds_bpermute via __shfl_down), the per-wave partial rows are staged in LDS and
// the first wave adds them in slice order.  The association order differs from the
// sequential sum, so results agree with the reference to rounding (1e-3 / 1e-2
// relative, fp32 / fp16), not bit-for-bit: the host API only selects this kernel
// when ReductionOrder::kAllowSplit has been requested.
// ---------------------------------------------------------------------------
template <typename AccT>
__device__ __forceinline__ AccT ShuffleDownAcc(AccT v, int delta) {
  if constexpr (sizeof(AccT) == 2) {
    const int bits = __shfl_down(static_cast<int>(__builtin_bit_cast(unsigned short, v)), delta);
    return __builtin_bit_cast(AccT, static_cast<unsigned short>(bits));
  } else {
    return __shfl_down(v, delta);
  }
}

constexpr int kSplitBlockThreads = 256;

template <typename ElemT, typename AccT, typename IndexT, typename OffsetT, int N, bool kWeighted>
__global__ void __launch_bounds__(kSplitBlockThreads)
GatherReduceSplitKernel(const ElemT* __restrict__ table,
                        const int width,
                        const int batch,
                        const IndexT* __restrict__ indices,
                        const OffsetT* __restrict__ offsets,  // null => fixed hotness
                        const int num_hots,
                        const ElemT* __restrict__ weights,
                        const bool is_mean,
                        ElemT* __restrict__ out) {
  using A = Arith<AccT>;
  (void)batch;  // grid == batch: one workgroup per sample
  constexpr int kWaves = kSplitBlockThreads / 64;
  __shared__ AccT partial[kWaves][N][64];
  __shared__ float partial_w[kWaves];
  const int lane_x = threadIdx.x;
  const int lanes = blockDim.x;
  const int slice = threadIdx.y;
  const int slices = blockDim.y;
  const int64_t sample = blockIdx.x;

  int64_t begin;
  int hot = num_hots;
  if (offsets != nullptr) {
    begin = static_cast<int64_t>(offsets[sample]);
    hot = static_cast<int>(static_cast<int64_t>(offsets[sample + 1]) - begin);
  } else {
    begin = sample * num_hots;
  }
  const int chunk = (hot + slices - 1) / slices;
  const int j0 = slice * chunk < hot ? slice * chunk : hot;
  const int j1 = j0 + chunk < hot ? j0 + chunk : hot;
  const IndexT* my_idx = indices + begin + j0;
  const ElemT* my_w = weights + begin + j0;

  RowPool<ElemT, AccT, N, kWeighted> pool;
  pool.template Gather<kForwardUnroll, false, false>(
      table + static_cast<int64_t>(lane_x) * N, width, j1 - j0,
      [&](int j) { return WidenIndex(my_idx[j]); }, [&](int j) { return my_w[j]; });

  // ---- fold the slices that live in the same wavefront (lanes < 64 only) ----
  const int tid = slice * lanes + lane_x;
  const int wave = tid >> 6;
  const int lane = tid & 63;
  if (lanes < 64) {
    for (int delta = 32; delta >= lanes; delta >>= 1) {
#pragma unroll
      for (int e = 0; e < N; ++e) pool.acc[e] = A::add(pool.acc[e], ShuffleDownAcc(pool.acc[e], delta));
      pool.weight_sum += __shfl_down(pool.weight_sum, delta);
    }
  }
  // ---- stage one partial pooled row per wave in LDS, add them in wave order ----
  // (lanes >= 64: a row spans whole waves; wave w then holds columns of row part w % (lanes/64))
  const int row_waves = lanes >= 64 ? lanes / 64 : 1;   // waves that make up one full row
  if (lanes >= 64 || lane < lanes) {
#pragma unroll
    for (int e = 0; e < N; ++e) partial[wave][e][lane] = pool.acc[e];
    if (lane == 0) partial_w[wave] = pool.weight_sum;
  }
  __syncthreads();
  if (wave >= row_waves) return;
  if (lanes < 64 && lane >= lanes) return;
  AccT sum[N];
#pragma unroll
  for (int e = 0; e < N; ++e) sum[e] = partial[wave][e][lane];
  float weight_sum = partial_w[wave];
  for (int w = wave + row_waves; w < kWaves; w += row_waves) {
#pragma unroll
    for (int e = 0; e < N; ++e) sum[e] = A::add(sum[e], partial[w][e][lane]);
    weight_sum += partial_w[w];
  }
  if (is_mean) {
    if constexpr (!kWeighted) weight_sum = static_cast<float>(hot);
    const float inv = (weight_sum == 0.f) ? 0.f : 1.0f / weight_sum;
    const AccT scale = static_cast<AccT>(inv);
#pragma unroll
    for (int e = 0; e < N; ++e) sum[e] = A::mul(sum[e], scale);
  }
  Pack<ElemT, N> result;
#pragma unroll
  for (int e = 0; e < N; ++e) result.v[e] = static_cast<ElemT>(sum[e]);
  const int column_lane = lanes >= 64 ? wave * 64 + lane : lane;
  StorePack<ElemT, N>(out + sample * width + static_cast<int64_t>(column_lane) * N, result);
}

// ---------------------------------------------------------------------------
// Sum / mean for SMALL batches, BIT-EXACT: one sample per 256-thread workgroup, the LOADS of a bag spread over the
// whole workgroup, the ADDS kept in lookup order.
//   block = (lanes_per_row, slices, samples) with lanes_per_row * slices * samples = 256; grid = ceil(batch / samples);
//   rows of at most 1 KiB; dynamic LDS = kForwardUnroll * slices rows (+ as many weights) per sample: 32 KB with
//   16-byte lanes whatever the split
// With few samples the sequential kernel leaves most of the chip idle and walks a bag in rounds of kForwardUnroll
// rows, one memory round trip each (1024 samples x 64 lookups of 128-byte rows: 8 rounds).  Here slice s of the
// workgroup requests lookups s, s + slices, ... -- kForwardUnroll x slices rows (64 for 512-byte rows, 256 for
// 128-byte ones) in flight at once, ONE round trip for most bags -- and parks them in LDS.  Then one thread per
// 32-BIT WORD of the row (not per 16-byte pack: four times the lanes, on all four SIMDs) pools its one or two elements
// out of LDS in lookup order with the very operations of the sequential kernel -- acc = acc + row [* w], one IEEE
// operation each (Arith) -- so the result has the same bits, which is why this kernel is taken by default where it
// pays (ForwardWideLoadPays) while the re-associating split kernel above stays an option.
// ---------------------------------------------------------------------------
constexpr int kWideLoadThreads = 256;
constexpr int kWideLoadMaxRowBytes = 4 * kWideLoadThreads;   // one 32-bit word per thread in the pooling phase

template <typename ElemT, typename AccT, typename IndexT, typename OffsetT, int N, bool kWeighted>
__global__ void __launch_bounds__(kWideLoadThreads)
GatherReduceWideLoadKernel(const ElemT* __restrict__ table, const int width, const int batch,
                           const IndexT* __restrict__ indices, const OffsetT* __restrict__ offsets,  // null => fixed hotness
                           const int num_hots, const ElemT* __restrict__ weights, const bool is_mean,
                           ElemT* __restrict__ out, const bool stream_rows_host,
                           const uint32_t* __restrict__ row_loads_device = nullptr) {
  using A = Arith<AccT>;
  const bool stream_rows = row_loads_device != nullptr ? (*row_loads_device != 0u) : stream_rows_host;
  constexpr int kWordElems = 4 / static_cast<int>(sizeof(ElemT));   // elements in a 32-bit word: 1 (fp32) or 2
  extern __shared__ __attribute__((aligned(16))) unsigned char wide_lds_raw[];
  __shared__ int longest_bag;
  // block = (lanes, slices, samples): `samples` > 1 (narrow rows, larger batches) gives every sample of the workgroup
  // lanes x slices threads of its own and its own part of the LDS; the samples only meet at the barriers
  const int lane_x = threadIdx.x;
  const int lanes = blockDim.x;
  const int slice = threadIdx.y;
  const int slices = blockDim.y;
  const int sub = threadIdx.z;
  const int samples = blockDim.z;
  const int team = lanes * slices;             // threads of one sample
  const int tid = slice * lanes + lane_x;      // ... and this thread among them
  const int64_t sample = static_cast<int64_t>(blockIdx.x) * samples + sub;
  const bool present = sample < batch;
  int64_t begin = 0;
  int hot = 0;
  if (present) {
    if (offsets != nullptr) {
      begin = static_cast<int64_t>(offsets[sample]);
      hot = static_cast<int>(static_cast<int64_t>(offsets[sample + 1]) - begin);
    } else {
      begin = sample * num_hots;
      hot = num_hots;
    }
  }
  int rounds_for = hot;                        // every sample of the workgroup takes part in every round's barriers
  if (samples > 1) {
    if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) longest_bag = 0;
    __syncthreads();
    if (tid == 0) atomicMax(&longest_bag, hot);
    __syncthreads();
    rounds_for = longest_bag;
  }
  const int chunk = kForwardUnroll * slices;   // lookups of one sample parked in LDS at a time
  Pack<ElemT, N>* stage = reinterpret_cast<Pack<ElemT, N>*>(wide_lds_raw) + static_cast<size_t>(sub) * chunk * lanes;  // [chunk][lanes]
  ElemT* stage_w = reinterpret_cast<ElemT*>(reinterpret_cast<Pack<ElemT, N>*>(wide_lds_raw) +
                                            static_cast<size_t>(samples) * chunk * lanes) + static_cast<size_t>(sub) * chunk;
  const Pack<ElemT, kWordElems>* stage_words = reinterpret_cast<const Pack<ElemT, kWordElems>*>(stage);
  const int row_words = width / kWordElems;    // <= team (the launcher's condition)
  const bool pools = present && tid < row_words;
  const IndexT* my_idx = indices + begin;
  const ElemT* my_w = weights + begin;
  const ElemT* lane_base = table + static_cast<int64_t>(lane_x) * N;
  AccT acc[kWordElems];
#pragma unroll
  for (int e = 0; e < kWordElems; ++e) acc[e] = static_cast<AccT>(0);
  float weight_sum = 0.f;
  for (int c0 = 0; c0 < rounds_for; c0 += chunk) {
    const int left = hot - c0;
    const int n = left < 0 ? 0 : (left < chunk ? left : chunk);
    if (n > 0) {
      Pack<ElemT, N> row[kForwardUnroll];
      // (unconditional loads on a clamped lookup: a predicate per element would make the compiler merge the register
      // array at every branch; what a clamped lookup fetched is simply not parked)
#pragma unroll
      for (int u = 0; u < kForwardUnroll; ++u) {
        const int j = slice + u * slices;
        const int64_t r = WidenIndex(my_idx[c0 + (j < n ? j : 0)]);
        const ElemT* p = RowPtr(lane_base, r, width);
        row[u] = stream_rows ? LoadPackStreaming<ElemT, N>(p) : LoadPack<ElemT, N>(p);
      }
      if constexpr (kWeighted) {
        for (int j = tid; j < n; j += team) stage_w[j] = my_w[c0 + j];
      }
#pragma unroll
      for (int u = 0; u < kForwardUnroll; ++u) {
        const int j = slice + u * slices;
        if (j < n) stage[static_cast<size_t>(j) * lanes + lane_x] = row[u];
      }
    }
    __syncthreads();
    if (pools) {
      const Pack<ElemT, kWordElems>* mine = stage_words + tid;
#pragma unroll 8
      for (int j = 0; j < n; ++j) {
        const Pack<ElemT, kWordElems> v = mine[static_cast<size_t>(j) * row_words];
        if constexpr (kWeighted) {
          const ElemT w = stage_w[j];
          const AccT wa = A::widen(w);
          weight_sum += static_cast<float>(w);
#pragma unroll
          for (int e = 0; e < kWordElems; ++e) acc[e] = A::add(acc[e], A::mul(A::widen(v.v[e]), wa));
        } else {
#pragma unroll
          for (int e = 0; e < kWordElems; ++e) acc[e] = A::add(acc[e], A::widen(v.v[e]));
        }
      }
    }
    __syncthreads();   // the next chunk overwrites the parked rows
  }
  if (!pools) return;
  // epilogue as FinishPooledRow (mean: scale by the reciprocal of the accumulated weight, zeros when that is 0)
  if (is_mean) {
    if constexpr (!kWeighted) weight_sum = static_cast<float>(hot);
    const float inv = (weight_sum == 0.f) ? 0.f : 1.0f / weight_sum;
    const AccT scale = static_cast<AccT>(inv);
#pragma unroll
    for (int e = 0; e < kWordElems; ++e) acc[e] = A::mul(acc[e], scale);
  }
  Pack<ElemT, kWordElems> result;
#pragma unroll
  for (int e = 0; e < kWordElems; ++e) result.v[e] = static_cast<ElemT>(acc[e]);
  StorePackStreaming<ElemT, kWordElems>(out + sample * width + static_cast<int64_t>(tid) * kWordElems, result);
}

//! acc + <row, g> over the N elements of a pack, fp32 accumulation.  16-bit tables use the packed
//! dot instructions (v_dot2_f32_f16 / v_dot2_f32_bf16: two products and the add per instruction,
//! no separate conversions).
template <typename ElemT, int N>
__device__ __forceinline__ float DotPack(const Pack<ElemT, N>& row, const Pack<ElemT, N>& g, float acc) {
  if constexpr (std::is_same<ElemT, _Float16>::value) {
    typedef _Float16 pair_t __attribute__((ext_vector_type(2)));
    const pair_t* r2 = reinterpret_cast<const pair_t*>(row.v);
    const pair_t* g2 = reinterpret_cast<const pair_t*>(g.v);
#pragma unroll
    for (int e = 0; e < N / 2; ++e) acc = __builtin_amdgcn_fdot2(r2[e], g2[e], acc, false);
  } else if constexpr (std::is_same<ElemT, __bf16>::value) {
    typedef __bf16 pair_t __attribute__((ext_vector_type(2)));
    const pair_t* r2 = reinterpret_cast<const pair_t*>(row.v);
    const pair_t* g2 = reinterpret_cast<const pair_t*>(g.v);
#pragma unroll
    for (int e = 0; e < N / 2; ++e) acc = __builtin_amdgcn_fdot2_f32_bf16(r2[e], g2[e], acc, false);
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) acc += static_cast<float>(row.v[e]) * static_cast<float>(g.v[e]);
  }
  return acc;
}

// ---------------------------------------------------------------------------
// Gradient with respect to the per-lookup weights (an extension: the reference's torch example
// returns None for it, cuembed_pyt.py:34-35):
//     grad_w[s, j] = < table[idx[s, j], :], grad_y[s, :] >
// Same mapping as the forward: a group of `blockDim.x` lanes (a power of two <= 64, so a group
// never straddles a wavefront) owns a sample, batches kForwardUnroll row loads, multiplies
// them with its slice of grad_y (fp32), folds the per-lane partial dots with a cross-lane
// butterfly (`__shfl_xor`) and lane 0 stores one value per lookup.  Rows wider than the group
// are walked in strides of the group.
//   block = (group, samples_per_block); grid = ceil(batch / samples_per_block)
// ---------------------------------------------------------------------------
template <typename ElemT, typename IndexT, typename OffsetT, int N>
__global__ void __launch_bounds__(kMaxBlockThreads)
WeightGradKernel(const ElemT* __restrict__ table,
                 const int width,
                 const int batch,
                 const IndexT* __restrict__ indices,
                 const OffsetT* __restrict__ offsets,  // null => fixed hotness
                 const int num_hots,
                 const ElemT* __restrict__ grad_y,
                 ElemT* __restrict__ grad_w) {
  const int lane_x = threadIdx.x;
  const int group = blockDim.x;
  const int64_t sample = static_cast<int64_t>(blockIdx.x) * blockDim.y + threadIdx.y;
  if (sample >= batch) return;
  int64_t begin;
  int hot = num_hots;
  if (offsets != nullptr) {
    begin = static_cast<int64_t>(offsets[sample]);
    hot = static_cast<int>(static_cast<int64_t>(offsets[sample + 1]) - begin);
  } else {
    begin = sample * num_hots;
  }
  const IndexT* my_idx = indices + begin;
  ElemT* my_out = grad_w + begin;
  const ElemT* my_gy = grad_y + sample * width;
  const int chunks = width / N;  // N-element slices per row

  static_assert(kForwardUnroll == 8, "the transposing reduction below is written for 8 partial dots");
  if (chunks <= group && group >= 8) {
    // ---- a row fits one pass of the group (every shape up to 64 lanes x 16 B) ----
    // Software-pipelined: the rows of batch k+1 (and the row ids of batch k+2) are requested
    // BEFORE the cross-lane reduction of batch k, so loads are in flight during the ~15 dependent
    // cross-lane steps and a row request never waits for its own id.
    // The 8 partial dots are folded with a transposing butterfly: each exchange step halves the
    // number of values a lane still carries (4 + 2 + 1 exchanges leave lane l with the sum, over
    // its 8-lane subgroup, of lookup l & 7), then log2(group / 8) plain steps finish it --
    // 7..10 cross-lane operations per 8 lookups instead of 8 x log2(group).
    const bool has_column = lane_x < chunks;
    Pack<ElemT, N> g;
    if (has_column) g = LoadPack<ElemT, N>(my_gy + static_cast<int64_t>(lane_x) * N);
    const ElemT* lane_base = table + static_cast<int64_t>(lane_x) * N;
    Pack<ElemT, N> row[kForwardUnroll];
    IndexT ahead[kForwardUnroll];  // row ids of the batch AFTER the one whose rows are in flight
    // full batches take the branch-free path; only the last, partial batch is predicated
    auto request_ids = [&](const int j0) {
      if (j0 + kForwardUnroll <= hot) {
#pragma unroll
        for (int u = 0; u < kForwardUnroll; ++u) ahead[u] = my_idx[j0 + u];
      } else {
#pragma unroll
        for (int u = 0; u < kForwardUnroll; ++u)
          if (j0 + u < hot) ahead[u] = my_idx[j0 + u];
      }
    };
    auto request_rows = [&](const int j0) {  // consumes `ahead`
      if (!has_column) return;
      if (j0 + kForwardUnroll <= hot) {
#pragma unroll
        for (int u = 0; u < kForwardUnroll; ++u)
          row[u] = LoadPack<ElemT, N>(RowPtr(lane_base, WidenIndex(ahead[u]), width));
      } else {
#pragma unroll
        for (int u = 0; u < kForwardUnroll; ++u)
          if (j0 + u < hot) row[u] = LoadPack<ElemT, N>(RowPtr(lane_base, WidenIndex(ahead[u]), width));
      }
    };
    request_ids(0);
    request_rows(0);
    request_ids(kForwardUnroll);
    for (int j0 = 0; j0 < hot; j0 += kForwardUnroll) {
      float dot[kForwardUnroll];
      if (j0 + kForwardUnroll <= hot) {
#pragma unroll
        for (int u = 0; u < kForwardUnroll; ++u) dot[u] = has_column ? DotPack<ElemT, N>(row[u], g, 0.f) : 0.f;
      } else {
#pragma unroll
        for (int u = 0; u < kForwardUnroll; ++u)
          dot[u] = (j0 + u < hot && has_column) ? DotPack<ElemT, N>(row[u], g, 0.f) : 0.f;
      }
      __builtin_amdgcn_sched_barrier(0);
      request_rows(j0 + kForwardUnroll);
      request_ids(j0 + 2 * kForwardUnroll);
      __builtin_amdgcn_sched_barrier(0);
      float four[4], two[2];
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // exchange with lane ^ 1: keep the even or the odd lookup of a pair
        const bool odd = lane_x & 1;
        const float got = __shfl_xor(odd ? dot[2 * k] : dot[2 * k + 1], 1, group);
        four[k] = (odd ? dot[2 * k + 1] : dot[2 * k]) + got;
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const bool odd = lane_x & 2;
        const float got = __shfl_xor(odd ? four[2 * k] : four[2 * k + 1], 2, group);
        two[k] = (odd ? four[2 * k + 1] : four[2 * k]) + got;
      }
      float one;
      {
        const bool odd = lane_x & 4;
        const float got = __shfl_xor(odd ? two[0] : two[1], 4, group);
        one = (odd ? two[1] : two[0]) + got;
      }
      for (int d = 8; d < group; d <<= 1) one += __shfl_xor(one, d, group);
      // lane l < 8 now holds lookup j0 + l: one coalesced store per batch
      if (lane_x < kForwardUnroll && j0 + lane_x < hot) my_out[j0 + lane_x] = static_cast<ElemT>(one);
    }
    return;
  }

  for (int j0 = 0; j0 < hot; j0 += kForwardUnroll) {
    const int nb = hot - j0 < kForwardUnroll ? hot - j0 : kForwardUnroll;
    float dot[kForwardUnroll];
#pragma unroll
    for (int u = 0; u < kForwardUnroll; ++u) dot[u] = 0.f;
    for (int c = lane_x; c < chunks; c += group) {
      const Pack<ElemT, N> g = LoadPack<ElemT, N>(my_gy + static_cast<int64_t>(c) * N);
      Pack<ElemT, N> row[kForwardUnroll];
#pragma unroll
      for (int u = 0; u < kForwardUnroll; ++u) {
        if (u < nb)
          row[u] = LoadPack<ElemT, N>(table + RowElems(WidenIndex(my_idx[j0 + u]), width) +
                                      static_cast<int64_t>(c) * N);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kForwardUnroll; ++u) {
        if (u < nb) dot[u] = DotPack<ElemT, N>(row[u], g, dot[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < kForwardUnroll; ++u) {
      for (int d = group >> 1; d > 0; d >>= 1) dot[u] += __shfl_xor(dot[u], d, group);
    }
    if (lane_x == 0) {
#pragma unroll
      for (int u = 0; u < kForwardUnroll; ++u)
        if (u < nb) my_out[j0 + u] = static_cast<ElemT>(dot[u]);
    }
  }
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
  // The output is written once and never read here: non-temporal stores keep it from displacing
  // table rows in L2.  The row ids of batch k+1 are requested before the rows of batch k are
  // stored, so a row request never waits for its own id.
  IndexT ahead[kForwardUnroll];
  auto request_ids = [&](const int j0) {
#pragma unroll
    for (int u = 0; u < kForwardUnroll; ++u)
      if (j0 + u < num_hots) ahead[u] = my_idx[j0 + u];
  };
  request_ids(0);
  for (int j = 0; j < num_hots; j += kForwardUnroll) {
    Pack<ElemT, N> row[kForwardUnroll];
#pragma unroll
    for (int u = 0; u < kForwardUnroll; ++u) {
      if (j + u < num_hots) row[u] = LoadPack<ElemT, N>(RowPtr(lane_base, WidenIndex(ahead[u]), width));
    }
    request_ids(j + kForwardUnroll);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < kForwardUnroll; ++u) {
      if (j + u < num_hots) StorePackStreaming<ElemT, N>(dst + static_cast<int64_t>(j + u) * width, row[u]);
    }
  }
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_GATHER_REDUCE_KERNELS_HPP_
