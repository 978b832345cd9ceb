// MI355X (gfx950 / CDNA4) kernels for EmbeddingBackward.
//
// Input is the transposed (index-sorted) COO produced by Transpose(): runs of
// equal row ids are contiguous.  The work is cut into fixed-length nz-segments so
// that it is balanced no matter how skewed the run lengths are (at the north-star
// shape one row owns a run of 65,528 lookups); `lanes_per_row` lanes walk one
// segment in nz order, gathering grad_y[sample_id] row slices with kWindow
// loads in flight (a rolling window) and keeping the running sum in fp32 registers.  When a run ends:
//   * the run lies entirely inside this segment  -> one plain vector store;
//   * the run continues from / into a neighbour segment -> the partial is combined with its
//     neighbours inside the workgroup through LDS; only what crosses a WORKGROUP boundary
//     goes out as hardware float atomics (global_atomic_add_f32 / global_atomic_pk_add_f16)
//     into the zeroed output.
// This is the reference's scheme (embedding_lookup_kernels.cuh:175-220,
// embedding_lookup_ops.cuh:518-564, :647-662) with three differences: shared-run
// detection looks at the real neighbours instead of treating every first/last
// run as shared (fewer atomics), partial sums are fp32 for fp16 gradients too
// (one rounding per flush instead of one per lookup), and all row addressing is
// 64-bit (the reference's int32 `row * embed_width`, ops.cuh:610-618, overflows
// beyond 2^31 elements -- a dense 10M x 256 gradient).
#ifndef CUEMBED_INCLUDE_SCATTER_ADD_KERNELS_HPP_
#define CUEMBED_INCLUDE_SCATTER_ADD_KERNELS_HPP_

#include "cuembed/include/blocked_order.hpp"
#include "cuembed/include/device_shape.hpp"
#include "cuembed/include/embedding_types.hpp"
#include "cuembed/include/gather_reduce_kernels.hpp"

namespace cuembed {
namespace detail {

//! Depth of the rolling window of gathers (lookups in flight per lane group): 4 where grad_y is mostly L2-resident
//! (column-sliced launches: 64 instead of 86 VGPRs = 7-8 instead of 5 wavefronts per SIMD; the blocked orders gain
//! 5 %), 8 where the gathers miss (unsliced launches -- narrow rows or few lookups: uniform indices at W = 32,
//! B = 131,072 measured 0.265 ms with 8 and 0.318 ms with 4).  Segment lengths are multiples of 8 either way.
constexpr int kBackwardWindowHits = 4;
constexpr int kBackwardWindowMisses = 8;

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

template <int N>
__device__ __forceinline__ void FlushAtomic(__bf16* dst, const float (&acc)[N]) {
  static_assert(N % 2 == 0, "bf16 rows are split in multiples of 4 bytes");
  typedef __bf16 __attribute__((ext_vector_type(2))) bf16x2_t;
#pragma unroll
  for (int e = 0; e < N; e += 2) {
    bf16x2_t v;
    v.x = static_cast<__bf16>(acc[e]);
    v.y = static_cast<__bf16>(acc[e + 1]);
    __builtin_amdgcn_global_atomic_fadd_v2bf16(
        (__attribute__((address_space(1))) bf16x2_t*)(dst + e), v);  // generic -> global pointer
  }
}

template <typename GradT, int N>
__device__ __forceinline__ void FlushStore(GradT* dst, const float (&acc)[N]) {
  Pack<GradT, N> p;
#pragma unroll
  for (int e = 0; e < N; ++e) p.v[e] = static_cast<GradT>(acc[e]);
  // gradient rows are written once and not read again by this kernel: a non-temporal store
  // keeps them from evicting the grad_y lines that the gather re-reads from L2 (measured:
  // -10 % on the write-heavy uniform-index case, -1..2 % at C4)
  typedef unsigned __attribute__((ext_vector_type(sizeof(Pack<GradT, N>) / 4))) raw_t;
  __builtin_nontemporal_store(*reinterpret_cast<raw_t*>(&p), reinterpret_cast<raw_t*>(dst));
}

//! Padding (words) of a segment's slice of packed sample ids: keeps every slice 16-byte aligned
//! (the walk fetches 4 lookups per ds_read_b128) and shifts consecutive segments by 4 LDS banks.
constexpr int kPackedPad = 4;
//! Bit 31 of a staged sample id: "this lookup is the last one of its run".
constexpr uint32_t kRunEndBit = 0x80000000u;
//! Bit 30 of a staged sample id (sample-blocked order only): "an earlier block already stored the row of this run".
constexpr uint32_t kSharedRunBit = 0x40000000u;

//! LDS needed by SegmentedScatterAddKernel for `segments_per_block` segments of
//! `segment_len` lookups handled by `lanes_per_row` lanes each:
//!   row ids for [first - 1, last + 1] (two sentinels), sample ids, weights (if any),
//!   one fp32 partial row per segment (head run; the tail run's re-uses the id area) and their bookkeeping.
template <typename GradT, typename IndexT>
__host__ __device__ inline size_t ScatterStageBytes(int segments_per_block, int segment_len,
                                                    int lanes_per_row, int elems_per_lane,
                                                    bool weighted) {
  // ids are staged as 32-bit whatever IndexT is: nnz and num_grad_embedding_rows are `int` in the
  // API (embedding_lookup.cuh:423-435), so every row id and sample id fits.  Every segment has its
  // own padded slice: a wavefront reads the SAME position of 8 different
  // segments at a time, and with a power-of-two stride all 8 would hit one LDS bank.
  const size_t segs = static_cast<size_t>(segments_per_block);
  size_t bytes = (segs * (segment_len + 2) * sizeof(int32_t) + 15) / 16 * 16;   // row ids
  bytes += segs * (segment_len + kPackedPad) * sizeof(uint32_t);                // packed sample ids
  bytes = (bytes + 15) / 16 * 16;
  if (weighted) bytes += segs * (segment_len + 2) * sizeof(GradT);
  bytes = (bytes + 15) / 16 * 16;
  // one fp32 partial row per segment for the run that came in (written during the walk) ...
  const size_t partial = segs * lanes_per_row * elems_per_lane * sizeof(float);
  // ... and one for the run that goes on, which is parked AFTER the walk on top of the staged ids
  // (dead by then): the staging area is as large as the larger of the two
  if (bytes < partial) bytes = partial;
  bytes += partial;
  bytes += segs * (2 * sizeof(int64_t) + sizeof(int));
  return (bytes + 15) / 16 * 16;
}

enum : int { kPartHead = 1, kPartTail = 2, kPartWhole = 4 };

//! block = (lanes_per_row, segments_per_block); grid = ceil(num_segments / segments_per_block)
//! dynamic LDS = ScatterStageBytes(...).
//!
//! 1. The workgroup's segments are consecutive, so their COO triples form ONE contiguous
//!    range of the sorted arrays: it is copied into LDS with coalesced loads once; the walk
//!    reads ids from LDS and only the grad_y row gathers stay on the global-memory path.
//!    Where runs end is decided here, once per lookup (kRunEndBit of the staged sample id).
//! 2. Each segment is walked in nz order by `lanes_per_row` lanes: a uniform loop over a rolling
//!    window of kWindow gathers (lookup i + K is requested into the registers of lookup
//!    i as soon as those are converted), fp32 partial sums in register pairs.  A run that lies
//!    inside the segment ends in a plain vector store.  The partial sums of the segment's FIRST
//!    run (when it continues from the previous segment) and LAST run (when it continues into
//!    the next one) are parked in LDS instead.
//! 2b. XCD-aware column slices (`column_slices` > 1): grad_y (batch x width) is usually larger
//!    than one XCD's 4 MiB L2 but far smaller than the 256 MiB Infinity Cache, so every L2
//!    would stream all of it from the fabric.  Workgroup b (which runs on XCD b % 8) therefore
//!    only handles column slice (b % 8) % slices of its lookups: each L2 then holds 1/slices of
//!    grad_y and the same random-row gather was measured 1.47x faster (8.4 -> 12.3 TB/s with 4
//!    slices of 128 B).  The price is that the COO triples are staged once per slice.
//! 3. After a barrier the parked partials of neighbouring segments that belong to the same
//!    row are chained and summed in nz order by the segment that ends the chain.  A chain that
//!    stays inside the workgroup is written with a plain store; only a chain that crosses the
//!    workgroup boundary uses float atomics -- at most two atomic flushes per workgroup.
//!    Device-scope atomics to one address serialise at roughly 0.5 us each on MI355X (the
//!    XCD L2s are not coherent, so they execute memory-side); a row with a 65,528-lookup run
//!    would otherwise be hit by 512 of them and alone take longer than the rest of the kernel.
//! 4. Sample-blocked order (block_row_ids != nullptr; blocked_order.hpp): `rows` holds (block, table row) pair
//!    numbers and block_row_ids[pair] the gradient row, with kSharedRowBit when an EARLIER block -- an earlier launch
//!    on the same stream -- already stored that row (ComputeCompressedGradIndicesBlocked); the staging translates.
//!    The first block has no such rows and runs the plain kernel; the later ones run the kBlocked variant, in which
//!    a run with the bit is ADDED to what is there: a read-modify-write when the run lies inside this workgroup (nobody else
//!    touches the row during this launch), the usual float atomics when it crosses workgroups.  The rows to be read
//!    are requested once while staging (they were written by another launch and come from HBM; that brings their
//!    lines into this XCD's L2) and again with the gather of the run's last lookup (see `old` below).
template <typename GradT, typename IndexT, int N, bool kWeighted, bool kBlocked = false, int kWindow = kBackwardWindowHits>
__global__ void __launch_bounds__(kMaxBlockThreads)
SegmentedScatterAddKernel(const GradT* __restrict__ grad_y,
                          const int width,
                          const IndexT* __restrict__ rows,        // sorted; remapped or raw ids
                          const IndexT* __restrict__ sample_ids,
                          const GradT* __restrict__ weights,
                          const int64_t nnz,
                          const int segment_len,
                          const int segment_shift,  // log2(segment_len), or -1 when it is not a power of two
                          GradT* __restrict__ grad_out,
                          const int column_slices,  // a divisor of xcds (1, 2, 4 or 8 on a full chip): see ColumnSlice
                          const int xcds,           // XCDs of the device (workgroup b runs on XCD b % xcds)
                          const IndexT* __restrict__ run_ids,        // compressed gradient: table row ids ...
                          IndexT* __restrict__ inverse_mapping,      // ... and where the id of every run goes
                          const uint32_t* __restrict__ block_row_ids,   // sample-blocked order: pair number -> row | bit
                          const int64_t capacity_rows,               // > 0: rows that grad_out / inverse_mapping hold
                          uint32_t* __restrict__ capacity_overflow) {  // ... and the word that says "not enough"
  // (a template parameter, not a run-time flag: the registers of the read-modify-write would cost the
  // reference-order kernel a wavefront per SIMD)
  constexpr uint32_t shared_row_bit = kBlocked ? kSharedRowBit : 0u;
  using A = Arith<float>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane_x = threadIdx.x;
  const int lanes = blockDim.x;  // lanes of ONE column slice
  const int seg = threadIdx.y;
  const int segments_per_block = blockDim.y;
  const int block_len = segments_per_block * segment_len;
  const ColumnSlice cs = ColumnSlice::Of(blockIdx.x, column_slices, xcds);
  const int64_t block_begin = cs.block * block_len;
  if (block_begin >= nnz) return;  // the grid is rounded up to whole rounds of `xcds` workgroups
  if (capacity_rows > 0) {
    // Compressed gradient whose row count only the device knows (num_grad_embedding_rows < 0): the ids ascend through
    // the sorted COO of this launch, so its LAST lookup holds the largest gradient row.  If the caller's buffers are
    // too small for it, nothing of this launch is written -- every workgroup takes the same decision from the same
    // word -- and the caller's overflow word is raised instead of an overrun.
    uint32_t last = static_cast<uint32_t>(rows[nnz - 1]);
    if (block_row_ids != nullptr) last = block_row_ids[last] & ~kSharedRowBit;
    if (static_cast<int64_t>(last) >= capacity_rows) {
      if (blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0 && capacity_overflow != nullptr)
        atomicOr(capacity_overflow, 1u);
      return;
    }
  }
  const int64_t column0 = (static_cast<int64_t>(cs.slice) * lanes + lane_x) * N;
  const uint32_t id_mask = ~shared_row_bit;   // staged ids keep the bit (equal inside a run); addresses drop it

  // ---- LDS carve-up (must match ScatterStageBytes) ----
  // per segment: rows [segment_len + 2] = (lookup before, the segment's row ids, lookup after);
  // packed sample ids [segment_len + kPackedPad]; weights [segment_len + 2] -- strides that are not
  // multiples of the 64 LDS banks, so the 8 segments of a wavefront read from different banks
  const int row_stride = segment_len + 2;
  const int pk_stride = segment_len + kPackedPad;
  const int w_stride = segment_len + 2;
  int32_t* st_rows = reinterpret_cast<int32_t*>(lds_raw);
  size_t off = (static_cast<size_t>(segments_per_block) * row_stride * sizeof(int32_t) + 15) / 16 * 16;
  uint32_t* st_pk = reinterpret_cast<uint32_t*>(lds_raw + off);
  off = (off + static_cast<size_t>(segments_per_block) * pk_stride * sizeof(uint32_t) + 15) / 16 * 16;
  GradT* st_w = reinterpret_cast<GradT*>(lds_raw + off);
  if (kWeighted) off += static_cast<size_t>(segments_per_block) * w_stride * sizeof(GradT);
  off = (off + 15) / 16 * 16;
  const size_t partial_bytes = static_cast<size_t>(segments_per_block) * lanes * N * sizeof(float);
  if (off < partial_bytes) off = partial_bytes;                     // (the id area also holds the tail partials, below)
  float* part_head = reinterpret_cast<float*>(lds_raw + off);       // [seg][N][lanes]: run that came in from before
  float* part_tail = reinterpret_cast<float*>(lds_raw);             // [seg][N][lanes]: run that goes on; ALIASES the staged ids
  off += partial_bytes;
  int64_t* part_row = reinterpret_cast<int64_t*>(lds_raw + off);   // [seg][2]
  off += static_cast<size_t>(segments_per_block) * 2 * sizeof(int64_t);
  int* part_flags = reinterpret_cast<int*>(lds_raw + off);         // [seg]

  {
    // The workgroup's COO triples are ONE contiguous range of the sorted arrays: all threads copy it
    // with fully coalesced loads (consecutive threads, consecutive lookups) into the per-segment
    // padded slices.  The run structure is decided HERE, once per lookup, not in the walk: a staged
    // sample id carries kRunEndBit when the next lookup belongs to another row (sample ids are
    // non-negative 32-bit values: nnz and the batch are `int` in the API).  Positions past nnz
    // hold sample 0 without the bit: their lanes gather a valid row and never flush it.
    const int tid = seg * lanes + lane_x;
    const int threads = lanes * segments_per_block;
    // kStageBatch lookups per thread and round: all their loads are requested before the first one is used (the
    // blocked order adds a dependent table lookup per lookup; one at a time, a thread's 8 lookups at C4 would pay
    // 8 x 2 memory latencies before the walk can start)
    constexpr int kStageBatch = 4;
    for (int e0 = tid; e0 < block_len; e0 += threads * kStageBatch) {
      int32_t r[kStageBatch], r_next[kStageBatch];
      uint32_t sid[kStageBatch];
      GradT wv[kStageBatch];
#pragma unroll
      for (int q = 0; q < kStageBatch; ++q) {
        const int64_t g = block_begin + e0 + q * threads;
        const bool in = e0 + q * threads < block_len && g < nnz;
        r[q] = in ? static_cast<int32_t>(rows[g]) : -1;
        r_next[q] = (in && g + 1 < nnz) ? static_cast<int32_t>(rows[g + 1]) : -1;
        sid[q] = in ? static_cast<uint32_t>(static_cast<int32_t>(sample_ids[g])) : 0u;
        if constexpr (kWeighted) wv[q] = in ? weights[g] : static_cast<GradT>(0);
      }
      int32_t row[kStageBatch];
#pragma unroll
      for (int q = 0; q < kStageBatch; ++q) {
        row[q] = r[q];
        if (block_row_ids != nullptr && r[q] >= 0)   // sample-blocked order: pair number -> gradient row | kSharedRowBit
          row[q] = static_cast<int32_t>(block_row_ids[r[q]]);
      }
#pragma unroll
      for (int q = 0; q < kStageBatch; ++q) {
        const int e = e0 + q * threads;
        if (e < block_len) {
          const int s = segment_shift >= 0 ? e >> segment_shift : e / segment_len;
          const int i = e - s * segment_len;
          uint32_t pk = 0;
          if (r[q] >= 0) {
            pk = sid[q] | (r_next[q] != r[q] ? kRunEndBit : 0u);
            if constexpr (kBlocked) {   // (sample ids are < 2^30 in a blocked order)
              if (static_cast<uint32_t>(row[q]) & shared_row_bit) pk |= kSharedRunBit;
            }
            if constexpr (kWeighted) st_w[s * w_stride + i] = wv[q];
          }
          st_rows[s * row_stride + 1 + i] = row[q];
          st_pk[s * pk_stride + i] = pk;
        }
      }
    }
    if constexpr (kBlocked) {
      // The rows that will be read and added to were written by another launch and come from HBM.  Touch this
      // slice's lines of them now -- all requests of the workgroup in flight together, one wait -- so that the reads
      // in the walk, which sit in the in-order return queue of the gathers, find them in L2.  Every thread looks at
      // the lookups it staged itself (no barrier needed); lookups without such a row touch line 0 of the gradient.
      uint32_t warmed = 0;
      const int slice_bytes = lanes * N * static_cast<int>(sizeof(GradT));
      const char* slice0 = reinterpret_cast<const char*>(grad_out + static_cast<int64_t>(cs.slice) * lanes * N);
      for (int e0 = tid; e0 < block_len; e0 += threads * kStageBatch) {
        const char* line[kStageBatch];
#pragma unroll
        for (int q = 0; q < kStageBatch; ++q) {
          const int e = e0 + q * threads;
          line[q] = slice0;
          if (e < block_len) {
            const int s = segment_shift >= 0 ? e >> segment_shift : e / segment_len;
            const int i = e - s * segment_len;
            const uint32_t pk = st_pk[s * pk_stride + i];
            if ((pk & (kRunEndBit | kSharedRunBit)) == (kRunEndBit | kSharedRunBit))
              line[q] = slice0 + static_cast<int64_t>(static_cast<uint32_t>(st_rows[s * row_stride + 1 + i]) & id_mask) *
                                     (static_cast<int64_t>(width) * static_cast<int64_t>(sizeof(GradT)));
          }
        }
        for (int b = 0; b < slice_bytes; b += 128) {
#pragma unroll
          for (int q = 0; q < kStageBatch; ++q) warmed |= *reinterpret_cast<const uint32_t*>(line[q] + b);
        }
      }
      asm volatile("" ::"v"(warmed));   // (keeps the loads; the barrier below waits for them)
    }
    if (lane_x == 0) part_flags[seg] = 0;
    __syncthreads();
    // the lookup before and after every segment: a neighbour's staged id, or -- at the two ends of
    // the workgroup's range -- one more element of the sorted array (-1 outside it)
    if (lane_x == 0) {
      int32_t* seg_rows = st_rows + seg * row_stride;
      const int64_t before = block_begin - 1, after = block_begin + block_len;
      auto row_at = [&](const int64_t g) {
        const int32_t r = static_cast<int32_t>(rows[g]);
        return block_row_ids != nullptr ? static_cast<int32_t>(block_row_ids[r]) : r;
      };
      seg_rows[0] = seg > 0 ? st_rows[(seg - 1) * row_stride + segment_len]
                            : ((before >= 0 && before < nnz) ? row_at(before) : -1);
      seg_rows[segment_len + 1] = seg + 1 < segments_per_block
                                      ? st_rows[(seg + 1) * row_stride + 1]
                                      : (after < nnz ? row_at(after) : -1);
    }
  }
  __syncthreads();
  // Compressed gradient: inverse_mapping[dense id] = table row id, written at the first lookup of
  // every run (the reference's CompactSparseIndicesKernel, embedding_lookup_kernels.cuh:289-302, is
  // a separate launch over all nnz; here the dense ids are already in LDS and only the run heads
  // load their table row id).
  if (run_ids != nullptr && cs.slice == 0) {
    const int32_t* seg_rows = st_rows + seg * row_stride;
    const int64_t seg_begin = block_begin + static_cast<int64_t>(seg) * segment_len;
    for (int i = lane_x; i < segment_len; i += lanes) {
      const int64_t g = seg_begin + i;
      if (g < nnz && seg_rows[1 + i] != seg_rows[i])
        inverse_mapping[static_cast<uint32_t>(seg_rows[1 + i]) & id_mask] = run_ids[g];
    }
  }
  const int seg_off = seg * segment_len;  // offset of this segment inside the block
  const int64_t begin = block_begin + seg_off;
  const bool active = begin < nnz;
  const GradT* lane_src = grad_y + column0;
  GradT* lane_dst = grad_out + column0;

  {
    // ---- the walk: every lane group runs the SAME segment_len iterations (uniform loop, scalar
    // branches); what differs per segment is only where runs end.
    const int32_t* my_rows = st_rows + seg * row_stride + 1;  // my_rows[-1] = lookup before the segment
    const uint32_t* my_pk = st_pk + seg * pk_stride;
    const GradT* my_w = st_w + seg * w_stride;
    typedef uint32_t __attribute__((ext_vector_type(4))) word4_t;
    constexpr int K = kWindow;
    static_assert(K % 4 == 0, "packed ids are fetched four at a time");

    // A run is "shared" when it also has lookups in a neighbouring segment (the -1 sentinels and
    // the -1 of positions past nnz never count).
    bool first_pending = my_rows[0] >= 0 && my_rows[-1] == my_rows[0];   // the first run came in from before
    const bool tail_shared = my_rows[segment_len - 1] >= 0 && my_rows[segment_len] == my_rows[segment_len - 1];

    // the running sum lives in register PAIRS: v_pk_add_f32 adds a pair per instruction and a
    // pair is cleared by one 64-bit move
    typedef float __attribute__((ext_vector_type(2))) pair_t;
    constexpr int kPairs = (N + 1) / 2;
    pair_t acc[kPairs];
#pragma unroll
    for (int j = 0; j < kPairs; ++j) acc[j] = pair_t{0.f, 0.f};
    auto unpair = [&](float (&dst)[N]) {
#pragma unroll
      for (int e = 0; e < N; ++e) dst[e] = (e & 1) ? acc[e / 2].y : acc[e / 2].x;
    };

    auto note = [&](const int slot, const int64_t row, const int flag) {
      if (lane_x == 0) {
        part_row[seg * 2 + slot] = row;
        part_flags[seg] |= flag;
      }
    };
    auto park_head = [&](const int64_t row) {   // the run that came in from earlier segments ends here
      float a[N];
      unpair(a);
#pragma unroll
      for (int e = 0; e < N; ++e) part_head[(seg * N + e) * lanes + lane_x] = a[e];
      note(0, row, kPartHead);
    };
    auto fetch_ids = [&](uint32_t (&dst)[K], const int i0) {
#pragma unroll
      for (int q = 0; q < K / 4; ++q) {
        const word4_t v = *reinterpret_cast<const word4_t*>(my_pk + i0 + 4 * q);
        dst[4 * q + 0] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
      }
    };
    // rows in flight are held as raw dwords (one register tuple per lookup, re-used by the
    // request that replaces it), not as N separate elements
    typedef uint32_t __attribute__((ext_vector_type(sizeof(Pack<GradT, N>) / 4))) raw_t;
    constexpr uint32_t kSampleMask = kBlocked ? ~(kRunEndBit | kSharedRunBit) : ~kRunEndBit;
    auto gather = [&](const uint32_t packed) {
      return *reinterpret_cast<const raw_t*>(RowPtr(lane_src, static_cast<int64_t>(packed & kSampleMask), width));
    };
    // Sample-blocked order: a run whose row an earlier block (an earlier launch) already stored is ADDED to it -- a
    // read-modify-write, not an atomic (13.7 M dword atomics at C4 cost 0.1 ms; the 55 MB they touch are 10 us of
    // traffic).  The old row is requested TOGETHER with the gather of the run's last lookup, kWindow
    // lookups ahead of its use, into a register tuple of its own per window slot: loads return in order, so it is
    // there when that lookup is consumed and the window never drains for it (a read issued at the run's end and
    // awaited at the next one made the compiler wait for vmcnt(1): 0.177 instead of 0.119 ms per block).
    raw_t old[kBlocked ? K : 1];
    auto request_old = [&](const uint32_t packed, const int pos, raw_t& into) {
      if constexpr (kBlocked) {
        if ((packed & (kRunEndBit | kSharedRunBit)) == (kRunEndBit | kSharedRunBit))
          into = *reinterpret_cast<const raw_t*>(
              RowPtr(lane_dst, static_cast<int64_t>(static_cast<uint32_t>(my_rows[pos]) & id_mask), width));
      }
    };

    // kWindow gathers stay in flight for the whole walk: the row of lookup i + K is
    // requested as soon as the registers of lookup i have been converted (a rolling window, not
    // load-8 / wait / add-8: the wave's own arithmetic overlaps its own memory latency).
    raw_t g[K];
    uint32_t cur[K], nxt[K];
    fetch_ids(cur, 0);
#pragma unroll
    for (int u = 0; u < K; ++u) {
      g[u] = gather(cur[u]);
      request_old(cur[u], u, old[kBlocked ? u : 0]);
    }
    auto batch = [&](const int i, auto more_tag) {
      constexpr bool kMore = decltype(more_tag)::value;
      if constexpr (kMore) fetch_ids(nxt, i + K);
      GradT w[K];
      if constexpr (kWeighted) {
#pragma unroll
        for (int u = 0; u < K; ++u) w[u] = my_w[i + u];
      }
#pragma unroll
      for (int u = 0; u < K; ++u) {
        pair_t f[kPairs];
        const Pack<GradT, N> row_u = __builtin_bit_cast(Pack<GradT, N>, g[u]);
#pragma unroll
        for (int e = 0; e < 2 * kPairs; ++e) {
          float x = 0.f;
          if (e < N) {
            x = static_cast<float>(row_u.v[e < N ? e : 0]);
            if constexpr (kWeighted) x = A::mul(x, static_cast<float>(w[u]));
          }
          if (e & 1) f[e / 2].y = x; else f[e / 2].x = x;
        }
        if constexpr (kMore) {
          __builtin_amdgcn_sched_barrier(0);   // the registers of lookup i are free: request i + K into them
          g[u] = gather(nxt[u]);
          __builtin_amdgcn_sched_barrier(0);
        }
        {
#pragma clang fp contract(off)
#pragma unroll
          for (int j = 0; j < kPairs; ++j) acc[j] = acc[j] + f[j];
        }
        if (static_cast<int32_t>(cur[u]) < 0) {   // kRunEndBit: the run ends with this lookup
          const uint32_t flagged = static_cast<uint32_t>(my_rows[i + u]);
          if (first_pending) {
            park_head(static_cast<int64_t>(flagged));   // its first lookups are in earlier segments
          } else {
            GradT* dst = const_cast<GradT*>(RowPtr(lane_dst, static_cast<int64_t>(flagged & id_mask), width));
            float a[N];
            unpair(a);
            if constexpr (kBlocked) {
              if (flagged & shared_row_bit) {   // an earlier block stored this row: add what it holds
                const Pack<GradT, N> was = __builtin_bit_cast(Pack<GradT, N>, old[u]);
#pragma unroll
                for (int e = 0; e < N; ++e) a[e] = A::add(static_cast<float>(was.v[e]), a[e]);
              }
            }
            FlushStore<GradT, N>(dst, a);
          }
          first_pending = false;
#pragma unroll
          for (int j = 0; j < kPairs; ++j) asm("v_mov_b64 %0, 0" : "=v"(acc[j]));   // one move per pair
        }
        if constexpr (kBlocked && kMore) request_old(nxt[u], i + K + u, old[u]);   // (the slot is free again)
      }
      if constexpr (kMore) {
#pragma unroll
        for (int u = 0; u < K; ++u) cur[u] = nxt[u];
      }
    };
    int i = 0;
    for (; i + K < segment_len; i += K) batch(i, std::true_type{});
    batch(i, std::false_type{});   // the last K lookups: nothing left to request
    // The segment's last run goes on into the next segment (kPartWhole: it also came in): its
    // partial stays in registers until every walk of the workgroup is over, then it is parked on
    // top of the staged ids, which nobody reads any more.
    if (tail_shared)
      note(1, static_cast<int64_t>(static_cast<uint32_t>(my_rows[segment_len - 1])), kPartTail | (first_pending ? kPartWhole : 0));
    __syncthreads();
    if (tail_shared) {
      float a[N];
      unpair(a);
#pragma unroll
      for (int e = 0; e < N; ++e) part_tail[(seg * N + e) * lanes + lane_x] = a[e];
    }
  }
  __syncthreads();
  if (!active) return;

  // ---- chain the parked partials (see 3. above) ----
  const int flags = part_flags[seg];
  const int64_t remaining = nnz - block_begin;
  const int last_seg = static_cast<int>(
      remaining >= block_len ? segments_per_block - 1 : (remaining + segment_len - 1) / segment_len - 1);
  auto add_part = [&](float (&sum)[N], int s, int slot) {
    const float* p = (slot == 0 ? part_head : part_tail) + static_cast<size_t>(s) * N * lanes;
#pragma unroll
    for (int e = 0; e < N; ++e) sum[e] = A::add(sum[e], p[e * lanes + lane_x]);
  };
  if (flags & kPartHead) {
    // a run that came in from earlier segments ends in this one
    float sum[N];
#pragma unroll
    for (int e = 0; e < N; ++e) sum[e] = 0.f;
    int k = seg - 1;
    while (k >= 0 && (part_flags[k] & kPartWhole)) --k;
    const bool from_previous_block = k < 0;
    for (int m = from_previous_block ? 0 : k; m < seg; ++m) add_part(sum, m, 1);
    add_part(sum, seg, 0);
    const uint32_t flagged = static_cast<uint32_t>(part_row[seg * 2 + 0]);
    GradT* dst = lane_dst + static_cast<int64_t>(flagged & id_mask) * width;
    // (sample-blocked order: an earlier block stored this row -> add to it.  An atomic, not a read-modify-write: this
    // is the end of the workgroup, a read would be waited for in full, and there is at most one chain per segment)
    if (from_previous_block || (flagged & shared_row_bit)) FlushAtomic<N>(dst, sum);
    else FlushStore<GradT, N>(dst, sum);
  }
  if ((flags & kPartTail) && seg == last_seg) {
    // the workgroup's last run goes on into the next workgroup
    float sum[N];
#pragma unroll
    for (int e = 0; e < N; ++e) sum[e] = 0.f;
    int k = seg;
    while (k >= 0 && (part_flags[k] & kPartWhole)) --k;
    for (int m = k < 0 ? 0 : k; m <= seg; ++m) add_part(sum, m, 1);
    FlushAtomic<N>(lane_dst + static_cast<int64_t>(static_cast<uint32_t>(part_row[seg * 2 + 1]) & id_mask) * width, sum);
  }
}

//! Zero-initialisation for a COMPRESSED gradient without touching all of it.  With dense row
//! ids 0..num_unique-1 every output row is written by SegmentedScatterAddKernel: rows whose run
//! lies inside one workgroup get a plain store, so only (a) the rows of the first and last
//! lookup of every workgroup -- the only ones that can receive atomics -- and (b) rows beyond
//! the last id, should the caller have over-allocated, must be zero beforehand.  That is a few
//! MB instead of a memset of the whole buffer (293 MB, ~45 us, at the north-star shape).
//!   grid = sample_blocks * blocks_per_sample_block + tail workgroups (ZeroTailBlocks: at most kZeroTailBlocksPerCu per
//!   compute unit, all striding over the tail with 16-byte stores), block = 256
//! Sample-blocked order (sample_blocks > 1): the COO is `sample_blocks` arrays of `sample_block_len` lookups that
//! are scattered one after the other, each cut into workgroup ranges from its own start; ONE call zeroes the edge
//! rows of all of them up front (a row that block 0 stores and block 1 adds to must not be zeroed in between), and
//! the last id is the largest of the blocks' last ids.  A row with kSharedRowBit that an EARLIER block never wrote
//! does not exist, so zeroing it here and letting block 0 store it afterwards is always right.
constexpr int kZeroTailRowsPerBlock = 64;
//! The tail (rows past the last id) is usually EMPTY -- the caller read num_unique back and allocated exactly -- but
//! only the device knows: a bounded number of workgroups stride over it instead of one workgroup per 64 rows of the
//! whole buffer (8,938 workgroups that found nothing to do cost 5 us of the 8 us this kernel took at C4).  The bound
//! follows the device (4 workgroups per compute unit: a caller that over-allocates by gigabytes -- min(nnz, table
//! rows) rows with few distinct ones -- gets the whole chip storing 16 bytes per lane, not 128 workgroups of scalars).
constexpr int kZeroTailBlocksPerCu = 4;
inline int64_t ZeroTailBlocks(const int64_t zero_rows, const DeviceShape& dev) {
  if (zero_rows <= 0) return 0;
  const int64_t pieces = (zero_rows + kZeroTailRowsPerBlock - 1) / kZeroTailRowsPerBlock;
  const int64_t cap = static_cast<int64_t>(kZeroTailBlocksPerCu) * dev.compute_units;
  return pieces < cap ? pieces : cap;
}

//! Padded gradient (EmbeddingBackward(..., pad_to_capacity)): the rows past the device-side count are zero (the kernel
//! below) and must NAME rows so that (inverse_mapping, grad) over all num_rows entries is a valid uncoalesced COO
//! gradient.  They name rows of the batch itself -- a consumer with per-row state then touches no row the batch did not --
//! and DIFFERENT ones: entry i names the row of entry (i - unique) mod unique, i.e. every looked-up row gets at most
//! ceil(padding / unique) extra zero entries.  (Naming one row for the whole tail, as a first version did, makes
//! torch's coalesce() and index_add_ serialise tens of thousands of adds on that row: 46 ms for a 65,536-entry gradient.)
//! Launched after the scatter, which writes inverse_mapping[0, unique).
template <typename IndexT>
__global__ void __launch_bounds__(256)
NamePaddedRowsKernel(const IndexT* __restrict__ dense_ids /* sorted; the last one is unique - 1 */, const int64_t nnz,
                     IndexT* __restrict__ inverse_mapping, const int64_t num_rows) {
  const int64_t unique = static_cast<int64_t>(dense_ids[nnz - 1]) + 1;
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t i = unique + static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < num_rows; i += stride)
    inverse_mapping[i] = inverse_mapping[(i - unique) % unique];
}

template <typename GradT, typename IndexT>
__global__ void __launch_bounds__(256)
ZeroSharedAndTailRowsKernel(const IndexT* __restrict__ rows, const int64_t nnz, const int block_len,
                            const int64_t blocks_per_sample_block, const int64_t sample_block_len,
                            const int sample_blocks, const int width, const int64_t num_rows,
                            GradT* __restrict__ grad_out, const uint32_t* __restrict__ block_row_ids,
                            const int64_t capacity_rows,
                            const bool pad_tail /* padded gradient: the rows past the last id are zeroed up to num_rows */) {
  // sample-blocked order: `rows` holds pair numbers, block_row_ids[pair] the gradient row | kSharedRowBit
  auto row_of = [&](const int64_t g) -> int64_t {
    const uint32_t r = static_cast<uint32_t>(rows[g]);
    return static_cast<int64_t>(block_row_ids != nullptr ? (block_row_ids[r] & ~kSharedRowBit) : r);
  };
  const int64_t b = blockIdx.x;
  const int64_t num_blocks = blocks_per_sample_block * sample_blocks;
  if (b < num_blocks) {
    const int64_t p = b / blocks_per_sample_block;
    const int64_t origin = p * sample_block_len;
    const int64_t end = origin + sample_block_len < nnz ? origin + sample_block_len : nnz;
    const int64_t first = origin + (b - p * blocks_per_sample_block) * block_len;
    if (first >= end) return;
    const int64_t last = (first + block_len < end ? first + block_len : end) - 1;
    const int64_t r0 = row_of(first);
    const int64_t r1 = row_of(last);
    // (capacity_rows > 0: a launch whose last id does not fit the buffer writes nothing at all -- the scatter takes the
    // same decision from the same word and raises the caller's flag)
    if (capacity_rows > 0 && row_of(end - 1) >= capacity_rows) return;
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
      grad_out[r0 * width + c] = static_cast<GradT>(0);
      grad_out[r1 * width + c] = static_cast<GradT>(0);
    }
    return;
  }
  int64_t last_id = 0;
  for (int p = 0; p < sample_blocks; ++p) {
    const int64_t end = (p + 1) * sample_block_len < nnz ? (p + 1) * sample_block_len : nnz;
    const int64_t id = row_of(end - 1);
    last_id = id > last_id ? id : last_id;
  }
  // the tail is ONE contiguous range of bytes: 16-byte stores between its aligned ends, all tail workgroups striding
  // over it; the (at most 15-byte) ragged ends go out element by element from one thread
  char* const p0 = reinterpret_cast<char*>(grad_out + (last_id + 1) * width);
  char* const p1 = reinterpret_cast<char*>(grad_out + num_rows * width);
  if (p0 >= p1) return;
  (void)pad_tail;   // (the tail's row ids are written after the scatter: NamePaddedRowsKernel)
  char* a0 = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(p0) + 15) & ~uintptr_t{15});
  char* a1 = reinterpret_cast<char*>(reinterpret_cast<uintptr_t>(p1) & ~uintptr_t{15});
  if (a0 > p1) a0 = p1;
  if (a1 < a0) a1 = a0;
  const int64_t tail_blocks = static_cast<int64_t>(gridDim.x) - num_blocks;
  const int64_t worker = (b - num_blocks) * blockDim.x + threadIdx.x;
  const int64_t step = tail_blocks * blockDim.x * 16;
  typedef unsigned __attribute__((ext_vector_type(4))) word4_t;
  for (char* p = a0 + worker * 16; p < a1; p += step) *reinterpret_cast<word4_t*>(p) = word4_t{0u, 0u, 0u, 0u};
  if (worker == 0) {
    for (char* p = p0; p < a0; p += sizeof(GradT)) *reinterpret_cast<GradT*>(p) = static_cast<GradT>(0);
    for (char* p = a1; p < p1; p += sizeof(GradT)) *reinterpret_cast<GradT*>(p) = static_cast<GradT>(0);
  }
}

//! EmbeddingBackward in the REFERENCE's arithmetic (EmbeddingBackwardReferenceSums; opt-in, for verification): one
//! lane group per run of equal row ids walks the run in nz order with the product and the running sum rounded to
//! GradT at every lookup -- `grad += grad_y * weight` in GradT, the CPU reference's loop
//! (embedding_lookup_cpu.hpp:131-143) -- so the result is bit-identical to it for ANY data.  A rounding chain cannot be
//! cut into partial sums, hence one run = one chain whatever its length: the hottest row of the C4 batch is 65,528
//! dependent additions.  What CAN be shared is everything around the chain:
//!   * short runs: a lane group chains the runs that START among its kReferenceSpan lookups in one pass over the span --
//!     row ids, sample ids, weights and the span's rows of grad_y are requested up front, unconditionally, at clamped
//!     positions (two memory round trips per span instead of six dependent ones per run);
//!   * runs longer than kReferenceLongRun lookups and than the workgroup's own span (so that at most one can START inside
//!     a workgroup's lookups): the whole workgroup walks it with its wavefronts SPECIALISED -- the gather groups fetch
//!     chunks of rows of grad_y into LDS through a pipeline two chunks deep (rows requested two chunks ahead, their ids
//!     four; nothing compared or counted per row: a chunk whose LAST lookup belongs to the run belongs to it entirely),
//!     the lanes of the chain wavefront run the chain out of LDS -- the wavefront's lane groups share it, each lane on
//!     N / 2 or N / 4 of its group's elements, packed half arithmetic for fp16, LDS addresses as immediate offsets for the
//!     common row widths; two LDS buffers, ONE barrier per chunk.  Up to 1,024 threads per workgroup, so that a gather
//!     thread handles one or two rows of a chunk.  tools/reference_sums_run_probe.py, ns per lookup of one run of 65,528:
//!     295 with one lane group per run, 50 with 256 threads sharing the work, 30 specialised, 22 with chunks of 60 rows,
//!     14.4 with the shared chain, the constant strides and the lighter, deeper gather side.
//!     C4 (fp16 / fp32): 26.4 / 23.7 -> 1.8 / 2.9 -> 1.2 / 1.8 ms.
//!   block = (lanes_per_row, groups); every group looks at kReferenceSpan consecutive lookups and walks the runs that
//!   START there.
constexpr int kReferenceSpan = 8;
constexpr int kReferenceLongRun = 256;
constexpr int kReferenceBlockThreads = 1024;  //!< lane groups x lanes of a workgroup of the reference-sums kernel
constexpr int kReferenceMaxPer = 2;           //!< rows of a chunk per gather group: with 1,024 threads one, two for rows of 4 KB and more

//! Long-run path: which lane groups of a workgroup chain and which gather.  The wavefront(s) of group 0 run the chain
//! (ReferenceChainSplit of its groups share it, the rest of that wavefront idles: work of theirs would be executed by
//! the chain wavefront too); every other group gathers `per` rows of each chunk.
struct ReferenceLongRunShape {
  int chain_groups;   //!< groups in the chain wavefront(s): lanes < 64 -> 64 / lanes, else 1
  int gather_groups;  //!< the rest
  int per;            //!< rows of a chunk per gather group (1 .. kReferenceMaxPer)
  int chunk_rows;     //!< per x gather_groups; 0 = no long-run path for this shape
};
__host__ __device__ inline ReferenceLongRunShape ReferenceLongRun(const size_t row_bytes, const int lanes, const int groups) {
  ReferenceLongRunShape s{1, 0, 0, 0};
  const bool aligned = lanes > 0 && (lanes < 64 ? 64 % lanes == 0 : lanes % 64 == 0);
  if (!aligned || (lanes * groups) % 64 != 0) return s;
  s.chain_groups = lanes < 64 ? 64 / lanes : 1;
  s.gather_groups = groups - s.chain_groups;
  if (s.gather_groups < 1) return s;
  // two LDS buffers (+ their weights) within 62 KB -- below the 64 KB a workgroup gets without asking: the barrier and
  // the bookkeeping of a chunk are paid once per chunk, so a chunk should be as long as LDS allows (512-byte rows: 60)
  const size_t max_rows = (size_t{62} << 10) / (2 * (row_bytes + 8));
  s.per = static_cast<int>(max_rows / static_cast<size_t>(s.gather_groups));
  if (s.per > kReferenceMaxPer) s.per = kReferenceMaxPer;
  if (s.per < 1) return s;
  s.chunk_rows = s.per * s.gather_groups;
  return s;
}

//! One step of the reference's rounding chain on the N elements of a lane: acc = GradT(acc + GradT(x * w)), every
//! operation rounded to GradT (embedding_lookup_cpu.hpp:139-142).  fp16 runs on packed half arithmetic (v_pk_mul_f16 /
//! v_pk_add_f16: one IEEE operation per element, the same bits as the float-and-round form -- a sum or product of two
//! binary16 values rounded through binary32 cannot round differently, 24 >= 2 x 11 + 2 -- at a fifth of the instructions;
//! never contracted into an FMA, which would skip the product's rounding).
template <typename GradT, int N, bool kWeighted>
__device__ __forceinline__ void ReferenceChainStep(Pack<GradT, N>& acc, const Pack<GradT, N>& x, const GradT w) {
  if constexpr (std::is_same<GradT, _Float16>::value && N % 2 == 0) {
    typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
    half2_t* a2 = reinterpret_cast<half2_t*>(&acc);
    const half2_t* x2 = reinterpret_cast<const half2_t*>(&x);
    const half2_t w2 = {w, w};
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
#pragma clang fp contract(off)
      half2_t v = x2[i];
      if constexpr (kWeighted) v = v * w2;
      a2[i] = a2[i] + v;
    }
  } else {
#pragma unroll
    for (int e = 0; e < N; ++e) {
#pragma clang fp contract(off)
      float v = static_cast<float>(x.v[e]);
      if constexpr (kWeighted) v = static_cast<float>(static_cast<GradT>(v * static_cast<float>(w)));   // product in GradT
      acc.v[e] = static_cast<GradT>(static_cast<float>(acc.v[e]) + v);                                  // sum in GradT
    }
  }
}

//! How many lane groups of the chain wavefront SHARE the chain of a long run: a wavefront pays for an instruction
//! whatever the number of its lanes that execute it, so with rows of 32 lanes (512 bytes) half of the chain wavefront
//! idled while the other half issued four packed additions per lookup; split 2 (4) ways, every lane chains N / 2 (N / 4)
//! of its group's elements and the wavefront issues half (a quarter of) the additions per lookup.  Pieces of at least
//! 4 bytes (one LDS word); the arithmetic per element is the same, so are the bits.
template <typename GradT, int N>
__host__ __device__ inline int ReferenceChainSplit(const int chain_groups) {
  int split = 1;
  while (split * 2 <= chain_groups && split * 2 <= 4 && N % (split * 2) == 0 &&
         (N / (split * 2)) * static_cast<int>(sizeof(GradT)) >= 4)
    split *= 2;
  return split;
}

//! The rows [0, count) of one LDS chunk chained onto `sum` (E elements of a lane; `mine` = the lane's piece of row 0).
//! kWidth > 0: the row width in elements is a compile-time constant -- the LDS addresses of a batch become immediate
//! offsets of ONE base register (no address arithmetic between the reads: a third of the chain wavefront's instructions).
template <typename GradT, int E, bool kWeighted, int kWidth = 0>
__device__ __forceinline__ void ReferenceChainChunk(Pack<GradT, E>& sum, const GradT* mine, const GradT* w_now,
                                                    const int count, const int runtime_width) {
  const int width = kWidth > 0 ? kWidth : runtime_width;
  typedef uint32_t __attribute__((ext_vector_type(sizeof(Pack<GradT, E>) / 4))) piece_t;
  int j = 0;
  // LDS reads requested together, then the dependent additions over them: batches of 8, then one of 4 (a full chunk of
  // 512-byte rows has 60: no row is left to the one-by-one loop, whose every read is a full LDS round trip)
  auto batch = [&](auto size) {
    constexpr int kBatch = decltype(size)::value;
    piece_t x[kBatch];
    GradT wj[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      x[u] = *reinterpret_cast<const piece_t*>(mine + static_cast<size_t>(j + u) * width);
      wj[u] = kWeighted ? w_now[j + u] : static_cast<GradT>(0);
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u)
      ReferenceChainStep<GradT, E, kWeighted>(sum, __builtin_bit_cast(Pack<GradT, E>, x[u]), wj[u]);
    j += kBatch;
  };
  while (j + 12 <= count) batch(std::integral_constant<int, 12>{});
  if (j + 8 <= count) batch(std::integral_constant<int, 8>{});
  if (j + 4 <= count) batch(std::integral_constant<int, 4>{});
  if (j + 2 <= count) batch(std::integral_constant<int, 2>{});
  if (j < count) batch(std::integral_constant<int, 1>{});
}

//! fn(std::integral_constant<int, E>) for E = N / split (the splits ReferenceChainSplit can return for this type).
template <typename GradT, int N, typename Fn>
__device__ __forceinline__ void ReferenceWithSplit(const int split, Fn&& fn) {
  if constexpr (N % 4 == 0 && (N / 4) * sizeof(GradT) >= 4) {
    if (split == 4) {
      fn(std::integral_constant<int, N / 4>{});
      return;
    }
  }
  if constexpr (N % 2 == 0 && (N / 2) * sizeof(GradT) >= 4) {
    if (split == 2) {
      fn(std::integral_constant<int, N / 2>{});
      return;
    }
  }
  fn(std::integral_constant<int, N>{});
}

template <typename GradT, typename IndexT, int N, bool kWeighted>
__global__ void __launch_bounds__(kMaxBlockThreads)
ReferenceSumsScatterKernel(const GradT* __restrict__ grad_y, const int width, const IndexT* __restrict__ rows,
                           const IndexT* __restrict__ sample_ids, const GradT* __restrict__ weights, const int64_t nnz,
                           GradT* __restrict__ grad_out, const bool add_to_output,
                           const IndexT* __restrict__ run_ids, IndexT* __restrict__ inverse_mapping,
                           const int chunk_rows /* > 0: long runs are walked by the whole workgroup */) {
  extern __shared__ __attribute__((aligned(16))) unsigned char reference_lds[];
  __shared__ long long long_head;          // first lookup of the long run that starts in this workgroup's span, or -1
  __shared__ int chunk_full[2];            // the chunk parked in this buffer lies entirely inside the run
  __shared__ int last_count;               // the run's rows in its last chunk
  const int lane_x = threadIdx.x;
  // With the long-run path the grid has TWO halves over the same lookups: the first half only looks for long runs and
  // walks them (a workgroup without one leaves at once), the second walks the short runs.  Workgroups are dispatched in
  // blockIdx order, so every long run -- the kernel's critical path is the hottest row's chain -- starts within
  // microseconds instead of queueing behind thousands of short-run workgroups (C4: the hottest run takes 2.0 ms, the
  // kernel took 4.0 with one mixed grid).
  const unsigned half = chunk_rows > 0 ? gridDim.x / 2 : gridDim.x;
  const bool long_phase = chunk_rows > 0 && blockIdx.x < half;
  const bool short_phase = !long_phase;
  const int64_t block_in_phase = blockIdx.x < half ? blockIdx.x : blockIdx.x - half;
  const int64_t group = block_in_phase * blockDim.y + threadIdx.y;
  const int64_t column0 = static_cast<int64_t>(lane_x) * N;
  typedef uint32_t __attribute__((ext_vector_type(sizeof(Pack<GradT, N>) / 4))) raw_t;
  constexpr int K = 8;
  // "long" = longer than the workgroup's own span of lookups (narrow rows: 64 and more groups of 8 lookups), so that at
  // most ONE long run can start inside it
  const int span = static_cast<int>(blockDim.y) * kReferenceSpan;
  const int long_run = span > kReferenceLongRun ? span : kReferenceLongRun;
  if (chunk_rows > 0) {
    if (threadIdx.x == 0 && threadIdx.y == 0) {
      long_head = -1;
      chunk_full[0] = 0;
      chunk_full[1] = 0;
      last_count = 0;
    }
    __syncthreads();
  }
  if (long_phase) {
    // does a long run START among this workgroup's lookups?  One thread per lookup, three loads at clamped positions
    // and one round trip for the whole workgroup (the lane groups used to walk their eight lookups one after the other,
    // a dependent load or two each: the hottest row's workgroup sits somewhere among thousands of these)
    const int threads = static_cast<int>(blockDim.x * blockDim.y);
    const int64_t first = block_in_phase * span;
    for (int i = static_cast<int>(threadIdx.y * blockDim.x + threadIdx.x); i < span; i += threads) {
      const int64_t at = first + i;
      const int64_t here = at < nnz ? at : nnz - 1;
      const int64_t far = at + long_run;
      const IndexT row = rows[here];
      const IndexT before = rows[here > 0 ? here - 1 : 0];
      const IndexT ahead = rows[far < nnz ? far : nnz - 1];
      if (at < nnz && (at == 0 || before != row) && far < nnz && ahead == row)
        long_head = at;                          // (one per workgroup at most: the run outlasts the workgroup's span)
    }
  }
  const int64_t p0 = group * kReferenceSpan;
  if (short_phase && p0 < nnz) {
    // ---- the short runs that START among this lane group's kReferenceSpan lookups ----
    // Everything the span can need is requested up front, UNCONDITIONALLY, at clamped positions -- the row ids around
    // the span, the ids `long_run` further on, sample ids, weights, run names, then the span's rows of grad_y: two memory
    // round trips per span.  (Walking the lookups one by one, every load behind the comparison before it, was six
    // dependent round trips per run: uniform indices, 3.4 M runs of one or two lookups, took 2.3 ms.)  One pass over the
    // span then chains every run that starts in it, in lookup order; a run that outlasts the span is finished in batches
    // of K lookups with the same discipline.
    constexpr int S = kReferenceSpan;
    const int64_t last_pos = nnz - 1;
    IndexT rid[S + 2];                           // rows[p0 - 1 .. p0 + S]
#pragma unroll
    for (int j = 0; j < S + 2; ++j) {
      const int64_t at = p0 - 1 + j;
      rid[j] = rows[at < 0 ? 0 : (at < nnz ? at : last_pos)];
    }
    IndexT ahead[S];                             // rows[p + long_run]: does a run that starts at p belong to the other half?
    IndexT sid[S];
    GradT w[S];
    IndexT name[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const int64_t at = p0 + j < nnz ? p0 + j : last_pos;
      const int64_t far = p0 + j + long_run;
      ahead[j] = chunk_rows > 0 ? rows[far < nnz ? far : last_pos] : static_cast<IndexT>(0);
      sid[j] = sample_ids[at];
      w[j] = kWeighted ? weights[at] : static_cast<GradT>(0);
      name[j] = run_ids != nullptr ? run_ids[at] : static_cast<IndexT>(0);
    }
    raw_t g[S];
#pragma unroll
    for (int j = 0; j < S; ++j)
      g[j] = *reinterpret_cast<const raw_t*>(RowPtr(grad_y + column0, static_cast<int64_t>(sid[j]), width));
    bool active = false;                         // a run that started in this span is being chained
    IndexT current = 0;
    Pack<GradT, N> acc;
#pragma unroll
    for (int e = 0; e < N; ++e) acc.v[e] = static_cast<GradT>(0);
    auto store = [&](const IndexT row) {
      *reinterpret_cast<Pack<GradT, N>*>(grad_out + static_cast<int64_t>(row) * width + column0) = acc;
    };
#pragma unroll
    for (int j = 0; j < S; ++j) {
      const int64_t p = p0 + j;
      const IndexT row = rid[j + 1];
      const bool head = p < nnz && (p == 0 || rid[j] != row);
      if (head) {
        if (active) store(current);              // the run before this one started in the span too, and ends here
        if (run_ids != nullptr && lane_x == 0) inverse_mapping[row] = name[j];
        // (a long run is walked by the first half of the grid)
        active = !(chunk_rows > 0 && p + long_run < nnz && ahead[j] == row);
        current = row;
        if (add_to_output && active) {
          acc = *reinterpret_cast<const Pack<GradT, N>*>(grad_out + static_cast<int64_t>(row) * width + column0);
        } else {
#pragma unroll
          for (int e = 0; e < N; ++e) acc.v[e] = static_cast<GradT>(0);
        }
      }
      if (active && p < nnz) ReferenceChainStep<GradT, N, kWeighted>(acc, __builtin_bit_cast(Pack<GradT, N>, g[j]), w[j]);
    }
    if (active) {
      // the run goes on past the span (rid[S + 1] = rows[p0 + S]): batches of K lookups, loads first, comparisons after
      bool more = p0 + S < nnz && rid[S + 1] == current;
      for (int64_t q = p0 + S; more; q += K) {
        IndexT r2[K], s2[K];
        GradT w2[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
          const int64_t at = q + j < nnz ? q + j : last_pos;
          r2[j] = rows[at];
          s2[j] = sample_ids[at];
          w2[j] = kWeighted ? weights[at] : static_cast<GradT>(0);
        }
        raw_t g2[K];
#pragma unroll
        for (int j = 0; j < K; ++j)
          g2[j] = *reinterpret_cast<const raw_t*>(RowPtr(grad_y + column0, static_cast<int64_t>(s2[j]), width));
        bool inside = true;                      // the run's lookups are a prefix of the batch (the COO is sorted)
#pragma unroll
        for (int j = 0; j < K; ++j) {
          inside = inside && q + j < nnz && r2[j] == current;
          if (inside) ReferenceChainStep<GradT, N, kWeighted>(acc, __builtin_bit_cast(Pack<GradT, N>, g2[j]), w2[j]);
        }
        more = inside;
      }
      store(current);
    }
  }
  if (!long_phase) return;
  __syncthreads();
  if (long_head < 0) return;
  // ---- the long run that starts here: gathered by the gather groups, chained out of LDS by group 0 ----
  const int64_t p = long_head;
  const IndexT row = rows[p];
  const ReferenceLongRunShape shape =
      ReferenceLongRun(static_cast<size_t>(width) * sizeof(GradT), static_cast<int>(blockDim.x), static_cast<int>(blockDim.y));
  const int gather_groups = shape.gather_groups, per = shape.per;
  // the groups of the chain wavefront that share the chain, each lane on N / split of its group's elements
  const int split = ReferenceChainSplit<GradT, N>(shape.chain_groups);
  const bool chains = static_cast<int>(threadIdx.y) < split;
  const int chain_column = static_cast<int>(column0) + (chains ? static_cast<int>(threadIdx.y) * (N / split) : 0);
  const bool gathers = static_cast<int>(threadIdx.y) >= shape.chain_groups;
  const int gi = static_cast<int>(threadIdx.y) - shape.chain_groups;      // this gather group's number
  GradT* stage = reinterpret_cast<GradT*>(reference_lds);      // [2][chunk_rows][width]
  GradT* stage_w = stage + 2 * static_cast<size_t>(chunk_rows) * width;   // [2][chunk_rows]
  GradT* dst = grad_out + static_cast<int64_t>(row) * width + chain_column;
  Pack<GradT, N> sum;                        // the chain's state lives in GradT: it is rounded to it at every step anyway
#pragma unroll                               // (its first N / split elements are this lane's)
  for (int e = 0; e < N; ++e) sum.v[e] = static_cast<GradT>(0);
  if (chains && add_to_output)
    ReferenceWithSplit<GradT, N>(split, [&](auto e) {
      constexpr int E = decltype(e)::value;
      const Pack<GradT, E> before = *reinterpret_cast<const Pack<GradT, E>*>(dst);
#pragma unroll
      for (int i = 0; i < E; ++i) sum.v[i] = before.v[i];
    });
  // The gather side is a pipeline TWO chunks deep, in two statically named register sets used in turn (A: even chunks,
  // B: odd ones -- no register that a load is still writing is ever copied):
  //   Ids   : sample ids and weights of a gather group's rows of a chunk, and the row id at the chunk's LAST position;
  //           requested four chunks ahead of the chunk being chained, consumed two iterations later;
  //   Stage : the rows themselves (16 bytes per lane and row), requested two chunks ahead, parked into LDS two
  //           iterations later -- with one chunk of slack the loop ran at the latency of a memory round trip per chunk
  //           (0.76 us without any chain: the streamed id arrays miss in every cache), whatever the chain cost.
  // The COO is sorted, so a chunk whose last lookup still belongs to the run belongs to it entirely: nothing is compared
  // or counted per row (that was half of the gather side's instructions); only the run's last chunk is counted, once.
  struct Ids {
    IndexT sid[kReferenceMaxPer];
    GradT w[kReferenceMaxPer];
    IndexT last_rid;                   //!< the row id at the chunk's last position (clamped)
    bool last_inside;                  //!< that position exists
  };
  struct Stage {
    raw_t held[kReferenceMaxPer];
    GradT w[kReferenceMaxPer];
    bool full;                         //!< the whole chunk belongs to the run
  };
  // UNCONDITIONAL loads at clamped positions, nothing decided here: a load under a runtime condition, or a comparison of a
  // loaded row id kept as a lane mask while the load is in flight, becomes a full wait behind every single load (eight
  // dependent round trips per chunk instead of one: 3.8 us per 64 lookups).  Slots beyond `per` repeat slot 0's position.
  auto look_up = [&](Ids& l, const int64_t first) {
#pragma unroll
    for (int k = 0; k < kReferenceMaxPer; ++k) {
      const int64_t at = first + gi + static_cast<int64_t>(k < per ? k : 0) * gather_groups;
      const int64_t safe = at < nnz ? at : nnz - 1;
      l.sid[k] = sample_ids[safe];
      if constexpr (kWeighted) l.w[k] = weights[safe];
    }
    const int64_t last = first + chunk_rows - 1;
    l.last_inside = last < nnz;
    l.last_rid = rows[last < nnz ? last : nnz - 1];
  };
  // (every slot loads -- a row of the run or, past its end, whatever row the clamped position names: a valid sample id)
  auto request = [&](Stage& st, const Ids& l) {
#pragma unroll
    for (int k = 0; k < kReferenceMaxPer; ++k) {
      st.held[k] = *reinterpret_cast<const raw_t*>(RowPtr(grad_y + column0, static_cast<int64_t>(l.sid[k]), width));
      st.w[k] = l.w[k];
    }
    st.full = l.last_inside && l.last_rid == row;
  };
  // registers -> LDS, every slot of the chunk (rows past the run's end are parked too and never read)
  auto park = [&](const Stage& st, const int buffer) {
#pragma unroll
    for (int k = 0; k < kReferenceMaxPer; ++k) {
      if (k < per) {
        const int slot = gi + k * gather_groups;
        *reinterpret_cast<raw_t*>(stage + (static_cast<size_t>(buffer) * chunk_rows + slot) * width + column0) = st.held[k];
        if constexpr (kWeighted)
          if (lane_x == 0) stage_w[buffer * chunk_rows + slot] = st.w[k];
      }
    }
    if (gi == 0 && lane_x == 0) chunk_full[buffer] = st.full ? 1 : 0;
  };
  int64_t q = p;                             // first lookup of the chunk that is parked next
  // one chunk: its rows into LDS, the requests of the chunk two further on, the barrier, the chain.  true = the run ended.
  auto step = [&](Stage& st, Ids& l, const int buffer) -> bool {
    if (gathers) {
      park(st, buffer);                      // (waits for this chunk's rows, requested two iterations ago)
      request(st, l);                        // the rows of chunk + 2 (their ids were requested two iterations ago)
      look_up(l, q + 4 * static_cast<int64_t>(chunk_rows));
    }
    __syncthreads();                         // the chunk is in LDS; the buffer written next is the other one
    int count = chunk_rows;
    if (chunk_full[buffer] == 0) {           // (the same for every thread) the run ends inside this chunk: count its rows
      if (threadIdx.y < static_cast<unsigned>(shape.chain_groups)) {      // the chain wavefront(s), all lanes
        const int lane = static_cast<int>(threadIdx.y) * static_cast<int>(blockDim.x) + lane_x;
        const int lanes_here = shape.chain_groups * static_cast<int>(blockDim.x);
        int inside = 0;
        for (int base = 0; base < chunk_rows; base += lanes_here) {
          const int64_t at = q + base + lane;
          const IndexT rid = rows[at < nnz ? at : nnz - 1];
          inside += (base + lane < chunk_rows && at < nnz && rid == row) ? 1 : 0;
        }
        if (inside > 0) atomicAdd(&last_count, inside);
      }
      __syncthreads();
      count = last_count;
    }
    if (chains) {
      const GradT* mine = stage + static_cast<size_t>(buffer) * chunk_rows * width + chain_column;
      const GradT* w_now = stage_w + buffer * chunk_rows;
      ReferenceWithSplit<GradT, N>(split, [&](auto e) {
        constexpr int E = decltype(e)::value;
        Pack<GradT, E> part;
#pragma unroll
        for (int i = 0; i < E; ++i) part.v[i] = sum.v[i];
        switch (width) {                       // (the common widths as constants; the same bits, fewer instructions)
          case 128: ReferenceChainChunk<GradT, E, kWeighted, 128>(part, mine, w_now, count, width); break;
          case 256: ReferenceChainChunk<GradT, E, kWeighted, 256>(part, mine, w_now, count, width); break;
          case 512: ReferenceChainChunk<GradT, E, kWeighted, 512>(part, mine, w_now, count, width); break;
          default: ReferenceChainChunk<GradT, E, kWeighted>(part, mine, w_now, count, width);
        }
#pragma unroll
        for (int i = 0; i < E; ++i) sum.v[i] = part.v[i];
      });
    }
    q += chunk_rows;
    return count < chunk_rows;               // the run ended inside this chunk (the same count for every thread)
  };
  Ids ids_a, ids_b;
  Stage rows_a, rows_b;
  if (gathers) {
    look_up(ids_a, p);
    look_up(ids_b, p + chunk_rows);
    request(rows_a, ids_a);
    request(rows_b, ids_b);
    look_up(ids_a, p + 2 * static_cast<int64_t>(chunk_rows));
    look_up(ids_b, p + 3 * static_cast<int64_t>(chunk_rows));
  }
  for (;;) {
    if (step(rows_a, ids_a, 0)) break;
    if (step(rows_b, ids_b, 1)) break;
  }
  if (chains)
    ReferenceWithSplit<GradT, N>(split, [&](auto e) {
      constexpr int E = decltype(e)::value;
      Pack<GradT, E> part;
#pragma unroll
      for (int i = 0; i < E; ++i) part.v[i] = sum.v[i];
      *reinterpret_cast<Pack<GradT, E>*>(dst) = part;
    });
}

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_SCATTER_ADD_KERNELS_HPP_
