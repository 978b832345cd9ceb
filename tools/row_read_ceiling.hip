// What can a random-row gather reach at all?  Loads only -- no reduction, no output rows: every wavefront reads
// rows of R bytes at uniformly random row ids of a table far larger than the caches, 16 bytes per lane (R / 16 lanes
// per row, 64 * 16 / R rows per wavefront instruction), K independent loads in flight per lane, and folds what it read
// into one word so that nothing is optimised away.  The GB/s of this loop is the ceiling EmbeddingForward's gather can be
// judged against for narrow rows (10M x 32 fp32, alpha = 0 reaches 0.49-0.58 of the HBM peak against
// 0.72 for 512-byte rows -- is that DRAM or the kernel?).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cuembed_amd/csrc tools/row_read_ceiling.hip \
//         cuembed_amd/csrc/utils/synthetic_inputs.cpp -o tools/row_read_ceiling
//   tools/row_read_ceiling            (prints one line per row size x loads in flight x load policy)
//   tools/row_read_ceiling --c2 [alpha]   the HEADLINE's access pattern with loads only: 10M x 512-byte rows, batches of
//                                     65,536 samples x 64 lookups from the reference's generator (alpha 1.15), two samples
//                                     per wavefront walking their lookups in order, K in flight -- what EmbeddingForward's
//                                     0.137 ms would be if pooling, index staging and the output row cost nothing.
#include <hip/hip_runtime.h>

#include "cuembed/include/embedding_lookup.hpp"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#define HIP_OK(x)                                                                                   \
  do {                                                                                              \
    hipError_t e_ = (x);                                                                            \
    if (e_ != hipSuccess) {                                                                         \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));        \
      std::exit(2);                                                                                 \
    }                                                                                               \
  } while (0)

typedef unsigned __attribute__((ext_vector_type(4))) word4_t;

// lookups: row ids; each group of (row_bytes / 16) lanes reads one row per load instruction.
template <int K, bool kNonTemporal>
__global__ void __launch_bounds__(256) GatherRowsKernel(const char* __restrict__ table, const int row_bytes,
                                                        const int* __restrict__ lookups, const int64_t num_lookups,
                                                        unsigned* __restrict__ sink) {
  const int lanes_per_row = row_bytes / 16;
  const int rows_per_wave = 64 / lanes_per_row;
  const int lane = threadIdx.x & 63;
  const int sub = lane / lanes_per_row, part = lane % lanes_per_row;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t waves = (static_cast<int64_t>(gridDim.x) * blockDim.x) >> 6;
  word4_t acc = word4_t{0u, 0u, 0u, 0u};
  // wavefront w takes lookups [w * chunk, (w + 1) * chunk), K * rows_per_wave of them per iteration
  const int64_t chunk = (num_lookups + waves - 1) / waves;
  const int64_t begin = wave * chunk, end = begin + chunk < num_lookups ? begin + chunk : num_lookups;
  for (int64_t i = begin; i + static_cast<int64_t>(K) * rows_per_wave <= end; i += static_cast<int64_t>(K) * rows_per_wave) {
    word4_t v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int row = lookups[i + k * rows_per_wave + sub];
      const word4_t* p = reinterpret_cast<const word4_t*>(table + static_cast<int64_t>(row) * row_bytes + part * 16);
      v[k] = kNonTemporal ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc ^= v[k];
  }
  const unsigned folded = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (folded == 0x12345678u) sink[0] = folded;   // (practically never: keeps the loads)
}

// The forward's launch shape with loads only: the half-wavefront `sub` of wavefront w walks the lookups of sample
// 2 w + sub in order, K rows in flight (32 lanes x 16 bytes = one 512-byte row per load instruction and half-wavefront).
template <int K, bool kNonTemporal>
__global__ void __launch_bounds__(256) GatherSamplesKernel(const char* __restrict__ table, const int* __restrict__ lookups,
                                                           const int batch, const int hotness, unsigned* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 5, part = lane & 31;
  const int64_t wave = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
  const int64_t sample = wave * 2 + sub;
  if (sample >= batch) return;
  const int* mine = lookups + sample * hotness;
  word4_t acc = word4_t{0u, 0u, 0u, 0u};
  for (int j = 0; j + K <= hotness; j += K) {
    word4_t v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const word4_t* p = reinterpret_cast<const word4_t*>(table + static_cast<int64_t>(mine[j + k]) * 512 + part * 16);
      v[k] = kNonTemporal ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) acc ^= v[k];
  }
  const unsigned folded = acc.x ^ acc.y ^ acc.z ^ acc.w;
  if (folded == 0x12345678u) sink[0] = folded;
}

// ... and the same loop with the forward's other ingredients switched on one at a time (`--c2-parts [alpha]`): where do
// the 7-12 % between the loads-only time and EmbeddingForward's go at alpha = 0?
//   kStage: the workgroup's 8 x 64 indices go through LDS first (one coalesced load + barrier), as kLdsStaged does
//   kPool : fp16 -> fp32 conversion and in-order adds of every row (8 elements per lane) instead of the XOR fold
//   kStore: the pooled row is written (512 bytes per sample, fp16)
template <int K, bool kNonTemporal, bool kStage, bool kPool, bool kStore>
__global__ void __launch_bounds__(256) GatherSamplesPartsKernel(const char* __restrict__ table,
                                                                const int* __restrict__ lookups, const int batch,
                                                                const int hotness, unsigned* __restrict__ sink,
                                                                _Float16* __restrict__ out) {
  __shared__ int staged[8 * 64];
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 5, part = lane & 31;
  const int slot = (threadIdx.x >> 6) * 2 + sub;      // sample of the workgroup, 0..7
  const int64_t sample = static_cast<int64_t>(blockIdx.x) * 8 + slot;
  const int* mine = lookups + sample * hotness;
  if constexpr (kStage) {
    const int64_t first = static_cast<int64_t>(blockIdx.x) * 8 * hotness;
    for (int i = threadIdx.x; i < 8 * hotness; i += 256) staged[i] = lookups[first + i];
    __syncthreads();
    mine = staged + slot * hotness;
  }
  if (sample >= batch) return;
  typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
  word4_t acc = word4_t{0u, 0u, 0u, 0u};
  float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j + K <= hotness; j += K) {
    word4_t v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const word4_t* p = reinterpret_cast<const word4_t*>(table + static_cast<int64_t>(mine[j + k]) * 512 + part * 16);
      v[k] = kNonTemporal ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      if constexpr (kPool) {
        const half8_t h = __builtin_bit_cast(half8_t, v[k]);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] += static_cast<float>(h[e]);
      } else {
        acc ^= v[k];
      }
    }
  }
  if constexpr (kPool) {
    half8_t r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = static_cast<_Float16>(sum[e]);
    acc = __builtin_bit_cast(word4_t, r);
  }
  if constexpr (kStore) {
    *reinterpret_cast<word4_t*>(out + sample * 256 + part * 8) = acc;
  } else {
    const unsigned folded = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (folded == 0x12345678u) sink[0] = folded;
  }
}

// ... and the full forward (staged indices, fp32 pooling, row stored) with every workgroup walking kGroups consecutive
// groups of 8 samples: the next group's indices are requested before the current group's rows, so a group's output store
// and the start-up of the next one (index load + barrier) overlap row loads instead of leaving the wave slot idle.
template <int K, int kGroups>
__global__ void __launch_bounds__(256) GatherSamplesLoopKernel(const char* __restrict__ table,
                                                               const int* __restrict__ lookups, const int batch,
                                                               _Float16* __restrict__ out) {
  constexpr int kHot = 64;
  __shared__ int staged[2][8 * kHot];
  const int lane = threadIdx.x & 63;
  const int sub = lane >> 5, part = lane & 31;
  const int slot = (threadIdx.x >> 6) * 2 + sub;
  const int64_t group0 = static_cast<int64_t>(blockIdx.x) * kGroups;
  const int64_t total = static_cast<int64_t>(batch) * kHot;
  typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
  int ahead[2];
  auto request = [&](const int64_t group) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int64_t i = group * 8 * kHot + r * 256 + threadIdx.x;
      ahead[r] = i < total ? lookups[i] : 0;
    }
  };
  request(group0);
  staged[0][threadIdx.x] = ahead[0];
  staged[0][256 + threadIdx.x] = ahead[1];
  __syncthreads();
  for (int g = 0; g < kGroups; ++g) {
    const int64_t sample = (group0 + g) * 8 + slot;
    if (g + 1 < kGroups) request(group0 + g + 1);
    const int* mine = staged[g & 1] + slot * kHot;
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (sample < batch) {
      for (int j = 0; j + K <= kHot; j += K) {
        word4_t v[K];
#pragma unroll
        for (int k = 0; k < K; ++k)
          v[k] = *reinterpret_cast<const word4_t*>(table + static_cast<int64_t>(mine[j + k]) * 512 + part * 16);
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const half8_t h = __builtin_bit_cast(half8_t, v[k]);
#pragma unroll
          for (int e = 0; e < 8; ++e) sum[e] += static_cast<float>(h[e]);
        }
      }
      half8_t r;
#pragma unroll
      for (int e = 0; e < 8; ++e) r[e] = static_cast<_Float16>(sum[e]);
      __builtin_nontemporal_store(__builtin_bit_cast(word4_t, r), reinterpret_cast<word4_t*>(out + sample * 256 + part * 8));
    }
    if (g + 1 < kGroups) {
      staged[(g + 1) & 1][threadIdx.x] = ahead[0];
      staged[(g + 1) & 1][256 + threadIdx.x] = ahead[1];
      __syncthreads();
    }
  }
}

extern "C" int64_t cuembed_harness_generate_indices(int64_t num_categories, int batch, int hotness, double alpha,
                                                    int shuffle, int permute, int index_is_64, const int32_t* offsets,
                                                    void* out);

template <int K, bool kNt>
double RunSamples(const char* table, const int* lookups, int batches, int batch, int hotness, unsigned* sink) {
  hipEvent_t a, z;
  HIP_OK(hipEventCreate(&a));
  HIP_OK(hipEventCreate(&z));
  const int grid = (batch / 2 + 3) / 4;
  const int64_t per_batch = static_cast<int64_t>(batch) * hotness;
  for (int t = 0; t < 5; ++t)
    GatherSamplesKernel<K, kNt><<<grid, 256>>>(table, lookups + (t % batches) * per_batch, batch, hotness, sink);
  HIP_OK(hipEventRecord(a));
  const int iters = 40;
  for (int t = 0; t < iters; ++t)
    GatherSamplesKernel<K, kNt><<<grid, 256>>>(table, lookups + (t % batches) * per_batch, batch, hotness, sink);
  HIP_OK(hipEventRecord(z));
  HIP_OK(hipEventSynchronize(z));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, a, z));
  return ms / iters;
}

template <int K, bool kNt, bool kStage, bool kPool, bool kStore>
double RunParts(const char* table, const int* lookups, int batches, int batch, int hotness, unsigned* sink, _Float16* out) {
  hipEvent_t a, z;
  HIP_OK(hipEventCreate(&a));
  HIP_OK(hipEventCreate(&z));
  const int grid = (batch + 7) / 8;
  const int64_t per_batch = static_cast<int64_t>(batch) * hotness;
  for (int t = 0; t < 5; ++t)
    GatherSamplesPartsKernel<K, kNt, kStage, kPool, kStore><<<grid, 256>>>(table, lookups + (t % batches) * per_batch, batch,
                                                                        hotness, sink, out);
  HIP_OK(hipEventRecord(a));
  const int iters = 40;
  for (int t = 0; t < iters; ++t)
    GatherSamplesPartsKernel<K, kNt, kStage, kPool, kStore><<<grid, 256>>>(table, lookups + (t % batches) * per_batch, batch,
                                                                        hotness, sink, out);
  HIP_OK(hipEventRecord(z));
  HIP_OK(hipEventSynchronize(z));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, a, z));
  return ms / iters;
}

int HeadlineParts(const double alpha) {
  const int64_t rows = 10000000;
  const int batch = 65536, hotness = 64, batches = 4;
  char* table = nullptr;
  int* lookups = nullptr;
  unsigned* sink = nullptr;
  _Float16* out = nullptr;
  HIP_OK(hipMalloc(&table, rows * 512));
  HIP_OK(hipMemset(table, 1, rows * 512));
  std::vector<int> h(static_cast<size_t>(batches) * batch * hotness);
  cuembed_harness_generate_indices(rows, batches * batch, hotness, alpha, 1, 1, 0, nullptr, h.data());
  HIP_OK(hipMalloc(&lookups, h.size() * sizeof(int)));
  HIP_OK(hipMemcpy(lookups, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&sink, 64));
  HIP_OK(hipMalloc(&out, static_cast<size_t>(batch) * 512));
  std::printf("pattern,alpha,policy,indices_through_lds,fp32_pooling,row_stored,ms_per_batch\n");
#define PARTS(NT, ST, PO, SO)                                                                                   \
  std::printf("c2_parts,%.2f,%s,%d,%d,%d,%.4f\n", alpha, NT ? "nt" : "default", ST, PO, SO,                       \
              RunParts<8, NT, ST, PO, SO>(table, lookups, batches, batch, hotness, sink, out));
  PARTS(false, false, false, false) PARTS(false, true, false, false) PARTS(false, false, true, false)
  PARTS(false, false, false, true) PARTS(false, true, true, false) PARTS(false, true, true, true)
  PARTS(true, false, false, false) PARTS(true, true, false, false) PARTS(true, false, true, false)
  PARTS(true, false, false, true) PARTS(true, true, true, false) PARTS(true, true, true, true)
#undef PARTS
  // the product on the same index stream, same protocol: EmbeddingForward, and its kernel with other template knobs
  {
    const int64_t per_batch = static_cast<int64_t>(batch) * hotness;
    hipEvent_t a, z;
    HIP_OK(hipEventCreate(&a));
    HIP_OK(hipEventCreate(&z));
    auto time_it = [&](auto&& launch) {
      for (int t = 0; t < 5; ++t) launch(lookups + (t % batches) * per_batch);
      HIP_OK(hipEventRecord(a));
      const int iters = 40;
      for (int t = 0; t < iters; ++t) launch(lookups + (t % batches) * per_batch);
      HIP_OK(hipEventRecord(z));
      HIP_OK(hipEventSynchronize(z));
      float ms = 0;
      HIP_OK(hipEventElapsedTime(&ms, a, z));
      return ms / iters;
    };
    const __half* tab = reinterpret_cast<const __half*>(table);
    __half* o = reinterpret_cast<__half*>(out);
    for (int streaming = 0; streaming < 2; ++streaming) {
      cuembed::ForwardOptions opt = cuembed::DefaultForwardOptions();
      opt.row_loads = streaming ? cuembed::RowLoadPolicy::kStreaming : cuembed::RowLoadPolicy::kDefault;
      const double ms = time_it([&](const int* idx) {
        cuembed::EmbeddingForward<__half, __half, int, int>(tab, 256, idx, nullptr, nullptr, batch, hotness,
                                                            cuembed::CombineMode::kSum, o, 0, opt);
      });
      std::printf("EmbeddingForward,%.2f,%s,1,1,1,%.4f\n", alpha, streaming ? "nt" : "default", ms);
    }
#define LOOPED(G)                                                                                              \
  {                                                                                                            \
    const double ms = time_it([&](const int* idx) {                                                            \
      GatherSamplesLoopKernel<8, G><<<((batch + 7) / 8 + G - 1) / G, 256>>>(table, idx, batch, out);            \
    });                                                                                                        \
    std::printf("looped_%d_groups_per_workgroup,%.2f,default,1,1,1,%.4f\n", G, alpha, ms);                     \
  }
    LOOPED(1) LOOPED(2) LOOPED(4) LOOPED(8) LOOPED(16)
#undef LOOPED
    using cuembed::detail::GatherReduceKernel;
    using cuembed::detail::IndexSource;
    const _Float16* dtab = reinterpret_cast<const _Float16*>(table);
#define VARIANT(U, P, BT, NAME)                                                                                       \
  {                                                                                                                   \
    const double ms = time_it([&](const int* idx) {                                                                   \
      GatherReduceKernel<_Float16, float, int, int, 8, false, IndexSource::kLdsStaged, U, P, BT>                      \
          <<<dim3((batch + 7) / 8), dim3(32, 8), 8 * hotness * sizeof(int)>>>(dtab, 256, batch, idx, nullptr, hotness, \
                                                                             nullptr, false, out, 1, false, nullptr);  \
    });                                                                                                               \
    std::printf("%s,%.2f,default,1,1,1,%.4f\n", NAME, alpha, ms);                                                     \
  }
    VARIANT(8, false, 1024, "kernel_unroll8_bounds1024")
    VARIANT(8, false, 256, "kernel_unroll8_bounds256")
    VARIANT(8, true, 256, "kernel_unroll8_pipelined_bounds256")
    VARIANT(16, false, 256, "kernel_unroll16_bounds256")
    VARIANT(4, false, 256, "kernel_unroll4_bounds256")
#undef VARIANT
  }
  std::fflush(stdout);
  return 0;
}

int HeadlinePattern(const double alpha) {
  const int64_t rows = 10000000;
  const int batch = 65536, hotness = 64, batches = 4;
  char* table = nullptr;
  int* lookups = nullptr;
  unsigned* sink = nullptr;
  HIP_OK(hipMalloc(&table, rows * 512));
  HIP_OK(hipMemset(table, 1, rows * 512));
  std::vector<int> h(static_cast<size_t>(batches) * batch * hotness);
  cuembed_harness_generate_indices(rows, batches * batch, hotness, alpha, 1, 1, 0, nullptr, h.data());
  HIP_OK(hipMalloc(&lookups, h.size() * sizeof(int)));
  HIP_OK(hipMemcpy(lookups, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&sink, 64));
  const double row_bytes = static_cast<double>(batch) * hotness * 512;
  std::printf("pattern,alpha,loads_in_flight,policy,ms_per_batch,row_GBps\n");
#define ROWS(K, NT)                                                                                          \
  {                                                                                                          \
    const double ms = RunSamples<K, NT>(table, lookups, batches, batch, hotness, sink);                      \
    std::printf("c2_samples,%.2f,%d,%s,%.4f,%.0f\n", alpha, K, NT ? "nt" : "default", ms, row_bytes / ms / 1e6); \
  }
  ROWS(4, false) ROWS(8, false) ROWS(16, false) ROWS(32, false) ROWS(8, true) ROWS(16, true)
#undef ROWS
  std::fflush(stdout);
  return 0;
}

template <int K, bool kNt>
double Run(const char* table, int row_bytes, const int* lookups, int64_t n, unsigned* sink, int grid) {
  hipEvent_t a, z;
  HIP_OK(hipEventCreate(&a));
  HIP_OK(hipEventCreate(&z));
  for (int t = 0; t < 3; ++t) GatherRowsKernel<K, kNt><<<grid, 256>>>(table, row_bytes, lookups, n, sink);
  HIP_OK(hipEventRecord(a));
  const int iters = 10;
  for (int t = 0; t < iters; ++t) GatherRowsKernel<K, kNt><<<grid, 256>>>(table, row_bytes, lookups, n, sink);
  HIP_OK(hipEventRecord(z));
  HIP_OK(hipEventSynchronize(z));
  float ms = 0;
  HIP_OK(hipEventElapsedTime(&ms, a, z));
  return ms / iters;
}

// Address translation (`--span`): the headline's launch shape, loads only, K = 8, with the SAME index streams spread
// over more and more address space -- row id * stride, the table grown to match -- so that reuse, L2 / MALL hit rates and
// DRAM bytes stay what they are and only the number of distinct pages behind them grows (alpha 1.15: 570 k distinct
// rows per batch; alpha 0: 4.1 M).  If translation mattered for random 512-byte gathers, the time would grow with the
// span; TCP_UTCL1_TRANSLATION_MISS of the same runs is in profiles/r06_translation_counters.txt.
int TranslationSpan() {
  const int64_t base_rows = 10000000;
  const int batch = 65536, hotness = 64, batches = 4;
  const int strides[] = {1, 2, 4, 8, 16, 32};
  const int64_t max_rows = base_rows * 32;
  char* table = nullptr;
  int* lookups = nullptr;
  unsigned* sink = nullptr;
  HIP_OK(hipMalloc(&table, max_rows * 512));          // 164 GB of the 288
  HIP_OK(hipMemset(table, 1, max_rows * 512));
  HIP_OK(hipMalloc(&sink, 64));
  std::vector<int> h(static_cast<size_t>(batches) * batch * hotness), scaled(h.size());
  HIP_OK(hipMalloc(&lookups, h.size() * sizeof(int)));
  std::printf("pattern,alpha,row_id_stride,span_GB,policy,ms_per_batch,row_GBps\n");
  for (const double alpha : {1.15, 0.0}) {
    cuembed_harness_generate_indices(base_rows, batches * batch, hotness, alpha, 1, 1, 0, nullptr, h.data());
    for (const int stride : strides) {
      for (size_t i = 0; i < h.size(); ++i) scaled[i] = h[i] * stride;
      HIP_OK(hipMemcpy(lookups, scaled.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
      const double row_bytes = static_cast<double>(batch) * hotness * 512;
      const double ms = RunSamples<8, false>(table, lookups, batches, batch, hotness, sink);
      const double ms_nt = RunSamples<8, true>(table, lookups, batches, batch, hotness, sink);
      std::printf("c2_samples_span,%.2f,%d,%.1f,default,%.4f,%.0f\n", alpha, stride, base_rows * 512.0 * stride / 1e9, ms,
                  row_bytes / ms / 1e6);
      std::printf("c2_samples_span,%.2f,%d,%.1f,nt,%.4f,%.0f\n", alpha, stride, base_rows * 512.0 * stride / 1e9, ms_nt,
                  row_bytes / ms_nt / 1e6);
      std::fflush(stdout);
    }
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && std::string(argv[1]) == "--span") return TranslationSpan();
  if (argc > 1 && std::string(argv[1]) == "--c2-parts") return HeadlineParts(argc > 2 ? std::atof(argv[2]) : 0.0);
  if (argc > 1 && std::string(argv[1]) == "--c2") return HeadlinePattern(argc > 2 ? std::atof(argv[2]) : 1.15);
  const int64_t table_bytes = int64_t{5} << 30;       // 5 GiB: far beyond L2 (32 MiB) and the Infinity Cache (256 MiB)
  const int64_t num_lookups = int64_t{1} << 24;       // 16.8 M rows per launch
  char* table = nullptr;
  int* lookups = nullptr;
  unsigned* sink = nullptr;
  HIP_OK(hipMalloc(&table, table_bytes));
  HIP_OK(hipMemset(table, 1, table_bytes));
  HIP_OK(hipMalloc(&lookups, num_lookups * sizeof(int)));
  HIP_OK(hipMalloc(&sink, 64));
  std::vector<int> h(num_lookups);
  std::mt19937_64 rng(7);
  std::printf("row_bytes,loads_in_flight,policy,grid,ms,GBps,frac_of_8TBps\n");
  for (int row_bytes : {64, 128, 256, 512, 1024}) {
    const int64_t rows = table_bytes / row_bytes;
    for (auto& x : h) x = static_cast<int>(rng() % static_cast<uint64_t>(rows));
    HIP_OK(hipMemcpy(lookups, h.data(), num_lookups * sizeof(int), hipMemcpyHostToDevice));
    const double bytes = static_cast<double>(num_lookups) * row_bytes;
    for (int grid : {2048, 8192}) {
#define ROW(K, NT)                                                                                              \
  {                                                                                                             \
    const double ms = Run<K, NT>(table, row_bytes, lookups, num_lookups, sink, grid);                           \
    std::printf("%d,%d,%s,%d,%.4f,%.0f,%.3f\n", row_bytes, K, NT ? "nt" : "default", grid, ms, bytes / ms / 1e6, \
                bytes / ms / 1e6 / 8000.0);                                                                     \
  }
      ROW(4, false) ROW(8, false) ROW(16, false) ROW(8, true) ROW(16, true)
#undef ROW
    }
  }
  std::fflush(stdout);
  return 0;
}
