// manual_benchmark -- C++ benchmark harness on the header-only API.
//
// Counterpart of the reference's benchmarks/manual_benchmark.cu: same flag names and defaults
// (:44-82), same synthetic workload recipe (utils/src/embedding_allocation.cu:96-247 via
// cuembed_amd/csrc/utils), same protocol (one warm-up call; every iteration timed alone with
// events after a 1.02 GB cache-flushing reduction, or the whole loop once with
// --clear_caches=false; :199-248) and the same "Application BW" formulas (:250-261, :340-354,
// :444-471).  RunForward / RunTranspose / RunBackward below are the launch wrappers of
// utils/src/embedding_gpu_{forward,transpose,backward}.cu on raw device pointers.
//
// build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I cuembed_amd/csrc \
//               benchmarks/manual_benchmark.hip cuembed_amd/csrc/utils/synthetic_inputs.cpp \
//               -o benchmarks/manual_benchmark
// run:    benchmarks/manual_benchmark --num_categories 10000000 --embed_width 256 \
//               --batch_size 65536 --alpha 1.15 --hotness 64 --half_embedding_type=true --iterations 100
#include <dlfcn.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "cuembed/include/embedding_lookup.hpp"
#include "cuembed/include/index_transforms.hpp"

// synthetic workload recipe (cuembed_amd/csrc/utils/synthetic_inputs.cpp)
extern "C" int64_t cuembed_harness_allocate_forward(int64_t num_categories, int embed_width, int batch,
                                                    int hotness, double alpha, int is_csr, int shuffle,
                                                    int permute, int elem_is_half, int index_is_64,
                                                    void* table, int consume_table_draws, int32_t* offsets,
                                                    void* indices, void* weights);
extern "C" void cuembed_harness_allocate_grad_y(int64_t count, int elem_is_half, void* grad_y);

#define HIP_OK(x)                                                                  \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                \
    }                                                                              \
  } while (0)

namespace {


struct Flags {
  int num_categories = 1048576, embed_width = 128, batch_size = 1024, hotness = 1, iterations = 1;
  int sample_blocks = 1;   // extension: Transpose in this many blocks of samples (0 = RecommendedSampleBlocks; compressed gradient only)
  float alpha = 0.f;
  bool use_int64_indices = false, check_result = false, half_embedding_type = false, csr_input = false,
       weighted_sum = false, fp16_math = false, compressed_grad = true, skip_grad_init = true,
       forward_only = false, enable_csv = false, enable_stderr = true, clear_caches = true,
       bounded_sort = false,      // extension: Transpose sorts ceil(log2(num_categories)) key bits only
       fused_row_ids = false,     // extension: TransposeFixedHotness (no sample-id array; fixed hotness only)
       fused_remap = false,       // extension: Transpose(..., transpose_remapped_indices): sort + remap in one call
       coalesce_blocks = false,   // extension: with --sample_blocks, the REFERENCE's compressed gradient from the blocked
                                  // order (ComputeCompressedGradIndicesBlocked + EmbeddingBackward(sample_blocks))
       bag_order = false;         // extension (CSR): forward with ForwardOptions::sample_order = bags by descending
                                  // length (BagOrderByLength, computed once outside the timed loop)
};

bool ParseBool(const std::string& v) { return v.empty() || v == "1" || v == "true" || v == "True" || v == "yes"; }

Flags ParseFlags(int argc, char** argv) {
  std::map<std::string, std::string> kv;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.rfind("--", 0) != 0) {
      std::fprintf(stderr, "unexpected argument %s\n", a.c_str());
      std::exit(1);
    }
    a = a.substr(2);
    const size_t eq = a.find('=');
    if (eq != std::string::npos) kv[a.substr(0, eq)] = a.substr(eq + 1);
    else if (i + 1 < argc && std::strncmp(argv[i + 1], "--", 2) != 0) kv[a] = argv[++i];
    else kv[a] = "";
  }
  Flags f;
  auto geti = [&](const char* n, int* dst) { if (kv.count(n)) { *dst = std::atoi(kv[n].c_str()); kv.erase(n); } };
  auto getb = [&](const char* n, bool* dst) { if (kv.count(n)) { *dst = ParseBool(kv[n]); kv.erase(n); } };
  geti("num_categories", &f.num_categories); geti("embed_width", &f.embed_width);
  geti("batch_size", &f.batch_size); geti("hotness", &f.hotness); geti("iterations", &f.iterations);
  if (kv.count("alpha")) { f.alpha = static_cast<float>(std::atof(kv["alpha"].c_str())); kv.erase("alpha"); }
  getb("use_int64_indices", &f.use_int64_indices); getb("check_result", &f.check_result);
  getb("half_embedding_type", &f.half_embedding_type); getb("csr_input", &f.csr_input);
  getb("weighted_sum", &f.weighted_sum); getb("fp16_math", &f.fp16_math);
  getb("compressed_grad", &f.compressed_grad); getb("skip_grad_init", &f.skip_grad_init);
  getb("forward_only", &f.forward_only); getb("enable_csv", &f.enable_csv);
  getb("enable_stderr", &f.enable_stderr); getb("clear_caches", &f.clear_caches);
  getb("bounded_sort", &f.bounded_sort);
  getb("fused_row_ids", &f.fused_row_ids);
  getb("fused_remap", &f.fused_remap);
  getb("coalesce_blocks", &f.coalesce_blocks);
  getb("bag_order", &f.bag_order);
  geti("sample_blocks", &f.sample_blocks);
  for (auto& e : kv) {
    std::fprintf(stderr, "unknown flag --%s\n", e.first.c_str());
    std::exit(1);
  }
  return f;
}

// ---- --check_result: the CPU checker (reference: ValidateResult + the CPU runs, manual_benchmark.cu:85-90,
// :278-285, :373-386, :495-507).  The checker is this repository's oracle (oracle/libcuembed_oracle.so, a
// restatement of utils/include/embedding_lookup_cpu.hpp / index_transforms_cpu.hpp), loaded with dlopen ONLY when
// the flag is given and called only after the timed loops: nothing of it is linked into or timed by the benchmark.
struct Checker {
  void* lib = nullptr;
  int (*forward)(const void*, int, int, int, int, const void*, int, const void*, int, const void*, void*, int, int, int) = nullptr;
  int (*backward)(const void*, int, int, int64_t, int64_t, const void*, const void*, const void*, int, const void*, int,
                  void*, void*) = nullptr;
  int (*transpose)(const void*, const void*, const void*, int64_t, int, int, void*, void*, void*, int) = nullptr;
  int (*row_ids_fixed)(int, int, int, void*) = nullptr;
  int (*row_ids_csr)(const void*, int, int, int, void*) = nullptr;
  int (*remap)(const void*, int64_t, int, void*) = nullptr;
  int (*max_threads)() = nullptr;
  int failures = 0;

  void Load(const char* argv0) {
    std::string path;
    if (const char* env = std::getenv("CUEMBED_ORACLE_LIB")) path = env;
    if (path.empty()) {
      char exe[4096];
      const ssize_t n = readlink("/proc/self/exe", exe, sizeof exe - 1);
      std::string dir = n > 0 ? std::string(exe, static_cast<size_t>(n)) : std::string(argv0);
      dir = dir.substr(0, dir.find_last_of('/'));
      path = dir + "/../oracle/libcuembed_oracle.so";
    }
    lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (lib == nullptr) {
      std::fprintf(stderr, "--check_result: cannot load the CPU checker %s (%s); build it with `make -C oracle` or "
                           "point CUEMBED_ORACLE_LIB at it\n", path.c_str(), dlerror());
      std::exit(3);
    }
    auto sym = [&](const char* name) {
      void* p = dlsym(lib, name);
      if (p == nullptr) {
        std::fprintf(stderr, "--check_result: %s has no %s\n", path.c_str(), name);
        std::exit(3);
      }
      return p;
    };
    forward = reinterpret_cast<decltype(forward)>(sym("oracle_embedding_forward"));
    backward = reinterpret_cast<decltype(backward)>(sym("oracle_embedding_backward"));
    transpose = reinterpret_cast<decltype(transpose)>(sym("oracle_transpose"));
    row_ids_fixed = reinterpret_cast<decltype(row_ids_fixed)>(sym("oracle_extract_row_ids_from_fixed"));
    row_ids_csr = reinterpret_cast<decltype(row_ids_csr)>(sym("oracle_extract_row_ids_from_csr"));
    remap = reinterpret_cast<decltype(remap)>(sym("oracle_compute_compressed_grad_indices"));
    max_threads = reinterpret_cast<decltype(max_threads)>(sym("oracle_max_threads"));
  }

  //! exact equality, like the reference's ValidateResult (manual_benchmark.cu:85-90) -- for floats too
  template <typename T>
  void Expect(const char* what, const std::vector<T>& got, const std::vector<T>& want) {
    size_t bad = got.size() == want.size() ? 0 : 1;
    size_t first = 0;
    for (size_t i = 0; bad == 0 && i < got.size(); ++i)
      if (std::memcmp(&got[i], &want[i], sizeof(T)) != 0) {
        bad = 1;
        first = i;
      }
    if (bad == 0) {
      std::fprintf(stderr, "check_result: %s matches the CPU result (%zu values, exact)\n", what, got.size());
      return;
    }
    size_t count = 0;
    for (size_t i = 0; i < got.size() && i < want.size(); ++i) count += std::memcmp(&got[i], &want[i], sizeof(T)) != 0;
    std::fprintf(stderr, "check_result: %s MISMATCH: %zu of %zu values differ (first at %zu)\n", what, count,
                 got.size(), first);
    ++failures;
  }
};

template <typename T>
std::vector<T> Download(const T* dev, size_t n) {
  std::vector<T> h(n);
  if (n) HIP_OK(hipMemcpy(h.data(), dev, n * sizeof(T), hipMemcpyDeviceToHost));
  return h;
}

// ---- device helpers ----------------------------------------------------------------------
template <typename T>
struct DeviceBuffer {
  T* ptr = nullptr;
  size_t n = 0;
  void Resize(size_t count) {
    if (ptr) (void)hipFree(ptr);
    n = count;
    HIP_OK(hipMalloc(&ptr, (count ? count : 1) * sizeof(T)));
  }
  void Upload(const std::vector<T>& h) {
    Resize(h.size());
    if (n) HIP_OK(hipMemcpy(ptr, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
  }
  ~DeviceBuffer() { if (ptr) (void)hipFree(ptr); }
};

// uniform(-1, 1) table fill on the device (values do not influence timing; the reference's
// host RNG would need rows * width sequential draws)
template <typename ElemT>
__global__ void FillTableKernel(ElemT* table, int64_t count) {
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
    uint64_t z = static_cast<uint64_t>(i) + 0x9e3779b97f4a7c15ull;  // splitmix64
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    const float u = static_cast<float>(z >> 40) * (1.0f / 16777216.0f);
    table[i] = static_cast<ElemT>(2.0f * u - 1.0f);
  }
}

// cache flush: max-reduction over 256,000,000 ints (manual_benchmark.cu:136-144)
__global__ void FlushKernel(const int* buf, int64_t count, int* sink) {
  int m = 0;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < count;
       i += static_cast<int64_t>(gridDim.x) * blockDim.x)
    m = buf[i] > m ? buf[i] : m;
  if (m == 0x7fffffff) atomicMax(sink, m);
}

template <typename T> struct DevElem { using type = T; };
template <> struct DevElem<__half> { using type = _Float16; };

// ---- the reference's launch wrappers on raw pointers ---------------------------------------
template <typename ElemT, typename IndexT, typename OffsetT>
struct Workload {
  Flags f;
  int64_t nnz = 0;
  DeviceBuffer<ElemT> table, weights, result, grad_y, grad_embedding, transpose_weights;
  DeviceBuffer<IndexT> indices, sample_ids, transpose_indices, transpose_remapped_indices,
      transpose_sample_ids, inverse_mapping;
  DeviceBuffer<OffsetT> offsets;
  DeviceBuffer<char> workspace;
  DeviceBuffer<uint32_t> block_row_ids, num_unique_dev;   // --coalesce_blocks
  DeviceBuffer<int32_t> sample_order;                      // --bag_order (CSR only)
  int blocks = 1;                                          // sample blocks of the last transpose
  size_t lwork = 0;
};

template <typename ElemT, typename IndexT, typename OffsetT, bool fp16_math>
void RunForward(Workload<ElemT, IndexT, OffsetT>& w) {
  const OffsetT* offsets = w.f.csr_input ? w.offsets.ptr : nullptr;
  const int hotness = w.f.csr_input ? 0 : w.f.hotness;
  const ElemT* weights = w.f.weighted_sum ? w.weights.ptr : nullptr;
  if (w.f.bag_order && w.f.csr_input) {
    cuembed::ForwardOptions options = cuembed::DefaultForwardOptions();
    options.sample_order = w.sample_order.ptr;
    cuembed::EmbeddingForward<ElemT, ElemT, IndexT, OffsetT, fp16_math>(
        w.table.ptr, w.f.embed_width, w.indices.ptr, offsets, weights, w.f.batch_size, hotness,
        cuembed::CombineMode::kSum, w.result.ptr, 0, options);
    return;
  }
  cuembed::EmbeddingForward<ElemT, ElemT, IndexT, OffsetT, fp16_math>(
      w.table.ptr, w.f.embed_width, w.indices.ptr, offsets, weights, w.f.batch_size, hotness,
      cuembed::CombineMode::kSum, w.result.ptr);
}

template <typename ElemT, typename IndexT, typename OffsetT>
void RunTranspose(Workload<ElemT, IndexT, OffsetT>& w) {
  const int nnz = static_cast<int>(w.nnz);
  const bool fused = w.f.fused_row_ids && !w.f.csr_input;
  if (w.f.csr_input)
    cuembed::ExtractRowIdsFromCSR<IndexT, OffsetT>(w.offsets.ptr, w.f.batch_size, w.sample_ids.ptr);
  else if (!fused)
    cuembed::ExtractRowIdsFromFixed<IndexT>(w.f.batch_size, w.f.hotness, w.sample_ids.ptr);
  const ElemT* weights = w.f.weighted_sum ? w.weights.ptr : nullptr;
  ElemT* t_weights = w.f.weighted_sum ? w.transpose_weights.ptr : nullptr;
  size_t lwork = w.lwork;
  int bits = static_cast<int>(sizeof(IndexT) * 8);
  if (w.f.bounded_sort) {
    bits = 1;
    while ((int64_t{1} << bits) < w.f.num_categories) ++bits;
  }
  int blocks = 1;
  if (w.f.compressed_grad)   // never with a dense gradient: a table row would be produced once per block
    blocks = w.f.sample_blocks > 0 ? w.f.sample_blocks
                                   : cuembed::RecommendedSampleBlocks<ElemT>(w.f.embed_width, w.f.batch_size, w.nnz);
  // --fused_remap: the remapped ids come out of the transpose call itself (plain remap only: not the blocked one)
  const bool remap_in_sort = w.f.fused_remap && w.f.compressed_grad && !w.f.coalesce_blocks;
  IndexT* remap_out = remap_in_sort ? w.transpose_remapped_indices.ptr : nullptr;
  if (fused)
    cuembed::TransposeFixedHotness<IndexT, ElemT>(w.indices.ptr, weights, w.f.batch_size, w.f.hotness,
                                                  w.transpose_indices.ptr, w.transpose_sample_ids.ptr, t_weights,
                                                  w.workspace.ptr, &lwork, 0, bits, blocks, remap_out);
  else
    cuembed::Transpose<IndexT, ElemT>(w.sample_ids.ptr, w.indices.ptr, weights, nnz, w.transpose_indices.ptr,
                                      w.transpose_sample_ids.ptr, t_weights, w.workspace.ptr, &lwork, 0, bits, 0, blocks,
                                      remap_out);
  w.blocks = blocks;
  if (remap_in_sort) return;
  if (w.f.compressed_grad && w.f.coalesce_blocks)
    cuembed::ComputeCompressedGradIndicesBlocked<IndexT>(w.transpose_indices.ptr, nnz, blocks,
                                                         w.transpose_remapped_indices.ptr, w.block_row_ids.ptr,
                                                         w.num_unique_dev.ptr, w.workspace.ptr, &lwork);
  else if (w.f.compressed_grad)
    cuembed::ComputeCompressedGradIndices<IndexT>(w.transpose_indices.ptr, nnz,
                                                  w.transpose_remapped_indices.ptr, w.workspace.ptr, &lwork);
}

template <typename ElemT, typename IndexT, typename OffsetT>
void RunBackward(Workload<ElemT, IndexT, OffsetT>& w, int num_unique) {
  cuembed::EmbeddingBackward<ElemT, IndexT>(
      w.grad_y.ptr, w.f.embed_width, w.f.compressed_grad ? num_unique : w.f.num_categories,
      static_cast<int>(w.nnz), w.transpose_indices.ptr, w.transpose_sample_ids.ptr,
      w.f.compressed_grad ? w.transpose_remapped_indices.ptr : nullptr,
      w.f.weighted_sum ? w.transpose_weights.ptr : nullptr, w.f.skip_grad_init, w.grad_embedding.ptr,
      w.f.compressed_grad ? w.inverse_mapping.ptr : nullptr, 0,
      (w.f.compressed_grad && w.f.coalesce_blocks) ? w.blocks : 1, w.block_row_ids.ptr);
}

struct Timer {
  hipEvent_t start, stop;
  DeviceBuffer<int> flush, sink;
  bool clear;
  explicit Timer(bool clear_caches) : clear(clear_caches) {
    HIP_OK(hipEventCreate(&start));
    HIP_OK(hipEventCreate(&stop));
    sink.Resize(1);
    HIP_OK(hipMemset(sink.ptr, 0, sizeof(int)));
    if (clear) {
      flush.Resize(256000000);
      HIP_OK(hipMemset(flush.ptr, 1, flush.n * sizeof(int)));
    }
  }
  void ClearCache() {
    if (clear) FlushKernel<<<2048, 256>>>(flush.ptr, static_cast<int64_t>(flush.n), sink.ptr);
  }
  //! per-iteration times of the last Run in the flushed protocol (every iteration is timed alone there): the sweep
  //! reports their minimum and median next to the reference's mean, so that one slow iteration is seen for what it is
  std::vector<float> samples;
  float Min() const { return samples.empty() ? 0.f : *std::min_element(samples.begin(), samples.end()); }
  float Median() const {
    if (samples.empty()) return 0.f;
    std::vector<float> v(samples);
    std::nth_element(v.begin(), v.begin() + v.size() / 2, v.end());
    return v[v.size() / 2];
  }
  template <typename Fn>
  float Run(int iterations, Fn fn) {
    fn();  // warm-up
    ClearCache();
    float total = 0.f;
    samples.clear();
    for (int it = 0; it < iterations; ++it) {
      if (clear || it == 0) HIP_OK(hipEventRecord(start));
      fn();
      if (clear || it == iterations - 1) {
        HIP_OK(hipEventRecord(stop));
        HIP_OK(hipEventSynchronize(stop));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, start, stop));
        total += ms;
        if (clear) samples.push_back(ms);
      }
      ClearCache();
    }
    HIP_OK(hipDeviceSynchronize());
    return total;
  }
};

void CsvLine(const Flags& f, const char* name, double ms, double bw_l2, double bw_dram, const Timer& timer) {
  if (!f.enable_csv) return;
  const char* fname = "manual_benchmark_out.csv";
  bool existed = std::ifstream(fname).good();
  std::ofstream out(fname, std::ios::app);
  if (!existed)
    out << "num_categories,batch_size,hotness,alpha,embed_width,combine_mode,is_csr,is_weighted,"
           "compressed_grad,skip_grad_init,name,iterations,elapsed_time_ms,avg_time_ms,algo_bw_l2,algo_bw_dram,"
           "min_time_ms,median_time_ms\n";      // (the reference's columns, then this benchmark's two)
  char buf[512];
  std::snprintf(buf, sizeof buf, "%d,%d,%d,%g,%d,kSum,%d,%d,%d,%d,%s,%d ,%.2f ,%.4f ,%.2f,%.2f,%.5f,%.5f\n",
                f.num_categories, f.batch_size, f.hotness, f.alpha, f.embed_width, f.csr_input, f.weighted_sum,
                f.compressed_grad, f.skip_grad_init, name, f.iterations, ms, ms / f.iterations, bw_l2, bw_dram,
                timer.Min(), timer.Median());
  out << buf;
}

template <typename ElemT, typename IndexT, typename OffsetT, bool fp16_math>
int EmbeddingLookupBenchmark(const Flags& f, const char* argv0) {
  using DevT = typename DevElem<ElemT>::type;
  Checker check;
  if (f.check_result) check.Load(argv0);
  const int etype = sizeof(ElemT) == 2 ? 1 : 0, itype = sizeof(IndexT) == 8 ? 1 : 0, otype = sizeof(OffsetT) == 8 ? 1 : 0;
  constexpr bool kHalf = sizeof(ElemT) == 2;
  constexpr bool kIdx64 = sizeof(IndexT) == 8;
  Workload<ElemT, IndexT, OffsetT> w;
  w.f = f;
  const int64_t cells = static_cast<int64_t>(f.num_categories) * f.embed_width;
  const bool device_fill = cells > (int64_t{64} << 20);

  // ---- host-side synthetic inputs (reference recipe), table on the device when it is large ----
  const size_t cap = static_cast<size_t>(f.batch_size) * f.hotness;
  std::vector<ElemT> h_table(device_fill ? 0 : cells), h_weights(cap);
  std::vector<IndexT> h_indices(cap);
  std::vector<int32_t> h_offsets(f.batch_size + 1);
  w.nnz = cuembed_harness_allocate_forward(f.num_categories, f.embed_width, f.batch_size, f.hotness, f.alpha,
                                           f.csr_input, 1, 1, kHalf, kIdx64, device_fill ? nullptr : h_table.data(),
                                           device_fill ? 0 : 1, h_offsets.data(), h_indices.data(),
                                           h_weights.data());
  h_indices.resize(w.nnz);
  h_weights.resize(w.nnz);
  if (device_fill) {
    w.table.Resize(cells);
    FillTableKernel<DevT><<<4096, 256>>>(reinterpret_cast<DevT*>(w.table.ptr), cells);
  } else {
    w.table.Upload(h_table);
  }
  w.indices.Upload(h_indices);
  w.weights.Upload(h_weights);
  std::vector<OffsetT> h_off(h_offsets.begin(), h_offsets.end());
  w.offsets.Upload(h_off);
  w.result.Resize(static_cast<size_t>(f.batch_size) * f.embed_width);
  HIP_OK(hipDeviceSynchronize());

  Timer timer(f.clear_caches);
  const double es = sizeof(ElemT), W = f.embed_width, B = f.batch_size, H = f.hotness, nnz = w.nnz;
  const double it = f.iterations;

  // ---- forward ----
  if (f.bag_order && f.csr_input) {   // the scheduling hint is prepared once, with the batch
    w.sample_order.Resize(f.batch_size);
    size_t lw_o = 0;
    cuembed::BagOrderByLength<OffsetT>(w.offsets.ptr, f.batch_size, f.hotness, w.sample_order.ptr, nullptr, &lw_o);
    DeviceBuffer<char> order_work;
    order_work.Resize(lw_o);
    cuembed::BagOrderByLength<OffsetT>(w.offsets.ptr, f.batch_size, f.hotness, w.sample_order.ptr, order_work.ptr, &lw_o);
    HIP_OK(hipDeviceSynchronize());
    // ... and what it costs to prepare: the same protocol as the kernels below (one launch up to 131,072 bags of <= 255 lookups)
    const float oms = timer.Run(f.iterations, [&] {
      cuembed::BagOrderByLength<OffsetT>(w.offsets.ptr, f.batch_size, f.hotness, w.sample_order.ptr, order_work.ptr, &lw_o);
    });
    std::fprintf(stderr, "Bag order. Iterations: %d , Total time [ms]: %.3f , Avg [ms]: %.5f\n", f.iterations, oms,
                 oms / f.iterations);
    CsvLine(f, "bag_order", oms, 0.0, 0.0, timer);
  }
  float ms = timer.Run(f.iterations, [&] { RunForward<ElemT, IndexT, OffsetT, fp16_math>(w); });
  double bytes = f.csr_input ? es * (nnz - 1 + B) * W : es * B * (H + 1) * W;
  double bw = bytes * it / 1e6 / ms;
  std::fprintf(stderr, "Embedding forward. Iterations: %d , Total time [ms]: %.2f , Avg [ms]: %.4f , "
                       "Application BW [GB/s]: %.2f (algorithmic bytes / time: cache hits count, NOT an HBM "
                       "rate -- bench.py's roofline block has the measured fabric traffic)\n",
               f.iterations, ms, ms / it, bw);
  CsvLine(f, "forward", ms, bw, 0.0, timer);
  if (f.check_result) {   // (manual_benchmark.cu:278-285)
    const std::vector<ElemT> table = device_fill ? Download(w.table.ptr, static_cast<size_t>(cells)) : h_table;
    std::vector<ElemT> want(static_cast<size_t>(f.batch_size) * f.embed_width);
    const int rc = check.forward(table.data(), etype, f.embed_width, f.batch_size, f.csr_input ? 0 : f.hotness,
                                 h_indices.data(), itype, f.csr_input ? h_off.data() : nullptr, otype,
                                 f.weighted_sum ? h_weights.data() : nullptr, want.data(), /*sum*/ 0, fp16_math ? 1 : 0,
                                 check.max_threads());
    if (rc != 0) {
      std::fprintf(stderr, "check_result: the CPU forward rejected the arguments\n");
      ++check.failures;
    }
    check.Expect("forward", Download(w.result.ptr, want.size()), want);
  }
  if (f.forward_only) return check.failures;

  // ---- transpose (+ compressed remap) ----
  w.sample_ids.Resize(w.nnz);
  w.transpose_indices.Resize(w.nnz);
  w.transpose_remapped_indices.Resize(w.nnz);
  w.transpose_sample_ids.Resize(w.nnz);
  w.transpose_weights.Resize(w.nnz);
  size_t lw_t = 0, lw_c = 0;
  cuembed::Transpose<IndexT, ElemT>(w.sample_ids.ptr, w.indices.ptr, f.weighted_sum ? w.weights.ptr : nullptr,
                                    static_cast<int>(w.nnz), w.transpose_indices.ptr, w.transpose_sample_ids.ptr,
                                    w.transpose_weights.ptr, nullptr, &lw_t);
  cuembed::ComputeCompressedGradIndices<IndexT>(w.transpose_indices.ptr, static_cast<int>(w.nnz),
                                                w.transpose_remapped_indices.ptr, nullptr, &lw_c);
  w.lwork = lw_t > lw_c ? lw_t : lw_c;
  if (f.coalesce_blocks && f.compressed_grad) {
    int blocks = f.sample_blocks > 0 ? f.sample_blocks
                                     : cuembed::RecommendedSampleBlocks<ElemT>(f.embed_width, f.batch_size, w.nnz);
    size_t lw_b = 0;
    cuembed::ComputeCompressedGradIndicesBlocked<IndexT>(w.transpose_indices.ptr, static_cast<int>(w.nnz), blocks,
                                                         w.transpose_remapped_indices.ptr, nullptr, nullptr, nullptr, &lw_b);
    w.lwork = w.lwork > lw_b ? w.lwork : lw_b;
    w.block_row_ids.Resize(w.nnz);
    w.num_unique_dev.Resize(1);
  }
  w.workspace.Resize(w.lwork);
  ms = timer.Run(f.iterations, [&] { RunTranspose<ElemT, IndexT, OffsetT>(w); });
  double tb = nnz * sizeof(IndexT) + (f.csr_input ? nnz * sizeof(OffsetT) : 0) + (f.weighted_sum ? nnz * es : 0) +
              (f.compressed_grad ? 3 : 2) * nnz * sizeof(IndexT) + (f.weighted_sum ? nnz * es : 0);
  bw = tb * it / 1e6 / ms;
  std::fprintf(stderr, "Transpose. Iterations: %d , Total time [ms]: %.2f , Avg [ms]: %.4f , "
                       "Application BW [GB/s]: %.2f\n", f.iterations, ms, ms / it, bw);
  CsvLine(f, "transpose", ms, 0.0, bw, timer);
  std::vector<IndexT> c_idx, c_sid, c_remap;
  std::vector<ElemT> c_w;
  if (f.check_result) {   // (manual_benchmark.cu:373-386; the device contract: a STABLE sort by index)
    int blocks = 1;
    if (f.compressed_grad)
      blocks = f.sample_blocks > 0 ? f.sample_blocks
                                   : cuembed::RecommendedSampleBlocks<ElemT>(f.embed_width, f.batch_size, w.nnz);
    if (cuembed::TransposeSampleBlockLength(w.nnz, blocks) < w.nnz) {
      std::fprintf(stderr, "check_result: --sample_blocks > 1 changes the order on purpose; transpose and backward are "
                           "not checked (tests/test_gpu_sample_blocks.py checks them block by block)\n");
      return check.failures;
    }
    std::vector<IndexT> sid(w.nnz);
    if (f.csr_input) check.row_ids_csr(h_off.data(), otype, f.batch_size, itype, sid.data());
    else check.row_ids_fixed(f.batch_size, f.hotness, itype, sid.data());
    c_idx.resize(w.nnz);
    c_sid.resize(w.nnz);
    c_w.resize(f.weighted_sum ? w.nnz : 0);
    check.transpose(sid.data(), h_indices.data(), f.weighted_sum ? h_weights.data() : nullptr, w.nnz, itype, etype,
                    c_idx.data(), c_sid.data(), f.weighted_sum ? c_w.data() : nullptr, /*stable=*/1);
    check.Expect("transpose indices", Download(w.transpose_indices.ptr, w.nnz), c_idx);
    check.Expect("transpose sample ids", Download(w.transpose_sample_ids.ptr, w.nnz), c_sid);
    if (f.weighted_sum) check.Expect("transpose weights", Download(w.transpose_weights.ptr, w.nnz), c_w);
    if (f.compressed_grad) {
      c_remap.resize(w.nnz);
      check.remap(c_idx.data(), w.nnz, itype, c_remap.data());
      check.Expect("remapped indices", Download(w.transpose_remapped_indices.ptr, w.nnz), c_remap);
    }
  }

  // ---- backward ----
  int num_unique = 0;
  if (f.compressed_grad && f.coalesce_blocks) {   // (a blocked order's last id is not the largest)
    uint32_t nu = 0;
    HIP_OK(hipMemcpy(&nu, w.num_unique_dev.ptr, sizeof nu, hipMemcpyDeviceToHost));
    num_unique = static_cast<int>(nu);
  } else if (f.compressed_grad) {
    IndexT last = 0;
    HIP_OK(hipMemcpy(&last, w.transpose_remapped_indices.ptr + (w.nnz - 1), sizeof(IndexT), hipMemcpyDeviceToHost));
    num_unique = static_cast<int>(last) + 1;
  }
  const int64_t grad_rows = f.compressed_grad ? num_unique : f.num_categories;
  std::vector<ElemT> h_gy(static_cast<size_t>(f.batch_size) * f.embed_width);
  cuembed_harness_allocate_grad_y(static_cast<int64_t>(h_gy.size()), kHalf, h_gy.data());
  w.grad_y.Upload(h_gy);
  w.grad_embedding.Resize(static_cast<size_t>(grad_rows) * f.embed_width);
  HIP_OK(hipMemset(w.grad_embedding.ptr, 0, w.grad_embedding.n * sizeof(ElemT)));
  w.inverse_mapping.Resize(f.compressed_grad ? num_unique : 0);
  ms = timer.Run(f.iterations, [&] { RunBackward<ElemT, IndexT, OffsetT>(w, num_unique); });
  // unique rows actually touched (the reference counts them with thrust::unique_count)
  std::vector<IndexT> h_t(w.nnz);
  HIP_OK(hipMemcpy(h_t.data(), w.transpose_indices.ptr, w.nnz * sizeof(IndexT), hipMemcpyDeviceToHost));
  int64_t uniq = w.nnz > 0 ? 1 : 0;
  for (int64_t i = 1; i < w.nnz; ++i) uniq += h_t[i] != h_t[i - 1];
  if (f.compressed_grad && f.coalesce_blocks) uniq = num_unique;   // (one gradient row per table row, not per run)
  double dram = es * W * uniq + 2.0 * sizeof(IndexT) * nnz + (f.weighted_sum ? es * nnz : 0) + es * W * B;
  double l2 = dram + es * W * nnz;
  std::fprintf(stderr, "Backward. Iterations: %d , Total time [ms]: %.2f , Avg [ms]: %.4f , "
                       "Application DRAM BW [GB/s]: %.2f , Application L2 BW [GB/s]: %.2f\n",
               f.iterations, ms, ms / it, dram * it / 1e6 / ms, l2 * it / 1e6 / ms);
  CsvLine(f, "backward", ms, l2 * it / 1e6 / ms, dram * it / 1e6 / ms, timer);
  if (f.check_result) {   // (manual_benchmark.cu:495-507)
    // the timed iterations may have run with skip_grad_init onto a buffer that earlier iterations had written
    // (only their time means anything): one more call into a zeroed buffer is what is compared
    HIP_OK(hipMemset(w.grad_embedding.ptr, 0, w.grad_embedding.n * sizeof(ElemT)));
    RunBackward<ElemT, IndexT, OffsetT>(w, num_unique);
    HIP_OK(hipDeviceSynchronize());
    std::vector<ElemT> want(w.grad_embedding.n);
    std::memset(want.data(), 0, want.size() * sizeof(ElemT));
    std::vector<IndexT> want_inv(f.compressed_grad ? num_unique : 0);
    check.backward(h_gy.data(), etype, f.embed_width, grad_rows, w.nnz, c_idx.data(), c_sid.data(),
                   f.compressed_grad ? c_remap.data() : nullptr, itype, f.weighted_sum ? c_w.data() : nullptr,
                   /*skip_grad_init=*/1, want.data(), f.compressed_grad ? want_inv.data() : nullptr);
    check.Expect(kHalf ? "backward (fp16: exact while every partial sum is exactly representable, the reference's "
                         "own test data; see include/cuembed_amd.h)" : "backward",
                 Download(w.grad_embedding.ptr, want.size()), want);
    if (f.compressed_grad) check.Expect("inverse mapping", Download(w.inverse_mapping.ptr, want_inv.size()), want_inv);
  }
  return check.failures;
}

}  // namespace

int main(int argc, char** argv) {
  const Flags f = ParseFlags(argc, argv);
  // type dispatch as in manual_benchmark.cu:563-659
  int failures = 0;
#define DISPATCH(ELEM, MATH)                                                                             \
  do {                                                                                                   \
    if (f.use_int64_indices) failures = EmbeddingLookupBenchmark<ELEM, int64_t, int, MATH>(f, argv[0]);  \
    else failures = EmbeddingLookupBenchmark<ELEM, int32_t, int, MATH>(f, argv[0]);                      \
  } while (0)
  if (f.half_embedding_type) {
    if (f.fp16_math) DISPATCH(__half, true);
    else DISPATCH(__half, false);
  } else {
    DISPATCH(float, false);
  }
  return failures == 0 ? 0 : 4;   // --check_result: a mismatch is an error
}
