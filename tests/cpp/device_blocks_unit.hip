// Unit tests of the gather / scatter BUILDING BLOCKS on the device, against host recomputation -- the counterpart of the
// reference's tests/test_embedding_ops.cu (Addresser :153-237, Combiner sum / mean vs host recomputation with random
// weights :239-315, IndexLoader :121-152, GradAddresser :317-346) for the pieces THIS design is made of:
//   * WidenIndex / RowElems / RowPtr : row addressing in bytes, 32-bit ids zero-extended, offsets past 2^31 elements;
//   * ColumnSlice::Of                : workgroup -> (column slice, sample group) is a bijection for every slice count;
//   * Pack + Arith + RowPool::Add    : one unfused IEEE operation per add / multiply, fp32 and fp16 accumulation;
//   * RowPool::Gather                : every lookup pooled exactly once and IN ORDER for every count around the unroll
//                                      and pipelining boundaries (0 .. 40 lookups), plain and weighted, both load kinds;
//   * FinishPooledRow                : mean = multiply by the reciprocal of the weight sum, zeros for an empty bag;
//   * AccumulateRow                  : the backward's fp32 partial sums of fp16 rows, weighted and not;
//   * PackRowsByOwner / FinishOwnerPiece (exchange_transforms.hpp, the header API itself): range starts, slots, padding
//                                      ids, untouched slack, the flag word; the piece's count, zeroed row, tail.
// Small kernels instantiate the blocks directly (one wavefront each); the host recomputes with the same
// single-rounding operations (Arith is __host__ __device__) and compares BITS.  Built by cuembed_amd.build, run by
// tests/test_gpu_device_blocks.py (-m gpu).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "cuembed/include/embedding_lookup.hpp"
#include "cuembed/include/exchange_transforms.hpp"

using namespace cuembed::detail;

#define HIP_OK(x)                                                                             \
  do {                                                                                        \
    hipError_t e_ = (x);                                                                      \
    if (e_ != hipSuccess) {                                                                   \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                           \
    }                                                                                         \
  } while (0)

static int g_fail = 0;
#define CHECK(cond, ...)                                        \
  do {                                                          \
    if (!(cond)) {                                              \
      if (g_fail < 20) {                                        \
        std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
        std::fprintf(stderr, __VA_ARGS__);                      \
        std::fprintf(stderr, "\n");                             \
      }                                                         \
      ++g_fail;                                                 \
    }                                                           \
  } while (0)

template <typename T>
struct DeviceArray {
  T* ptr = nullptr;
  size_t n = 0;
  explicit DeviceArray(size_t count) : n(count) { HIP_OK(hipMalloc(&ptr, (count ? count : 1) * sizeof(T))); }
  explicit DeviceArray(const std::vector<T>& h) : DeviceArray(h.size()) {
    if (n) HIP_OK(hipMemcpy(ptr, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
  }
  ~DeviceArray() { (void)hipFree(ptr); }
  std::vector<T> Download() const {
    std::vector<T> h(n);
    if (n) HIP_OK(hipMemcpy(h.data(), ptr, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }
};

// ---- addressing ------------------------------------------------------------------------------------------------------
template <typename ElemT, typename IndexT>
__global__ void AddressKernel(const IndexT* ids, const int* widths, const int n, int64_t* elems, int64_t* bytes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t r = WidenIndex(ids[i]);
  elems[i] = RowElems(r, widths[i]);
  const ElemT* base = reinterpret_cast<const ElemT*>(uintptr_t{0x100000});
  bytes[i] = reinterpret_cast<const char*>(RowPtr(base, r, widths[i])) - reinterpret_cast<const char*>(base);
}

template <typename ElemT, typename IndexT>
static void Addressing(const char* name) {
  std::mt19937_64 rng(11);
  std::vector<IndexT> ids = {0, 1, 2, 1000, static_cast<IndexT>(0x7fffffff)};
  std::vector<int> widths = {1, 32, 256, 514, 1 << 20};
  if (sizeof(IndexT) == 8) ids.push_back(static_cast<IndexT>(int64_t{1} << 33));     // beyond 32 bits
  if (sizeof(IndexT) == 4) ids.push_back(static_cast<IndexT>(0x80000001u));          // 32-bit ids are unsigned row numbers
  std::vector<IndexT> h_ids;
  std::vector<int> h_w;
  for (const IndexT r : ids)
    for (const int w : widths) {
      h_ids.push_back(r);
      h_w.push_back(w);
    }
  for (int t = 0; t < 500; ++t) {
    h_ids.push_back(static_cast<IndexT>(rng() % (sizeof(IndexT) == 8 ? (uint64_t{1} << 40) : (uint64_t{1} << 31))));
    h_w.push_back(1 + static_cast<int>(rng() % 4096));
  }
  const int n = static_cast<int>(h_ids.size());
  DeviceArray<IndexT> d_ids(h_ids);
  DeviceArray<int> d_w(h_w);
  DeviceArray<int64_t> d_e(n), d_b(n);
  AddressKernel<ElemT, IndexT><<<(n + 63) / 64, 64>>>(d_ids.ptr, d_w.ptr, n, d_e.ptr, d_b.ptr);
  HIP_OK(hipDeviceSynchronize());
  const auto e = d_e.Download(), b = d_b.Download();
  for (int i = 0; i < n; ++i) {
    const int64_t r = sizeof(IndexT) == 4 ? static_cast<int64_t>(static_cast<uint32_t>(h_ids[i])) : static_cast<int64_t>(h_ids[i]);
    CHECK(e[i] == r * h_w[i], "%s: RowElems(%lld, %d) = %lld", name, (long long)r, h_w[i], (long long)e[i]);
    CHECK(b[i] == r * h_w[i] * static_cast<int64_t>(sizeof(ElemT)), "%s: RowPtr(%lld, %d) is %lld bytes in", name,
          (long long)r, h_w[i], (long long)b[i]);
  }
}

// ---- column slices ---------------------------------------------------------------------------------------------------
__global__ void SliceKernel(const int blocks, const int slices, const int xcds, int* slice_of, int64_t* group_of) {
  const unsigned b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= static_cast<unsigned>(blocks)) return;
  const ColumnSlice c = ColumnSlice::Of(b, slices, xcds);
  slice_of[b] = c.slice;
  group_of[b] = c.block;
}

static void ColumnSlices() {
  for (const int xcds : {1, 2, 4, 8})
    for (int slices = 1; slices <= xcds; slices *= 2)
      for (const int groups : {1, 5, 64, 1000}) {
        const int blocks = groups * slices;
        if (slices > 1 && blocks % xcds != 0) continue;          // (the launcher rounds the grid to whole XCD rounds)
        DeviceArray<int> d_s(blocks);
        DeviceArray<int64_t> d_g(blocks);
        SliceKernel<<<(blocks + 63) / 64, 64>>>(blocks, slices, xcds, d_s.ptr, d_g.ptr);
        HIP_OK(hipDeviceSynchronize());
        const auto s = d_s.Download();
        const auto g = d_g.Download();
        std::vector<int> seen(static_cast<size_t>(blocks), 0);
        for (int b = 0; b < blocks; ++b) {
          CHECK(s[b] >= 0 && s[b] < slices && g[b] >= 0 && g[b] < groups, "ColumnSlice(%d; %d slices, %d xcds) = (%d, %lld)",
                b, slices, xcds, s[b], (long long)g[b]);
          if (s[b] >= 0 && s[b] < slices && g[b] >= 0 && g[b] < groups) ++seen[static_cast<size_t>(g[b]) * slices + s[b]];
          if (slices > 1) CHECK(s[b] == (b % xcds) % slices, "slice follows the XCD of the workgroup");
        }
        for (int k = 0; k < blocks; ++k)
          CHECK(seen[k] == 1, "%d slices on %d xcds, %d groups: (group %d, slice %d) taken %d times", slices, xcds, groups,
                k / slices, k % slices, seen[k]);
      }
}

// ---- pooling ---------------------------------------------------------------------------------------------------------
// One lane (N elements of the row) pools `count` lookups of a small table through RowPool::Gather and finishes the row.
template <typename ElemT, typename AccT, int N, bool kWeighted, int kUnroll, bool kPipelined, bool kStream>
__global__ void PoolKernel(const ElemT* table, const int width, const int* lookups, const ElemT* weights, const int count,
                           const bool is_mean, ElemT* out) {
  const int lane = threadIdx.x;                       // lane l owns elements [l * N, l * N + N)
  if (lane * N >= width) return;
  RowPool<ElemT, AccT, N, kWeighted> pool;
  const auto idx_at = [&](int j) { return WidenIndex(lookups[j]); };
  const auto w_at = [&](int j) { return weights[j]; };
  pool.template Gather<kUnroll, kPipelined, kStream>(table + lane * N, width, count, idx_at, w_at);
  FinishPooledRow<ElemT, AccT, N, kWeighted>(pool, count, is_mean, out + lane * N);
}

template <typename T>
static bool SameBits(const T a, const T b) { return std::memcmp(&a, &b, sizeof(T)) == 0; }

template <typename ElemT, typename AccT, int N, bool kWeighted, int kUnroll, bool kPipelined, bool kStream>
static void Pooling(const char* name) {
  using A = Arith<AccT>;
  const int rows = 97, width = 4 * N;                 // four lanes
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> val(-2.f, 2.f), wgt(0.f, 1.f);
  std::vector<ElemT> table(static_cast<size_t>(rows) * width);
  for (auto& x : table) x = static_cast<ElemT>(val(rng));
  DeviceArray<ElemT> d_table(table);
  DeviceArray<ElemT> d_out(width);
  for (int count = 0; count <= 40; ++count) {
    std::vector<int> lookups(count);
    std::vector<ElemT> weights(count);
    for (int j = 0; j < count; ++j) {
      lookups[j] = static_cast<int>(rng() % rows);
      weights[j] = static_cast<ElemT>(wgt(rng));
    }
    if (count > 3) weights[2] = static_cast<ElemT>(0);
    DeviceArray<int> d_l(lookups);
    DeviceArray<ElemT> d_w(weights);
    for (const bool is_mean : {false, true}) {
      PoolKernel<ElemT, AccT, N, kWeighted, kUnroll, kPipelined, kStream><<<1, 64>>>(d_table.ptr, width, d_l.ptr, d_w.ptr,
                                                                                    count, is_mean, d_out.ptr);
      HIP_OK(hipDeviceSynchronize());
      const auto got = d_out.Download();
      for (int c = 0; c < width; ++c) {       // the host loop of the reference combiner, one rounding per operation
        AccT acc = static_cast<AccT>(0);
        float weight_sum = 0.f;
        for (int j = 0; j < count; ++j) {
          const ElemT x = table[static_cast<size_t>(lookups[j]) * width + c];
          if (kWeighted) {
            weight_sum += static_cast<float>(weights[j]);
            acc = A::add(acc, A::mul(A::widen(x), A::widen(weights[j])));
          } else {
            acc = A::add(acc, A::widen(x));
          }
        }
        if (is_mean) {
          if (!kWeighted) weight_sum = static_cast<float>(count);
          const float inv = weight_sum == 0.f ? 0.f : 1.0f / weight_sum;
          acc = A::mul(acc, static_cast<AccT>(inv));
        }
        const ElemT want = static_cast<ElemT>(acc);
        CHECK(SameBits(got[c], want), "%s: count %d mean %d column %d: %g, host %g", name, count, int(is_mean), c,
              static_cast<double>(got[c]), static_cast<double>(want));
      }
    }
  }
}

// ---- the backward's partial sums -------------------------------------------------------------------------------------
template <typename GradT, int N, bool kWeighted>
__global__ void AccumulateKernel(const GradT* rows, const float* weights, const int count, float* out) {
  float acc[N];
#pragma unroll
  for (int e = 0; e < N; ++e) acc[e] = 0.f;
  for (int j = 0; j < count; ++j) {
    Pack<GradT, N> row;
#pragma unroll
    for (int e = 0; e < N; ++e) row.v[e] = rows[j * N + e];
    AccumulateRow<GradT, N, kWeighted>(acc, row, weights[j]);
  }
#pragma unroll
  for (int e = 0; e < N; ++e) out[e] = acc[e];
}

template <typename GradT, int N, bool kWeighted>
static void BackwardPartialSums(const char* name) {
  std::mt19937 rng(9);
  std::uniform_real_distribution<float> val(-3.f, 3.f);
  const int count = 300;
  std::vector<GradT> rows(static_cast<size_t>(count) * N);
  std::vector<float> weights(count);
  for (auto& x : rows) x = static_cast<GradT>(val(rng));
  for (auto& w : weights) w = static_cast<float>(static_cast<GradT>(val(rng)));
  DeviceArray<GradT> d_rows(rows);
  DeviceArray<float> d_w(weights);
  DeviceArray<float> d_out(N);
  AccumulateKernel<GradT, N, kWeighted><<<1, 1>>>(d_rows.ptr, d_w.ptr, count, d_out.ptr);
  HIP_OK(hipDeviceSynchronize());
  const auto got = d_out.Download();
  for (int e = 0; e < N; ++e) {
    float acc = 0.f;
    for (int j = 0; j < count; ++j) {
      const float x = static_cast<float>(rows[static_cast<size_t>(j) * N + e]);
      acc = Arith<float>::add(acc, kWeighted ? Arith<float>::mul(x, weights[j]) : x);
    }
    CHECK(SameBits(got[e], acc), "%s: element %d: %g, host %g", name, e, got[e], acc);
  }
}

// ---- the exchange's device-side halves ---------------------------------------------------------------------------------
template <typename IndexT>
static void ExchangePack(const char* name, const int world, const int64_t n, const int64_t slot, const int64_t given,
                         const int width) {
  std::mt19937_64 rng(91 + world + n);
  const int64_t num_categories = 40000;
  std::vector<IndexT> ids(n);
  {   // ascending, distinct
    std::vector<int64_t> pool(num_categories);
    for (int64_t i = 0; i < num_categories; ++i) pool[i] = i;
    for (int64_t i = 0; i < n; ++i) std::swap(pool[i], pool[i + rng() % (num_categories - i)]);
    std::sort(pool.begin(), pool.begin() + n);
    for (int64_t i = 0; i < n; ++i) ids[i] = static_cast<IndexT>(pool[i]);
  }
  std::vector<float> rows(n * width);
  for (auto& x : rows) x = static_cast<float>(static_cast<int>(rng() % 17) - 8);
  std::vector<int64_t> cuts(world + 1);
  for (int r = 0; r <= world; ++r) cuts[r] = num_categories * r / world;
  const int64_t valid = given < 0 ? n : std::min(given, n);
  const DeviceArray<IndexT> d_ids(ids);
  const DeviceArray<float> d_rows(rows);
  const DeviceArray<int64_t> d_cuts(cuts);
  const std::vector<IndexT> count_word(1, static_cast<IndexT>(given));
  const DeviceArray<IndexT> d_count(count_word);
  DeviceArray<int64_t> d_send_ids(std::vector<int64_t>(world * slot, -7));
  DeviceArray<float> d_send_rows(std::vector<float>(world * slot * width, 3.0f));
  DeviceArray<int64_t> d_starts(std::vector<int64_t>(world + 1, -1));
  DeviceArray<int64_t> d_flag(std::vector<int64_t>(1, 0));
  cuembed::PackRowsByOwner<IndexT, float>(d_ids.ptr, d_rows.ptr, n, width, given < 0 ? nullptr : d_count.ptr, d_cuts.ptr,
                                          world, slot, 0, num_categories, d_send_ids.ptr, d_send_rows.ptr, d_starts.ptr,
                                          d_flag.ptr, nullptr);
  HIP_OK(hipDeviceSynchronize());
  const auto send_ids = d_send_ids.Download();
  const auto send_rows = d_send_rows.Download();
  const auto starts = d_starts.Download();
  bool too_many = false;
  for (int r = 0; r < world; ++r) {
    const int64_t lo = std::lower_bound(ids.begin(), ids.begin() + valid, static_cast<IndexT>(cuts[r])) - ids.begin();
    const int64_t hi = r + 1 == world ? valid
                                      : std::lower_bound(ids.begin(), ids.begin() + valid, static_cast<IndexT>(cuts[r + 1])) - ids.begin();
    CHECK(starts[r] == lo, "%s: range %d starts at %lld, host %lld", name, r, (long long)starts[r], (long long)lo);
    too_many = too_many || hi - lo > slot;
    for (int64_t j = 0; j < slot; ++j) {
      const bool real = j < hi - lo;
      const int64_t want_id = real ? static_cast<int64_t>(ids[lo + j]) : num_categories;
      CHECK(send_ids[r * slot + j] == want_id, "%s: slot %d entry %lld holds id %lld, host %lld", name, r, (long long)j,
            (long long)send_ids[r * slot + j], (long long)want_id);
      for (int c = 0; c < width; ++c) {
        const float got = send_rows[(r * slot + j) * width + c];
        const float want = real ? rows[(lo + j) * width + c] : 3.0f;      // slack: not written
        CHECK(got == want, "%s: slot %d entry %lld column %d: %g, host %g", name, r, (long long)j, c, got, want);
      }
    }
  }
  CHECK(starts[world] == valid, "%s: the last start is the count", name);
  CHECK(d_flag.Download()[0] == (too_many ? 1 : 0), "%s: flag word", name);
}

static void ExchangeFinish(const char* name, const int64_t distinct, const int64_t padding, const int64_t capacity) {
  // what the owner's merge leaves: sorted ids (padding id last), their compressed ids, ids / rows from EmbeddingBackward
  const int64_t num_categories = 1000, pad_lo = 250, pad_len = 7;
  const int width = 24;
  std::vector<int64_t> sorted, remap;
  for (int64_t u = 0; u < distinct; ++u)
    for (int rep = 0; rep <= u % 3; ++rep) { sorted.push_back(3 * u + 1); remap.push_back(u); }
  for (int64_t k = 0; k < padding; ++k) { sorted.push_back(num_categories); remap.push_back(distinct); }
  const int64_t nnz = static_cast<int64_t>(sorted.size());
  std::vector<int64_t> ids(capacity + 1, 555);
  std::vector<float> rows((capacity + 1) * width, 2.0f);
  const bool fits = distinct + (padding > 0 ? 1 : 0) <= capacity + 1;
  if (fits) for (int64_t u = 0; u < distinct; ++u) ids[u] = 3 * u + 1;     // (the merge wrote its rows)
  DeviceArray<int64_t> d_sorted(sorted), d_remap(remap), d_ids(ids), d_tail(std::vector<int64_t>(capacity + 2, -1));
  DeviceArray<float> d_rows(rows);
  DeviceArray<int64_t> d_flag(std::vector<int64_t>(1, 0)), d_count(std::vector<int64_t>(1, -1));
  cuembed::FinishOwnerPiece<float>(d_sorted.ptr, d_remap.ptr, nnz, capacity, num_categories, pad_lo, pad_len, d_ids.ptr,
                                   d_rows.ptr, width, d_tail.ptr, d_flag.ptr, d_count.ptr, nullptr);
  HIP_OK(hipDeviceSynchronize());
  const auto got_ids = d_ids.Download();
  const auto got_rows = d_rows.Download();
  const auto tail = d_tail.Download();
  const bool overflow = distinct > capacity;
  CHECK(d_count.Download()[0] == distinct, "%s: count %lld, host %lld", name, (long long)d_count.Download()[0], (long long)distinct);
  CHECK(d_flag.Download()[0] == (overflow ? 1 : 0), "%s: flag", name);
  const int64_t zeroed = std::min(padding > 0 ? distinct : capacity, capacity);
  for (int64_t i = 0; i <= capacity; ++i) {
    const int64_t want = i < distinct ? ids[i] : pad_lo + i % pad_len;
    CHECK(got_ids[i] == want, "%s: id %lld is %lld, host %lld", name, (long long)i, (long long)got_ids[i], (long long)want);
    if (i < capacity) CHECK(tail[i] == want, "%s: tail id %lld", name, (long long)i);
    for (int c = 0; c < width; ++c)
      CHECK(got_rows[i * width + c] == (i == zeroed ? 0.0f : 2.0f), "%s: row %lld column %d is %g", name, (long long)i, c,
            got_rows[i * width + c]);
  }
  CHECK(tail[capacity] == std::min(distinct, capacity) && tail[capacity + 1] == (overflow ? 1 : 0), "%s: tail words", name);
}

int main() {
  Addressing<float, int32_t>("f32 rows, i32 ids");
  Addressing<float, int64_t>("f32 rows, i64 ids");
  Addressing<_Float16, int32_t>("f16 rows, i32 ids");
  Addressing<_Float16, int64_t>("f16 rows, i64 ids");
  ColumnSlices();
  Pooling<float, float, 4, false, 8, false, false>("f32 sum");
  Pooling<float, float, 4, true, 8, false, false>("f32 weighted");
  Pooling<float, float, 4, true, 8, true, false>("f32 weighted, pipelined loads");
  Pooling<float, float, 1, false, 4, false, true>("f32 sum, 4-byte lanes, unroll 4, non-temporal loads");
  Pooling<_Float16, float, 8, false, 8, false, false>("f16 rows, fp32 accumulate");
  Pooling<_Float16, float, 8, true, 8, true, true>("f16 rows weighted, fp32 accumulate, pipelined non-temporal loads");
  Pooling<_Float16, _Float16, 8, false, 8, false, false>("f16 rows, fp16 accumulate (fp16_math)");
  Pooling<_Float16, _Float16, 8, true, 16, false, false>("f16 rows weighted, fp16 accumulate, unroll 16");
  Pooling<_Float16, float, 2, true, 8, false, false>("f16 rows weighted, 4-byte lanes");
  BackwardPartialSums<_Float16, 8, false>("fp16 grad_y rows, fp32 partial sums");
  BackwardPartialSums<_Float16, 8, true>("fp16 grad_y rows x weight, fp32 partial sums");
  BackwardPartialSums<float, 4, true>("fp32 grad_y rows x weight");
  ExchangePack<int32_t>("pack: 8 owners, i32 ids", 8, 5000, 700, -1, 16);
  ExchangePack<int64_t>("pack: 3 owners, i64 ids, a device-side count", 3, 4000, 1500, 3100, 5);
  ExchangePack<int64_t>("pack: a slot that overflows", 4, 3000, 600, -1, 8);
  ExchangePack<int32_t>("pack: nothing valid", 5, 64, 16, 0, 4);
  ExchangePack<int32_t>("pack: one owner, 1 KiB rows", 1, 300, 512, 257, 256);
  ExchangeFinish("finish: padding run present", 40, 13, 64);
  ExchangeFinish("finish: no padding, exactly full", 64, 0, 64);
  ExchangeFinish("finish: overflow", 70, 5, 64);
  ExchangeFinish("finish: one row", 1, 900, 4);
  if (g_fail != 0) {
    std::fprintf(stderr, "%d device building-block checks FAILED\n", g_fail);
    return 1;
  }
  std::printf("all device building-block checks passed\n");
  return 0;
}
