// Header-only API check (C++ side of the drop-in boundary): instantiates every public template
// with every type combination the reference instantiates (utils/src/embedding_gpu_*.cu), through
// the reference-named include paths, and runs the reference's known-answer vectors
// (tests/test_embedding_forward.cu:120-160, test_embedding_transpose.cu:112-122,
// test_embedding_backward.cu:162-202) on the GPU.  Built by __graft_entry__.build() (compile
// check, no GPU needed) and executed by tests/test_gpu_cpp_header_api.py.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cuembed/include/embedding_lookup.cuh"   // forwarding names of the reference
#include "cuembed/include/index_transforms.cuh"

#define HIP_OK(x)                                                              \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));      \
      std::exit(2);                                                            \
    }                                                                          \
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
  std::vector<T> host() const {
    std::vector<T> h(n);
    if (n) HIP_OK(hipMemcpy(h.data(), ptr, n * sizeof(T), hipMemcpyDeviceToHost));
    return h;
  }
};

template <typename T> T From(float v);
template <> float From<float>(float v) { return v; }
template <> __half From<__half>(float v) { return __float2half(v); }
inline float ToF(float v) { return v; }
inline float ToF(__half v) { return __half2float(v); }

template <typename T>
std::vector<T> Vec(std::initializer_list<float> v) {
  std::vector<T> out;
  for (float x : v) out.push_back(From<T>(x));
  return out;
}

static int g_failures = 0;
template <typename T>
void Expect(const char* what, const std::vector<T>& got, std::initializer_list<float> want) {
  bool ok = got.size() == want.size();
  size_t i = 0;
  for (float w : want) {
    if (ok && ToF(got[i]) != w) ok = false;
    ++i;
  }
  if (!ok) {
    ++g_failures;
    std::fprintf(stderr, "MISMATCH %s\n", what);
  }
}
template <typename T, typename U>
void ExpectAll(const char* what, const std::vector<T>& got, const std::vector<U>& want) {
  bool ok = got.size() == want.size();
  for (size_t i = 0; ok && i < got.size(); ++i) ok = static_cast<double>(got[i]) == static_cast<double>(want[i]);
  if (!ok) {
    ++g_failures;
    std::fprintf(stderr, "MISMATCH %s\n", what);
  }
}
template <typename T>
void ExpectInt(const char* what, const std::vector<T>& got, std::initializer_list<long long> want) {
  bool ok = got.size() == want.size();
  size_t i = 0;
  for (long long w : want) {
    if (ok && static_cast<long long>(got[i]) != w) ok = false;
    ++i;
  }
  if (!ok) {
    ++g_failures;
    std::fprintf(stderr, "MISMATCH %s\n", what);
  }
}

template <typename ElemT, typename IndexT, typename OffsetT, bool kFp16Math>
void ForwardKat(hipStream_t stream) {
  using cuembed::CombineMode;
  DeviceArray<ElemT> table(Vec<ElemT>({1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20}));
  DeviceArray<IndexT> idx(std::vector<IndexT>{1, 3, 0, 4});
  DeviceArray<OffsetT> off(std::vector<OffsetT>{0, 2, 4});
  DeviceArray<ElemT> w(Vec<ElemT>({1.f, .5f, 1.f, .5f}));
  DeviceArray<ElemT> out(16);
  const OffsetT* no_off = nullptr;
  const ElemT* no_w = nullptr;
  auto run = [&](const OffsetT* o, const ElemT* ww, int hots, CombineMode m) {
    cuembed::EmbeddingForward<ElemT, ElemT, IndexT, OffsetT, kFp16Math>(table.ptr, 4, idx.ptr, o, ww, 2, hots, m,
                                                                        out.ptr, stream);
    HIP_OK(hipStreamSynchronize(stream));
    return out.host();
  };
  auto first8 = [](std::vector<ElemT> v) { v.resize(8); return v; };
  Expect("fwd fixed sum", first8(run(no_off, no_w, 2, CombineMode::kSum)), {18, 20, 22, 24, 18, 20, 22, 24});
  Expect("fwd fixed weighted", first8(run(no_off, w.ptr, 2, CombineMode::kSum)), {11.5, 13, 14.5, 16, 9.5, 11, 12.5, 14});
  Expect("fwd fixed mean", first8(run(no_off, no_w, 2, CombineMode::kMean)), {9, 10, 11, 12, 9, 10, 11, 12});
  Expect("fwd concat", run(no_off, no_w, 2, CombineMode::kConcat), {5, 6, 7, 8, 13, 14, 15, 16, 1, 2, 3, 4, 17, 18, 19, 20});
  Expect("fwd csr sum", first8(run(off.ptr, no_w, 0, CombineMode::kSum)), {18, 20, 22, 24, 18, 20, 22, 24});
  Expect("fwd csr weighted", first8(run(off.ptr, w.ptr, 0, CombineMode::kSum)), {11.5, 13, 14.5, 16, 9.5, 11, 12.5, 14});
  Expect("fwd csr mean", first8(run(off.ptr, no_w, 0, CombineMode::kMean)), {9, 10, 11, 12, 9, 10, 11, 12});
}

template <typename IndexT, typename WeightT>
void TransposeKat(hipStream_t stream) {
  DeviceArray<IndexT> sid(std::vector<IndexT>{0, 0, 1, 1});
  DeviceArray<IndexT> idx(std::vector<IndexT>{1, 3, 0, 4});
  DeviceArray<WeightT> w(Vec<WeightT>({1.f, .5f, 1.f, .5f}));
  DeviceArray<IndexT> t_idx(4), t_sid(4), remap(4);
  DeviceArray<WeightT> t_w(4);
  for (int weighted = 0; weighted < 2; ++weighted) {
    const WeightT* wp = weighted ? w.ptr : nullptr;
    size_t lwork = 0;
    cuembed::Transpose<IndexT, WeightT>(sid.ptr, idx.ptr, wp, 4, t_idx.ptr, t_sid.ptr, t_w.ptr, nullptr, &lwork, stream);
    size_t lwork2 = 0;
    cuembed::ComputeCompressedGradIndices<IndexT>(t_idx.ptr, 4, remap.ptr, nullptr, &lwork2, stream);
    if (lwork2 > lwork) lwork = lwork2;
    DeviceArray<char> work(lwork);
    cuembed::Transpose<IndexT, WeightT>(sid.ptr, idx.ptr, wp, 4, t_idx.ptr, t_sid.ptr, t_w.ptr, work.ptr, &lwork, stream);
    cuembed::ComputeCompressedGradIndices<IndexT>(t_idx.ptr, 4, remap.ptr, work.ptr, &lwork, stream);
    HIP_OK(hipStreamSynchronize(stream));
    ExpectInt("transpose indices", t_idx.host(), {0, 1, 3, 4});
    ExpectInt("transpose sample ids", t_sid.host(), {1, 0, 0, 1});
    ExpectInt("remap", remap.host(), {0, 1, 2, 3});
    if (weighted) Expect("transpose weights", t_w.host(), {1, 1, .5, .5});
  }
  DeviceArray<IndexT> rid(9);
  cuembed::ExtractRowIdsFromFixed<IndexT>(3, 3, rid.ptr, stream);
  HIP_OK(hipStreamSynchronize(stream));
  ExpectInt("row ids fixed", rid.host(), {0, 0, 0, 1, 1, 1, 2, 2, 2});
  DeviceArray<IndexT> rid2(4);
  cuembed::ExtractRowIdsForConcat<IndexT>(4, rid2.ptr, stream);
  HIP_OK(hipStreamSynchronize(stream));
  ExpectInt("row ids concat", rid2.host(), {0, 1, 2, 3});
}

template <typename IndexT, typename OffsetT>
void CsrRowIdsKat(hipStream_t stream) {
  DeviceArray<OffsetT> off(std::vector<OffsetT>{0, 2, 3, 5});
  DeviceArray<IndexT> rid(5);
  cuembed::ExtractRowIdsFromCSR<IndexT, OffsetT>(off.ptr, 3, rid.ptr, stream);
  HIP_OK(hipStreamSynchronize(stream));
  ExpectInt("row ids csr", rid.host(), {0, 0, 1, 2, 2});
}

template <typename GradT, typename IndexT>
void BackwardKat(hipStream_t stream) {
  DeviceArray<IndexT> t_idx(std::vector<IndexT>{0, 1, 3, 3});
  DeviceArray<IndexT> remap(std::vector<IndexT>{0, 1, 2, 2});
  DeviceArray<IndexT> t_sid(std::vector<IndexT>{1, 0, 0, 1});
  DeviceArray<GradT> t_w(Vec<GradT>({3.f, 1.f, .5f, 3.f}));
  DeviceArray<GradT> gy(Vec<GradT>({1, 2, 3, 4, 5, 6, 7, 8}));
  DeviceArray<GradT> grad(20);
  DeviceArray<IndexT> inv(3);
  const IndexT* no_remap = nullptr;
  const GradT* no_w = nullptr;
  cuembed::EmbeddingBackward<GradT, IndexT>(gy.ptr, 4, 5, 4, t_idx.ptr, t_sid.ptr, no_remap, no_w, false, grad.ptr,
                                            nullptr, stream);
  HIP_OK(hipStreamSynchronize(stream));
  Expect("bwd full", grad.host(), {5, 6, 7, 8, 1, 2, 3, 4, 0, 0, 0, 0, 6, 8, 10, 12, 0, 0, 0, 0});
  cuembed::EmbeddingBackward<GradT, IndexT>(gy.ptr, 4, 5, 4, t_idx.ptr, t_sid.ptr, no_remap, t_w.ptr, false,
                                            grad.ptr, nullptr, stream);
  HIP_OK(hipStreamSynchronize(stream));
  Expect("bwd full weighted", grad.host(), {15, 18, 21, 24, 1, 2, 3, 4, 0, 0, 0, 0, 15.5, 19, 22.5, 26, 0, 0, 0, 0});
  DeviceArray<GradT> cgrad(12);
  cuembed::EmbeddingBackward<GradT, IndexT>(gy.ptr, 4, 3, 4, t_idx.ptr, t_sid.ptr, remap.ptr, no_w, false, cgrad.ptr,
                                            inv.ptr, stream);
  HIP_OK(hipStreamSynchronize(stream));
  Expect("bwd compressed", cgrad.host(), {5, 6, 7, 8, 1, 2, 3, 4, 6, 8, 10, 12});
  ExpectInt("bwd inverse mapping", inv.host(), {0, 1, 3});
  // extension: num_unique left on the device (num_grad_embedding_rows < 0): worst-case buffers (nnz rows),
  // the rows past the last id keep what they held
  DeviceArray<GradT> wgrad(Vec<GradT>({9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9, 9}));
  DeviceArray<IndexT> winv(std::vector<IndexT>{7, 7, 7, 7});
  cuembed::EmbeddingBackward<GradT, IndexT>(gy.ptr, 4, -1, 4, t_idx.ptr, t_sid.ptr, remap.ptr, no_w, false, wgrad.ptr,
                                            winv.ptr, stream);
  HIP_OK(hipStreamSynchronize(stream));
  Expect("bwd compressed, num_unique on the device", wgrad.host(), {5, 6, 7, 8, 1, 2, 3, 4, 6, 8, 10, 12, 9, 9, 9, 9});
  ExpectInt("bwd inverse mapping, num_unique on the device", winv.host(), {0, 1, 3, 7});
}

// ---- this library's extensions of the header-only API (same known answers through other doors) ----
template <typename IndexT, typename WeightT>
void ExtensionTransposeKat(hipStream_t stream) {
  // TransposeFixedHotness == ExtractRowIdsFromFixed + Transpose: batch 2, hotness 2 (the transpose KAT)
  DeviceArray<IndexT> idx(std::vector<IndexT>{1, 3, 0, 4});
  DeviceArray<WeightT> w(Vec<WeightT>({1.f, .5f, 1.f, .5f}));
  DeviceArray<IndexT> t_idx(4), t_sid(4);
  DeviceArray<WeightT> t_w(4);
  size_t lwork = 0;
  cuembed::TransposeFixedHotness<IndexT, WeightT>(idx.ptr, w.ptr, 2, 2, t_idx.ptr, t_sid.ptr, t_w.ptr, nullptr, &lwork,
                                                  stream, /*index_bits=*/3);
  DeviceArray<char> work(lwork);
  cuembed::TransposeFixedHotness<IndexT, WeightT>(idx.ptr, w.ptr, 2, 2, t_idx.ptr, t_sid.ptr, t_w.ptr, work.ptr, &lwork,
                                                  stream, /*index_bits=*/3);
  HIP_OK(hipStreamSynchronize(stream));
  ExpectInt("fixed-hotness transpose indices", t_idx.host(), {0, 1, 3, 4});
  ExpectInt("fixed-hotness transpose sample ids", t_sid.host(), {1, 0, 0, 1});
  Expect("fixed-hotness transpose weights", t_w.host(), {1, 1, .5, .5});
  // Transpose is a generic COO transpose: signed keys (negative first), arbitrary payloads
  DeviceArray<IndexT> cols(std::vector<IndexT>{3, -2, 0, -2, 7});
  DeviceArray<IndexT> rows(std::vector<IndexT>{-9, 5, static_cast<IndexT>(sizeof(IndexT) == 8 ? (1ll << 40) : 70000), 6, 0});
  DeviceArray<IndexT> t_rows(5), t_cols(5);
  const WeightT* no_w = nullptr;
  lwork = 0;
  cuembed::Transpose<IndexT, WeightT>(rows.ptr, cols.ptr, no_w, 5, t_rows.ptr, t_cols.ptr, nullptr, nullptr, &lwork, stream);
  DeviceArray<char> work2(lwork);
  cuembed::Transpose<IndexT, WeightT>(rows.ptr, cols.ptr, no_w, 5, t_rows.ptr, t_cols.ptr, nullptr, work2.ptr, &lwork, stream);
  HIP_OK(hipStreamSynchronize(stream));
  ExpectInt("signed keys", t_rows.host(), {-2, -2, 0, 3, 7});
  ExpectInt("wide payloads", t_cols.host(), {5, 6, sizeof(IndexT) == 8 ? (1ll << 40) : 70000, -9, 0});
  // Transpose in sample blocks: 200,000 lookups in 2 blocks -- each block sorted and stable on its own
  {
    const int n = 200000, blocks = 2;
    std::vector<IndexT> h_cols(n), h_rows(n);
    for (int i = 0; i < n; ++i) {
      h_cols[i] = static_cast<IndexT>((static_cast<int64_t>(i) * 7919) % 1000);
      h_rows[i] = static_cast<IndexT>(i / 4);
    }
    DeviceArray<IndexT> b_cols(h_cols), b_rows(h_rows), o_keys(n), o_rows(n);
    size_t lw = 0;
    cuembed::Transpose<IndexT, WeightT>(b_rows.ptr, b_cols.ptr, no_w, n, o_keys.ptr, o_rows.ptr, nullptr, nullptr, &lw, stream,
                                        10, 31, blocks);
    DeviceArray<char> work3(lw);
    cuembed::Transpose<IndexT, WeightT>(b_rows.ptr, b_cols.ptr, no_w, n, o_keys.ptr, o_rows.ptr, nullptr, work3.ptr, &lw, stream,
                                        10, 31, blocks);
    HIP_OK(hipStreamSynchronize(stream));
    const int64_t L = cuembed::TransposeSampleBlockLength(n, blocks);
    const std::vector<IndexT> k = o_keys.host(), r = o_rows.host();
    bool ok = L % 4096 == 0 && L < n && 2 * L >= n;
    long long sum_in = 0, sum_out = 0;
    for (int i = 0; i < n; ++i) {
      sum_in += static_cast<long long>(h_cols[i]) * 31 + h_rows[i];
      sum_out += static_cast<long long>(k[i]) * 31 + r[i];
      if (i % L != 0) ok = ok && (k[i - 1] < k[i] || (k[i - 1] == k[i] && r[i - 1] <= r[i]));   // sorted, stable inside a block
      ok = ok && (r[i] * 4 / L == i / L || (r[i] * 4 + 3) / L == i / L);                          // the pair stayed in its block
    }
    ok = ok && sum_in == sum_out && cuembed::RecommendedSampleBlocks<float>(128, 65536, 65536 * 64) == 2;
    if (!ok) {
      std::fprintf(stderr, "FAIL: transpose in sample blocks\n");
      ++g_failures;
    }
    // ... and the REFERENCE's compressed gradient from that blocked order: every key 0..999 occurs 200 times (100 per
    // block), grad_y is all ones -> 1000 ascending rows of 200s, inverse_mapping = 0..999
    const int width = 4, samples = n / 4;
    DeviceArray<float> gy(std::vector<float>(static_cast<size_t>(samples) * width, 1.0f));
    DeviceArray<IndexT> pairs(n), inv(1000);
    DeviceArray<uint32_t> pair_rows(n), num_unique(1);
    DeviceArray<float> grad(1000 * width);
    size_t lb = 0;
    cuembed::ComputeCompressedGradIndicesBlocked<IndexT>(o_keys.ptr, n, blocks, pairs.ptr, pair_rows.ptr, num_unique.ptr,
                                                         nullptr, &lb, stream);
    DeviceArray<char> work4(lb);
    cuembed::ComputeCompressedGradIndicesBlocked<IndexT>(o_keys.ptr, n, blocks, pairs.ptr, pair_rows.ptr, num_unique.ptr,
                                                         work4.ptr, &lb, stream);
    const float* no_weights = nullptr;
    cuembed::EmbeddingBackward<float, IndexT>(gy.ptr, width, 1000, n, o_keys.ptr, o_rows.ptr, pairs.ptr, no_weights,
                                              /*skip_grad_init=*/false, grad.ptr, inv.ptr, stream, blocks, pair_rows.ptr);
    HIP_OK(hipStreamSynchronize(stream));
    ExpectInt("blocked remap: num_unique", num_unique.host(), {1000});
    std::vector<long long> ids(1000);
    for (int i = 0; i < 1000; ++i) ids[i] = i;
    ExpectAll("blocked backward: inverse mapping", inv.host(), ids);
    ExpectAll("blocked backward: gradient rows", grad.host(), std::vector<float>(1000 * width, 200.0f));
    // per-call forward options (no process-wide state): streaming row loads give the same bits
    DeviceArray<float> table(std::vector<float>(1000 * width, 0.5f)), pooled(static_cast<size_t>(samples) * width);
    cuembed::ForwardOptions options;
    options.row_loads = cuembed::RowLoadPolicy::kStreaming;
    const int* no_offsets = nullptr;
    cuembed::EmbeddingForward<float, float, IndexT, int>(table.ptr, width, b_cols.ptr, no_offsets, no_weights, samples, 4,
                                                          cuembed::CombineMode::kSum, pooled.ptr, stream, options);
    HIP_OK(hipStreamSynchronize(stream));
    ExpectAll("forward with ForwardOptions{kStreaming}", pooled.host(), std::vector<float>(static_cast<size_t>(samples) * width, 2.0f));
    // ... and a sample order (CSR: bags of 0, 1, 2, 3, 0, 1, ... lookups, handed out back to front) moves no result
    std::vector<int> offsets(samples + 1, 0);
    std::vector<int32_t> back_to_front(samples);
    std::vector<float> want(static_cast<size_t>(samples) * width);
    for (int s = 0; s < samples; ++s) {
      offsets[s + 1] = offsets[s] + s % 4;
      back_to_front[s] = samples - 1 - s;
      for (int c = 0; c < width; ++c) want[static_cast<size_t>(s) * width + c] = 0.5f * (s % 4);
    }
    DeviceArray<int> d_offsets(offsets);
    DeviceArray<int32_t> d_order(back_to_front);
    cuembed::ForwardOptions ordered;
    ordered.sample_order = d_order.ptr;
    cuembed::EmbeddingForward<float, float, IndexT, int>(table.ptr, width, b_cols.ptr, d_offsets.ptr, no_weights, samples, 0,
                                                          cuembed::CombineMode::kSum, pooled.ptr, stream, ordered);
    HIP_OK(hipStreamSynchronize(stream));
    ExpectAll("forward with ForwardOptions{sample_order}", pooled.host(), want);
  }
  // row-cache index translation: rows 3 and 0 are cached in slots 0 and 1, cache 1000 rows above the table
  DeviceArray<int32_t> slot_of_row(std::vector<int32_t>{1, -1, -1, 0, -1});
  DeviceArray<int64_t> translated(4);
  cuembed::TranslateIndicesForRowCache<IndexT>(idx.ptr, 4, slot_of_row.ptr, 5, 1000, translated.ptr, stream);
  HIP_OK(hipStreamSynchronize(stream));
  ExpectInt("row cache translation", translated.host(), {1, 1000, 1001, 4});
}

// Round-5 extensions of the header API: the remapped ids from the transpose call itself (one launch at this size), the
// capacity check and the padded gradient of a device-side row count, the reference's GradT arithmetic.
template <typename IndexT>
void OneCallAndBoundedKat(hipStream_t stream) {
  // README example through ONE call: indices [[4, 8], [18, 4], [8, 7], [8, 0]] (4 samples x 2) -> sorted + ids
  DeviceArray<IndexT> idx(std::vector<IndexT>{4, 8, 18, 4, 8, 7, 8, 0});
  DeviceArray<IndexT> t_idx(8), t_sid(8), remap(8);
  const float* no_w = nullptr;
  float* no_tw = nullptr;
  size_t lwork = 0;
  cuembed::TransposeFixedHotness<IndexT, float>(idx.ptr, no_w, 4, 2, t_idx.ptr, t_sid.ptr, no_tw, nullptr, &lwork, stream,
                                                static_cast<int>(sizeof(IndexT) * 8), 1, remap.ptr);
  DeviceArray<char> work(lwork);
  cuembed::TransposeFixedHotness<IndexT, float>(idx.ptr, no_w, 4, 2, t_idx.ptr, t_sid.ptr, no_tw, work.ptr, &lwork, stream,
                                                static_cast<int>(sizeof(IndexT) * 8), 1, remap.ptr);
  HIP_OK(hipStreamSynchronize(stream));
  ExpectInt("one call: sorted indices", t_idx.host(), {0, 4, 4, 7, 8, 8, 8, 18});
  ExpectInt("one call: sample ids", t_sid.host(), {3, 0, 1, 2, 0, 2, 3, 1});
  ExpectInt("one call: remapped ids", remap.host(), {0, 1, 1, 2, 3, 3, 3, 4});
  // compressed backward with the row count left on the device: 5 rows; buffers of 4 rows raise the flag and stay untouched
  const int width = 4;
  DeviceArray<float> gy(std::vector<float>(4 * width, 1.0f));
  DeviceArray<uint32_t> overflow(std::vector<uint32_t>{0u});
  {
    DeviceArray<float> grad(std::vector<float>(5 * width, 7.0f));
    DeviceArray<IndexT> inv(std::vector<IndexT>(5, static_cast<IndexT>(-3)));
    cuembed::EmbeddingBackward<float, IndexT>(gy.ptr, width, -1, 8, t_idx.ptr, t_sid.ptr, remap.ptr, no_w, false, grad.ptr,
                                              inv.ptr, stream, 1, nullptr, /*capacity_rows=*/4, overflow.ptr);
    HIP_OK(hipStreamSynchronize(stream));
    ExpectInt("capacity 4 < 5 rows: flag raised", overflow.host(), {1});
    ExpectAll("capacity 4 < 5 rows: nothing written", grad.host(), std::vector<float>(5 * width, 7.0f));
  }
  {
    // capacity 7 with padding: rows 5 and 6 are zero and name rows of the batch in turn (the first two: 0 and 4)
    DeviceArray<float> grad(std::vector<float>(7 * width, 7.0f));
    DeviceArray<IndexT> inv(std::vector<IndexT>(7, static_cast<IndexT>(-3)));
    cuembed::EmbeddingBackward<float, IndexT>(gy.ptr, width, -1, 8, t_idx.ptr, t_sid.ptr, remap.ptr, no_w, false, grad.ptr,
                                              inv.ptr, stream, 1, nullptr, /*capacity_rows=*/7, overflow.ptr,
                                              /*pad_to_capacity=*/true);
    HIP_OK(hipStreamSynchronize(stream));
    ExpectInt("padded: inverse mapping", inv.host(), {0, 4, 7, 8, 18, 0, 4});
    std::vector<float> want;
    for (float v : {1.f, 2.f, 1.f, 3.f, 1.f, 0.f, 0.f})
      for (int c = 0; c < width; ++c) want.push_back(v);
    ExpectAll("padded: gradient rows", grad.host(), want);
  }
  {
    // the reference's GradT arithmetic (fp16: 2049 does not exist, 2048 + 1 stays 2048 -- as in the CPU reference)
    std::vector<__half> h_gy(3 * width, __float2half(1.0f));
    for (int c = 0; c < width; ++c) h_gy[c] = __float2half(2048.0f);
    DeviceArray<__half> gy16(h_gy), grad16(std::vector<__half>(1 * width, __float2half(9.0f)));
    DeviceArray<IndexT> one_row(std::vector<IndexT>{5, 5, 5}), samples(std::vector<IndexT>{0, 1, 2}), ids(std::vector<IndexT>{0, 0, 0});
    DeviceArray<IndexT> inv(1);
    const __half* no_w16 = nullptr;
    cuembed::EmbeddingBackwardReferenceSums<__half, IndexT>(gy16.ptr, width, 1, 3, one_row.ptr, samples.ptr, ids.ptr, no_w16,
                                                            false, grad16.ptr, inv.ptr, stream);
    HIP_OK(hipStreamSynchronize(stream));
    Expect("reference sums: 2048 + 1 + 1 in fp16", grad16.host(), {2048, 2048, 2048, 2048});
    ExpectInt("reference sums: inverse mapping", inv.host(), {5});
    cuembed::EmbeddingBackward<__half, IndexT>(gy16.ptr, width, 1, 3, one_row.ptr, samples.ptr, ids.ptr, no_w16, false,
                                               grad16.ptr, inv.ptr, stream);
    HIP_OK(hipStreamSynchronize(stream));
    Expect("default: fp32 partial sums, one rounding", grad16.host(), {2050, 2050, 2050, 2050});
  }
}

int main() {
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  // forward: utils/src/embedding_gpu_forward.cu:69-76 plus the int64 offsets of the torch binding
  ForwardKat<float, int32_t, int, false>(stream);
  ForwardKat<float, int64_t, int, false>(stream);
  ForwardKat<__half, int32_t, int, false>(stream);
  ForwardKat<__half, int64_t, int, false>(stream);
  ForwardKat<float, int32_t, int, true>(stream);
  ForwardKat<float, int64_t, int, true>(stream);
  ForwardKat<__half, int32_t, int, true>(stream);
  ForwardKat<__half, int64_t, int, true>(stream);
  ForwardKat<float, int64_t, int64_t, false>(stream);
  // transpose: utils/src/embedding_gpu_transpose.cu:95-98
  TransposeKat<int32_t, float>(stream);
  TransposeKat<int64_t, float>(stream);
  TransposeKat<int32_t, __half>(stream);
  TransposeKat<int64_t, __half>(stream);
  CsrRowIdsKat<int32_t, int>(stream);
  CsrRowIdsKat<int64_t, int>(stream);
  CsrRowIdsKat<int64_t, int64_t>(stream);
  // backward: utils/src/embedding_gpu_backward.cu:84-87
  BackwardKat<float, int32_t>(stream);
  BackwardKat<float, int64_t>(stream);
  BackwardKat<__half, int32_t>(stream);
  BackwardKat<__half, int64_t>(stream);
  // extensions
  ExtensionTransposeKat<int32_t, float>(stream);
  ExtensionTransposeKat<int64_t, __half>(stream);
  OneCallAndBoundedKat<int32_t>(stream);
  OneCallAndBoundedKat<int64_t>(stream);
  cuembed::SetBackwardTuning(cuembed::BackwardTuning{0, 0});
  HIP_OK(hipStreamDestroy(stream));
  if (g_failures) {
    std::fprintf(stderr, "%d known-answer checks failed\n", g_failures);
    return 1;
  }
  std::printf("header-only API: all known-answer checks passed\n");
  return 0;
}
