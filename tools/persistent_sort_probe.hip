// What does a radix pass cost inside ONE persistent launch (grid-wide barriers between histogram, scan and scatter)
// against three launches?  Transpose() of nnz int64 keys that use ALL 64 bits (so the four passes over the high word
// really work), sample ids as payload; knobs through the environment (read once per process):
//   CUEMBED_SORT_HIGH_WORD_LAUNCHES=1   every pass as launches (24 for 8 passes)
//   (default)                          passes 4..7 inside RadixHighPassesKernel, one workgroup per compute unit
// and, for scale, the same keys with the high word cleared (passes 4..7 skipped on the device).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cuembed_amd/csrc tools/persistent_sort_probe.hip -o tools/persistent_sort_probe
//   tools/persistent_sort_probe [nnz = 4194304]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

#include "cuembed/include/index_transforms.hpp"

#define HIP_OK(x)                                                                            \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) {                                                                  \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                          \
    }                                                                                        \
  } while (0)

int main(int argc, char** argv) {
  const int nnz = argc > 1 ? std::atoi(argv[1]) : 4194304;
  std::vector<int64_t> keys(nnz), rows(nnz);
  std::mt19937_64 rng(5);
  for (auto& k : keys) k = static_cast<int64_t>(rng());
  std::iota(rows.begin(), rows.end(), int64_t{0});
  int64_t *d_keys, *d_rows, *o_keys, *o_rows;
  HIP_OK(hipMalloc(&d_keys, nnz * 8));
  HIP_OK(hipMalloc(&d_rows, nnz * 8));
  HIP_OK(hipMalloc(&o_keys, nnz * 8));
  HIP_OK(hipMalloc(&o_rows, nnz * 8));
  HIP_OK(hipMemcpy(d_rows, rows.data(), nnz * 8, hipMemcpyHostToDevice));
  size_t lwork = 0;
  cuembed::Transpose<int64_t, float>(d_rows, d_keys, nullptr, nnz, o_keys, o_rows, nullptr, nullptr, &lwork);
  char* work;
  HIP_OK(hipMalloc(&work, lwork));
  hipEvent_t a, z;
  HIP_OK(hipEventCreate(&a));
  HIP_OK(hipEventCreate(&z));
  const char* knob = std::getenv("CUEMBED_SORT_HIGH_WORD_LAUNCHES");
  for (int full = 1; full >= 0; --full) {
    if (!full)
      for (auto& k : keys) k &= 0x7fffffffll;
    HIP_OK(hipMemcpy(d_keys, keys.data(), nnz * 8, hipMemcpyHostToDevice));
    auto run = [&] {
      cuembed::Transpose<int64_t, float>(d_rows, d_keys, nullptr, nnz, o_keys, o_rows, nullptr, work, &lwork, 0, 64, 32);
    };
    for (int t = 0; t < 5; ++t) run();
    HIP_OK(hipEventRecord(a));
    const int iters = 30;
    for (int t = 0; t < iters; ++t) run();
    HIP_OK(hipEventRecord(z));
    HIP_OK(hipEventSynchronize(z));
    float ms = 0;
    HIP_OK(hipEventElapsedTime(&ms, a, z));
    // check: keys ascending (signed), payload of equal keys ascending
    std::vector<int64_t> hk(nnz), hr(nnz);
    HIP_OK(hipMemcpy(hk.data(), o_keys, nnz * 8, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(hr.data(), o_rows, nnz * 8, hipMemcpyDeviceToHost));
    bool ok = true;
    for (int i = 1; i < nnz && ok; ++i) ok = hk[i - 1] < hk[i] || (hk[i - 1] == hk[i] && hr[i - 1] < hr[i]);
    for (int i = 0; i < nnz && ok; i += 997) ok = keys[hr[i]] == hk[i];
    std::printf("nnz=%d high_word_launches=%s keys=%s ms=%.4f sorted=%s\n", nnz, knob ? knob : "(default)",
                full ? "64-bit" : "31-bit", ms / iters, ok ? "yes" : "NO");
  }
  return 0;
}
