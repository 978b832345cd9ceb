// Host-side synthetic workload generation for the benchmark harness and the
// tests (C ABI, loaded with ctypes as cuembed_amd/lib/libcuembed_harness.so).
//
// Counterpart of the reference's utils/src/embedding_allocation.cu
// (AllocateForward :96-169, AllocateBackward :221-247): same seeds, same
// distributions, same draw order, so a workload is defined by its flags alone.
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>

#include "utils/datagen.hpp"

namespace {

// IEEE binary32 -> binary16, round-to-nearest-even.
uint16_t FloatToHalfBits(float f) {
  uint32_t x;
  std::memcpy(&x, &f, sizeof x);
  const uint16_t sign = static_cast<uint16_t>((x >> 16) & 0x8000u);
  const uint32_t mag = x & 0x7fffffffu;
  if (mag > 0x7f800000u) return sign | 0x7e00u;          // NaN
  if (mag >= 0x477ff000u) return sign | 0x7c00u;         // overflow -> inf
  if (mag <= 0x33000000u) return sign;                   // underflow -> 0
  const int exp = static_cast<int>(mag >> 23);           // biased fp32 exponent
  uint32_t sig = (mag & 0x007fffffu) | 0x00800000u;      // 24-bit significand
  // target: 10 fraction bits for normals; subnormal halves lose more bits
  const int drop = exp >= 113 ? 13 : 13 + (113 - exp);
  const uint32_t kept = sig >> drop;
  const uint32_t rest = sig & ((1u << drop) - 1u);
  const uint32_t tie = 1u << (drop - 1);
  uint32_t rounded = kept + ((rest > tie || (rest == tie && (kept & 1u))) ? 1u : 0u);
  if (exp >= 113) rounded += static_cast<uint32_t>(exp - 113) << 10;  // kept has the hidden bit
  return sign | static_cast<uint16_t>(rounded);
}

template <typename IndexT>
int64_t FillIndices(int64_t num_categories, int batch, int hotness, double alpha, bool shuffle,
                    bool permute, const int32_t* offsets, IndexT* out) {
  cuembed::index_generators::PowerLawFeatureGenerator<IndexT> gen(
      static_cast<IndexT>(num_categories - 1), hotness, alpha, shuffle, permute);
  int64_t n = 0;
  for (int s = 0; s < batch; ++s) {
    const std::vector<IndexT> ids = gen.getCategoryIndices();
    const int keep = offsets ? offsets[s + 1] - offsets[s] : hotness;
    std::memcpy(out + n, ids.data(), sizeof(IndexT) * keep);
    n += keep;
  }
  return n;
}

}  // namespace

extern "C" {

// Lookup indices only: `batch` samples of `hotness` distinct ids in
// [0, num_categories).  offsets (int32[batch+1]) truncates sample s to
// offsets[s+1]-offsets[s] ids (CSR); NULL keeps all.  Returns the id count.
int64_t cuembed_harness_generate_indices(int64_t num_categories, int batch, int hotness,
                                         double alpha, int shuffle, int permute, int index_is_64,
                                         const int32_t* offsets, void* out) {
  if (index_is_64)
    return FillIndices<int64_t>(num_categories, batch, hotness, alpha, shuffle, permute, offsets,
                                static_cast<int64_t*>(out));
  return FillIndices<int32_t>(num_categories, batch, hotness, alpha, shuffle, permute, offsets,
                              static_cast<int32_t*>(out));
}

// The whole forward workload with the engine threading of AllocateForward:
// engine(123456) -> table values U(-1,1) [skipped when table == NULL, but the
// draws are still consumed so offsets/weights stay reproducible only if
// `consume_table_draws` != 0] -> batch offset increments U{0..hotness} ->
// (indices from their own engines) -> one Bernoulli(0.5) per id: 0.5 or 0.25.
int64_t cuembed_harness_allocate_forward(int64_t num_categories, int embed_width, int batch,
                                         int hotness, double alpha, int is_csr, int shuffle,
                                         int permute, int elem_is_half, int index_is_64,
                                         void* table, int consume_table_draws, int32_t* offsets,
                                         void* indices, void* weights) {
  std::default_random_engine rng(123456);
  std::uniform_real_distribution<float> value(-1, 1);
  const int64_t cells = num_categories * static_cast<int64_t>(embed_width);
  if (table != nullptr) {
    for (int64_t i = 0; i < cells; ++i) {
      const float v = value(rng);
      if (elem_is_half) static_cast<uint16_t*>(table)[i] = FloatToHalfBits(v);
      else static_cast<float*>(table)[i] = v;
    }
  } else if (consume_table_draws) {
    for (int64_t i = 0; i < cells; ++i) (void)value(rng);
  }
  offsets[0] = 0;
  std::uniform_int_distribution<> increment(0, hotness);
  for (int s = 0; s < batch; ++s) offsets[s + 1] = offsets[s] + increment(rng);

  const int64_t nnz = cuembed_harness_generate_indices(num_categories, batch, hotness, alpha,
                                                       shuffle, permute, index_is_64,
                                                       is_csr ? offsets : nullptr, indices);
  if (weights != nullptr) {
    std::bernoulli_distribution coin(0.5);
    for (int64_t i = 0; i < nnz; ++i) {
      const float w = coin(rng) ? 0.5f : 0.25f;
      if (elem_is_half) static_cast<uint16_t*>(weights)[i] = FloatToHalfBits(w);
      else static_cast<float*>(weights)[i] = w;
    }
  }
  return nnz;
}

// Incoming gradient of AllocateBackward: engine(654321), integers U{-10..10}.
void cuembed_harness_allocate_grad_y(int64_t count, int elem_is_half, void* grad_y) {
  std::default_random_engine rng(654321);
  std::uniform_int_distribution<int> value(-10, 10);
  for (int64_t i = 0; i < count; ++i) {
    const float v = static_cast<float>(value(rng));
    if (elem_is_half) static_cast<uint16_t*>(grad_y)[i] = FloatToHalfBits(v);
    else static_cast<float*>(grad_y)[i] = v;
  }
}

}  // extern "C"
