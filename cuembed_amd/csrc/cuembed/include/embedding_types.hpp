// MI355X (gfx950 / CDNA4) embedding lookup -- element, pack and trait types.
//
// Counterpart of the reference's cuembed/include/embedding_lookup_types.cuh
// (CombineMode :29, GetElemT :576-586, the vector structs :32-39 and their
// casts/operators :45-319).  Here a row slice owned by one lane is a `Pack<T,N>`
// (N elements, naturally aligned to 4/8/16 bytes so that it moves with one
// global_load_dword / dwordx2 / dwordx4), and all arithmetic goes through two
// tiny functors so that every add and multiply is a single, unfused IEEE
// operation -- the results are then identical to a sequential host loop.
#ifndef CUEMBED_INCLUDE_EMBEDDING_TYPES_HPP_
#define CUEMBED_INCLUDE_EMBEDDING_TYPES_HPP_

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

namespace cuembed {

//! How the rows looked up for one sample are combined
//! (reference: embedding_lookup_types.cuh:29).
enum class CombineMode { kSum, kMean, kConcat };

//! Customisation point kept from the reference (embedding_lookup_types.cuh
//! :576-586): the scalar element type behind a (possibly structured) table type.
template <typename T>
struct GetElemType {
  using Type = T;
};
template <typename T>
using GetElemT = typename GetElemType<T>::Type;

namespace detail {

// Device-side storage type: `__half` tables are handled as `_Float16`, which
// the compiler lowers to native v_*_f16 / v_cvt_f32_f16.
template <typename T>
struct DeviceElem {
  using type = T;
};
template <>
struct DeviceElem<__half> {
  using type = _Float16;
};
// bf16 tables (an extension: the reference lists bf16 as future work, README.md:111).
// `__bf16` converts to/from float with round-to-nearest-even in hardware on gfx950
// (v_cvt_pk_bf16_f32); accumulation is always fp32.
template <>
struct DeviceElem<__hip_bfloat16> {
  using type = __bf16;
};
template <typename T>
using DeviceElemT = typename DeviceElem<T>::type;

//! N consecutive row elements owned by one lane.
template <typename T, int N>
struct alignas(sizeof(T) * N) Pack {
  T v[N];
};

//! Single-rounding arithmetic.  `#pragma clang fp contract(off)` keeps the
//! compiler from fusing a*b+c into an FMA, so `acc = acc + v*w` rounds twice
//! exactly like the reference's host loop (embedding_lookup_cpu.hpp:75).
template <typename AccT>
struct Arith {
  template <typename ElemT>
  static __host__ __device__ __forceinline__ AccT widen(ElemT x) {
    return static_cast<AccT>(x);
  }
  static __host__ __device__ __forceinline__ AccT add(AccT a, AccT b) {
#pragma clang fp contract(off)
    return a + b;
  }
  static __host__ __device__ __forceinline__ AccT mul(AccT a, AccT b) {
#pragma clang fp contract(off)
    return a * b;
  }
};

template <typename T>
struct IsHalf : std::false_type {};
template <>
struct IsHalf<_Float16> : std::true_type {};
template <>
struct IsHalf<__half> : std::true_type {};

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_EMBEDDING_TYPES_HPP_
