// MI355X (gfx950 / CDNA4) -- what the sorts of this library share: the wave-synchronous digit match, the sort's
// treatment of its arrays (SortMode) and the implicit first payload.  See radix_sort_kernels.hpp (many workgroups,
// three launches per pass) and block_sort_kernels.hpp (one workgroup, one launch for the whole sort).
#ifndef CUEMBED_INCLUDE_SORT_COMMON_HPP_
#define CUEMBED_INCLUDE_SORT_COMMON_HPP_

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <type_traits>

#include "cuembed/include/blocked_order.hpp"

namespace cuembed {
namespace detail {

constexpr int kSortThreads = 256;
constexpr int kSortWaves = kSortThreads / 64;
constexpr int kSortItems = 16;                          // keys per lane (8 measured 10 % slower)
static_assert(kSortTile == kSortThreads * kSortItems, "4096 keys per workgroup (blocked_order.hpp)");
constexpr int kSortBins = 256;                          // 8-bit digits

struct NoPayload {};

//! Lanes of the wavefront (among `valid` ones) whose 8-bit digit equals this lane's.
//! Per digit bit b: m = ballot(bit b set); a lane keeps the peers that agree with it on bit b,
//! peers &= ~(m ^ sel) with sel = all-ones if its own bit is set.  That three-input function is
//! ONE v_bitop3_b32 per mask half on gfx950 (truth table 0x90 = S0 & ~(S1 ^ S2)).
__device__ __forceinline__ unsigned long long MatchDigit(const unsigned digit, const bool valid) {
  const unsigned long long v = __ballot(valid);
  unsigned lo = static_cast<unsigned>(v);
  unsigned hi = static_cast<unsigned>(v >> 32);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const int sel = __builtin_amdgcn_sbfe(static_cast<int>(digit), b, 1);  // v_bfe_i32: bit b ? -1 : 0
    const unsigned long long m = __ballot(sel != 0);
    lo = __builtin_amdgcn_bitop3_b32(lo, static_cast<unsigned>(m), static_cast<unsigned>(sel), 0x90);
    hi = __builtin_amdgcn_bitop3_b32(hi, static_cast<unsigned>(m >> 32), static_cast<unsigned>(sel), 0x90);
  }
  return (static_cast<unsigned long long>(hi) << 32) | lo;
}

__device__ __forceinline__ unsigned long long LanesBelow(const int lane) {
  return (1ull << lane) - 1ull;
}

//! Number of set bits of `mask` in the lanes BELOW this one: v_mbcnt_lo + v_mbcnt_hi.
__device__ __forceinline__ unsigned CountBelow(const unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u));
}

//! How one sort treats its arrays; the same for every pass and kernel of the sort.
struct SortMode {
  int narrow_keys;   //!< NarrowKeys: 64-bit keys stored as 32 bits in the scratch
  int narrow_v1;     //!< NarrowKeys: 64-bit first payload stored as 32 bits (IfConstantHigh: if all values < 2^32)
  int use_varying;   //!< passes whose digit is the same for every key are skipped on the device
  int sign_pass;     //!< pass whose digit holds the sign bit of the keys, or -1
  // Implicit first payload: v1_div > 0 means "the first payload of input element i is i / v1_div"
  // (the sample id of lookup i of a fixed-hotness batch) -- pass 0 computes it instead of loading an
  // array that somebody would have had to write first.  i / d = (i * magic) >> shift for i < 2^31.
  int v1_div;
  unsigned v1_magic;
  int v1_shift;
};

//! magic / shift with (uint64(i) * magic) >> shift == i / d for every 0 <= i < 2^31, d >= 1:
//! s = ceil(log2 d), magic = ceil(2^(31+s) / d) < 2^32; the error term i * e / (d * 2^(31+s)) with
//! e < d <= 2^s stays below 1 / d.
inline void ImplicitPayloadDivisor(const int d, SortMode* mode) {
  int s = 0;
  while ((int64_t{1} << s) < d) ++s;
  const unsigned __int128 one = static_cast<unsigned __int128>(1) << (31 + s);
  mode->v1_div = d;
  mode->v1_magic = static_cast<unsigned>((one + d - 1) / d);
  mode->v1_shift = 31 + s;
}

__host__ __device__ __forceinline__ unsigned ImplicitPayload(const SortMode& mode, const int64_t i) {
  return static_cast<unsigned>((static_cast<unsigned long long>(static_cast<unsigned>(i)) * mode.v1_magic) >> mode.v1_shift);
}

__device__ __forceinline__ unsigned SignFlip(const SortMode& mode, const int pass) {
  return pass == mode.sign_pass ? 0x80u : 0u;
}

inline size_t SortAlign(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace detail
}  // namespace cuembed

#endif  // CUEMBED_INCLUDE_SORT_COMMON_HPP_
