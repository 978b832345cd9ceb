// Shared bits of the C-ABI translation units (see include/cuembed_amd.h).
#ifndef CUEMBED_AMD_C_API_COMMON_HPP_
#define CUEMBED_AMD_C_API_COMMON_HPP_

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <iostream>

#include "cuembed_amd.h"

namespace cuembed_c_api {
inline hipStream_t Stream(cuembed_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
}  // namespace cuembed_c_api

#define CUEMBED_C_API_BAD_TYPE()                                                     \
  do {                                                                               \
    std::cerr << "Check failed: unsupported type code at " << __FILE__ << ":"        \
              << __LINE__ << std::endl;                                              \
    std::abort();                                                                    \
  } while (0)

#endif  // CUEMBED_AMD_C_API_COMMON_HPP_
