// Contract violations print the failed condition and abort, exactly like the
// reference (embedding_lookup.cuh:151-158).
#ifndef CUEMBED_INCLUDE_CUEMBED_ASSERT_HPP_
#define CUEMBED_INCLUDE_CUEMBED_ASSERT_HPP_

#include <cstdlib>
#include <iostream>

#define CUEMBED_ASSERT(condition)                                           \
  do {                                                                      \
    if (!(condition)) {                                                     \
      std::cerr << "Check failed: " #condition << " at " << __FILE__ << ":" \
                << __LINE__ << std::endl;                                   \
      std::abort();                                                         \
    }                                                                       \
  } while (0)

#endif  // CUEMBED_INCLUDE_CUEMBED_ASSERT_HPP_
