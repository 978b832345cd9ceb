// ============================================================================
// ref_datagen_capi.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A C entry point in front of the REFERENCE's own index generator
// (utils/src/datagen.cpp + utils/include/datagen.h), which is compiled
// unmodified, from where it lies under /root/reference, into
// oracle/_ref/libref_datagen.so by `make -C oracle ref`.  No reference source
// is copied into this repository; this file only calls the reference's public
// class.  It is used (a) to pin oracle/cuembed_oracle.cpp's restated generator
// and (b) by tests/golden/make_golden.py to produce index fixtures.
// ============================================================================
#include <cstdint>
#include <cstring>
#include <vector>

#include "utils/include/datagen.h"

extern "C" int ref_psx_samples(int64_t num_categories_arg, int hot, double alpha,
                               int shuffle, int permute, int n_samples,
                               int index_type, void* out) {
  using cuembed::index_generators::PowerLawFeatureGenerator;
  using cuembed::index_generators::PowerLawType;
  if (index_type == 0) {
    PowerLawFeatureGenerator<int32_t> g((int32_t)num_categories_arg, hot, alpha,
                                        shuffle != 0, permute != 0,
                                        PowerLawType::kPsx);
    for (int i = 0; i < n_samples; ++i) {
      std::vector<int32_t> s = g.getCategoryIndices();
      std::memcpy((int32_t*)out + (size_t)i * hot, s.data(), sizeof(int32_t) * hot);
    }
  } else {
    PowerLawFeatureGenerator<int64_t> g((int64_t)num_categories_arg, hot, alpha,
                                        shuffle != 0, permute != 0,
                                        PowerLawType::kPsx);
    for (int i = 0; i < n_samples; ++i) {
      std::vector<int64_t> s = g.getCategoryIndices();
      std::memcpy((int64_t*)out + (size_t)i * hot, s.data(), sizeof(int64_t) * hot);
    }
  }
  return 0;
}
