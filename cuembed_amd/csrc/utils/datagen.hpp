// Synthetic lookup-index generator of the benchmark / test harness.
//
// Host-side counterpart of the reference's utils/include/datagen.h +
// utils/src/datagen.cpp ("Psx" power-law features): category ids are drawn from
// p(x) ~ x^-alpha on [1, N], optionally relabelled by a fixed random permutation
// of [0, N], collected without repetition until a sample has `hotness` ids, and
// optionally shuffled inside the sample.  alpha = 0 degenerates to uniform.
// The draws go through libstdc++'s <random> in the reference's order, so for a
// given (N, hotness, alpha, flags) the stream is the reference's stream.
#ifndef CUEMBED_AMD_UTILS_DATAGEN_HPP_
#define CUEMBED_AMD_UTILS_DATAGEN_HPP_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <random>
#include <set>
#include <vector>

namespace cuembed {
namespace index_generators {

template <typename IndexType = int>
class PowerLawFeatureGenerator {
 public:
  //! \param num_categories largest id that can be drawn (ids lie in [1, N] before
  //!        the permutation, [0, N] after); \param num_hot ids per sample.
  PowerLawFeatureGenerator(IndexType num_categories, int num_hot, double alpha,
                           bool shuffle = false, bool permute = false)
      : num_categories_(num_categories), num_hot_(num_hot), alpha_(alpha),
        shuffle_(shuffle), permute_(permute), uniform_(0.0, 1.0) {
    if (permute_) {
      relabel_.resize(static_cast<size_t>(num_categories_) + 1);
      std::iota(relabel_.begin(), relabel_.end(), IndexType(0));
      std::shuffle(relabel_.begin(), relabel_.end(), layout_rng_);
    }
    const double gamma = 1.0 - alpha_;
    span_ = std::pow(static_cast<double>(num_categories_) + 1.0, gamma) - std::pow(1.0, gamma);
    inv_gamma_ = 1.0 / gamma;
  }

  //! One id; inverse-CDF sampling, narrowed through float like the reference
  //! (datagen.cpp:39-50, :120-132).
  IndexType generateIndex() {
    const double u = uniform_(value_rng_);
    const float y = static_cast<float>(std::pow(u * span_ + 1.0, inv_gamma_));
    const IndexType raw = static_cast<IndexType>(y);
    return permute_ ? relabel_[static_cast<int>(raw)] : raw;
  }

  //! `num_hot` distinct ids for one sample (datagen.cpp:86-104).
  std::vector<IndexType> getCategoryIndices() {
    std::set<IndexType> distinct;
    while (distinct.size() < static_cast<size_t>(num_hot_)) distinct.insert(generateIndex());
    std::vector<IndexType> ids(distinct.begin(), distinct.end());
    if (shuffle_) std::shuffle(ids.begin(), ids.end(), layout_rng_);
    return ids;
  }

 private:
  IndexType num_categories_;
  int num_hot_;
  double alpha_;
  bool shuffle_, permute_;
  double span_, inv_gamma_;
  std::vector<IndexType> relabel_;
  std::default_random_engine layout_rng_;  // permutation + per-sample shuffles
  std::default_random_engine value_rng_;   // uniform draws
  std::uniform_real_distribution<double> uniform_;
};

}  // namespace index_generators
}  // namespace cuembed

#endif  // CUEMBED_AMD_UTILS_DATAGEN_HPP_
