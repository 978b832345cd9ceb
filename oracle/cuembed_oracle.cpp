// ============================================================================
// cuembed_oracle.cpp -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
//
// A plain, scalar, single-source restatement of the reference's algorithms for
// the gather-reduce hot path (forward / backward / transpose / row-id
// extraction / compressed-gradient remap / synthetic-input recipe).  It exists
// to CHECK the HIP path and to serve as bench.py's `cpu_baseline` leg.  Nothing
// under cuembed_amd/ may import, link or call it; only tests/, __graft_entry__.
// smoke() and bench.py's cpu_baseline leg do.
//
// Parity status: PINNED.
//   * every known-answer vector of the reference's own tests
//     (tests/test_embedding_forward.cu:120-160, tests/test_embedding_backward.cu
//     :162-202, tests/test_embedding_transpose.cu:112-122, cuembed/README.md
//     :132,:141,:150,:202) -- see tests/golden/reference_kats.json;
//   * the FNV-1a-64 digests of the reference CPU code's own outputs recorded in
//     SURVEY.md section 8(c) (tests/golden/survey_digests.json);
//   * the index generator is pinned against the reference's datagen.cpp compiled
//     unmodified into oracle/_ref/ (oracle/Makefile target `ref`).
//
// All functions below are written from the reference's documented behaviour;
// each one cites the reference file:line it follows (paths relative to the
// reference checkout).  Arithmetic is done with explicit single operations
// (compile with -ffp-contract=off) so that results are the sequential,
// unfused IEEE results the reference's host code produces.
// ============================================================================
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <numeric>
#include <random>
#include <set>
#include <type_traits>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------
// IEEE binary16 <-> binary32, round-to-nearest-even (what __float2half /
// __half2float do; embedding_lookup_types.cuh:50-62).
// ---------------------------------------------------------------------------
typedef uint16_t h16;

inline float h2f(h16 h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal -> normalise
      int e = -1;
      do {
        man <<= 1;
        ++e;
      } while ((man & 0x400u) == 0);
      man &= 0x3ffu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7f800000u | (man << 13);
  } else {
    bits = sign | ((exp + 112u) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

inline h16 f2h(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7fffffffu;
  if (x >= 0x7f800000u) {  // inf / nan
    return (h16)(sign | 0x7c00u | ((x > 0x7f800000u) ? 0x200u : 0u));
  }
  if (x >= 0x477ff000u) {  // rounds to >= 65520 -> inf
    return (h16)(sign | 0x7c00u);
  }
  if (x < 0x33000001u) {  // <= 2^-25 -> +-0
    return (h16)sign;
  }
  const int e = (int)(x >> 23) - 127;
  uint32_t man = (x & 0x7fffffu) | 0x800000u;
  int shift;
  uint32_t base;
  if (e < -14) {  // subnormal half
    shift = 13 + (-14 - e);
    base = 0;
  } else {
    shift = 13;
    base = (uint32_t)(e + 15) << 10;
    man &= 0x7fffffu;
  }
  const uint32_t lsb = 1u << shift;
  const uint32_t half = lsb >> 1;
  uint32_t q = man >> shift;
  const uint32_t rem = man & (lsb - 1);
  if (rem > half || (rem == half && (q & 1u))) ++q;
  return (h16)(sign | (base + q));  // mantissa carry rolls into the exponent
}

// fp16 arithmetic = exact op in fp32, one rounding to fp16 (fp32 has >= 2*11+2
// significand bits, so the double rounding is innocuous): what __hadd/__hmul do.
inline h16 hadd(h16 a, h16 b) { return f2h(h2f(a) + h2f(b)); }
inline h16 hmul(h16 a, h16 b) { return f2h(h2f(a) * h2f(b)); }

// bfloat16 storage (an extension of this library; the reference has no bf16): the upper 16
// bits of a binary32, converted with round-to-nearest-even (what v_cvt_pk_bf16_f32 does).
struct b16 {
  uint16_t bits;
};
inline float b2f(b16 v) {
  const uint32_t x = (uint32_t)v.bits << 16;
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}
inline b16 f2b(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  b16 r;
  if ((x & 0x7fffffffu) > 0x7f800000u) {  // NaN: keep it a NaN
    r.bits = (uint16_t)((x >> 16) | 0x0040u);
    return r;
  }
  x += 0x7fffu + ((x >> 16) & 1u);  // round to nearest, ties to even
  r.bits = (uint16_t)(x >> 16);
  return r;
}

// ---------------------------------------------------------------------------
// Element traits: T is `float`, `h16` or `b16` (storage type).
// ---------------------------------------------------------------------------
template <typename T> struct El;
template <> struct El<b16> {
  static float to_f(b16 v) { return b2f(v); }
  static b16 from_f(float v) { return f2b(v); }
};
template <> struct El<float> {
  static float to_f(float v) { return v; }
  static float from_f(float v) { return v; }
};
template <> struct El<h16> {
  static float to_f(h16 v) { return h2f(v); }
  static h16 from_f(float v) { return f2h(v); }
};

enum Mode { kSum = 0, kMean = 1, kConcat = 2 };  // embedding_lookup_types.cuh:29

// ---------------------------------------------------------------------------
// Forward.  Follows utils/include/embedding_lookup_cpu.hpp:35-94 (loop order
// batch -> width -> hotness, sequential accumulation in SumT = float unless
// fp16_math with a half table).  Weighted mean is rejected by the CPU reference
// (:51) but defined by the GPU combiner (embedding_lookup_ops.cuh:259-285:
// out = sum * (1.0f / sum_of_weights), zeros when the weight sum is 0); that
// GPU definition is what is restated for (mean, weights != null).
// ---------------------------------------------------------------------------
template <typename T, typename IndexT, typename OffsetT, bool kFp16Math>
void forward_impl(const T* params, int W, int B, int H, const IndexT* indices,
                  const OffsetT* offsets, const T* weights, T* ret, int mode,
                  int64_t b_begin, int64_t b_end) {
  for (int64_t i = b_begin; i < b_end; ++i) {
    const int64_t start = offsets ? (int64_t)offsets[i] : i * (int64_t)H;
    const int hot = offsets ? (int)(offsets[i + 1] - offsets[i]) : H;
    for (int k = 0; k < W; ++k) {
      if (mode == kConcat) {  // :68-70
        for (int j = 0; j < hot; ++j) {
          ret[(start + j) * (int64_t)W + k] =
              params[(int64_t)indices[start + j] * W + k];
        }
        continue;
      }
      if (kFp16Math && std::is_same<T, h16>::value) {
        // SumT = half (:59): every product and every partial sum is rounded
        // to fp16 (embedding_lookup_types.cuh:140-150, :269-291).
        h16 sum = 0;
        float wsum = 0.f;
        for (int j = 0; j < hot; ++j) {
          h16 v;
          std::memcpy(&v, &params[(int64_t)indices[start + j] * W + k], 2);
          if (weights) {
            h16 w;
            std::memcpy(&w, &weights[start + j], 2);
            v = hmul(v, w);
            wsum += h2f(w);
          } else {
            wsum += 1.0f;
          }
          sum = hadd(sum, v);
        }
        if (mode == kMean) {
          // half * float -> half * __float2half(float)
          // (embedding_lookup_types.cuh:301-319; cpu.hpp:82-90)
          if (wsum == 0.f) sum = hmul(sum, f2h(0.0f));
          else sum = hmul(sum, f2h(1.0f / wsum));
        }
        std::memcpy(&ret[i * (int64_t)W + k], &sum, 2);
      } else {
        float sum = 0.0f;
        float wsum = 0.f;
        for (int j = 0; j < hot; ++j) {
          const float v =
              El<T>::to_f(params[(int64_t)indices[start + j] * W + k]);
          if (weights) {
            const float w = El<T>::to_f(weights[start + j]);
            const float p = v * w;  // float * __half2float(w): types.cuh:264-267
            sum = sum + p;
            wsum += w;
          } else {
            sum = sum + v;
            wsum += 1.0f;
          }
        }
        if (mode == kMean) {
          if (wsum == 0.f) sum = sum * 0.0f;          // cpu.hpp:83-86
          else sum = sum * (1.0f / wsum);             // cpu.hpp:88-89
        }
        ret[i * (int64_t)W + k] = El<T>::from_f(sum);
      }
    }
  }
}

template <typename T, typename IndexT, typename OffsetT>
void forward_dispatch(const void* params, int W, int B, int H,
                      const void* indices, const void* offsets,
                      const void* weights, void* ret, int mode, int fp16_math,
                      int threads) {
  const T* p = (const T*)params;
  const IndexT* idx = (const IndexT*)indices;
  const OffsetT* off = (const OffsetT*)offsets;
  const T* w = (const T*)weights;
  T* r = (T*)ret;
  const bool f16m = fp16_math && std::is_same<T, h16>::value;
  if (threads <= 1) {
    if (f16m) forward_impl<T, IndexT, OffsetT, true>(p, W, B, H, idx, off, w, r, mode, 0, B);
    else forward_impl<T, IndexT, OffsetT, false>(p, W, B, H, idx, off, w, r, mode, 0, B);
    return;
  }
#ifdef _OPENMP
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
  for (int t = 0; t < threads; ++t) {
    const int64_t lo = (int64_t)B * t / threads;
    const int64_t hi = (int64_t)B * (t + 1) / threads;
    if (f16m) forward_impl<T, IndexT, OffsetT, true>(p, W, B, H, idx, off, w, r, mode, lo, hi);
    else forward_impl<T, IndexT, OffsetT, false>(p, W, B, H, idx, off, w, r, mode, lo, hi);
  }
}

// ---------------------------------------------------------------------------
// Backward.  Follows utils/include/embedding_lookup_cpu.hpp:96-144: inverse
// mapping at first occurrence of each remapped id (:110-123), optional memset
// (:124-129), then nz-sequential `grad[e + row*W] += grad_y[e + sid*W] * w`
// (:131-143) with the product and the sum in GradT (fp16 rounds twice).
// ---------------------------------------------------------------------------
template <typename T, typename IndexT>
void backward_impl(const T* grad_y, int W, int64_t num_rows, int64_t nnz,
                   const IndexT* t_idx, const IndexT* t_sid,
                   const IndexT* t_remap, const T* t_w, int skip_init,
                   T* grad, IndexT* inverse_mapping) {
  if (nnz == 0) return;
  if (t_remap) {
    inverse_mapping[0] = t_idx[0];
    int64_t cnt = 1;
    for (int64_t i = 1; i < nnz; ++i) {
      if (t_remap[i - 1] != t_remap[i]) inverse_mapping[cnt++] = t_idx[i];
    }
  }
  if (!skip_init) std::memset(grad, 0, (size_t)num_rows * W * sizeof(T));
  for (int64_t nz = 0; nz < nnz; ++nz) {
    const int64_t row = t_remap ? (int64_t)t_remap[nz] : (int64_t)t_idx[nz];
    const int64_t sid = (int64_t)t_sid[nz];
    for (int e = 0; e < W; ++e) {
      // product and sum are both rounded to GradT, like `grad += result_grad * weight`
      // in GradT arithmetic (embedding_lookup_cpu.hpp:139-142)
      T& dst = grad[e + row * W];
      float p = El<T>::to_f(grad_y[e + sid * W]);
      if (t_w) p = El<T>::to_f(El<T>::from_f(p * El<T>::to_f(t_w[nz])));
      dst = El<T>::from_f(El<T>::to_f(dst) + p);
    }
  }
}

// ---------------------------------------------------------------------------
// Transpose.  The device contract (index_transforms.cuh:95-137) is a STABLE
// sort of (sample id [, weight]) by lookup index; the CPU reference
// (index_transforms_cpu.hpp:86-125) sorts by (index, sample id, weight).  Both
// are provided; they coincide whenever a sample never repeats an index, which
// the generator guarantees (datagen.cpp:86-104).
// ---------------------------------------------------------------------------
template <typename IndexT, typename WT>
void transpose_impl(const IndexT* rows, const IndexT* cols, const WT* weights,
                    int64_t nnz, IndexT* t_rows, IndexT* t_cols, WT* t_weights,
                    int full_tuple_order) {
  std::vector<int64_t> perm((size_t)nnz);
  std::iota(perm.begin(), perm.end(), (int64_t)0);
  if (!full_tuple_order) {
    std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) {
      return cols[a] < cols[b];
    });
  } else {
    std::sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) {
      if (cols[a] != cols[b]) return cols[a] < cols[b];
      if (rows[a] != rows[b]) return rows[a] < rows[b];
      if (weights) {
        float wa, wb;
        if (sizeof(WT) == 2) {
          h16 ha, hb;
          std::memcpy(&ha, &weights[a], 2);
          std::memcpy(&hb, &weights[b], 2);
          wa = h2f(ha);
          wb = h2f(hb);
        } else {
          std::memcpy(&wa, &weights[a], 4);
          std::memcpy(&wb, &weights[b], 4);
        }
        if (wa != wb) return wa < wb;
      }
      return a < b;
    });
  }
  for (int64_t i = 0; i < nnz; ++i) {
    t_rows[i] = cols[perm[i]];
    t_cols[i] = rows[perm[i]];
    if (weights && t_weights) t_weights[i] = weights[perm[i]];
  }
}

// FNV-1a 64 over raw bytes (the digest SURVEY.md section 8(c) quotes).
uint64_t fnv1a64(const void* data, size_t n) {
  const unsigned char* p = (const unsigned char*)data;
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) {
    h ^= p[i];
    h *= 1099511628211ull;
  }
  return h;
}

// ---------------------------------------------------------------------------
// Synthetic inputs.  Power-law ("Psx") lookup-index generator, restating
// utils/src/datagen.cpp:39-132 + utils/include/datagen.h:77,:114: two
// default-seeded std::default_random_engine instances (one for the permutation
// and the per-sample shuffle, one for the uniform draws); u in [0,1) double ->
// y = float(pow(u * ((N+1)^g - 1) + 1, 1/g)), g = 1 - alpha, truncated to the
// index type; mapped through a random permutation of [0..N]; distinct values
// collected in an ordered set until `hot` of them exist; emitted ascending and
// then std::shuffle'd.  libstdc++'s <random> algorithms are implementation-
// defined, so the same standard-library calls are made in the same order.
// ---------------------------------------------------------------------------
template <typename IndexT>
class PsxGenerator {
 public:
  PsxGenerator(IndexT n_cat, int hot, double alpha, bool shuffle, bool permute)
      : n_(n_cat), hot_(hot), alpha_(alpha), shuffle_(shuffle), permute_(permute),
        unit_(0.0, 1.0) {
    if (permute_) {  // datagen.cpp:64-74
      perm_.resize((size_t)n_ + 1);
      std::iota(perm_.begin(), perm_.end(), (IndexT)0);
      std::shuffle(perm_.begin(), perm_.end(), order_rng_);
    }
  }
  IndexT draw() {  // datagen.cpp:39-50, :120-132
    const double u = unit_(draw_rng_);
    const double g = 1.0 - alpha_;
    const double hi = (double)(n_ + 1);
    const float y = (float)std::pow(u * (std::pow(hi, g) - std::pow(1.0, g)) +
                                        std::pow(1.0, g),
                                    1.0 / g);
    const IndexT raw = (IndexT)y;
    return permute_ ? perm_[(int)raw] : raw;
  }
  void sample(std::vector<IndexT>* out) {  // datagen.cpp:86-104
    std::set<IndexT> seen;
    while (seen.size() < (size_t)hot_) seen.insert(draw());
    out->assign(seen.begin(), seen.end());
    if (shuffle_) std::shuffle(out->begin(), out->end(), order_rng_);
  }

 private:
  IndexT n_;
  int hot_;
  double alpha_;
  bool shuffle_, permute_;
  std::vector<IndexT> perm_;
  std::default_random_engine order_rng_;
  std::default_random_engine draw_rng_;
  std::uniform_real_distribution<double> unit_;
};

template <typename IndexT>
int64_t gen_indices_impl(int64_t num_categories, int B, int H, double alpha,
                         int shuffle, int permute, const int32_t* offsets,
                         IndexT* out) {
  // utils/src/embedding_allocation.cu:139-158: generator over
  // num_categories - 1; every sample draws a full H-set and keeps the first
  // `hotness_for_sample` entries (all H for fixed hotness).
  PsxGenerator<IndexT> gen((IndexT)(num_categories - 1), H, alpha, shuffle != 0,
                           permute != 0);
  std::vector<IndexT> s;
  int64_t n = 0;
  for (int i = 0; i < B; ++i) {
    gen.sample(&s);
    const int keep = offsets ? (offsets[i + 1] - offsets[i]) : H;
    for (int j = 0; j < keep; ++j) out[n++] = s[j];
  }
  return n;
}

}  // namespace

// ===========================================================================
// C interface (ctypes).  type codes: elem 0 = f32, 1 = f16, 2 = bf16 (bits in uint16);
// index/offset 0 = int32, 1 = int64.  Return 0 on success, -1 on a contract
// violation (the reference CHECK-fails / aborts in those cases).
// ===========================================================================
extern "C" {

int oracle_embedding_forward(const void* params, int elem_type, int embed_width,
                             int batch_size, int num_hots, const void* indices,
                             int index_type, const void* offsets,
                             int offset_type, const void* weights, void* ret,
                             int mode, int fp16_math, int threads) {
  // embedding_lookup_cpu.hpp:50-56 / embedding_lookup.cuh:261-267
  if (weights && mode == kConcat) return -1;
  if (!((offsets && num_hots == 0) || (!offsets && num_hots > 0))) return -1;
  if (offsets && mode == kConcat) return -1;
#define FWD(T, I, O)                                                         \
  forward_dispatch<T, I, O>(params, embed_width, batch_size, num_hots,       \
                            indices, offsets, weights, ret, mode, fp16_math, \
                            threads)
  const int key = elem_type * 4 + index_type * 2 + offset_type;
  switch (key) {
    case 0: FWD(float, int32_t, int32_t); break;
    case 1: FWD(float, int32_t, int64_t); break;
    case 2: FWD(float, int64_t, int32_t); break;
    case 3: FWD(float, int64_t, int64_t); break;
    case 4: FWD(h16, int32_t, int32_t); break;
    case 5: FWD(h16, int32_t, int64_t); break;
    case 6: FWD(h16, int64_t, int32_t); break;
    case 7: FWD(h16, int64_t, int64_t); break;
    case 8: FWD(b16, int32_t, int32_t); break;
    case 9: FWD(b16, int32_t, int64_t); break;
    case 10: FWD(b16, int64_t, int32_t); break;
    case 11: FWD(b16, int64_t, int64_t); break;
    default: return -1;
  }
#undef FWD
  return 0;
}

int oracle_embedding_backward(const void* grad_y, int elem_type,
                              int embed_width, int64_t num_grad_rows,
                              int64_t nnz, const void* t_indices,
                              const void* t_sample_ids, const void* t_remapped,
                              int index_type, const void* t_weights,
                              int skip_grad_init, void* grad_embedding,
                              void* inverse_mapping) {
#define BWD(T, I)                                                              \
  backward_impl<T, I>((const T*)grad_y, embed_width, num_grad_rows, nnz,       \
                      (const I*)t_indices, (const I*)t_sample_ids,             \
                      (const I*)t_remapped, (const T*)t_weights,               \
                      skip_grad_init, (T*)grad_embedding, (I*)inverse_mapping)
  switch (elem_type * 2 + index_type) {
    case 0: BWD(float, int32_t); break;
    case 1: BWD(float, int64_t); break;
    case 2: BWD(h16, int32_t); break;
    case 3: BWD(h16, int64_t); break;
    case 4: BWD(b16, int32_t); break;
    case 5: BWD(b16, int64_t); break;
    default: return -1;
  }
#undef BWD
  return 0;
}

// stable != 0: device contract (stable by index).  stable == 0: the CPU
// reference's (index, sample id, weight) order.
int oracle_transpose(const void* rows, const void* cols, const void* weights,
                     int64_t nnz, int index_type, int weight_type, void* t_rows,
                     void* t_cols, void* t_weights, int stable) {
#define TR(I, W)                                                           \
  transpose_impl<I, W>((const I*)rows, (const I*)cols, (const W*)weights,  \
                       nnz, (I*)t_rows, (I*)t_cols, (W*)t_weights, !stable)
  switch (index_type * 2 + (weight_type != 0 ? 1 : 0)) {  // 2-byte weights are only moved
    case 0: TR(int32_t, float); break;
    case 1: TR(int32_t, h16); break;
    case 2: TR(int64_t, float); break;
    case 3: TR(int64_t, h16); break;
    default: return -1;
  }
#undef TR
  return 0;
}

// index_transforms_cpu.hpp:35-44
int oracle_extract_row_ids_from_fixed(int batch_size, int num_hots,
                                      int index_type, void* row_ids) {
  int64_t n = 0;
  for (int b = 0; b < batch_size; ++b)
    for (int h = 0; h < num_hots; ++h, ++n) {
      if (index_type) ((int64_t*)row_ids)[n] = b;
      else ((int32_t*)row_ids)[n] = b;
    }
  return 0;
}

// index_transforms_cpu.hpp:46-57
int oracle_extract_row_ids_from_csr(const void* offsets, int offset_type,
                                    int batch_size, int index_type,
                                    void* row_ids) {
  int64_t n = 0;
  for (int b = 0; b < batch_size; ++b) {
    const int64_t lo = offset_type ? ((const int64_t*)offsets)[b]
                                   : ((const int32_t*)offsets)[b];
    const int64_t hi = offset_type ? ((const int64_t*)offsets)[b + 1]
                                   : ((const int32_t*)offsets)[b + 1];
    for (int64_t o = lo; o < hi; ++o, ++n) {
      if (index_type) ((int64_t*)row_ids)[n] = b;
      else ((int32_t*)row_ids)[n] = b;
    }
  }
  return 0;
}

// index_transforms_cpu.hpp:59-64
int oracle_extract_row_ids_for_concat(int64_t nnz, int index_type,
                                      void* row_ids) {
  for (int64_t i = 0; i < nnz; ++i) {
    if (index_type) ((int64_t*)row_ids)[i] = i;
    else ((int32_t*)row_ids)[i] = (int32_t)i;
  }
  return 0;
}

// index_transforms_cpu.hpp:66-77
int oracle_compute_compressed_grad_indices(const void* indices, int64_t nnz,
                                           int index_type, void* remapped) {
  int64_t uniq = 0;
  for (int64_t i = 0; i < nnz; ++i) {
    if (index_type) {
      const int64_t* p = (const int64_t*)indices;
      if (i > 0 && p[i] != p[i - 1]) ++uniq;
      ((int64_t*)remapped)[i] = uniq;
    } else {
      const int32_t* p = (const int32_t*)indices;
      if (i > 0 && p[i] != p[i - 1]) ++uniq;
      ((int32_t*)remapped)[i] = (int32_t)uniq;
    }
  }
  return 0;
}

uint64_t oracle_fnv1a64(const void* data, uint64_t nbytes) {
  return fnv1a64(data, (size_t)nbytes);
}

// ---- synthetic inputs (utils/src/embedding_allocation.cu:96-169, :221-247) --
// All of table / offsets / weights come from ONE engine seeded 123456, drawn in
// this order: rows*W table values, B offset increments, (indices use their own
// engines), nnz weight coin flips.  The caller gets them through one call so
// the engine state threads through exactly as in AllocateForward.
//
// table     : [num_categories * W] of elem_type (uniform_real<float>(-1,1) cast)
// offsets   : [B + 1] int32 (increments uniform_int(0, H)); always drawn
// indices   : capacity B*H of index_type; *nnz_out receives the count
// weights   : capacity B*H of elem_type: bernoulli(0.5) ? 0.5 : 0.25
int oracle_allocate_forward(int64_t num_categories, int embed_width,
                            int batch_size, int hotness, double alpha,
                            int is_csr, int shuffle, int permute, int elem_type,
                            int index_type, void* table, int32_t* offsets,
                            void* indices, void* weights, int64_t* nnz_out) {
  std::default_random_engine rng(123456);  // allocation.cu:113
  {
    std::uniform_real_distribution<float> dist(-1, 1);
    const int64_t n = num_categories * (int64_t)embed_width;
    if (elem_type == 0) {
      float* t = (float*)table;
      for (int64_t i = 0; i < n; ++i) t[i] = dist(rng);
    } else {
      h16* t = (h16*)table;
      for (int64_t i = 0; i < n; ++i) t[i] = f2h(dist(rng));
    }
  }
  offsets[0] = 0;  // allocation.cu:128-136
  {
    std::uniform_int_distribution<> inc(0, hotness);
    for (int i = 0; i < batch_size; ++i) offsets[i + 1] = offsets[i] + inc(rng);
  }
  const int32_t* off = is_csr ? offsets : nullptr;
  int64_t nnz;
  if (index_type == 0)
    nnz = gen_indices_impl<int32_t>(num_categories, batch_size, hotness, alpha,
                                    shuffle, permute, off, (int32_t*)indices);
  else
    nnz = gen_indices_impl<int64_t>(num_categories, batch_size, hotness, alpha,
                                    shuffle, permute, off, (int64_t*)indices);
  {  // allocation.cu:160-168
    std::bernoulli_distribution coin(0.5);
    for (int64_t i = 0; i < nnz; ++i) {
      const float w = coin(rng) ? 0.5f : 0.25f;
      if (elem_type == 0) ((float*)weights)[i] = w;
      else ((h16*)weights)[i] = f2h(w);
    }
  }
  *nnz_out = nnz;
  return 0;
}

// Indices only (no table): the generator by itself, for shapes whose table is
// produced elsewhere (bench.py fills the 10M-row table on the GPU).
int64_t oracle_generate_indices(int64_t num_categories, int batch_size,
                                int hotness, double alpha, int shuffle,
                                int permute, int index_type,
                                const int32_t* offsets_or_null, void* indices) {
  if (index_type == 0)
    return gen_indices_impl<int32_t>(num_categories, batch_size, hotness, alpha,
                                     shuffle, permute, offsets_or_null,
                                     (int32_t*)indices);
  return gen_indices_impl<int64_t>(num_categories, batch_size, hotness, alpha,
                                   shuffle, permute, offsets_or_null,
                                   (int64_t*)indices);
}

// Raw generator: `n_samples` consecutive getCategoryIndices() results of a
// PowerLawFeatureGenerator(num_categories_arg, hot, alpha, shuffle, permute).
int oracle_psx_samples(int64_t num_categories_arg, int hot, double alpha,
                       int shuffle, int permute, int n_samples, int index_type,
                       void* out) {
  if (index_type == 0) {
    PsxGenerator<int32_t> g((int32_t)num_categories_arg, hot, alpha, shuffle, permute);
    std::vector<int32_t> s;
    for (int i = 0; i < n_samples; ++i) {
      g.sample(&s);
      std::memcpy((int32_t*)out + (size_t)i * hot, s.data(), sizeof(int32_t) * hot);
    }
  } else {
    PsxGenerator<int64_t> g((int64_t)num_categories_arg, hot, alpha, shuffle, permute);
    std::vector<int64_t> s;
    for (int i = 0; i < n_samples; ++i) {
      g.sample(&s);
      std::memcpy((int64_t*)out + (size_t)i * hot, s.data(), sizeof(int64_t) * hot);
    }
  }
  return 0;
}

// allocation.cu:234-237: engine(654321), uniform_int<int>(-10,10) cast to GradT.
int oracle_allocate_grad_y(int64_t count, int elem_type, void* grad_y) {
  std::default_random_engine rng(654321);
  std::uniform_int_distribution<int> dist(-10, 10);
  for (int64_t i = 0; i < count; ++i) {
    const float v = (float)dist(rng);
    if (elem_type == 0) ((float*)grad_y)[i] = v;
    else ((h16*)grad_y)[i] = f2h(v);
  }
  return 0;
}

// conversions exposed for tests (checked against numpy.float16)
uint16_t oracle_f2b(float f) { return f2b(f).bits; }
uint16_t oracle_f2h(float f) { return f2h(f); }
float oracle_h2f(uint16_t h) { return h2f(h); }

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

}  // extern "C"
