// C ABI: forward entry points = explicit instantiations of
// cuembed::EmbeddingForward (reference instantiation list:
// utils/src/embedding_gpu_forward.cu:69-76; int64 offsets as used by
// examples/pytorch/cuembed_embedding.cu:39-49).
#include "c_api_common.hpp"
#include "cuembed/include/embedding_lookup.hpp"

using cuembed::CombineMode;
using cuembed_c_api::Stream;

namespace {
template <typename ElemT, typename IndexT, typename OffsetT>
void Forward(const void* params, int embed_width, const IndexT* indices, const OffsetT* offsets,
             const void* weights, int batch_size, int num_hots, int mode, int fp16_math,
             void* ret, cuembed_stream_t stream, int reduction_order = -1, int row_load_policy = -1,
             const int32_t* sample_order = nullptr, const uint32_t* row_loads_device = nullptr) {
  // per-call options (cuembed_embedding_forward_with_options); < 0 = the process-wide default
  cuembed::ForwardOptions options = cuembed::DefaultForwardOptions();
  options.sample_order = sample_order;
  options.row_loads_device = row_loads_device;
  CUEMBED_ASSERT(reduction_order <= 1 && row_load_policy <= 1);
  if (reduction_order >= 0) options.reduction_order = static_cast<cuembed::ReductionOrder>(reduction_order);
  if (row_load_policy >= 0) options.row_loads = static_cast<cuembed::RowLoadPolicy>(row_load_policy);
  CUEMBED_ASSERT(mode == CUEMBED_SUM || mode == CUEMBED_MEAN || mode == CUEMBED_CONCAT);
  const CombineMode m = mode == CUEMBED_SUM    ? CombineMode::kSum
                        : mode == CUEMBED_MEAN ? CombineMode::kMean
                                               : CombineMode::kConcat;
  const ElemT* p = static_cast<const ElemT*>(params);
  const ElemT* w = static_cast<const ElemT*>(weights);
  ElemT* r = static_cast<ElemT*>(ret);
  if (fp16_math)
    cuembed::EmbeddingForward<ElemT, ElemT, IndexT, OffsetT, true>(
        p, embed_width, indices, offsets, w, batch_size, num_hots, m, r, Stream(stream), options);
  else
    cuembed::EmbeddingForward<ElemT, ElemT, IndexT, OffsetT, false>(
        p, embed_width, indices, offsets, w, batch_size, num_hots, m, r, Stream(stream), options);
}
}  // namespace

extern "C" {

#define CUEMBED_DEFINE_FORWARD(SUFFIX, CELEM, ELEM, INDEX, OFFSET)                          \
  void cuembed_embedding_forward_##SUFFIX(                                                  \
      const CELEM* params, int embed_width, const INDEX* indices, const OFFSET* offsets,    \
      const CELEM* weights, int batch_size, int num_hots, int mode, int fp16_math,          \
      CELEM* ret, cuembed_stream_t stream) {                                                \
    Forward<ELEM, INDEX, OFFSET>(params, embed_width, indices, offsets, weights, batch_size, \
                                 num_hots, mode, fp16_math, ret, stream);                   \
  }
CUEMBED_DEFINE_FORWARD(f32_i32_o32, float, float, int32_t, int32_t)
CUEMBED_DEFINE_FORWARD(f32_i32_o64, float, float, int32_t, int64_t)
CUEMBED_DEFINE_FORWARD(f32_i64_o32, float, float, int64_t, int32_t)
CUEMBED_DEFINE_FORWARD(f32_i64_o64, float, float, int64_t, int64_t)
CUEMBED_DEFINE_FORWARD(f16_i32_o32, void, __half, int32_t, int32_t)
CUEMBED_DEFINE_FORWARD(f16_i32_o64, void, __half, int32_t, int64_t)
CUEMBED_DEFINE_FORWARD(f16_i64_o32, void, __half, int64_t, int32_t)
CUEMBED_DEFINE_FORWARD(f16_i64_o64, void, __half, int64_t, int64_t)
CUEMBED_DEFINE_FORWARD(bf16_i32_o32, void, __hip_bfloat16, int32_t, int32_t)
CUEMBED_DEFINE_FORWARD(bf16_i32_o64, void, __hip_bfloat16, int32_t, int64_t)
CUEMBED_DEFINE_FORWARD(bf16_i64_o32, void, __hip_bfloat16, int64_t, int32_t)
CUEMBED_DEFINE_FORWARD(bf16_i64_o64, void, __hip_bfloat16, int64_t, int64_t)
#undef CUEMBED_DEFINE_FORWARD

void cuembed_embedding_forward(const void* params, int elem_type, int embed_width,
                               const void* indices, int index_type, const void* offsets,
                               int offset_type, const void* weights, int batch_size,
                               int num_hots, int mode, int fp16_math, void* ret,
                               cuembed_stream_t stream) {
  cuembed_embedding_forward_with_options(params, elem_type, embed_width, indices, index_type, offsets, offset_type,
                                         weights, batch_size, num_hots, mode, fp16_math, ret, -1, -1, stream);
}

void cuembed_embedding_forward_with_options(const void* params, int elem_type, int embed_width,
                                            const void* indices, int index_type, const void* offsets,
                                            int offset_type, const void* weights, int batch_size,
                                            int num_hots, int mode, int fp16_math, void* ret,
                                            int reduction_order, int row_load_policy,
                                            cuembed_stream_t stream) {
  cuembed_embedding_forward_ordered(params, elem_type, embed_width, indices, index_type, offsets, offset_type, weights,
                                    batch_size, num_hots, mode, fp16_math, ret, reduction_order, row_load_policy,
                                    nullptr, stream);
}

void cuembed_embedding_forward_ordered(const void* params, int elem_type, int embed_width,
                                       const void* indices, int index_type, const void* offsets,
                                       int offset_type, const void* weights, int batch_size,
                                       int num_hots, int mode, int fp16_math, void* ret,
                                       int reduction_order, int row_load_policy,
                                       const int32_t* sample_order, cuembed_stream_t stream) {
  cuembed_embedding_forward_device_hints(params, elem_type, embed_width, indices, index_type, offsets, offset_type,
                                         weights, batch_size, num_hots, mode, fp16_math, ret, reduction_order,
                                         row_load_policy, sample_order, nullptr, stream);
}

void cuembed_decide_row_loads(const void* indices, int index_type, int64_t nnz, int64_t table_bytes,
                              uint32_t* decision, unsigned distinct_per_65536, cuembed_stream_t stream) {
  const unsigned th = distinct_per_65536 != 0u ? distinct_per_65536 : cuembed::kStreamingDistinctPer65536;
  if (index_type == CUEMBED_I32)
    cuembed::DecideRowLoads<int32_t>(static_cast<const int32_t*>(indices), nnz, table_bytes, decision, Stream(stream), th);
  else if (index_type == CUEMBED_I64)
    cuembed::DecideRowLoads<int64_t>(static_cast<const int64_t*>(indices), nnz, table_bytes, decision, Stream(stream), th);
  else
    CUEMBED_C_API_BAD_TYPE();
}

void cuembed_embedding_forward_device_hints(const void* params, int elem_type, int embed_width,
                                            const void* indices, int index_type, const void* offsets,
                                            int offset_type, const void* weights, int batch_size,
                                            int num_hots, int mode, int fp16_math, void* ret,
                                            int reduction_order, int row_load_policy,
                                            const int32_t* sample_order, const uint32_t* row_loads_device,
                                            cuembed_stream_t stream) {
#define FWD(E, I, O)                                                                          \
  Forward<E, I, O>(params, embed_width, static_cast<const I*>(indices),                       \
                   static_cast<const O*>(offsets), weights, batch_size, num_hots, mode,       \
                   fp16_math, ret, stream, reduction_order, row_load_policy, sample_order, row_loads_device)
  const int key = (elem_type << 2) | (index_type << 1) | (offsets ? offset_type : 0);
  switch (key) {
    case 0: FWD(float, int32_t, int32_t); break;
    case 1: FWD(float, int32_t, int64_t); break;
    case 2: FWD(float, int64_t, int32_t); break;
    case 3: FWD(float, int64_t, int64_t); break;
    case 4: FWD(__half, int32_t, int32_t); break;
    case 5: FWD(__half, int32_t, int64_t); break;
    case 6: FWD(__half, int64_t, int32_t); break;
    case 7: FWD(__half, int64_t, int64_t); break;
    case 8: FWD(__hip_bfloat16, int32_t, int32_t); break;
    case 9: FWD(__hip_bfloat16, int32_t, int64_t); break;
    case 10: FWD(__hip_bfloat16, int64_t, int32_t); break;
    case 11: FWD(__hip_bfloat16, int64_t, int64_t); break;
    default: CUEMBED_C_API_BAD_TYPE();
  }
#undef FWD
}

void cuembed_forward_launch_shape(int elem_type, int index_type, int embed_width, int batch_size,
                                  int num_hots, int is_csr, int is_weighted, int mode,
                                  int* out) {
  cuembed::detail::ForwardLaunch f;
  const bool concat = mode == CUEMBED_CONCAT;
  if (elem_type == CUEMBED_F32) {
    f = index_type == CUEMBED_I32
            ? cuembed::detail::PlanForward<float, int32_t>(embed_width, nullptr, nullptr, batch_size,
                                                           num_hots, is_csr, is_weighted, concat)
            : cuembed::detail::PlanForward<float, int64_t>(embed_width, nullptr, nullptr, batch_size,
                                                           num_hots, is_csr, is_weighted, concat);
  } else {  // 2-byte elements (fp16 and bf16 plan identically)
    f = index_type == CUEMBED_I32
            ? cuembed::detail::PlanForward<_Float16, int32_t>(embed_width, nullptr, nullptr,
                                                              batch_size, num_hots, is_csr,
                                                              is_weighted, concat)
            : cuembed::detail::PlanForward<_Float16, int64_t>(embed_width, nullptr, nullptr,
                                                              batch_size, num_hots, is_csr,
                                                              is_weighted, concat);
  }
  out[0] = f.split.elems_per_lane;
  out[1] = f.split.lanes_per_row;
  out[2] = f.split.rows_per_block;
  out[3] = static_cast<int>(f.grid);
  out[4] = static_cast<int>(f.stage_bytes);
  out[5] = f.staged ? 1 : 0;
  // small batches of sum / mean lookups: the wide-load kernel (one sample per workgroup, rows parked in LDS)
  const size_t elem_bytes = elem_type == CUEMBED_F32 ? 4 : 2;
  const size_t row_bytes = static_cast<size_t>(embed_width) * elem_bytes;
  const int wide = (!concat && cuembed::GetForwardReductionOrder() == cuembed::ReductionOrder::kSequential)
                       ? cuembed::detail::ForwardWideLoadSamples(f.split.lanes_per_row, row_bytes, batch_size, num_hots,
                                                                 is_csr != 0)
                       : 0;
  if (wide > 0) {
    const size_t chunk = static_cast<size_t>(cuembed::detail::kForwardUnroll) *
                         (cuembed::detail::kWideLoadThreads / (f.split.lanes_per_row * wide));
    out[2] = wide;
    out[3] = (batch_size + wide - 1) / wide;
    out[4] = static_cast<int>(wide * (chunk * row_bytes + (is_weighted ? chunk * elem_bytes : 0)));
    out[5] = 2;
  }
}

void cuembed_embedding_weight_grad(const void* params, int elem_type, int embed_width,
                                   const void* indices, int index_type, const void* offsets,
                                   int offset_type, const void* grad_y, int batch_size, int num_hots,
                                   void* grad_weights, cuembed_stream_t stream) {
#define WG(E, I, O)                                                                              \
  cuembed::EmbeddingWeightGrad<E, I, O>(static_cast<const E*>(params), embed_width,              \
                                        static_cast<const I*>(indices),                          \
                                        static_cast<const O*>(offsets),                          \
                                        static_cast<const E*>(grad_y), batch_size, num_hots,     \
                                        static_cast<E*>(grad_weights), Stream(stream))
  const int key = (elem_type << 2) | (index_type << 1) | (offsets ? offset_type : 0);
  switch (key) {
    case 0: WG(float, int32_t, int32_t); break;
    case 1: WG(float, int32_t, int64_t); break;
    case 2: WG(float, int64_t, int32_t); break;
    case 3: WG(float, int64_t, int64_t); break;
    case 4: WG(__half, int32_t, int32_t); break;
    case 5: WG(__half, int32_t, int64_t); break;
    case 6: WG(__half, int64_t, int32_t); break;
    case 7: WG(__half, int64_t, int64_t); break;
    case 8: WG(__hip_bfloat16, int32_t, int32_t); break;
    case 9: WG(__hip_bfloat16, int32_t, int64_t); break;
    case 10: WG(__hip_bfloat16, int64_t, int32_t); break;
    case 11: WG(__hip_bfloat16, int64_t, int64_t); break;
    default: CUEMBED_C_API_BAD_TYPE();
  }
#undef WG
}

void cuembed_set_forward_reduction_order(int order) {
  CUEMBED_ASSERT(order == 0 || order == 1);
  cuembed::SetForwardReductionOrder(static_cast<cuembed::ReductionOrder>(order));
}
int cuembed_get_forward_reduction_order(void) {
  return static_cast<int>(cuembed::GetForwardReductionOrder());
}
void cuembed_set_forward_row_load_policy(int policy) {
  CUEMBED_ASSERT(policy == 0 || policy == 1);
  cuembed::SetForwardRowLoadPolicy(static_cast<cuembed::RowLoadPolicy>(policy));
}
int cuembed_get_forward_row_load_policy(void) {
  return static_cast<int>(cuembed::GetForwardRowLoadPolicy());
}
void cuembed_set_forward_wide_load(int mode) {
  CUEMBED_ASSERT(mode >= 0 && mode <= 8);
  cuembed::SetForwardWideLoad(mode);
}

int cuembed_peek_last_error(void) { return static_cast<int>(hipPeekAtLastError()); }

const char* cuembed_version(void) { return "cuembed_amd 0.1.0 gfx950"; }

}  // extern "C"
