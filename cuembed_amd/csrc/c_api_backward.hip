// C ABI: backward entry points = explicit instantiations of
// cuembed::EmbeddingBackward (reference instantiation list:
// utils/src/embedding_gpu_backward.cu:84-87).
#include "c_api_common.hpp"
#include "cuembed/include/embedding_lookup.hpp"

using cuembed_c_api::Stream;

namespace {
template <typename ElemT, typename IndexT>
void Backward(const void* grad_y, int embed_width, int num_rows, int nnz, const IndexT* t_idx,
              const IndexT* t_sid, const IndexT* t_remap, const void* t_w, int skip_init,
              void* grad, IndexT* inverse_mapping, cuembed_stream_t stream, int sample_blocks = 1,
              const uint32_t* block_row_ids = nullptr, int capacity_rows = 0, uint32_t* capacity_overflow = nullptr,
              int pad_to_capacity = 0) {
  cuembed::EmbeddingBackward<ElemT, IndexT>(
      static_cast<const ElemT*>(grad_y), embed_width, num_rows, nnz, t_idx, t_sid, t_remap,
      static_cast<const ElemT*>(t_w), skip_init != 0, static_cast<ElemT*>(grad), inverse_mapping,
      Stream(stream), sample_blocks, block_row_ids, capacity_rows, capacity_overflow, pad_to_capacity != 0);
}
}  // namespace

namespace {
cuembed::detail::DeviceShape ShapeFrom(int compute_units, int xcds, int64_t l2_bytes_per_xcd) {
  cuembed::detail::DeviceShape dev =
      compute_units > 0 ? cuembed::detail::Mi355xShape() : cuembed::detail::CurrentDeviceShape();
  if (compute_units > 0) dev.compute_units = compute_units;
  if (xcds > 0) dev.xcds = xcds;
  if (l2_bytes_per_xcd > 0) dev.l2_bytes_per_xcd = static_cast<size_t>(l2_bytes_per_xcd);
  return dev;
}

template <typename ElemT, int N>
cuembed::detail::ScatterShape PlanFor(int index_type, int width, int64_t nnz, bool weighted,
                                      const cuembed::detail::RowSplit& split,
                                      const cuembed::detail::DeviceShape& dev) {
  return index_type == CUEMBED_I32
             ? cuembed::detail::PlanScatter<ElemT, int32_t, N>(width, nnz, split, weighted, dev)
             : cuembed::detail::PlanScatter<ElemT, int64_t, N>(width, nnz, split, weighted, dev);
}
}  // namespace

extern "C" {

#define CUEMBED_DEFINE_BACKWARD(SUFFIX, CELEM, ELEM, INDEX)                                  \
  void cuembed_embedding_backward_##SUFFIX(                                                  \
      const CELEM* grad_y, int embed_width, int num_grad_embedding_rows, int nnz,            \
      const INDEX* transpose_indices, const INDEX* transpose_sample_ids,                     \
      const INDEX* transpose_remapped_indices, const CELEM* transpose_weights,               \
      int skip_grad_init, CELEM* grad_embedding, INDEX* inverse_mapping,                     \
      cuembed_stream_t stream) {                                                             \
    Backward<ELEM, INDEX>(grad_y, embed_width, num_grad_embedding_rows, nnz,                 \
                          transpose_indices, transpose_sample_ids,                           \
                          transpose_remapped_indices, transpose_weights, skip_grad_init,     \
                          grad_embedding, inverse_mapping, stream);                          \
  }
CUEMBED_DEFINE_BACKWARD(f32_i32, float, float, int32_t)
CUEMBED_DEFINE_BACKWARD(f32_i64, float, float, int64_t)
CUEMBED_DEFINE_BACKWARD(f16_i32, void, __half, int32_t)
CUEMBED_DEFINE_BACKWARD(f16_i64, void, __half, int64_t)
CUEMBED_DEFINE_BACKWARD(bf16_i32, void, __hip_bfloat16, int32_t)
CUEMBED_DEFINE_BACKWARD(bf16_i64, void, __hip_bfloat16, int64_t)
#undef CUEMBED_DEFINE_BACKWARD

void cuembed_embedding_backward(const void* grad_y, int elem_type, int embed_width,
                                int num_grad_embedding_rows, int nnz,
                                const void* transpose_indices, const void* transpose_sample_ids,
                                const void* transpose_remapped_indices, int index_type,
                                const void* transpose_weights, int skip_grad_init,
                                void* grad_embedding, void* inverse_mapping,
                                cuembed_stream_t stream) {
  cuembed_embedding_backward_blocked(grad_y, elem_type, embed_width, num_grad_embedding_rows, nnz,
                                     transpose_indices, transpose_sample_ids, transpose_remapped_indices,
                                     index_type, transpose_weights, skip_grad_init, grad_embedding,
                                     inverse_mapping, 1, nullptr, stream);
}

void cuembed_embedding_backward_blocked(const void* grad_y, int elem_type, int embed_width,
                                        int num_grad_embedding_rows, int nnz,
                                        const void* transpose_indices, const void* transpose_sample_ids,
                                        const void* transpose_remapped_indices, int index_type,
                                        const void* transpose_weights, int skip_grad_init,
                                        void* grad_embedding, void* inverse_mapping, int sample_blocks,
                                        const uint32_t* block_row_ids, cuembed_stream_t stream) {
  cuembed_embedding_backward_bounded(grad_y, elem_type, embed_width, num_grad_embedding_rows, nnz, transpose_indices,
                                     transpose_sample_ids, transpose_remapped_indices, index_type, transpose_weights,
                                     skip_grad_init, grad_embedding, inverse_mapping, sample_blocks, block_row_ids,
                                     0, nullptr, 0, stream);
}

void cuembed_embedding_backward_bounded(const void* grad_y, int elem_type, int embed_width,
                                        int num_grad_embedding_rows, int nnz,
                                        const void* transpose_indices, const void* transpose_sample_ids,
                                        const void* transpose_remapped_indices, int index_type,
                                        const void* transpose_weights, int skip_grad_init,
                                        void* grad_embedding, void* inverse_mapping, int sample_blocks,
                                        const uint32_t* block_row_ids, int capacity_rows,
                                        uint32_t* capacity_overflow, int pad_to_capacity, cuembed_stream_t stream) {
#define BWD(E, I)                                                                             \
  Backward<E, I>(grad_y, embed_width, num_grad_embedding_rows, nnz,                           \
                 static_cast<const I*>(transpose_indices),                                    \
                 static_cast<const I*>(transpose_sample_ids),                                 \
                 static_cast<const I*>(transpose_remapped_indices), transpose_weights,        \
                 skip_grad_init, grad_embedding, static_cast<I*>(inverse_mapping), stream, sample_blocks, \
                 block_row_ids, capacity_rows, capacity_overflow, pad_to_capacity)
  switch ((elem_type << 1) | index_type) {
    case 0: BWD(float, int32_t); break;
    case 1: BWD(float, int64_t); break;
    case 2: BWD(__half, int32_t); break;
    case 3: BWD(__half, int64_t); break;
    case 4: BWD(__hip_bfloat16, int32_t); break;
    case 5: BWD(__hip_bfloat16, int64_t); break;
    default: CUEMBED_C_API_BAD_TYPE();
  }
#undef BWD
}

void cuembed_embedding_backward_reference_sums(const void* grad_y, int elem_type, int embed_width,
                                               int num_grad_embedding_rows, int nnz,
                                               const void* transpose_indices, const void* transpose_sample_ids,
                                               const void* transpose_remapped_indices, int index_type,
                                               const void* transpose_weights, int skip_grad_init,
                                               void* grad_embedding, void* inverse_mapping,
                                               cuembed_stream_t stream) {
#define BWDR(E, I)                                                                                     \
  cuembed::EmbeddingBackwardReferenceSums<E, I>(                                                       \
      static_cast<const E*>(grad_y), embed_width, num_grad_embedding_rows, nnz,                        \
      static_cast<const I*>(transpose_indices), static_cast<const I*>(transpose_sample_ids),           \
      static_cast<const I*>(transpose_remapped_indices), static_cast<const E*>(transpose_weights),     \
      skip_grad_init != 0, static_cast<E*>(grad_embedding), static_cast<I*>(inverse_mapping), Stream(stream))
  switch ((elem_type << 1) | index_type) {
    case 0: BWDR(float, int32_t); break;
    case 1: BWDR(float, int64_t); break;
    case 2: BWDR(__half, int32_t); break;
    case 3: BWDR(__half, int64_t); break;
    case 4: BWDR(__hip_bfloat16, int32_t); break;
    case 5: BWDR(__hip_bfloat16, int64_t); break;
    default: CUEMBED_C_API_BAD_TYPE();
  }
#undef BWDR
}

int cuembed_recommended_sample_blocks(int elem_type, int embed_width, int batch_size, int64_t nnz) {
  switch (elem_type) {
    case 0: return cuembed::RecommendedSampleBlocks<float>(embed_width, batch_size, nnz);
    case 1: return cuembed::RecommendedSampleBlocks<__half>(embed_width, batch_size, nnz);
    case 2: return cuembed::RecommendedSampleBlocks<__hip_bfloat16>(embed_width, batch_size, nnz);
    default: CUEMBED_C_API_BAD_TYPE();
  }
  return 1;
}

void cuembed_device_shape(int* out) {
  const cuembed::detail::DeviceShape dev = cuembed::detail::CurrentDeviceShape();
  out[0] = dev.compute_units;
  out[1] = dev.xcds;
  out[2] = dev.lanes_per_cu;
  out[3] = static_cast<int>(dev.l2_bytes_per_xcd);
}

void cuembed_backward_launch_shape(int elem_type, int index_type, int embed_width, int64_t nnz, int is_weighted,
                                   int compute_units, int xcds, int* out) {
  const cuembed::detail::DeviceShape dev = ShapeFrom(compute_units, xcds, 0);
  cuembed::detail::ScatterShape s;
  if (elem_type == CUEMBED_F32) {
    const auto split = cuembed::detail::SplitRow<float>(embed_width, nullptr, nullptr);
    s = split.elems_per_lane == 4   ? PlanFor<float, 4>(index_type, embed_width, nnz, is_weighted != 0, split, dev)
        : split.elems_per_lane == 2 ? PlanFor<float, 2>(index_type, embed_width, nnz, is_weighted != 0, split, dev)
                                    : PlanFor<float, 1>(index_type, embed_width, nnz, is_weighted != 0, split, dev);
  } else {  // 2-byte elements (fp16 and bf16 plan identically)
    const auto split = cuembed::detail::SplitRow<_Float16>(embed_width, nullptr, nullptr);
    s = split.elems_per_lane == 8   ? PlanFor<_Float16, 8>(index_type, embed_width, nnz, is_weighted != 0, split, dev)
        : split.elems_per_lane == 4 ? PlanFor<_Float16, 4>(index_type, embed_width, nnz, is_weighted != 0, split, dev)
                                    : PlanFor<_Float16, 2>(index_type, embed_width, nnz, is_weighted != 0, split, dev);
  }
  out[0] = s.slices;
  out[1] = s.lanes;
  out[2] = s.segments_per_block;
  out[3] = s.segment_len;
  out[4] = static_cast<int>(s.nz_blocks);
  out[5] = static_cast<int>(s.grid_blocks);
  out[6] = static_cast<int>(s.lds);
  out[7] = s.xcds;
}

int cuembed_recommended_sample_blocks_on(int elem_type, int embed_width, int batch_size, int64_t nnz,
                                         int compute_units, int xcds, int64_t l2_bytes_per_xcd) {
  const cuembed::detail::DeviceShape dev = ShapeFrom(compute_units, xcds, l2_bytes_per_xcd);
  switch (elem_type) {
    case 0: return cuembed::RecommendedSampleBlocks<float>(embed_width, batch_size, nnz, dev);
    case 1: return cuembed::RecommendedSampleBlocks<__half>(embed_width, batch_size, nnz, dev);
    case 2: return cuembed::RecommendedSampleBlocks<__hip_bfloat16>(embed_width, batch_size, nnz, dev);
    default: CUEMBED_C_API_BAD_TYPE();
  }
  return 1;
}

void cuembed_set_backward_tuning(int segment_len, int column_slices) {
  cuembed::SetBackwardTuning(cuembed::BackwardTuning{segment_len, column_slices});
}

void cuembed_get_backward_tuning(int* out) {
  const cuembed::BackwardTuning t = cuembed::GetBackwardTuning();
  out[0] = t.segment_len;
  out[1] = t.column_slices;
}

}  // extern "C"
