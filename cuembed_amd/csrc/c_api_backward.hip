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
              const uint32_t* block_row_ids = nullptr) {
  cuembed::EmbeddingBackward<ElemT, IndexT>(
      static_cast<const ElemT*>(grad_y), embed_width, num_rows, nnz, t_idx, t_sid, t_remap,
      static_cast<const ElemT*>(t_w), skip_init != 0, static_cast<ElemT*>(grad), inverse_mapping,
      Stream(stream), sample_blocks, block_row_ids);
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
#define BWD(E, I)                                                                             \
  Backward<E, I>(grad_y, embed_width, num_grad_embedding_rows, nnz,                           \
                 static_cast<const I*>(transpose_indices),                                    \
                 static_cast<const I*>(transpose_sample_ids),                                 \
                 static_cast<const I*>(transpose_remapped_indices), transpose_weights,        \
                 skip_grad_init, grad_embedding, static_cast<I*>(inverse_mapping), stream, sample_blocks, \
                 block_row_ids)
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

int cuembed_recommended_sample_blocks(int elem_type, int embed_width, int batch_size, int64_t nnz) {
  switch (elem_type) {
    case 0: return cuembed::RecommendedSampleBlocks<float>(embed_width, batch_size, nnz);
    case 1: return cuembed::RecommendedSampleBlocks<__half>(embed_width, batch_size, nnz);
    case 2: return cuembed::RecommendedSampleBlocks<__hip_bfloat16>(embed_width, batch_size, nnz);
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
