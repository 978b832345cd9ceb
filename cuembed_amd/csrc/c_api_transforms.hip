// C ABI: index transformations = explicit instantiations of cuembed::Transpose,
// ComputeCompressedGradIndices and ExtractRowIds* (reference instantiation list:
// utils/src/embedding_gpu_transpose.cu:95-98).
#include "c_api_common.hpp"
#include "cuembed/include/index_transforms.hpp"

using cuembed_c_api::Stream;

extern "C" {

#define CUEMBED_DEFINE_TRANSPOSE(SUFFIX, INDEX, CWEIGHT, WEIGHT)                              \
  void cuembed_transpose_##SUFFIX(const INDEX* rows, const INDEX* cols,                       \
                                  const CWEIGHT* weights, int nnz, INDEX* transpose_rows,     \
                                  INDEX* transpose_cols, CWEIGHT* transpose_weights,          \
                                  char* work, size_t* lwork, cuembed_stream_t stream) {       \
    cuembed::Transpose<INDEX, WEIGHT>(rows, cols, static_cast<const WEIGHT*>(weights), nnz,   \
                                      transpose_rows, transpose_cols,                         \
                                      static_cast<WEIGHT*>(transpose_weights), work, lwork,   \
                                      Stream(stream));                                        \
  }
CUEMBED_DEFINE_TRANSPOSE(i32_f32, int32_t, float, float)
CUEMBED_DEFINE_TRANSPOSE(i64_f32, int64_t, float, float)
CUEMBED_DEFINE_TRANSPOSE(i32_f16, int32_t, void, __half)
CUEMBED_DEFINE_TRANSPOSE(i64_f16, int64_t, void, __half)
CUEMBED_DEFINE_TRANSPOSE(i32_bf16, int32_t, void, __hip_bfloat16)
CUEMBED_DEFINE_TRANSPOSE(i64_bf16, int64_t, void, __hip_bfloat16)
#undef CUEMBED_DEFINE_TRANSPOSE

#define CUEMBED_DEFINE_COMPRESS(SUFFIX, INDEX)                                                \
  void cuembed_compute_compressed_grad_indices_##SUFFIX(                                      \
      const INDEX* indices, int nnz, INDEX* remapped_indices, char* work, size_t* lwork,      \
      cuembed_stream_t stream) {                                                              \
    cuembed::ComputeCompressedGradIndices<INDEX>(indices, nnz, remapped_indices, work, lwork, \
                                                 Stream(stream));                             \
  }
CUEMBED_DEFINE_COMPRESS(i32, int32_t)
CUEMBED_DEFINE_COMPRESS(i64, int64_t)
#undef CUEMBED_DEFINE_COMPRESS

#define CUEMBED_DEFINE_EXTRACT(SUFFIX, INDEX)                                                 \
  void cuembed_extract_row_ids_from_fixed_##SUFFIX(int batch_size, int num_hots,              \
                                                   INDEX* row_ids, cuembed_stream_t stream) { \
    cuembed::ExtractRowIdsFromFixed<INDEX>(batch_size, num_hots, row_ids, Stream(stream));    \
  }                                                                                           \
  void cuembed_extract_row_ids_for_concat_##SUFFIX(int nnz, INDEX* row_ids,                   \
                                                   cuembed_stream_t stream) {                 \
    cuembed::ExtractRowIdsForConcat<INDEX>(nnz, row_ids, Stream(stream));                     \
  }
CUEMBED_DEFINE_EXTRACT(i32, int32_t)
CUEMBED_DEFINE_EXTRACT(i64, int64_t)
#undef CUEMBED_DEFINE_EXTRACT

#define CUEMBED_DEFINE_EXTRACT_CSR(SUFFIX, INDEX, OFFSET)                                     \
  void cuembed_extract_row_ids_from_csr_##SUFFIX(const OFFSET* offsets, int batch_size,       \
                                                 INDEX* row_ids, cuembed_stream_t stream) {   \
    cuembed::ExtractRowIdsFromCSR<INDEX, OFFSET>(offsets, batch_size, row_ids,                \
                                                 Stream(stream));                             \
  }
CUEMBED_DEFINE_EXTRACT_CSR(i32_o32, int32_t, int32_t)
CUEMBED_DEFINE_EXTRACT_CSR(i32_o64, int32_t, int64_t)
CUEMBED_DEFINE_EXTRACT_CSR(i64_o32, int64_t, int32_t)
CUEMBED_DEFINE_EXTRACT_CSR(i64_o64, int64_t, int64_t)
#undef CUEMBED_DEFINE_EXTRACT_CSR

void cuembed_transpose(const void* rows, const void* cols, const void* weights, int nnz,
                       int index_type, int weight_type, void* transpose_rows,
                       void* transpose_cols, void* transpose_weights, char* work, size_t* lwork,
                       cuembed_stream_t stream) {
  cuembed_transpose_bounded(rows, cols, weights, nnz, index_type, weight_type, transpose_rows,
                            transpose_cols, transpose_weights, work, lwork, 0, stream);
}

void cuembed_transpose_bounded(const void* rows, const void* cols, const void* weights, int nnz,
                               int index_type, int weight_type, void* transpose_rows,
                               void* transpose_cols, void* transpose_weights, char* work,
                               size_t* lwork, int index_bits, cuembed_stream_t stream) {
  cuembed_transpose_hinted(rows, cols, weights, nnz, index_type, weight_type, transpose_rows,
                           transpose_cols, transpose_weights, work, lwork, index_bits, 0, stream);
}

void cuembed_transpose_hinted(const void* rows, const void* cols, const void* weights, int nnz,
                              int index_type, int weight_type, void* transpose_rows,
                              void* transpose_cols, void* transpose_weights, char* work,
                              size_t* lwork, int index_bits, int row_bits, cuembed_stream_t stream) {
#define TR(I, W)                                                                              \
  cuembed::Transpose<I, W>(static_cast<const I*>(rows), static_cast<const I*>(cols),          \
                           static_cast<const W*>(weights), nnz, static_cast<I*>(transpose_rows), \
                           static_cast<I*>(transpose_cols), static_cast<W*>(transpose_weights), \
                           work, lwork, Stream(stream), index_bits, row_bits)
  // weights are only moved, never computed on: fp16 and bf16 share the 2-byte instantiation
  switch ((index_type << 1) | (weight_type != CUEMBED_F32 ? 1 : 0)) {
    case 0: TR(int32_t, float); break;
    case 1: TR(int32_t, __half); break;
    case 2: TR(int64_t, float); break;
    case 3: TR(int64_t, __half); break;
    default: CUEMBED_C_API_BAD_TYPE();
  }
#undef TR
}

void cuembed_transpose_fixed_hotness(const void* indices, const void* weights, int batch_size,
                                     int num_hots, int index_type, int weight_type,
                                     void* transpose_indices, void* transpose_sample_ids,
                                     void* transpose_weights, char* work, size_t* lwork, int index_bits,
                                     cuembed_stream_t stream) {
#define TR(I, W)                                                                              \
  cuembed::TransposeFixedHotness<I, W>(static_cast<const I*>(indices), static_cast<const W*>(weights), \
                                       batch_size, num_hots, static_cast<I*>(transpose_indices), \
                                       static_cast<I*>(transpose_sample_ids),                  \
                                       static_cast<W*>(transpose_weights), work, lwork, Stream(stream), index_bits)
  switch ((index_type << 1) | (weight_type != CUEMBED_F32 ? 1 : 0)) {
    case 0: TR(int32_t, float); break;
    case 1: TR(int32_t, __half); break;
    case 2: TR(int64_t, float); break;
    case 3: TR(int64_t, __half); break;
    default: CUEMBED_C_API_BAD_TYPE();
  }
#undef TR
}

void cuembed_translate_indices_for_row_cache(const void* indices, int index_type, int64_t nnz,
                                             const int32_t* slot_of_row, int64_t num_rows,
                                             int64_t cache_row_offset, int64_t* translated,
                                             cuembed_stream_t stream) {
  if (index_type == CUEMBED_I32)
    cuembed::TranslateIndicesForRowCache<int32_t>(static_cast<const int32_t*>(indices), nnz, slot_of_row,
                                                  num_rows, cache_row_offset, translated, Stream(stream));
  else if (index_type == CUEMBED_I64)
    cuembed::TranslateIndicesForRowCache<int64_t>(static_cast<const int64_t*>(indices), nnz, slot_of_row,
                                                  num_rows, cache_row_offset, translated, Stream(stream));
  else
    CUEMBED_C_API_BAD_TYPE();
}

void cuembed_compute_compressed_grad_indices(const void* indices, int nnz, int index_type,
                                             void* remapped_indices, char* work, size_t* lwork,
                                             cuembed_stream_t stream) {
  if (index_type == CUEMBED_I32)
    cuembed::ComputeCompressedGradIndices<int32_t>(static_cast<const int32_t*>(indices), nnz,
                                                   static_cast<int32_t*>(remapped_indices), work,
                                                   lwork, Stream(stream));
  else if (index_type == CUEMBED_I64)
    cuembed::ComputeCompressedGradIndices<int64_t>(static_cast<const int64_t*>(indices), nnz,
                                                   static_cast<int64_t*>(remapped_indices), work,
                                                   lwork, Stream(stream));
  else
    CUEMBED_C_API_BAD_TYPE();
}

void cuembed_compute_compressed_grad_indices_blocked(const void* indices, int nnz, int index_type,
                                                     int sample_blocks, void* remapped_indices,
                                                     uint32_t* block_row_ids, uint32_t* num_unique,
                                                     char* work, size_t* lwork, cuembed_stream_t stream) {
  if (index_type == CUEMBED_I32)
    cuembed::ComputeCompressedGradIndicesBlocked<int32_t>(static_cast<const int32_t*>(indices), nnz, sample_blocks,
                                                          static_cast<int32_t*>(remapped_indices), block_row_ids,
                                                          num_unique, work, lwork, Stream(stream));
  else if (index_type == CUEMBED_I64)
    cuembed::ComputeCompressedGradIndicesBlocked<int64_t>(static_cast<const int64_t*>(indices), nnz, sample_blocks,
                                                          static_cast<int64_t*>(remapped_indices), block_row_ids,
                                                          num_unique, work, lwork, Stream(stream));
  else
    CUEMBED_C_API_BAD_TYPE();
}

void cuembed_extract_row_ids_from_fixed(int batch_size, int num_hots, int index_type,
                                        void* row_ids, cuembed_stream_t stream) {
  if (index_type == CUEMBED_I32)
    cuembed::ExtractRowIdsFromFixed<int32_t>(batch_size, num_hots,
                                             static_cast<int32_t*>(row_ids), Stream(stream));
  else if (index_type == CUEMBED_I64)
    cuembed::ExtractRowIdsFromFixed<int64_t>(batch_size, num_hots,
                                             static_cast<int64_t*>(row_ids), Stream(stream));
  else
    CUEMBED_C_API_BAD_TYPE();
}

void cuembed_extract_row_ids_for_concat(int nnz, int index_type, void* row_ids,
                                        cuembed_stream_t stream) {
  if (index_type == CUEMBED_I32)
    cuembed::ExtractRowIdsForConcat<int32_t>(nnz, static_cast<int32_t*>(row_ids), Stream(stream));
  else if (index_type == CUEMBED_I64)
    cuembed::ExtractRowIdsForConcat<int64_t>(nnz, static_cast<int64_t*>(row_ids), Stream(stream));
  else
    CUEMBED_C_API_BAD_TYPE();
}

void cuembed_extract_row_ids_from_csr(const void* offsets, int offset_type, int batch_size,
                                      int index_type, void* row_ids, cuembed_stream_t stream) {
#define EX(I, O)                                                                              \
  cuembed::ExtractRowIdsFromCSR<I, O>(static_cast<const O*>(offsets), batch_size,             \
                                      static_cast<I*>(row_ids), Stream(stream))
  switch ((index_type << 1) | offset_type) {
    case 0: EX(int32_t, int32_t); break;
    case 1: EX(int32_t, int64_t); break;
    case 2: EX(int64_t, int32_t); break;
    case 3: EX(int64_t, int64_t); break;
    default: CUEMBED_C_API_BAD_TYPE();
  }
#undef EX
}

// Extension: Transpose in `sample_blocks` blocks of the (sample-major) input, each sorted on its own -- for the
// compressed gradient path only (cuembed::Transpose, sample_blocks).
void cuembed_transpose_sample_blocks(const void* rows, const void* cols, const void* weights, int nnz,
                                     int index_type, int weight_type, void* transpose_rows,
                                     void* transpose_cols, void* transpose_weights, char* work, size_t* lwork,
                                     int index_bits, int row_bits, int sample_blocks, cuembed_stream_t stream) {
  cuembed_transpose_remapped(rows, cols, weights, nnz, index_type, weight_type, transpose_rows, transpose_cols,
                             transpose_weights, nullptr, work, lwork, index_bits, row_bits, sample_blocks, stream);
}

// Extension: ... and ComputeCompressedGradIndices' output from the same call (cuembed::Transpose,
// transpose_remapped_indices): ONE launch for the whole index work of a batch of up to 4,096 lookups.
void cuembed_transpose_remapped(const void* rows, const void* cols, const void* weights, int nnz,
                                int index_type, int weight_type, void* transpose_rows,
                                void* transpose_cols, void* transpose_weights, void* transpose_remapped_indices,
                                char* work, size_t* lwork, int index_bits, int row_bits, int sample_blocks,
                                cuembed_stream_t stream) {
#define TRB(I, W)                                                                                      \
  cuembed::Transpose<I, W>(static_cast<const I*>(rows), static_cast<const I*>(cols),                   \
                           static_cast<const W*>(weights), nnz, static_cast<I*>(transpose_rows),       \
                           static_cast<I*>(transpose_cols), static_cast<W*>(transpose_weights), work,  \
                           lwork, Stream(stream), index_bits, row_bits, sample_blocks,                 \
                           static_cast<I*>(transpose_remapped_indices))
  switch ((index_type << 1) | (weight_type != CUEMBED_F32 ? 1 : 0)) {   // 16-bit weights move as bit patterns
    case 0: TRB(int32_t, float); break;
    case 1: TRB(int32_t, __half); break;
    case 2: TRB(int64_t, float); break;
    case 3: TRB(int64_t, __half); break;
    default: CUEMBED_C_API_BAD_TYPE();
  }
#undef TRB
}

int64_t cuembed_transpose_sample_block_length(int64_t nnz, int sample_blocks) {
  return cuembed::TransposeSampleBlockLength(nnz, sample_blocks);
}

void cuembed_transpose_fixed_hotness_sample_blocks(const void* indices, const void* weights, int batch_size,
                                                   int num_hots, int index_type, int weight_type,
                                                   void* transpose_indices, void* transpose_sample_ids,
                                                   void* transpose_weights, char* work, size_t* lwork,
                                                   int index_bits, int sample_blocks, cuembed_stream_t stream) {
  cuembed_transpose_fixed_hotness_remapped(indices, weights, batch_size, num_hots, index_type, weight_type,
                                           transpose_indices, transpose_sample_ids, transpose_weights, nullptr, work,
                                           lwork, index_bits, sample_blocks, stream);
}

void cuembed_transpose_fixed_hotness_remapped(const void* indices, const void* weights, int batch_size,
                                              int num_hots, int index_type, int weight_type,
                                              void* transpose_indices, void* transpose_sample_ids,
                                              void* transpose_weights, void* transpose_remapped_indices, char* work,
                                              size_t* lwork, int index_bits, int sample_blocks,
                                              cuembed_stream_t stream) {
#define TFB(I, W)                                                                                               \
  cuembed::TransposeFixedHotness<I, W>(static_cast<const I*>(indices), static_cast<const W*>(weights), batch_size, \
                                       num_hots, static_cast<I*>(transpose_indices),                             \
                                       static_cast<I*>(transpose_sample_ids), static_cast<W*>(transpose_weights), \
                                       work, lwork, Stream(stream), index_bits, sample_blocks,                   \
                                       static_cast<I*>(transpose_remapped_indices))
  switch ((index_type << 1) | (weight_type != CUEMBED_F32 ? 1 : 0)) {   // 16-bit weights move as bit patterns
    case 0: TFB(int32_t, float); break;
    case 1: TFB(int32_t, __half); break;
    case 2: TFB(int64_t, float); break;
    case 3: TFB(int64_t, __half); break;
    default: CUEMBED_C_API_BAD_TYPE();
  }
#undef TFB
}

void cuembed_bag_order_by_length(const void* offsets, int offset_type, int batch_size, int max_length,
                                 int32_t* sample_order, char* work, size_t* lwork, cuembed_stream_t stream) {
  if (offset_type == CUEMBED_I32)
    cuembed::BagOrderByLength<int32_t>(static_cast<const int32_t*>(offsets), batch_size, max_length, sample_order, work,
                                       lwork, Stream(stream));
  else if (offset_type == CUEMBED_I64)
    cuembed::BagOrderByLength<int64_t>(static_cast<const int64_t*>(offsets), batch_size, max_length, sample_order, work,
                                       lwork, Stream(stream));
  else
    CUEMBED_C_API_BAD_TYPE();
}

}  // extern "C"
