// C ABI: the device-side halves of the sparse gradient exchange = explicit instantiations of
// cuembed::PackRowsByOwner / FinishOwnerPiece (extensions; the reference is single-GPU, README.md:108-119).
#include "c_api_common.hpp"
#include "cuembed/include/exchange_transforms.hpp"

using cuembed_c_api::Stream;

namespace {
template <typename IndexT, typename ElemT>
void Pack(const void* ids, const void* rows, int64_t num_rows, int embed_width, const void* count, const int64_t* cuts,
          int world, int64_t slot_capacity, int64_t input_capacity, int64_t num_categories, int64_t* send_ids,
          void* send_rows, int64_t* range_starts, int64_t* flag, cuembed_stream_t stream) {
  cuembed::PackRowsByOwner<IndexT, ElemT>(static_cast<const IndexT*>(ids), static_cast<const ElemT*>(rows), num_rows,
                                          embed_width, static_cast<const IndexT*>(count), cuts, world, slot_capacity,
                                          input_capacity, num_categories, send_ids, static_cast<ElemT*>(send_rows),
                                          range_starts, flag, Stream(stream));
}
}  // namespace

extern "C" {

void cuembed_exchange_pack_rows(const void* ids, int index_type, const void* rows, int elem_type, int64_t num_rows,
                                int embed_width, const void* count, const int64_t* cuts, int world,
                                int64_t slot_capacity, int64_t input_capacity, int64_t num_categories,
                                int64_t* send_ids, void* send_rows, int64_t* range_starts, int64_t* flag,
                                cuembed_stream_t stream) {
#define CUEMBED_PACK(INDEX, ELEM)                                                                                    \
  Pack<INDEX, ELEM>(ids, rows, num_rows, embed_width, count, cuts, world, slot_capacity, input_capacity,             \
                    num_categories, send_ids, send_rows, range_starts, flag, stream)
  // (rows are only moved: one instantiation per element SIZE)
  if (index_type != CUEMBED_I32 && index_type != CUEMBED_I64) CUEMBED_C_API_BAD_TYPE();
  const bool narrow = index_type == CUEMBED_I32;
  switch (elem_type) {
    case CUEMBED_F32:
      if (narrow) CUEMBED_PACK(int32_t, float);
      else CUEMBED_PACK(int64_t, float);
      break;
    case CUEMBED_F16:
    case CUEMBED_BF16:
      if (narrow) CUEMBED_PACK(int32_t, __half);
      else CUEMBED_PACK(int64_t, __half);
      break;
    default: CUEMBED_C_API_BAD_TYPE();
  }
#undef CUEMBED_PACK
}

void cuembed_exchange_finish_piece(const int64_t* sorted_ids, const int64_t* remapped_ids, int64_t nnz, int64_t capacity,
                                   int64_t num_categories, int64_t pad_lo, int64_t pad_len, int64_t* ids, void* rows,
                                   int elem_type, int embed_width, int64_t* tail, int64_t* flag, int64_t* count,
                                   cuembed_stream_t stream) {
  // (the rows are only zeroed: all-zero bits are 0.0 in every element type)
  switch (elem_type) {
    case CUEMBED_F32:
      cuembed::FinishOwnerPiece<float>(sorted_ids, remapped_ids, nnz, capacity, num_categories, pad_lo, pad_len, ids,
                                       static_cast<float*>(rows), embed_width, tail, flag, count, Stream(stream));
      break;
    case CUEMBED_F16:
    case CUEMBED_BF16:
      cuembed::FinishOwnerPiece<__half>(sorted_ids, remapped_ids, nnz, capacity, num_categories, pad_lo, pad_len, ids,
                                        static_cast<__half*>(rows), embed_width, tail, flag, count, Stream(stream));
      break;
    default: CUEMBED_C_API_BAD_TYPE();
  }
}

}  // extern "C"
