/* ============================================================================
 * cuembed_amd.h -- C ABI of the MI355X-native embedding-lookup library.
 *
 * This is the drop-in boundary for foreign-function bindings (ctypes, cgo, JNI,
 * a torch extension ...).  Every entry point is an explicit instantiation of one
 * of the header-only C++ templates in cuembed_amd/csrc/cuembed/include/, which in
 * turn mirror the reference's host API one to one:
 *
 *   cuembed_embedding_forward_*              cuembed::EmbeddingForward
 *        reference: cuembed/include/embedding_lookup.cuh:245-308
 *        (instantiation list: utils/src/embedding_gpu_forward.cu:69-76,
 *         int64 offsets: examples/pytorch/cuembed_embedding.cu:39-49)
 *   cuembed_embedding_backward_*             cuembed::EmbeddingBackward
 *        reference: cuembed/include/embedding_lookup.cuh:423-483
 *        (instantiations: utils/src/embedding_gpu_backward.cu:84-87)
 *   cuembed_transpose_*                      cuembed::Transpose
 *        reference: cuembed/include/index_transforms.cuh:224-250
 *        (instantiations: utils/src/embedding_gpu_transpose.cu:95-98)
 *   cuembed_compute_compressed_grad_indices_* cuembed::ComputeCompressedGradIndices
 *        reference: cuembed/include/index_transforms.cuh:278-323
 *   cuembed_extract_row_ids_from_fixed_*     cuembed::ExtractRowIdsFromFixed
 *        reference: cuembed/include/index_transforms.cuh:45-55
 *   cuembed_extract_row_ids_from_csr_*       cuembed::ExtractRowIdsFromCSR
 *        reference: cuembed/include/index_transforms.cuh:66-74
 *   cuembed_extract_row_ids_for_concat_*     cuembed::ExtractRowIdsForConcat
 *        reference: cuembed/include/index_transforms.cuh:85-93
 *
 * Conventions (identical to the reference's):
 *   - all data pointers are DEVICE pointers owned by the caller; the library
 *     allocates nothing and keeps no state between calls;
 *   - every call only enqueues work on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream) and returns immediately;
 *   - functions return void; violating the argument contract prints
 *     "Check failed: ..." on stderr and aborts the process;
 *   - fp16 data is IEEE binary16 (`__half`), bf16 data is bfloat16 (`__hip_bfloat16`), both
 *     passed as void pointers here;
 *   - Transpose / ComputeCompressedGradIndices are two-phase: call with
 *     work == NULL to receive the scratch size in *lwork, then again with a
 *     scratch buffer of at least that size.
 *
 * Suffix grammar:  _{f32|f16|bf16}  element type of table / gradient / weights
 *                  _{i32|i64}  lookup-index and sample-id type
 *                  _{o32|o64}  CSR offset type
 * ==========================================================================*/
#ifndef CUEMBED_AMD_H_
#define CUEMBED_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* CombineMode values (reference: embedding_lookup_types.cuh:29). */
enum { CUEMBED_SUM = 0, CUEMBED_MEAN = 1, CUEMBED_CONCAT = 2 };
/* type codes for the generic (runtime-dispatched) entry points */
enum { CUEMBED_F32 = 0, CUEMBED_F16 = 1, CUEMBED_BF16 = 2 /* extension */ };
enum { CUEMBED_I32 = 0, CUEMBED_I64 = 1 };

typedef void* cuembed_stream_t; /* hipStream_t */

/* ---- forward ------------------------------------------------------------ */
/* fixed hotness: offsets == NULL, num_hots > 0; CSR: offsets[batch_size + 1],
 * num_hots == 0.  weights may be NULL.  ret: [batch x width] for sum/mean,
 * [batch x num_hots x width] for concat (fixed hotness, unweighted only).
 * fp16_math != 0 accumulates an fp16 table in fp16 (ignored for fp32). */
#define CUEMBED_DECLARE_FORWARD(SUFFIX, ELEM, INDEX, OFFSET)                           \
  void cuembed_embedding_forward_##SUFFIX(                                             \
      const ELEM* params, int embed_width, const INDEX* indices, const OFFSET* offsets, \
      const ELEM* weights, int batch_size, int num_hots, int mode, int fp16_math,      \
      ELEM* ret, cuembed_stream_t stream);
CUEMBED_DECLARE_FORWARD(f32_i32_o32, float, int32_t, int32_t)
CUEMBED_DECLARE_FORWARD(f32_i32_o64, float, int32_t, int64_t)
CUEMBED_DECLARE_FORWARD(f32_i64_o32, float, int64_t, int32_t)
CUEMBED_DECLARE_FORWARD(f32_i64_o64, float, int64_t, int64_t)
CUEMBED_DECLARE_FORWARD(f16_i32_o32, void, int32_t, int32_t)
CUEMBED_DECLARE_FORWARD(f16_i32_o64, void, int32_t, int64_t)
CUEMBED_DECLARE_FORWARD(f16_i64_o32, void, int64_t, int32_t)
CUEMBED_DECLARE_FORWARD(f16_i64_o64, void, int64_t, int64_t)
/* bf16 tables (extension; fp32 accumulation, fp16_math ignored) */
CUEMBED_DECLARE_FORWARD(bf16_i32_o32, void, int32_t, int32_t)
CUEMBED_DECLARE_FORWARD(bf16_i32_o64, void, int32_t, int64_t)
CUEMBED_DECLARE_FORWARD(bf16_i64_o32, void, int64_t, int32_t)
CUEMBED_DECLARE_FORWARD(bf16_i64_o64, void, int64_t, int64_t)
#undef CUEMBED_DECLARE_FORWARD

/* ---- backward ----------------------------------------------------------- */
/* COO lookups sorted by index (output of transpose).  Full gradient:
 * transpose_remapped_indices == NULL, grad_embedding has num_grad_embedding_rows
 * = table rows.  Compressed: remapped indices given, grad_embedding has
 * num_unique rows and inverse_mapping[num_unique] is written.  The gradient
 * buffer must be zero before the scatter: skip_grad_init != 0 means the caller
 * already zeroed it, otherwise the call zeroes it first.
 *
 * ARITHMETIC (deviation from the reference for 16-bit gradients).  The reference
 * accumulates in GradT: with fp16 every product grad_y * weight and every partial sum
 * is rounded to fp16 (embedding_lookup_ops.cuh:636-645, embedding_lookup_cpu.hpp:139-142).
 * Here the products and the partial sum of a run are kept in fp32 and rounded to GradT
 * once per flush: one flush for a run inside a workgroup; a run that crosses workgroups
 * (>= 256 lookups each) is combined with one GradT hardware atomic per workgroup.
 *   - fp32 gradients: same fp32 arithmetic in the same nz order, equal up to where a run
 *     is cut into partial sums (a few ulps of the summed magnitudes);
 *   - fp16/bf16 gradients: bit-identical to the reference whenever all partial sums are
 *     exactly representable in GradT (the reference's own test data: integer grad_y,
 *     weights 0.5/0.25); otherwise within 1e-2 (fp16) of the reference relative to
 *     sum_j |grad_y_j * w_j|, and never further from the exact sum than the reference is.
 *     Runs longer than ~2^11 lookups are where the two part for good: the reference's fp16
 *     running sum stalls, this one does not.  Measured in tests/test_gpu_backward_tolerance.py. */
#define CUEMBED_DECLARE_BACKWARD(SUFFIX, ELEM, INDEX)                                     \
  void cuembed_embedding_backward_##SUFFIX(                                               \
      const ELEM* grad_y, int embed_width, int num_grad_embedding_rows, int nnz,          \
      const INDEX* transpose_indices, const INDEX* transpose_sample_ids,                  \
      const INDEX* transpose_remapped_indices, const ELEM* transpose_weights,             \
      int skip_grad_init, ELEM* grad_embedding, INDEX* inverse_mapping,                   \
      cuembed_stream_t stream);
CUEMBED_DECLARE_BACKWARD(f32_i32, float, int32_t)
CUEMBED_DECLARE_BACKWARD(f32_i64, float, int64_t)
CUEMBED_DECLARE_BACKWARD(f16_i32, void, int32_t)
CUEMBED_DECLARE_BACKWARD(f16_i64, void, int64_t)
CUEMBED_DECLARE_BACKWARD(bf16_i32, void, int32_t)
CUEMBED_DECLARE_BACKWARD(bf16_i64, void, int64_t)
#undef CUEMBED_DECLARE_BACKWARD

/* ---- transpose ---------------------------------------------------------- */
/* Stable sort of (rows[i] [, weights[i]]) by key cols[i]; callers pass
 * rows = sample ids, cols = lookup indices.  transpose_rows receives the sorted
 * lookup indices, transpose_cols the sample ids, transpose_weights the weights
 * (untouched when weights == NULL).  The suffix names the index type and the
 * WEIGHT type.
 * Generic like the reference's (cub::DeviceRadixSort::SortPairs over all key bits,
 * index_transforms.cuh:108-136): `cols` may hold any value of the index type, keys are
 * ordered as SIGNED numbers (negative keys first); `rows` may hold any value of the
 * index type (int64 rows that are negative or >= 2^32 are carried at full width; the
 * library finds that out on the device).  nnz <= INT_MAX. */
#define CUEMBED_DECLARE_TRANSPOSE(SUFFIX, INDEX, WEIGHT)                                   \
  void cuembed_transpose_##SUFFIX(const INDEX* rows, const INDEX* cols,                    \
                                  const WEIGHT* weights, int nnz, INDEX* transpose_rows,   \
                                  INDEX* transpose_cols, WEIGHT* transpose_weights,        \
                                  char* work, size_t* lwork, cuembed_stream_t stream);
CUEMBED_DECLARE_TRANSPOSE(i32_f32, int32_t, float)
CUEMBED_DECLARE_TRANSPOSE(i64_f32, int64_t, float)
CUEMBED_DECLARE_TRANSPOSE(i32_f16, int32_t, void)
CUEMBED_DECLARE_TRANSPOSE(i64_f16, int64_t, void)
CUEMBED_DECLARE_TRANSPOSE(i32_bf16, int32_t, void)
CUEMBED_DECLARE_TRANSPOSE(i64_bf16, int64_t, void)
#undef CUEMBED_DECLARE_TRANSPOSE

/* ---- compressed-gradient remap ------------------------------------------ */
#define CUEMBED_DECLARE_COMPRESS(SUFFIX, INDEX)                                           \
  void cuembed_compute_compressed_grad_indices_##SUFFIX(                                  \
      const INDEX* indices, int nnz, INDEX* remapped_indices, char* work, size_t* lwork,  \
      cuembed_stream_t stream);
CUEMBED_DECLARE_COMPRESS(i32, int32_t)
CUEMBED_DECLARE_COMPRESS(i64, int64_t)
#undef CUEMBED_DECLARE_COMPRESS

/* ---- row-id extraction -------------------------------------------------- */
#define CUEMBED_DECLARE_EXTRACT(SUFFIX, INDEX)                                            \
  void cuembed_extract_row_ids_from_fixed_##SUFFIX(int batch_size, int num_hots,          \
                                                   INDEX* row_ids, cuembed_stream_t stream); \
  void cuembed_extract_row_ids_for_concat_##SUFFIX(int nnz, INDEX* row_ids,               \
                                                   cuembed_stream_t stream);
CUEMBED_DECLARE_EXTRACT(i32, int32_t)
CUEMBED_DECLARE_EXTRACT(i64, int64_t)
#undef CUEMBED_DECLARE_EXTRACT

#define CUEMBED_DECLARE_EXTRACT_CSR(SUFFIX, INDEX, OFFSET)                                \
  void cuembed_extract_row_ids_from_csr_##SUFFIX(const OFFSET* offsets, int batch_size,   \
                                                 INDEX* row_ids, cuembed_stream_t stream);
CUEMBED_DECLARE_EXTRACT_CSR(i32_o32, int32_t, int32_t)
CUEMBED_DECLARE_EXTRACT_CSR(i32_o64, int32_t, int64_t)
CUEMBED_DECLARE_EXTRACT_CSR(i64_o32, int64_t, int32_t)
CUEMBED_DECLARE_EXTRACT_CSR(i64_o64, int64_t, int64_t)
#undef CUEMBED_DECLARE_EXTRACT_CSR

/* ---- generic entry points (runtime type codes; same semantics) ---------- */
void cuembed_embedding_forward(const void* params, int elem_type, int embed_width,
                               const void* indices, int index_type, const void* offsets,
                               int offset_type, const void* weights, int batch_size,
                               int num_hots, int mode, int fp16_math, void* ret,
                               cuembed_stream_t stream);
/* Extension: with a compressed gradient (transpose_remapped_indices != NULL) num_grad_embedding_rows may
 * be negative = "num_unique is only known on the device" (it is transpose_remapped_indices[nnz-1] + 1):
 * grad_embedding / inverse_mapping must hold at least that many rows (nnz always suffices), rows past
 * the last id are left untouched, and no host read-back is needed between ComputeCompressedGradIndices
 * and EmbeddingBackward.  Applies to the typed entry points as well. */
void cuembed_embedding_backward(const void* grad_y, int elem_type, int embed_width,
                                int num_grad_embedding_rows, int nnz,
                                const void* transpose_indices, const void* transpose_sample_ids,
                                const void* transpose_remapped_indices, int index_type,
                                const void* transpose_weights, int skip_grad_init,
                                void* grad_embedding, void* inverse_mapping,
                                cuembed_stream_t stream);
/* Extension (cuembed::EmbeddingBackwardReferenceSums; opt-in, for verification): cuembed_embedding_backward in the
 * REFERENCE's arithmetic -- product and running sum rounded to the gradient type at every lookup, in nz order, as the
 * CPU reference's loop does (embedding_lookup_cpu.hpp:131-143) -- and therefore bit-identical to it for ANY data
 * (fp16 / bf16 gradients that are not exactly representable, runs of any length), where the default entry points
 * keep fp32 partial sums (ARITHMETIC note above).  WHAT EXACTNESS COSTS: a rounding chain cannot be cut into partial
 * sums, so a table row's run is one chain of dependent additions (short runs: one lane group each; runs of 257 lookups
 * and more: a whole workgroup stages the rows in LDS and one wavefront chains them from there).  At BASELINE config 4
 * (10M x 256, 65,536 x 64 lookups, alpha 1.15: the hottest row is a chain of 65,528) 1.2 ms in fp16 and 1.8 ms in fp32
 * against 0.26 / 0.54 ms for the default entry point on the same data -- 4.5 x and 3.4 x (bench.py: roofline.other_kernels,
 * "cost_of_exactness", with bit_identical_to_oracle checked on non-representable data).
 * num_grad_embedding_rows >= 0; skip_grad_init != 0 adds to what grad_embedding holds. */
void cuembed_embedding_backward_reference_sums(const void* grad_y, int elem_type, int embed_width,
                                               int num_grad_embedding_rows, int nnz,
                                               const void* transpose_indices, const void* transpose_sample_ids,
                                               const void* transpose_remapped_indices, int index_type,
                                               const void* transpose_weights, int skip_grad_init,
                                               void* grad_embedding, void* inverse_mapping,
                                               cuembed_stream_t stream);
void cuembed_transpose(const void* rows, const void* cols, const void* weights, int nnz,
                       int index_type, int weight_type, void* transpose_rows,
                       void* transpose_cols, void* transpose_weights, char* work, size_t* lwork,
                       cuembed_stream_t stream);
/* Extension: as cuembed_transpose, for callers that know all lookup indices are
 * < 2^index_bits (e.g. ceil(log2(num_categories))): the radix sort then skips the
 * always-zero high digits.  index_bits <= 0 means "all bits" (= cuembed_transpose). */
void cuembed_transpose_bounded(const void* rows, const void* cols, const void* weights, int nnz,
                               int index_type, int weight_type, void* transpose_rows,
                               void* transpose_cols, void* transpose_weights, char* work,
                               size_t* lwork, int index_bits, cuembed_stream_t stream);
/* Extension: as cuembed_transpose_bounded, plus a bound on the values in `rows`: int64 rows
 * known to lie in [0, 2^row_bits), row_bits <= 32 (sample ids always do), are moved as 32 bits
 * between the radix passes without the library reading them once more to find out.
 * row_bits <= 0 means "unknown" (= cuembed_transpose_bounded). */
void cuembed_transpose_hinted(const void* rows, const void* cols, const void* weights, int nnz,
                              int index_type, int weight_type, void* transpose_rows,
                              void* transpose_cols, void* transpose_weights, char* work,
                              size_t* lwork, int index_bits, int row_bits, cuembed_stream_t stream);
/* Extension (cuembed::TransposeFixedHotness): cuembed_extract_row_ids_from_fixed +
 * cuembed_transpose_bounded in one call, without materialising the sample ids -- the first radix
 * pass derives the sample id of lookup i as i / num_hots.  Same outputs.  num_hots = 1 is the
 * concat layout (row id = position).  index_bits <= 0: all bits. */
void cuembed_transpose_fixed_hotness(const void* indices, const void* weights, int batch_size,
                                     int num_hots, int index_type, int weight_type,
                                     void* transpose_indices, void* transpose_sample_ids,
                                     void* transpose_weights, char* work, size_t* lwork, int index_bits,
                                     cuembed_stream_t stream);
/* Extension (cuembed::Transpose / TransposeFixedHotness, `sample_blocks`): sample_blocks > 1 CHANGES the
 * result -- the sample-major input is cut into that many consecutive blocks of equal length (whole 4096-element
 * tiles; at most 64) and each block is transposed on its own; the output is the concatenation of the sorted
 * blocks.  For the COMPRESSED gradient only: cuembed_compute_compressed_grad_indices and
 * cuembed_embedding_backward then produce one gradient row per (block, table row) -- a table row looked up from
 * several blocks appears once per block in inverse_mapping (an uncoalesced compressed gradient) -- and while a
 * block is scattered every L2 gathers from 1 / sample_blocks of grad_y (C4: backward 0.258 -> 0.191 ms with 2
 * blocks; 572 k -> 679 k gradient rows).  cuembed_recommended_sample_blocks picks the count (1 = no gain).
 * Never combine with a dense gradient.  Otherwise as cuembed_transpose_hinted / cuembed_transpose_fixed_hotness. */
void cuembed_transpose_sample_blocks(const void* rows, const void* cols, const void* weights, int nnz,
                                     int index_type, int weight_type, void* transpose_rows,
                                     void* transpose_cols, void* transpose_weights, char* work, size_t* lwork,
                                     int index_bits, int row_bits, int sample_blocks, cuembed_stream_t stream);
void cuembed_transpose_fixed_hotness_sample_blocks(const void* indices, const void* weights, int batch_size,
                                                   int num_hots, int index_type, int weight_type,
                                                   void* transpose_indices, void* transpose_sample_ids,
                                                   void* transpose_weights, char* work, size_t* lwork,
                                                   int index_bits, int sample_blocks, cuembed_stream_t stream);
/* Extension (cuembed::Transpose / TransposeFixedHotness, `transpose_remapped_indices`): as the two calls above, and
 * transpose_remapped_indices (nnz entries of the index type; NULL: not wanted) also receives what
 * cuembed_compute_compressed_grad_indices would compute from transpose_rows / transpose_indices -- the same values
 * from the same call.  Up to 4,096 lookups the WHOLE index work (row ids of a fixed-hotness batch, stable sort, remap)
 * is ONE launch of one 1024-thread workgroup (a dependent launch costs 3.5-5 us at these sizes and the
 * reference's sequence is about ten of them, index_transforms.cuh:95-137, :278-323); beyond that the run-head scan's
 * launches follow the sort's on the stream and share `work`. */
void cuembed_transpose_remapped(const void* rows, const void* cols, const void* weights, int nnz,
                                int index_type, int weight_type, void* transpose_rows,
                                void* transpose_cols, void* transpose_weights, void* transpose_remapped_indices,
                                char* work, size_t* lwork, int index_bits, int row_bits, int sample_blocks,
                                cuembed_stream_t stream);
void cuembed_transpose_fixed_hotness_remapped(const void* indices, const void* weights, int batch_size,
                                              int num_hots, int index_type, int weight_type,
                                              void* transpose_indices, void* transpose_sample_ids,
                                              void* transpose_weights, void* transpose_remapped_indices, char* work,
                                              size_t* lwork, int index_bits, int sample_blocks,
                                              cuembed_stream_t stream);
int cuembed_recommended_sample_blocks(int elem_type, int embed_width, int batch_size, int64_t nnz);
/* Lookups per block that the two calls above use: block k = lookups [k * L, (k + 1) * L), L a multiple of 4096
 * (inputs of up to 131,072 lookups are always ONE block). */
int64_t cuembed_transpose_sample_block_length(int64_t nnz, int sample_blocks);
void cuembed_compute_compressed_grad_indices(const void* indices, int nnz, int index_type,
                                             void* remapped_indices, char* work, size_t* lwork,
                                             cuembed_stream_t stream);
/* Extension (cuembed::ComputeCompressedGradIndicesBlocked + cuembed::EmbeddingBackward(..., sample_blocks,
 * block_row_ids)): the REFERENCE's compressed gradient -- num_unique ascending rows, the inverse_mapping of the
 * fully sorted order (embedding_lookup.cuh:423-483, index_transforms.cuh:278-323) -- computed from a
 * sample-blocked order, so that the backward gathers grad_y block by block (see cuembed_transpose_sample_blocks).
 *   cuembed_compute_compressed_grad_indices_blocked: `indices` = transpose_rows of a transpose with the SAME nnz
 *     and sample_blocks.  remapped_indices[i] = number of the (block, table row) pair of lookup i (ids count up
 *     through the array; a new one where the index changes or a block begins).  block_row_ids[pair] (room for nnz
 *     uint32 always suffices) = rank of that table row among all distinct rows of the array (= the id the fully
 *     sorted order assigns), with bit 30 set when the row also occurs in an earlier block.  *num_unique (device
 *     word, may be NULL) = number of distinct rows.  At most 8 blocks, nnz < 2^30.  With one block (sample_blocks
 *     <= 1 or nnz <= 131072) remapped_indices is cuembed_compute_compressed_grad_indices' and block_row_ids is
 *     left alone.  Two-phase workspace query.
 *   cuembed_embedding_backward_blocked: as cuembed_embedding_backward with a compressed gradient, for that COO,
 *     those remapped indices and that table: one stream-ordered launch per block; a run whose row an earlier block
 *     already stored is added to it (read-modify-write; float atomic when the run crosses workgroups).
 *     num_grad_embedding_rows = num_unique, or negative when it is only known on the device (buffers of
 *     min(nnz, table rows) rows then suffice).  fp32: the sum of a table row is taken block by block; fp16 /
 *     bf16: one rounding per block (ARITHMETIC note above applies per block). */
void cuembed_compute_compressed_grad_indices_blocked(const void* indices, int nnz, int index_type,
                                                     int sample_blocks, void* remapped_indices,
                                                     uint32_t* block_row_ids, uint32_t* num_unique,
                                                     char* work, size_t* lwork, cuembed_stream_t stream);
void cuembed_embedding_backward_blocked(const void* grad_y, int elem_type, int embed_width,
                                        int num_grad_embedding_rows, int nnz,
                                        const void* transpose_indices, const void* transpose_sample_ids,
                                        const void* transpose_remapped_indices, int index_type,
                                        const void* transpose_weights, int skip_grad_init,
                                        void* grad_embedding, void* inverse_mapping, int sample_blocks,
                                        const uint32_t* block_row_ids, cuembed_stream_t stream);
/* Extension: cuembed_embedding_backward_blocked (sample_blocks <= 1, block_row_ids NULL: cuembed_embedding_backward)
 * with a CAPACITY for the device-side row count.  With num_grad_embedding_rows < 0 only the device knows how many rows
 * the compressed gradient has; capacity_rows > 0 states how many rows grad_embedding and inverse_mapping really hold.
 * If the count exceeds it, nothing is written (per launch: a sample-blocked call may have written its earlier blocks)
 * and *capacity_overflow -- a device word the caller zeroed once; may be NULL -- is OR-ed with 1: a sticky flag to read
 * back whenever convenient instead of a silent overrun.  capacity_rows = 0: unchecked (the other entry points).
 * pad_to_capacity != 0 (needs capacity_rows > 0, num_grad_embedding_rows < 0, skip_grad_init == 0, one block): the
 * rows from the device-side count up to capacity_rows are zeroed and their inverse_mapping entries set to rows of
 * the batch, different ones in turn (entry i names the row of entry (i - count) mod count) -- (inverse_mapping, grad_embedding) over all capacity_rows entries is then a valid uncoalesced
 * COO gradient (coalescing it gives the reference's) that needs no read-back of the count at all. */
void cuembed_embedding_backward_bounded(const void* grad_y, int elem_type, int embed_width,
                                        int num_grad_embedding_rows, int nnz,
                                        const void* transpose_indices, const void* transpose_sample_ids,
                                        const void* transpose_remapped_indices, int index_type,
                                        const void* transpose_weights, int skip_grad_init,
                                        void* grad_embedding, void* inverse_mapping, int sample_blocks,
                                        const uint32_t* block_row_ids, int capacity_rows,
                                        uint32_t* capacity_overflow, int pad_to_capacity, cuembed_stream_t stream);
void cuembed_extract_row_ids_from_fixed(int batch_size, int num_hots, int index_type,
                                        void* row_ids, cuembed_stream_t stream);
void cuembed_extract_row_ids_from_csr(const void* offsets, int offset_type, int batch_size,
                                      int index_type, void* row_ids, cuembed_stream_t stream);
void cuembed_extract_row_ids_for_concat(int nnz, int index_type, void* row_ids,
                                        cuembed_stream_t stream);

/* ---- extension: table-row caching hook (cuembed::TranslateIndicesForRowCache) ------------ */
/* For tables that live outside this GPU's HBM (pinned host memory, a peer) with hot rows copied
 * to a device buffer `cache_rows`: translated[i] = slot_of_row[indices[i]] >= 0 ?
 * cache_row_offset + slot : indices[i], with cache_row_offset = (cache_rows - params) / embed_width
 * in elements (the buffers must differ by a whole number of rows).  cuembed_embedding_forward with
 * index_type = CUEMBED_I64 on `translated` and the table's own `params` pointer then reads cached
 * rows from HBM and the rest from the table -- same kernel, same bits.  slot_of_row: one int32 per
 * table row (num_rows entries), -1 = not cached; indices outside [0, num_rows) pass through unchanged. */
void cuembed_translate_indices_for_row_cache(const void* indices, int index_type, int64_t nnz,
                                             const int32_t* slot_of_row, int64_t num_rows,
                                             int64_t cache_row_offset, int64_t* translated,
                                             cuembed_stream_t stream);

/* ---- extension: gradient w.r.t. the per-lookup weights ---------------------- */
/* grad_weights[s, j] = dot(params[indices[s, j], :], grad_y[s, :]); one entry per lookup;
 * index layouts as in cuembed_embedding_forward (cuembed::EmbeddingWeightGrad). */
void cuembed_embedding_weight_grad(const void* params, int elem_type, int embed_width,
                                   const void* indices, int index_type, const void* offsets,
                                   int offset_type, const void* grad_y, int batch_size, int num_hots,
                                   void* grad_weights, cuembed_stream_t stream);

/* ---- options ------------------------------------------------------------- */
/* cuembed::SetForwardReductionOrder / GetForwardReductionOrder (this library's addition):
 * 0 = sequential (default; bit-identical to the reference for every batch size),
 * 1 = small batches may split a sample's hotness loop over several wavefronts and combine
 *     partial rows through LDS (same result up to fp rounding, much faster for small batches). */
void cuembed_set_forward_reduction_order(int order);
int cuembed_get_forward_reduction_order(void);
/* cuembed::SetForwardRowLoadPolicy / GetForwardRowLoadPolicy (extension; the reference has no such knob,
 * embedding_lookup_kernels.cuh:34-77).  Never changes a result.
 * 0 = default: ordinary table-row loads (rows stay in L2 / Infinity Cache; right whenever rows are re-used),
 * 1 = streaming: non-temporal row loads for batches in which (nearly) every lookup hits a different row of a table
 *     far larger than the caches -- the HBM-bound case (C2 shape with uniform indices: 0.379 -> 0.355 ms); with
 *     re-use it is much slower (alpha = 1.15: 0.136 -> 0.222 ms).  sum / mean only.
 * Both setters change PROCESS-WIDE defaults used by the reference-shaped entry points (initial values:
 * CUEMBED_FORWARD_ORDER=split, CUEMBED_FORWARD_ROW_LOADS=streaming).  A caller that shares its process with
 * other users of the library passes the options per call instead: */
void cuembed_set_forward_row_load_policy(int policy);
int cuembed_get_forward_row_load_policy(void);
/* cuembed::SetForwardWideLoad (tuning / tests; never changes a result).  Small batches of sum / mean lookups take a
 * kernel with one sample per workgroup that requests a whole bag's rows at once and pools them in lookup order out of LDS
 * (bit-identical to the sequential kernel; the reference has one mapping for every batch size,
 * embedding_lookup.cuh:186-208).  0 = the launcher decides (default), 1 = never, 2 = whenever the row shape allows it,
 * 3 / 4 / 5 / ... = likewise with 2 / 4 / 8 / ... samples sharing a workgroup (narrow rows; as far as the row allows). */
void cuembed_set_forward_wide_load(int mode);
/* cuembed_embedding_forward with per-call options (cuembed::ForwardOptions): reduction_order 0 / 1,
 * row_load_policy 0 / 1, or -1 = the process-wide default.  No state is read or written when both are >= 0. */
void cuembed_embedding_forward_with_options(const void* params, int elem_type, int embed_width,
                                            const void* indices, int index_type, const void* offsets,
                                            int offset_type, const void* weights, int batch_size,
                                            int num_hots, int mode, int fp16_math, void* ret,
                                            int reduction_order, int row_load_policy,
                                            cuembed_stream_t stream);
/* ... and with cuembed::ForwardOptions::sample_order (extension, CSR only; NULL = none): `sample_order` is a device
 * array holding a permutation of [0, batch_size) -- the order in which the samples are handed to the wavefronts.
 * A scheduling hint: every sample is still pooled in lookup order into its own output row, so results do not
 * depend on it.  With ragged bags in descending order of length (the two bags of a wavefront run in lockstep;
 * wavefronts with unequal bags end at different times) BASELINE config 3 takes 0.148 instead of 0.170 ms.
 * Aborts when given with fixed hotness (nothing to balance there). */
void cuembed_embedding_forward_ordered(const void* params, int elem_type, int embed_width,
                                       const void* indices, int index_type, const void* offsets,
                                       int offset_type, const void* weights, int batch_size,
                                       int num_hots, int mode, int fp16_math, void* ret,
                                       int reduction_order, int row_load_policy,
                                       const int32_t* sample_order, cuembed_stream_t stream);
/* ... and with cuembed::ForwardOptions::row_loads_device (extension; NULL = none): the row-load decision that
 * cuembed_decide_row_loads left in device memory is read by the kernels instead of row_load_policy. */
void cuembed_embedding_forward_device_hints(const void* params, int elem_type, int embed_width,
                                            const void* indices, int index_type, const void* offsets,
                                            int offset_type, const void* weights, int batch_size,
                                            int num_hots, int mode, int fp16_math, void* ret,
                                            int reduction_order, int row_load_policy,
                                            const int32_t* sample_order, const uint32_t* row_loads_device,
                                            cuembed_stream_t stream);
/* cuembed::DecideRowLoads (extension; the reference has one launch rule for every index distribution,
 * embedding_lookup.cuh:186-208): the row-load policy decided ON THE DEVICE from the batch's own indices, no read-back,
 * one launch, capturable.  decision[0] = 1 (non-temporal row loads) when at least distinct_per_65536 / 65536 of an
 * evenly strided sample of up to 65,536 lookups names distinct rows (counted exactly, in groups of 4,096), the table
 * has table_bytes >= 1 GiB and the batch nnz >= 2^18 lookups; else 0.  distinct_per_65536 = 0: the built-in 0.998
 * (8 repeats per group of 4,096: the measured crossover -- streaming costs 20 % as soon as 3 % of a group repeats).
 * `decision` = FOUR 32-bit device words zeroed once by the caller (words 1..3 are the kernel's own and stay zero);
 * calls sharing them must be stream-ordered.  Never changes a result. */
void cuembed_decide_row_loads(const void* indices, int index_type, int64_t nnz, int64_t table_bytes,
                              uint32_t* decision, unsigned distinct_per_65536, cuembed_stream_t stream);
/* cuembed::BagOrderByLength (extension): sample_order[batch_size] = the samples of a CSR batch by descending bag
 * length, ties in input order.  max_length > 0: a bound on the bag length (longer bags rank as max_length), 0 =
 * unknown, < 0 = bags of 255 lookups and more rank alike.  With a bound <= 255 and batch_size <= 131,072 it is two small
 * launches (a stable counting sort in chunks of 1,024 samples; 7 us for 65,536 bags -- cheap enough for every fresh offsets
 * array); otherwise a key kernel + the library's stable sort.  Two-phase workspace query: work == NULL writes the bytes needed to *lwork. */
void cuembed_bag_order_by_length(const void* offsets, int offset_type, int batch_size, int max_length,
                                 int32_t* sample_order, char* work, size_t* lwork, cuembed_stream_t stream);
/* cuembed::SetBackwardTuning / GetBackwardTuning (tuning and tests; 0 = built-in heuristic):
 * lookups per nz-segment (rounded down to a multiple of 8), XCD column slices of the gather
 * (1, 2, 4, 8).  Process-wide; initial values from CUEMBED_BWD_SEGMENT_LEN / CUEMBED_BWD_SLICES,
 * read once.  Results never depend on these. */
void cuembed_set_backward_tuning(int segment_len, int column_slices);
void cuembed_get_backward_tuning(int* out2);

/* ---- multi-GPU: the device-side halves of the sparse gradient exchange (extensions) ------
 * The reference is single-GPU (README.md:108-119: multi-device is future work).  A data-parallel step sums the ranks'
 * compressed gradients with an owner-partitioned exchange of FIXED-size buffers (cuembed_amd/distributed.py:
 * SparseGradExchange; the collectives are the caller's); these two calls are the index work around the collectives,
 * without any read-back (cuembed::PackRowsByOwner / FinishOwnerPiece, exchange_transforms.hpp).
 *
 * cuembed_exchange_pack_rows: ids[num_rows] (index_type) ascending in their first *count entries (count: one device
 * word of index_type; NULL = all), rows[num_rows, embed_width]; cuts[world + 1] device words (cuts[r] = first row id
 * of owner r, cuts[world] = num_categories).  Slot r of send_ids[world * slot_capacity] /
 * send_rows[world * slot_capacity, embed_width] gets the ids / rows of owner r's range, then the padding id
 * num_categories.  range_starts[world + 1]: device scratch.  *flag |= 1 (a 64-bit device word the caller zeroes) when
 * a range exceeds its slot or, with input_capacity > 0, *count exceeds input_capacity.  Two launches. */
void cuembed_exchange_pack_rows(const void* ids, int index_type, const void* rows, int elem_type, int64_t num_rows,
                                int embed_width, const void* count, const int64_t* cuts, int world,
                                int64_t slot_capacity, int64_t input_capacity, int64_t num_categories,
                                int64_t* send_ids, void* send_rows, int64_t* range_starts, int64_t* flag,
                                cuembed_stream_t stream);
/* cuembed_exchange_finish_piece: after the owner merged what it received -- cuembed_transpose_fixed_hotness_remapped
 * (the received int64 ids as nnz samples of hotness 1, index bits of num_categories + 1) and
 * cuembed_embedding_backward_bounded (capacity_rows = capacity + 1, pad_to_capacity) into ids[capacity + 1] /
 * rows[capacity + 1, embed_width] -- this finishes the piece on the device: *count (may be NULL) = distinct real ids;
 * the padding ids' run (or the spare last row) zeroed; ids behind the count = pad_lo + i % pad_len; and, when `tail`
 * (capacity + 2 words) is given, the id buffer of the all-gather: the ids, min(count, capacity), *flag.
 * *flag |= 1 when count > capacity (the merge then wrote nothing).  One launch. */
void cuembed_exchange_finish_piece(const int64_t* sorted_ids, const int64_t* remapped_ids, int64_t nnz, int64_t capacity,
                                   int64_t num_categories, int64_t pad_lo, int64_t pad_len, int64_t* ids, void* rows,
                                   int elem_type, int embed_width, int64_t* tail, int64_t* flag, int64_t* count,
                                   cuembed_stream_t stream);

/* ---- introspection ------------------------------------------------------- */
/* Launch shape the forward would use (no launch): out[0] = elements per lane,
 * out[1] = lanes per row, out[2] = samples per workgroup, out[3] = grid size,
 * out[4] = dynamic LDS bytes, out[5] = 1 if indices are staged in LDS, 2 if the batch is small enough for the
 * wide-load kernel (out[2] = 1, or a few samples of narrow rows, per workgroup; the bags' rows parked in LDS). */
void cuembed_forward_launch_shape(int elem_type, int index_type, int embed_width, int batch_size,
                                  int num_hots, int is_csr, int is_weighted, int mode,
                                  int* out);
/* What the launch heuristics know about the current device (read once per device id, cuembed::detail::DeviceShape;
 * the reference queries the runtime per call, embedding_lookup.cuh:355-363): out[0] = compute units, out[1] = XCDs,
 * out[2] = resident lanes per compute unit, out[3] = L2 bytes per XCD. */
void cuembed_device_shape(int* out);
/* Launch shape EmbeddingBackward would use (no launch): out[0] = XCD column slices, out[1] = lanes per slice,
 * out[2] = nz-segments per workgroup, out[3] = lookups per segment, out[4] = workgroup ranges, out[5] = grid size,
 * out[6] = dynamic LDS bytes, out[7] = XCDs assumed.  compute_units <= 0: the current device; > 0: a device of that
 * many compute units in `xcds` XCDs (host arithmetic only: e.g. 32 CUs in 1 XCD = a CPX partition). */
void cuembed_backward_launch_shape(int elem_type, int index_type, int embed_width, int64_t nnz, int is_weighted,
                                   int compute_units, int xcds, int* out);
/* cuembed_recommended_sample_blocks for a described device (compute_units <= 0: the current one). */
int cuembed_recommended_sample_blocks_on(int elem_type, int embed_width, int batch_size, int64_t nnz,
                                         int compute_units, int xcds, int64_t l2_bytes_per_xcd);
/* hipPeekAtLastError() as an int (0 = hipSuccess); launches themselves never
 * report errors, exactly like the reference. */
int cuembed_peek_last_error(void);
/* "cuembed_amd <version> gfx950" */
const char* cuembed_version(void);

#ifdef __cplusplus
}
#endif

#endif /* CUEMBED_AMD_H_ */
