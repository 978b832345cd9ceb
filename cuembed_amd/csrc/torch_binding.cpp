// PyTorch-ROCm binding of the embedding-lookup library: the reference's torch library
// `cuembed_pyt` (examples/pytorch/cuembed_embedding.cu:169-190) rebuilt natively.
//
//   TORCH_LIBRARY(cuembed_pyt, m)            the reference's four schemas, verbatim, + extensions
//   TORCH_LIBRARY_IMPL(cuembed_pyt, CUDA, m) HIP tensors dispatch on the CUDA key on ROCm
//
// Every op validates like the reference's binding (AT_ASSERT -> TORCH_CHECK: a Python exception,
// not an abort), makes its inputs contiguous, allocates its outputs with ATen and enqueues the
// HIP kernels on torch's CURRENT stream (c10::hip::getCurrentHIPStream(), the counterpart of
// at::cuda::getCurrentCUDAStream() at cuembed_embedding.cu:49) of the tensors' device.  The
// kernels are reached through the library's C ABI (include/cuembed_amd.h; every entry point is an
// explicit instantiation of the header-only templates), so this translation unit holds no device
// code and builds with the host compiler in seconds.
//
// Relative to the reference binding (fp32 / int64 / sum only, cuembed_embedding.cu:15-32) the
// ops accept fp16 and bf16 tables, int32 indices / offsets and mode = "mean"; the extension ops
// are listed in INTEGRATION.md section 4.
#include <ATen/ATen.h>
#include <ATen/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <string>
#include <tuple>

#include "cuembed_amd.h"

namespace {

int ElemCode(const at::Tensor& t, const char* what) {
  switch (t.scalar_type()) {
    case at::kFloat: return CUEMBED_F32;
    case at::kHalf: return CUEMBED_F16;
    case at::kBFloat16: return CUEMBED_BF16;
    default: TORCH_CHECK(false, "cuembed_pyt: ", what, " must be float32, float16 or bfloat16");
  }
  return 0;
}

int IndexCode(const at::Tensor& t, const char* what) {
  switch (t.scalar_type()) {
    case at::kInt: return CUEMBED_I32;
    case at::kLong: return CUEMBED_I64;
    default: TORCH_CHECK(false, "cuembed_pyt: ", what, " must be int32 or int64");
  }
  return 0;
}

void CheckGpu(const at::Tensor& t, const char* what) {
  TORCH_CHECK(t.is_cuda(), "cuembed_pyt: ", what, " must be on the GPU (there is no CPU path)");
}

const void* Ptr(const at::Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }
void* MutPtr(at::Tensor& t) { return t.defined() ? t.data_ptr() : nullptr; }

at::Tensor ContiguousOrUndefined(const at::Tensor& t) { return t.defined() ? t.contiguous() : t; }

cuembed_stream_t CurrentStream(const at::Tensor& t) {
  return static_cast<cuembed_stream_t>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

int Mode(const std::string& mode, bool allow_concat) {
  if (mode == "sum") return CUEMBED_SUM;
  if (mode == "mean") return CUEMBED_MEAN;
  TORCH_CHECK(allow_concat && mode == "concat", "cuembed_pyt: mode must be 'sum', 'mean'",
              allow_concat ? " or 'concat'" : "");
  return CUEMBED_CONCAT;
}

int IndexBits(const int64_t num_categories) {
  if (num_categories <= 0) return 0;
  int bits = 1;
  while (bits < 63 && (int64_t{1} << bits) < num_categories) ++bits;
  return bits;
}

// ---- the reference's four ops --------------------------------------------------------------

// reference: cuembed_embedding.cu:10-52 (CSR layout, offsets has batch + 1 entries)
at::Tensor cuembed_embedding_forward_op(const at::Tensor& params, const at::Tensor& indices,
                                     const at::Tensor& offsets, const at::Tensor& weights,
                                     const std::string& mode) {
  CheckGpu(params, "params");
  CheckGpu(indices, "indices");
  CheckGpu(offsets, "offsets");
  TORCH_CHECK(params.dim() == 2, "cuembed_pyt: params must be [num_categories, embed_width]");
  const int elem = ElemCode(params, "params");
  const int idx = IndexCode(indices, "indices");
  const int off = IndexCode(offsets, "offsets");
  if (weights.defined()) {
    CheckGpu(weights, "weights");
    TORCH_CHECK(weights.scalar_type() == params.scalar_type(), "cuembed_pyt: weights must have the dtype of params");
    TORCH_CHECK(weights.numel() >= indices.numel(), "cuembed_pyt: weights must have one entry per index");
  }
  const int m = Mode(mode, false);
  const at::DeviceGuard guard(params.device());
  const at::Tensor p = params.contiguous(), i = indices.contiguous(), o = offsets.contiguous();
  const at::Tensor w = ContiguousOrUndefined(weights);
  const int64_t batch = o.numel() - 1;
  TORCH_CHECK(batch >= 0, "cuembed_pyt: offsets must hold batch_size + 1 entries");
  at::Tensor out = at::empty({batch, p.size(1)}, p.options());
  if (batch > 0)
    ::cuembed_embedding_forward(Ptr(p), elem, static_cast<int>(p.size(1)), Ptr(i), idx, Ptr(o), off, Ptr(w),
                                static_cast<int>(batch), 0, m, 0, MutPtr(out), CurrentStream(p));
  return out;
}

// reference: cuembed_embedding.cu:54-68.  The reference's Python passes offsets[:-1] and the
// kernel reads one element past the slice; here the end of the last bag is `nnz`, which is right
// for the sliced and for the full offsets tensor alike.
at::Tensor cuembed_extract_row_ids_from_csr_op(const at::Tensor& offsets, const int64_t nnz) {
  CheckGpu(offsets, "offsets");
  const int off = IndexCode(offsets, "offsets");
  const at::DeviceGuard guard(offsets.device());
  const int64_t batch = offsets.numel();
  at::Tensor closed = at::empty({batch + 1}, offsets.options());
  closed.narrow(0, 0, batch).copy_(offsets.reshape({-1}));
  closed.narrow(0, batch, 1).fill_(nnz);
  at::Tensor row_ids = at::empty({nnz}, offsets.options());
  if (batch > 0 && nnz > 0)
    ::cuembed_extract_row_ids_from_csr(Ptr(closed), off, static_cast<int>(batch), off, MutPtr(row_ids),
                                       CurrentStream(offsets));
  return row_ids;
}

// Extension: the same for an offsets tensor that already holds batch + 1 entries (what callers of
// cuembed_embedding_forward have anyway): no copy, one launch.
at::Tensor cuembed_extract_row_ids_from_offsets_op(const at::Tensor& offsets, const int64_t nnz) {
  CheckGpu(offsets, "offsets");
  const int off = IndexCode(offsets, "offsets");
  TORCH_CHECK(offsets.numel() >= 1, "cuembed_pyt: offsets must hold batch_size + 1 entries");
  const at::DeviceGuard guard(offsets.device());
  const at::Tensor o = offsets.contiguous();
  const int64_t batch = o.numel() - 1;
  at::Tensor row_ids = at::empty({nnz}, o.options());
  if (batch > 0 && nnz > 0)
    ::cuembed_extract_row_ids_from_csr(Ptr(o), off, static_cast<int>(batch), off, MutPtr(row_ids), CurrentStream(o));
  return row_ids;
}

std::tuple<at::Tensor, at::Tensor, at::Tensor> TransposeImpl(const at::Tensor& rows, const at::Tensor& cols,
                                                             const at::Tensor& weights, const int index_bits,
                                                             const int row_bits, const int sample_blocks = 1) {
  CheckGpu(rows, "rows");
  CheckGpu(cols, "cols");
  const int idx = IndexCode(rows, "rows");
  TORCH_CHECK(cols.scalar_type() == rows.scalar_type() && cols.numel() == rows.numel(),
              "cuembed_pyt: rows and cols must have the same dtype and length");
  int wt = CUEMBED_F32;
  if (weights.defined()) {
    CheckGpu(weights, "weights");
    wt = ElemCode(weights, "weights");
    TORCH_CHECK(weights.numel() == rows.numel(), "cuembed_pyt: weights must have nnz entries");
  }
  const at::DeviceGuard guard(rows.device());
  const at::Tensor r = rows.contiguous(), c = cols.contiguous(), w = ContiguousOrUndefined(weights);
  const int64_t nnz = r.numel();
  TORCH_CHECK(nnz <= INT32_MAX, "cuembed_pyt: nnz must fit an int (reference API, index_transforms.cuh:224-234)");
  at::Tensor t_rows = at::empty_like(c), t_cols = at::empty_like(r);
  // the reference returns a 0-length float tensor when there are no weights (cuembed_embedding.cu:90-93)
  at::Tensor t_w = w.defined() ? at::empty_like(w) : at::empty({0}, r.options().dtype(at::kFloat));
  if (nnz == 0) return {t_rows, t_cols, t_w};
  size_t lwork = 0;
  const void* query_weights = w.defined() ? reinterpret_cast<const void*>(256) : nullptr;  // only its nullness matters
  ::cuembed_transpose_sample_blocks(nullptr, nullptr, query_weights, static_cast<int>(nnz), idx, wt, nullptr, nullptr,
                                    nullptr, nullptr, &lwork, index_bits, row_bits, sample_blocks, nullptr);
  at::Tensor work = at::empty({static_cast<int64_t>(lwork)}, r.options().dtype(at::kByte));
  ::cuembed_transpose_sample_blocks(Ptr(r), Ptr(c), Ptr(w), static_cast<int>(nnz), idx, wt, MutPtr(t_rows),
                                    MutPtr(t_cols), w.defined() ? MutPtr(t_w) : nullptr,
                                    static_cast<char*>(work.data_ptr()), &lwork, index_bits, row_bits, sample_blocks,
                                    CurrentStream(r));
  return {t_rows, t_cols, t_w};
}

// reference: cuembed_embedding.cu:70-120
std::tuple<at::Tensor, at::Tensor, at::Tensor> cuembed_transpose_op(const at::Tensor& rows, const at::Tensor& cols,
                                                                 const at::Tensor& weights) {
  return TransposeImpl(rows, cols, weights, 0, 0);
}

// reference: cuembed_embedding.cu:122-167 (dense gradient: torch::zeros + skip_grad_init)
at::Tensor cuembed_embedding_backward_op(const at::Tensor& y_grad, const int64_t num_categories,
                                      const at::Tensor& transpose_indices, const at::Tensor& transpose_sample_ids,
                                      const at::Tensor& transpose_weights) {
  CheckGpu(y_grad, "y_grad");
  CheckGpu(transpose_indices, "transpose_indices");
  CheckGpu(transpose_sample_ids, "transpose_sample_ids");
  TORCH_CHECK(y_grad.dim() == 2, "cuembed_pyt: y_grad must be [rows, embed_width]");
  const int elem = ElemCode(y_grad, "y_grad");
  const int idx = IndexCode(transpose_indices, "transpose_indices");
  TORCH_CHECK(transpose_sample_ids.scalar_type() == transpose_indices.scalar_type() &&
                  transpose_sample_ids.numel() == transpose_indices.numel(),
              "cuembed_pyt: transpose_sample_ids must match transpose_indices");
  if (transpose_weights.defined())
    TORCH_CHECK(transpose_weights.scalar_type() == y_grad.scalar_type() &&
                    transpose_weights.numel() == transpose_indices.numel(),
                "cuembed_pyt: transpose_weights must be nnz entries of y_grad's dtype");
  TORCH_CHECK(num_categories <= INT32_MAX && transpose_indices.numel() <= INT32_MAX, "cuembed_pyt: sizes must fit an int");
  const at::DeviceGuard guard(y_grad.device());
  const at::Tensor g = y_grad.contiguous(), ti = transpose_indices.contiguous(),
                   ts = transpose_sample_ids.contiguous(), tw = ContiguousOrUndefined(transpose_weights);
  at::Tensor grad = at::zeros({num_categories, g.size(1)}, g.options());
  ::cuembed_embedding_backward(Ptr(g), elem, static_cast<int>(g.size(1)), static_cast<int>(num_categories),
                               static_cast<int>(ti.numel()), Ptr(ti), Ptr(ts), nullptr, idx, Ptr(tw),
                               /*skip_grad_init=*/1, MutPtr(grad), nullptr, CurrentStream(g));
  return grad;
}

// ---- extensions ------------------------------------------------------------------------------

std::tuple<at::Tensor, at::Tensor, at::Tensor> cuembed_transpose_bounded_op(const at::Tensor& rows, const at::Tensor& cols,
                                                                         const at::Tensor& weights,
                                                                         const int64_t num_categories) {
  return TransposeImpl(rows, cols, weights, IndexBits(num_categories), 0);
}

// sample ids of a CSR / fixed-hotness batch are < nnz: int64 ids travel as 32 bits without a look
std::tuple<at::Tensor, at::Tensor, at::Tensor> cuembed_transpose_sample_ids_op(const at::Tensor& sample_ids,
                                                                            const at::Tensor& indices,
                                                                            const at::Tensor& weights,
                                                                            const int64_t num_categories) {
  return TransposeImpl(sample_ids, indices, weights, IndexBits(num_categories), 31);
}

// Transpose in blocks of samples (cuembed::Transpose, sample_blocks): for the compressed gradient only -- with the
// plain remap the result is an UNCOALESCED compressed gradient, one row per (block, table row).  sample_blocks is
// forwarded as given (<= 1: one block, the reference's fully sorted order); the caller asks
// cuembed_recommended_sample_blocks for a recommendation (cuembed_pyt.py does).
std::tuple<at::Tensor, at::Tensor, at::Tensor> cuembed_transpose_sample_blocks_op(const at::Tensor& sample_ids,
                                                                               const at::Tensor& indices,
                                                                               const at::Tensor& weights,
                                                                               const int64_t num_categories,
                                                                               const int64_t sample_blocks) {
  return TransposeImpl(sample_ids, indices, weights, IndexBits(num_categories), 31, static_cast<int>(sample_blocks));
}

at::Tensor cuembed_compute_compressed_grad_indices_op(const at::Tensor& transpose_indices) {
  CheckGpu(transpose_indices, "transpose_indices");
  const int idx = IndexCode(transpose_indices, "transpose_indices");
  const at::DeviceGuard guard(transpose_indices.device());
  const at::Tensor ti = transpose_indices.contiguous();
  TORCH_CHECK(ti.numel() <= INT32_MAX, "cuembed_pyt: nnz must fit an int");
  at::Tensor remapped = at::empty_like(ti);
  if (ti.numel() == 0) return remapped;
  size_t lwork = 0;
  ::cuembed_compute_compressed_grad_indices(nullptr, static_cast<int>(ti.numel()), idx, nullptr, nullptr, &lwork, nullptr);
  at::Tensor work = at::empty({static_cast<int64_t>(lwork > 0 ? lwork : 1)}, ti.options().dtype(at::kByte));
  ::cuembed_compute_compressed_grad_indices(Ptr(ti), static_cast<int>(ti.numel()), idx, MutPtr(remapped),
                                            static_cast<char*>(work.data_ptr()), &lwork, CurrentStream(ti));
  return remapped;
}

// compressed gradient: (rows[num_unique, W], inverse_mapping[num_unique])
std::tuple<at::Tensor, at::Tensor> cuembed_embedding_backward_compressed_op(
    const at::Tensor& y_grad, const int64_t num_unique, const at::Tensor& transpose_indices,
    const at::Tensor& transpose_sample_ids, const at::Tensor& transpose_remapped_indices,
    const at::Tensor& transpose_weights) {
  CheckGpu(y_grad, "y_grad");
  TORCH_CHECK(y_grad.dim() == 2, "cuembed_pyt: y_grad must be [rows, embed_width]");
  const int elem = ElemCode(y_grad, "y_grad");
  const int idx = IndexCode(transpose_indices, "transpose_indices");
  TORCH_CHECK(transpose_sample_ids.scalar_type() == transpose_indices.scalar_type() &&
                  transpose_remapped_indices.scalar_type() == transpose_indices.scalar_type() &&
                  transpose_sample_ids.numel() == transpose_indices.numel() &&
                  transpose_remapped_indices.numel() == transpose_indices.numel(),
              "cuembed_pyt: the transposed index tensors must agree in dtype and length");
  if (transpose_weights.defined())
    TORCH_CHECK(transpose_weights.scalar_type() == y_grad.scalar_type() &&
                    transpose_weights.numel() == transpose_indices.numel(),
                "cuembed_pyt: transpose_weights must be nnz entries of y_grad's dtype");
  TORCH_CHECK(num_unique <= INT32_MAX && transpose_indices.numel() <= INT32_MAX, "cuembed_pyt: sizes must fit an int");
  const at::DeviceGuard guard(y_grad.device());
  const at::Tensor g = y_grad.contiguous(), ti = transpose_indices.contiguous(),
                   ts = transpose_sample_ids.contiguous(), tr = transpose_remapped_indices.contiguous(),
                   tw = ContiguousOrUndefined(transpose_weights);
  at::Tensor grad = at::empty({num_unique, g.size(1)}, g.options());
  at::Tensor inv = at::empty({num_unique}, ti.options());
  const int width = static_cast<int>(g.size(1)), nnz = static_cast<int>(ti.numel());
  ::cuembed_embedding_backward(Ptr(g), elem, width, static_cast<int>(num_unique), nnz, Ptr(ti), Ptr(ts), Ptr(tr), idx,
                               Ptr(tw), /*skip_grad_init=*/0, MutPtr(grad), MutPtr(inv), CurrentStream(g));
  return {grad, inv};
}

at::Tensor cuembed_embedding_forward_fixed_op(const at::Tensor& params, const at::Tensor& indices,
                                           const at::Tensor& weights, const std::string& mode) {
  CheckGpu(params, "params");
  CheckGpu(indices, "indices");
  TORCH_CHECK(params.dim() == 2 && indices.dim() == 2, "cuembed_pyt: params [rows, width], indices [batch, hotness]");
  const int elem = ElemCode(params, "params");
  const int idx = IndexCode(indices, "indices");
  const int m = Mode(mode, true);
  if (weights.defined()) {
    TORCH_CHECK(m != CUEMBED_CONCAT, "cuembed_pyt: concat does not take weights");
    TORCH_CHECK(weights.scalar_type() == params.scalar_type() && weights.sizes() == indices.sizes(),
                "cuembed_pyt: weights must match indices in shape and params in dtype");
  }
  const at::DeviceGuard guard(params.device());
  const at::Tensor p = params.contiguous(), i = indices.contiguous(), w = ContiguousOrUndefined(weights);
  const int64_t batch = i.size(0), hot = i.size(1), width = p.size(1);
  TORCH_CHECK(hot > 0, "cuembed_pyt: hotness must be positive");
  at::Tensor out = m == CUEMBED_CONCAT ? at::empty({batch, hot, width}, p.options()) : at::empty({batch, width}, p.options());
  if (batch > 0)
    ::cuembed_embedding_forward(Ptr(p), elem, static_cast<int>(width), Ptr(i), idx, nullptr, 0, Ptr(w),
                                static_cast<int>(batch), static_cast<int>(hot), m, 0, MutPtr(out), CurrentStream(p));
  return out;
}

at::Tensor cuembed_embedding_weight_grad_op(const at::Tensor& params, const at::Tensor& indices,
                                         const at::Tensor& offsets, const at::Tensor& y_grad) {
  CheckGpu(params, "params");
  CheckGpu(indices, "indices");
  CheckGpu(offsets, "offsets");
  CheckGpu(y_grad, "y_grad");
  const int elem = ElemCode(params, "params");
  const int idx = IndexCode(indices, "indices");
  const int off = IndexCode(offsets, "offsets");
  TORCH_CHECK(y_grad.scalar_type() == params.scalar_type() && y_grad.dim() == 2 && y_grad.size(1) == params.size(1),
              "cuembed_pyt: y_grad must be [batch, width] of the table's dtype");
  const at::DeviceGuard guard(params.device());
  const at::Tensor p = params.contiguous(), i = indices.contiguous(), o = offsets.contiguous(), g = y_grad.contiguous();
  const int64_t batch = o.numel() - 1;
  at::Tensor out = at::empty({i.numel()}, p.options());
  if (batch > 0 && i.numel() > 0)
    ::cuembed_embedding_weight_grad(Ptr(p), elem, static_cast<int>(p.size(1)), Ptr(i), idx, Ptr(o), off, Ptr(g),
                                    static_cast<int>(batch), 0, MutPtr(out), CurrentStream(p));
  return out;
}

// One call for the index work of a fixed-hotness training step: (sorted indices, sample ids,
// weights, dense ids) = TransposeFixedHotness + ComputeCompressedGradIndices.
std::tuple<at::Tensor, at::Tensor, at::Tensor, at::Tensor> cuembed_transpose_fixed_hotness_op(
    const at::Tensor& indices, const at::Tensor& weights, const int64_t num_categories, const bool compressed) {
  CheckGpu(indices, "indices");
  TORCH_CHECK(indices.dim() == 2, "cuembed_pyt: indices must be [batch, hotness]");
  const int idx = IndexCode(indices, "indices");
  int wt = CUEMBED_F32;
  if (weights.defined()) {
    wt = ElemCode(weights, "weights");
    TORCH_CHECK(weights.numel() == indices.numel(), "cuembed_pyt: weights must match indices");
  }
  const at::DeviceGuard guard(indices.device());
  const at::Tensor i = indices.contiguous(), w = ContiguousOrUndefined(weights);
  const int64_t batch = i.size(0), hot = i.size(1), nnz = batch * hot;
  TORCH_CHECK(nnz <= INT32_MAX && hot > 0, "cuembed_pyt: batch * hotness must fit an int");
  const auto flat = i.options();
  at::Tensor t_idx = at::empty({nnz}, flat), t_sid = at::empty({nnz}, flat);
  at::Tensor t_w = w.defined() ? at::empty({nnz}, w.options()) : at::empty({0}, flat.dtype(at::kFloat));
  at::Tensor remap = at::empty({compressed ? nnz : 0}, flat);
  if (nnz == 0) return {t_idx, t_sid, t_w, remap};
  size_t lwork = 0, lwork2 = 0;
  const void* query_weights = w.defined() ? reinterpret_cast<const void*>(256) : nullptr;
  ::cuembed_transpose_fixed_hotness(nullptr, query_weights, static_cast<int>(batch), static_cast<int>(hot), idx, wt,
                                    nullptr, nullptr, nullptr, nullptr, &lwork, IndexBits(num_categories), nullptr);
  ::cuembed_compute_compressed_grad_indices(nullptr, static_cast<int>(nnz), idx, nullptr, nullptr, &lwork2, nullptr);
  size_t both = lwork > lwork2 ? lwork : lwork2;
  at::Tensor work = at::empty({static_cast<int64_t>(both)}, flat.dtype(at::kByte));
  const cuembed_stream_t stream = CurrentStream(i);
  ::cuembed_transpose_fixed_hotness(Ptr(i), Ptr(w), static_cast<int>(batch), static_cast<int>(hot), idx, wt,
                                    MutPtr(t_idx), MutPtr(t_sid), w.defined() ? MutPtr(t_w) : nullptr,
                                    static_cast<char*>(work.data_ptr()), &both, IndexBits(num_categories), stream);
  if (compressed)
    ::cuembed_compute_compressed_grad_indices(Ptr(t_idx), static_cast<int>(nnz), idx, MutPtr(remap),
                                              static_cast<char*>(work.data_ptr()), &both, stream);
  return {t_idx, t_sid, t_w, remap};
}

}  // namespace

TORCH_LIBRARY(cuembed_pyt, m) {
  // the reference's schemas, verbatim (examples/pytorch/cuembed_embedding.cu:169-183)
  m.def("cuembed_extract_row_ids_from_csr(Tensor offsets, int nnz) ->Tensor");
  m.def("cuembed_transpose(Tensor rows, Tensor cols, Tensor weights) -> (Tensor, Tensor, Tensor)");
  m.def("cuembed_embedding_forward(Tensor params, Tensor indices, Tensor offsets, Tensor weights, str mode) -> Tensor");
  m.def(
      "cuembed_embedding_backward(Tensor y_grad, int num_categories, Tensor transpose_indices, Tensor "
      "transpose_sample_ids, Tensor transpose_weights) -> Tensor");
  // this library's extensions
  m.def("cuembed_extract_row_ids_from_offsets(Tensor offsets, int nnz) -> Tensor");
  m.def("cuembed_transpose_bounded(Tensor rows, Tensor cols, Tensor weights, int num_categories) -> (Tensor, Tensor, Tensor)");
  m.def(
      "cuembed_transpose_sample_ids(Tensor sample_ids, Tensor indices, Tensor weights, int num_categories) -> "
      "(Tensor, Tensor, Tensor)");
  m.def(
      "cuembed_transpose_fixed_hotness(Tensor indices, Tensor weights, int num_categories, bool compressed) -> "
      "(Tensor, Tensor, Tensor, Tensor)");
  m.def(
      "cuembed_transpose_sample_blocks(Tensor sample_ids, Tensor indices, Tensor weights, int num_categories, int "
      "sample_blocks) -> (Tensor, Tensor, Tensor)");
  m.def("cuembed_compute_compressed_grad_indices(Tensor transpose_indices) -> Tensor");
  m.def(
      "cuembed_embedding_backward_compressed(Tensor y_grad, int num_unique, Tensor transpose_indices, Tensor "
      "transpose_sample_ids, Tensor transpose_remapped_indices, Tensor transpose_weights) -> (Tensor, Tensor)");
  m.def("cuembed_embedding_forward_fixed(Tensor params, Tensor indices, Tensor weights, str mode) -> Tensor");
  m.def("cuembed_embedding_weight_grad(Tensor params, Tensor indices, Tensor offsets, Tensor y_grad) -> Tensor");
}

TORCH_LIBRARY_IMPL(cuembed_pyt, CUDA, m) {  // HIP tensors use the CUDA dispatch key on PyTorch-ROCm
  m.impl("cuembed_extract_row_ids_from_csr", cuembed_extract_row_ids_from_csr_op);
  m.impl("cuembed_transpose", cuembed_transpose_op);
  m.impl("cuembed_embedding_forward", cuembed_embedding_forward_op);
  m.impl("cuembed_embedding_backward", cuembed_embedding_backward_op);
  m.impl("cuembed_extract_row_ids_from_offsets", cuembed_extract_row_ids_from_offsets_op);
  m.impl("cuembed_transpose_bounded", cuembed_transpose_bounded_op);
  m.impl("cuembed_transpose_sample_ids", cuembed_transpose_sample_ids_op);
  m.impl("cuembed_transpose_sample_blocks", cuembed_transpose_sample_blocks_op);
  m.impl("cuembed_transpose_fixed_hotness", cuembed_transpose_fixed_hotness_op);
  m.impl("cuembed_compute_compressed_grad_indices", cuembed_compute_compressed_grad_indices_op);
  m.impl("cuembed_embedding_backward_compressed", cuembed_embedding_backward_compressed_op);
  m.impl("cuembed_embedding_forward_fixed", cuembed_embedding_forward_fixed_op);
  m.impl("cuembed_embedding_weight_grad", cuembed_embedding_weight_grad_op);
}
