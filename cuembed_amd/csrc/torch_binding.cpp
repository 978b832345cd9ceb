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
#include <c10/hip/HIPGraphsC10Utils.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <cstdlib>
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
// `row_loads` -1 = the process-wide default, 0 / 1 = RowLoadPolicy; `sample_order` = ForwardOptions::sample_order
// (a permutation of the samples; undefined = none): scheduling hints, never a different result.
at::Tensor ForwardImpl(const at::Tensor& params, const at::Tensor& indices, const at::Tensor& offsets,
                       const at::Tensor& weights, const std::string& mode, const int row_loads,
                       const at::Tensor& sample_order, const at::Tensor& row_loads_device = at::Tensor()) {
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
  if (sample_order.defined())
    TORCH_CHECK(sample_order.is_cuda() && sample_order.scalar_type() == at::kInt && sample_order.numel() == batch &&
                    sample_order.is_contiguous(),
                "cuembed_pyt: sample_order must be a contiguous int32 permutation of the samples on the GPU");
  if (row_loads_device.defined())      // the words cuembed_decide_row_loads works on: the kernels read the decision themselves
    TORCH_CHECK(row_loads_device.is_cuda() && row_loads_device.scalar_type() == at::kInt && row_loads_device.numel() >= 4 &&
                    row_loads_device.is_contiguous(),
                "cuembed_pyt: row_loads_device must be the contiguous 4-word int32 tensor of cuembed_decide_row_loads");
  if (batch > 0)
    ::cuembed_embedding_forward_device_hints(Ptr(p), elem, static_cast<int>(p.size(1)), Ptr(i), idx, Ptr(o), off, Ptr(w),
                                             static_cast<int>(batch), 0, m, 0, MutPtr(out), /*reduction_order=*/-1, row_loads,
                                             static_cast<const int32_t*>(Ptr(sample_order)),
                                             static_cast<const uint32_t*>(Ptr(row_loads_device)), CurrentStream(p));
  return out;
}

at::Tensor cuembed_embedding_forward_op(const at::Tensor& params, const at::Tensor& indices,
                                     const at::Tensor& offsets, const at::Tensor& weights,
                                     const std::string& mode) {
  return ForwardImpl(params, indices, offsets, weights, mode, -1, at::Tensor());
}

// Extension: ... with the per-call scheduling hints of cuembed::ForwardOptions.
at::Tensor cuembed_embedding_forward_hinted_op(const at::Tensor& params, const at::Tensor& indices,
                                            const at::Tensor& offsets, const c10::optional<at::Tensor>& weights,
                                            const std::string& mode, const int64_t row_loads,
                                            const c10::optional<at::Tensor>& sample_order,
                                            const c10::optional<at::Tensor>& row_loads_device) {
  return ForwardImpl(params, indices, offsets, weights.has_value() ? *weights : at::Tensor(), mode,
                     static_cast<int>(row_loads), sample_order.has_value() ? *sample_order : at::Tensor(),
                     row_loads_device.has_value() ? *row_loads_device : at::Tensor());
}

// Extension (cuembed::DecideRowLoads): the row-load policy decided on the device from the batch's indices; `decision` is
// four int32 words zeroed once by the caller, handed to the forward as row_loads_device.  One launch, no read-back.
void cuembed_decide_row_loads_op(const at::Tensor& indices, const int64_t table_bytes, at::Tensor decision) {
  CheckGpu(indices, "indices");
  CheckGpu(decision, "decision");
  const int idx = IndexCode(indices, "indices");
  TORCH_CHECK(decision.scalar_type() == at::kInt && decision.numel() >= 4 && decision.is_contiguous(),
              "cuembed_pyt: decision must be a contiguous int32 tensor of 4 words");
  const at::DeviceGuard guard(indices.device());
  const at::Tensor i = indices.contiguous();
  ::cuembed_decide_row_loads(Ptr(i), idx, i.numel(), table_bytes, static_cast<uint32_t*>(decision.data_ptr()), 0u,
                             CurrentStream(i));
}

// Extension (cuembed::BagOrderByLength): the samples of a CSR batch by descending bag length, for sample_order.
at::Tensor cuembed_bag_order_by_length_op(const at::Tensor& offsets, const int64_t max_length) {
  CheckGpu(offsets, "offsets");
  const int off = IndexCode(offsets, "offsets");
  const at::DeviceGuard guard(offsets.device());
  const at::Tensor o = offsets.contiguous();
  const int64_t batch = o.numel() - 1;
  TORCH_CHECK(batch >= 0, "cuembed_pyt: offsets must hold batch_size + 1 entries");
  at::Tensor order = at::empty({batch}, o.options().dtype(at::kInt));
  if (batch == 0) return order;
  size_t lwork = 0;
  ::cuembed_bag_order_by_length(nullptr, off, static_cast<int>(batch), static_cast<int>(max_length), nullptr, nullptr,
                                &lwork, nullptr);
  at::Tensor work = at::empty({static_cast<int64_t>(lwork > 0 ? lwork : 1)}, o.options().dtype(at::kByte));
  ::cuembed_bag_order_by_length(Ptr(o), off, static_cast<int>(batch), static_cast<int>(max_length),
                                static_cast<int32_t*>(order.data_ptr()), static_cast<char*>(work.data_ptr()), &lwork,
                                CurrentStream(o));
  return order;
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


// ---- the whole training step of cuemb_embedding as ONE autograd node -------------------------------------------
// examples/pytorch/cuembed_pyt.py:12-51 runs forward and backward as a Python autograd.Function over four ops; at small
// batches that is host-bound (B = 1024, H = 64: 0.236 ms per step for ~0.09 ms of kernels, most of it the Python
// dispatch of six ops and a read-back of num_unique in the MIDDLE of the backward).  Here forward and backward are one
// C++ node each: the backward enqueues row ids -> transpose (+ remap in the same call) -> scatter-add without returning
// to Python, and for a sparse gradient the row count is read back AFTER everything is enqueued (the scatter runs into
// buffers of capacity min(nnz, rows) under cuembed_embedding_backward_bounded and the result is narrowed to
// num_unique rows), so the host waits for the device once, at the end, instead of stalling the pipeline in the middle.
// Above kCapacityBytes of worst-case gradient the count is read back first, as before (2.1 GB at the C4 shape; there
// the device is the bottleneck, not the host: running the scatter into 5/4 of the table's LAST row count under the
// capacity check and reading the count at the end was built and measured 0.53-0.63 ms against 0.49 -- 435 MB
// allocations of changing size churn the caching allocator).
constexpr int64_t kCapacityBytes = int64_t{192} << 20;
// Up to this much worst-case gradient the "fastest order" kinds do not read the count back AT ALL: the gradient is
// handed on padded to min(nnz, rows) entries (zero rows that name a row of the batch: an uncoalesced COO tensor of the
// same value), so the host never waits for the device inside a step -- at B = 1024 the wait was a third of the step.
// Why not more: with the limit raised (tools/torch_padded_limit_probe.py, profiles/r05_torch_padded_limit_probe.txt) the step
// alone gains further (4,096 samples, 128 MB: 0.193 -> 0.122 ms; 8,192: 0.187 -> 0.164; slower from 16,384 on), but whoever
// consumes the gradient pays for min(lookups, rows) entries instead of num_unique (tools/padded_gradient_consumer_probe.py:
// step + torch.optim.SGD at 4,096 samples 0.236 -> 0.448 ms, at 1,024 samples 0.16 -> 0.22; step + coalesce() 0.29 -> 0.23 at
// 1,024, 0.32 -> 0.37 at 4,096) -- the entry count of torch's own EmbeddingBag(sparse=True) gradient, but not what
// "reference" hands on.  CUEMBED_PYT_PADDED_MB moves the limit (0: never pad).
constexpr int64_t kPaddedBytesDefault = int64_t{64} << 20;
// (tuning: CUEMBED_PYT_PADDED_MB, read once)
inline int64_t PaddedBytes() {
  static const int64_t v = [] {
    const char* e = std::getenv("CUEMBED_PYT_PADDED_MB");
    return e != nullptr ? (static_cast<int64_t>(std::atoll(e)) << 20) : kPaddedBytesDefault;
  }();
  return v;
}

// What the caller asked for (cuembed_pyt.py: sparse_grad=).  kGradSparseReference is the CONTRACT of sparse_grad=True: a
// coalesced tensor of exactly num_unique ascending rows, whatever the backend and whether or not torch.compile traces
// the call; the other sparse kinds are explicit opt-ins whose entry count differs from that (same dense value).
enum GradKind : int64_t {
  kGradDense = 0,
  kGradSparseFastest = 1,      // sample blocks when the shape gains from them, padded when the step would wait on the count
  kGradSparseReference = 2,    // fully sorted order: coalesced, num_unique rows
  kGradSparseUncoalesced = 3,  // sample blocks when the shape gains from them, exactly one entry per (block, row): never padded
  kGradSparsePadded = 4        // one block, min(lookups, rows) entries at ANY size: never reads the count, capturable
};

struct NarrowedIndices {
  at::Tensor idx, offsets;
};
// int64 indices of a table with < 2^31 rows: the index work of the backward runs on int32 copies (half the bytes through
// every sorting pass) -- from 2^18 lookups up, where the two conversions cost less than they save (cuembed_pyt.py).
NarrowedIndices NarrowForIndexWork(const at::Tensor& idx, const at::Tensor& offsets, const int64_t num_categories) {
  if (idx.scalar_type() == at::kLong && num_categories < (int64_t{1} << 31) && idx.numel() >= (int64_t{1} << 18) &&
      idx.numel() < (int64_t{1} << 31))
    return {idx.to(at::kInt), offsets.to(at::kInt)};
  // The forward takes indices and offsets of different integer types (ForwardImpl passes a type code for each); the index
  // work below has ONE type for lookups and sample ids, the one of the indices: offsets follow (their values are <= nnz,
  // which fits whatever type holds the lookups' count; int64 -> int32 is checked by the caller against INT32_MAX).
  if (offsets.scalar_type() != idx.scalar_type()) return {idx, offsets.to(idx.scalar_type())};
  return {idx, offsets};
}

struct TransposedLookups {
  at::Tensor t_idx, t_sid, t_w, remap;
};
// Transpose (+ ComputeCompressedGradIndices from the same call) of (sample_ids, indices[, weights]).
TransposedLookups TransposeWithRemap(const at::Tensor& sample_ids, const at::Tensor& indices, const at::Tensor& weights,
                                     const int64_t num_categories, const int sample_blocks, const bool want_remap) {
  CheckGpu(sample_ids, "sample_ids");
  CheckGpu(indices, "indices");
  const int idx = IndexCode(indices, "indices");
  // one type code names both arrays below: a pair of different integer types would be read and written past its end
  TORCH_CHECK(sample_ids.scalar_type() == indices.scalar_type() && sample_ids.numel() == indices.numel(),
              "cuembed_pyt: sample ids and indices must have the same dtype and length");
  TORCH_CHECK(indices.numel() <= INT32_MAX, "cuembed_pyt: the number of lookups must fit an int");
  TORCH_CHECK(sample_ids.is_contiguous() && indices.is_contiguous(), "cuembed_pyt: sample ids and indices must be contiguous");
  int wt = CUEMBED_F32;
  if (weights.defined()) {
    CheckGpu(weights, "weights");
    wt = ElemCode(weights, "weights");
    TORCH_CHECK(weights.numel() == indices.numel() && weights.is_contiguous(),
                "cuembed_pyt: weights must be contiguous with one entry per lookup");
  }
  const int nnz = static_cast<int>(indices.numel());
  TransposedLookups t;
  t.t_idx = at::empty_like(indices);
  t.t_sid = at::empty_like(sample_ids);
  if (weights.defined()) t.t_w = at::empty_like(weights);
  if (want_remap) t.remap = at::empty_like(indices);
  size_t lwork = 0;
  const void* query_weights = weights.defined() ? reinterpret_cast<const void*>(256) : nullptr;
  const int bits = IndexBits(num_categories);
  ::cuembed_transpose_remapped(nullptr, nullptr, query_weights, nnz, idx, wt, nullptr, nullptr, nullptr, nullptr, nullptr,
                               &lwork, bits, 31, sample_blocks, nullptr);
  at::Tensor work = at::empty({static_cast<int64_t>(lwork > 0 ? lwork : 1)}, indices.options().dtype(at::kByte));
  ::cuembed_transpose_remapped(Ptr(sample_ids), Ptr(indices), Ptr(weights), nnz, idx, wt, MutPtr(t.t_idx), MutPtr(t.t_sid),
                               weights.defined() ? MutPtr(t.t_w) : nullptr, want_remap ? MutPtr(t.remap) : nullptr,
                               static_cast<char*>(work.data_ptr()), &lwork, bits, 31, sample_blocks, CurrentStream(indices));
  return t;
}

class CuEmbEmbeddingNode : public torch::autograd::Function<CuEmbEmbeddingNode> {
 public:
  // (optional tensors travel as std::optional: autograd's argument scan cannot take an undefined at::Tensor)
  static at::Tensor forward(torch::autograd::AutogradContext* ctx, const at::Tensor& params, const at::Tensor& indices,
                            const at::Tensor& offsets, const c10::optional<at::Tensor>& optional_weights,
                            const int64_t grad_kind, const int64_t row_loads,
                            const c10::optional<at::Tensor>& optional_sample_order,
                            const c10::optional<at::Tensor>& optional_row_loads_device) {
    at::AutoDispatchBelowADInplaceOrView below;
    const at::Tensor weights = optional_weights.has_value() ? *optional_weights : at::Tensor();
    const at::Tensor sample_order = optional_sample_order.has_value() ? *optional_sample_order : at::Tensor();
    ctx->saved_data["num_categories"] = params.size(0);
    ctx->saved_data["grad_kind"] = grad_kind;
    ctx->saved_data["weighted"] = weights.defined();
    if (weights.defined()) ctx->save_for_backward({indices, offsets, weights});
    else ctx->save_for_backward({indices, offsets});
    return ForwardImpl(params, indices, offsets, weights, "sum", static_cast<int>(row_loads), sample_order,
                       optional_row_loads_device.has_value() ? *optional_row_loads_device : at::Tensor());
  }

  static torch::autograd::variable_list backward(torch::autograd::AutogradContext* ctx,
                                                 torch::autograd::variable_list grad_outputs) {
    at::AutoDispatchBelowADInplaceOrView below;
    const auto saved = ctx->get_saved_variables();
    const bool weighted = ctx->saved_data["weighted"].toBool();
    const int64_t num_categories = ctx->saved_data["num_categories"].toInt();
    const int64_t grad_kind = ctx->saved_data["grad_kind"].toInt();
    at::Tensor out_grad = grad_outputs[0];
    torch::autograd::variable_list grads(8);   // (params, indices, offsets, weights, grad_kind, row_loads, sample_order, row_loads_device)
    if (!ctx->needs_input_grad(0)) return grads;
    CheckGpu(out_grad, "the incoming gradient");
    const at::DeviceGuard guard(out_grad.device());
    out_grad = out_grad.contiguous();
    const int64_t width = out_grad.size(1);
    const int64_t nnz = saved[0].numel();
    TORCH_CHECK(nnz <= INT32_MAX, "cuembed_pyt: the number of lookups must fit an int");
    const at::Tensor weights = weighted ? saved[2].contiguous() : at::Tensor();
    // (the forward reads the first nnz weights of a longer tensor; the transpose moves exactly nnz)
    TORCH_CHECK(!weighted || weights.numel() == nnz, "cuembed_pyt: weights must have one entry per index");
    if (nnz == 0) {
      if (grad_kind == kGradDense) {
        grads[0] = at::zeros({num_categories, width}, out_grad.options());
      } else {
        grads[0] = at::_sparse_coo_tensor_unsafe(at::empty({1, 0}, out_grad.options().dtype(at::kLong)),
                                                 at::empty({0, width}, out_grad.options()), {num_categories, width},
                                                 out_grad.options().layout(at::kSparse));
      }
      return grads;
    }
    const NarrowedIndices nw = NarrowForIndexWork(saved[0].contiguous(), saved[1].contiguous(), num_categories);
    const at::Tensor sample_ids = cuembed_extract_row_ids_from_offsets_op(nw.offsets, nnz);
    const int elem = ElemCode(out_grad, "the incoming gradient");
    const int idx = IndexCode(nw.idx, "indices");
    const cuembed_stream_t stream = CurrentStream(out_grad);
    if (grad_kind == kGradDense) {   // the reference's gradient: a zero-filled table-sized tensor (cuembed_embedding.cu:122-167)
      const TransposedLookups t = TransposeWithRemap(sample_ids, nw.idx, weights, num_categories, 1, false);
      grads[0] = cuembed_embedding_backward_op(out_grad, num_categories, t.t_idx, t.t_sid, t.t_w);
      return grads;
    }
    // ---- compressed gradient as a sparse COO tensor ----
    int blocks = 1;
    if (grad_kind == kGradSparseFastest || grad_kind == kGradSparseUncoalesced)   // while a block of samples is scattered every L2 gathers from 1 / blocks of out_grad
      blocks = ::cuembed_recommended_sample_blocks(elem, static_cast<int>(width), static_cast<int>(out_grad.size(0)), nnz);
    const TransposedLookups t = TransposeWithRemap(sample_ids, nw.idx, weights, num_categories, blocks, true);
    const bool one_block = ::cuembed_transpose_sample_block_length(nnz, blocks) >= nnz;
    const int64_t row_bound = one_block ? num_categories : static_cast<int64_t>(blocks) * num_categories;
    const int64_t capacity = nnz < row_bound ? nnz : row_bound;
    const int64_t row_bytes = width * static_cast<int64_t>(out_grad.element_size());
    at::Tensor rows, inv;
    int64_t num_unique = -1;
    const int64_t room = (grad_kind == kGradSparsePadded ||
                          capacity * row_bytes <= (kCapacityBytes > PaddedBytes() ? kCapacityBytes : PaddedBytes()))
                             ? capacity : 0;   // rows the scatter may write blind
    const auto exact_backward = [&]() {
      rows = at::empty({num_unique, width}, out_grad.options());
      inv = at::empty({num_unique}, nw.idx.options());
      ::cuembed_embedding_backward(Ptr(out_grad), elem, static_cast<int>(width), static_cast<int>(num_unique),
                                   static_cast<int>(nnz), Ptr(t.t_idx), Ptr(t.t_sid), Ptr(t.remap), idx, Ptr(t.t_w),
                                   /*skip_grad_init=*/0, MutPtr(rows), MutPtr(inv), stream);
      if (inv.scalar_type() != at::kLong) inv = inv.to(at::kLong);
    };
    const bool padded = grad_kind == kGradSparsePadded ||
                        (grad_kind == kGradSparseFastest && one_block && room > 0 && room * row_bytes <= PaddedBytes());
    if (padded) {
      rows = at::empty({room, width}, out_grad.options());
      inv = at::empty({room}, nw.idx.options());
      ::cuembed_embedding_backward_bounded(Ptr(out_grad), elem, static_cast<int>(width), -1, static_cast<int>(nnz),
                                           Ptr(t.t_idx), Ptr(t.t_sid), Ptr(t.remap), idx, Ptr(t.t_w), /*skip_grad_init=*/0,
                                           MutPtr(rows), MutPtr(inv), 1, nullptr, static_cast<int>(room), nullptr,
                                           /*pad_to_capacity=*/1, stream);
      if (inv.scalar_type() != at::kLong) inv = inv.to(at::kLong);
      grads[0] = at::_sparse_coo_tensor_unsafe(inv.unsqueeze(0), rows, {num_categories, width},
                                               out_grad.options().layout(at::kSparse), /*is_coalesced=*/false);
      return grads;
    }
    // (from here on the host has to read the row count: not something a HIP graph can hold)
    TORCH_CHECK(c10::hip::currentStreamCaptureStatusMayInitCtx() == c10::hip::CaptureStatus::None,
                "cuembed_pyt: this sparse gradient needs its row count on the host and cannot be captured into a graph: "
                "sparse_grad=\"padded\" never reads it (min(lookups, rows) entries, here ", (capacity * row_bytes) >> 20,
                " MiB), sparse_grad=\"fastest\" does not while those fit ", PaddedBytes() >> 20,
                " MiB; sparse_grad=True / \"reference\" / \"uncoalesced\" always read the count; sparse_grad=False never does");
    if (room > 0) {
      // everything is enqueued before the host looks at the device: rows for `room`, narrowed afterwards
      rows = at::empty({room, width}, out_grad.options());
      inv = at::empty({room}, nw.idx.options());
      ::cuembed_embedding_backward_bounded(Ptr(out_grad), elem, static_cast<int>(width), -1, static_cast<int>(nnz),
                                           Ptr(t.t_idx), Ptr(t.t_sid), Ptr(t.remap), idx, Ptr(t.t_w), /*skip_grad_init=*/0,
                                           MutPtr(rows), MutPtr(inv), 1, nullptr, static_cast<int>(room), nullptr, 0, stream);
      at::Tensor inv64 = inv.scalar_type() == at::kLong ? inv : inv.to(at::kLong);
      num_unique = t.remap.narrow(0, nnz - 1, 1).item().toLong() + 1;   // (the one wait of the step)
      rows = rows.narrow(0, 0, num_unique);
      inv = inv64.narrow(0, 0, num_unique);
    } else {
      num_unique = t.remap.narrow(0, nnz - 1, 1).item().toLong() + 1;
      exact_backward();
    }
    grads[0] = at::_sparse_coo_tensor_unsafe(inv.unsqueeze(0), rows, {num_categories, width},
                                             out_grad.options().layout(at::kSparse), /*is_coalesced=*/one_block);
    return grads;
  }
};

// ---- multi-GPU: the device-side halves of the sparse gradient exchange (cuembed_amd/distributed.py) ----------------
// As a chain of tensor operations (searchsorted, where, index_select, index_fill_, ...) the index work around the
// collectives was ~60 launches and more host time than the whole forward + backward step; here it is two ops that
// enqueue 2 and ~13 launches without returning to Python.  Nothing is read back.

// Extension (cuembed::PackRowsByOwner): the rank's compressed gradient into the fixed slots of the all-to-all.
void cuembed_exchange_pack_op(const at::Tensor& ids, const at::Tensor& rows, const c10::optional<at::Tensor>& count,
                              const at::Tensor& cuts, const int64_t slot_capacity, const int64_t input_capacity,
                              const int64_t num_categories, at::Tensor send_ids, at::Tensor send_rows,
                              at::Tensor range_starts, at::Tensor flag) {
  CheckGpu(ids, "ids");
  CheckGpu(rows, "rows");
  CheckGpu(cuts, "cuts");
  CheckGpu(send_ids, "send_ids");
  CheckGpu(send_rows, "send_rows");
  CheckGpu(range_starts, "range_starts");
  CheckGpu(flag, "flag");
  const int idx = IndexCode(ids, "ids");
  const int elem = ElemCode(rows, "rows");
  TORCH_CHECK(rows.dim() == 2 && ids.dim() == 1 && rows.size(0) == ids.numel() && ids.is_contiguous() && rows.is_contiguous(),
              "cuembed_pyt: rows must be a contiguous [n, width] tensor with one contiguous id per row");
  const int64_t world = cuts.numel() - 1;
  TORCH_CHECK(world >= 1 && world <= 1024 && cuts.scalar_type() == at::kLong && cuts.is_contiguous(),
              "cuembed_pyt: cuts must be world + 1 contiguous int64 words (world <= 1024)");
  TORCH_CHECK(slot_capacity >= 1, "cuembed_pyt: slot_capacity must be at least one row");
  TORCH_CHECK(send_ids.scalar_type() == at::kLong && send_ids.is_contiguous() && send_ids.numel() == world * slot_capacity,
              "cuembed_pyt: send_ids must be world * slot_capacity contiguous int64 words");
  TORCH_CHECK(send_rows.scalar_type() == rows.scalar_type() && send_rows.is_contiguous() &&
                  send_rows.numel() == world * slot_capacity * rows.size(1),
              "cuembed_pyt: send_rows must be a contiguous [world * slot_capacity, width] tensor of rows' dtype");
  TORCH_CHECK(range_starts.scalar_type() == at::kLong && range_starts.is_contiguous() && range_starts.numel() >= world + 1,
              "cuembed_pyt: range_starts must hold world + 1 contiguous int64 words");
  TORCH_CHECK(flag.scalar_type() == at::kLong && flag.numel() >= 1 && flag.is_contiguous(),
              "cuembed_pyt: flag must be a contiguous int64 word");
  const at::DeviceGuard guard(rows.device());
  at::Tensor cnt;
  if (count.has_value() && count->defined()) {
    CheckGpu(*count, "count");
    TORCH_CHECK(count->numel() >= 1, "cuembed_pyt: count must hold one word");
    // (one word, read by the kernel: a count of the other integer type is converted here, one tiny launch)
    cnt = count->scalar_type() == ids.scalar_type() ? count->contiguous() : count->to(ids.scalar_type());
  }
  ::cuembed_exchange_pack_rows(Ptr(ids), idx, Ptr(rows), elem, ids.numel(), static_cast<int>(rows.size(1)), Ptr(cnt),
                               static_cast<const int64_t*>(cuts.data_ptr()), static_cast<int>(world), slot_capacity,
                               input_capacity, num_categories, static_cast<int64_t*>(send_ids.data_ptr()),
                               send_rows.data_ptr(), static_cast<int64_t*>(range_starts.data_ptr()),
                               static_cast<int64_t*>(flag.data_ptr()), CurrentStream(rows));
}

// Extension: the owner's fixed-capacity merge -- TransposeFixedHotness (+ remap) of the received ids, EmbeddingBackward
// with a device-side row count and pad_to_capacity into out_ids[capacity + 1] / out_rows[capacity + 1, width],
// cuembed::FinishOwnerPiece -- as one op.  ids >= num_categories are padding and dropped.  `tail`: capacity + 2 words
// (the ids, the count, the flag word), `count`: one word; either may be absent.
void cuembed_exchange_merge_op(const at::Tensor& ids, const at::Tensor& rows, const int64_t num_categories,
                               const int64_t pad_lo, const int64_t pad_len, at::Tensor out_ids, at::Tensor out_rows,
                               const c10::optional<at::Tensor>& tail, at::Tensor flag,
                               const c10::optional<at::Tensor>& count) {
  CheckGpu(ids, "ids");
  CheckGpu(rows, "rows");
  CheckGpu(out_ids, "out_ids");
  CheckGpu(out_rows, "out_rows");
  CheckGpu(flag, "flag");
  const int elem = ElemCode(rows, "rows");
  const int64_t nnz = ids.numel();
  TORCH_CHECK(ids.scalar_type() == at::kLong && ids.is_contiguous() && nnz >= 1 && nnz <= INT32_MAX,
              "cuembed_pyt: ids must be 1 .. 2^31 - 1 contiguous int64 words");
  TORCH_CHECK(rows.dim() == 2 && rows.size(0) == nnz && rows.is_contiguous(),
              "cuembed_pyt: rows must be a contiguous [n, width] tensor with one id per row");
  const int64_t width = rows.size(1);
  const int64_t capacity = out_ids.numel() - 1;
  TORCH_CHECK(capacity >= 1 && capacity < INT32_MAX && out_ids.scalar_type() == at::kLong && out_ids.is_contiguous(),
              "cuembed_pyt: out_ids must be capacity + 1 contiguous int64 words");
  TORCH_CHECK(out_rows.scalar_type() == rows.scalar_type() && out_rows.is_contiguous() && out_rows.dim() == 2 &&
                  out_rows.size(0) == capacity + 1 && out_rows.size(1) == width,
              "cuembed_pyt: out_rows must be a contiguous [capacity + 1, width] tensor of rows' dtype");
  TORCH_CHECK(flag.scalar_type() == at::kLong && flag.numel() >= 1 && flag.is_contiguous(),
              "cuembed_pyt: flag must be a contiguous int64 word");
  TORCH_CHECK(pad_len >= 1, "cuembed_pyt: pad_len must be at least 1");
  int64_t* tail_ptr = nullptr;
  if (tail.has_value() && tail->defined()) {
    CheckGpu(*tail, "tail");
    TORCH_CHECK(tail->scalar_type() == at::kLong && tail->is_contiguous() && tail->numel() == capacity + 2,
                "cuembed_pyt: tail must be capacity + 2 contiguous int64 words");
    tail_ptr = static_cast<int64_t*>(tail->data_ptr());
  }
  int64_t* count_ptr = nullptr;
  if (count.has_value() && count->defined()) {
    CheckGpu(*count, "count");
    TORCH_CHECK(count->scalar_type() == at::kLong && count->numel() >= 1 && count->is_contiguous(),
                "cuembed_pyt: count must be a contiguous int64 word");
    count_ptr = static_cast<int64_t*>(count->data_ptr());
  }
  const at::DeviceGuard guard(rows.device());
  const cuembed_stream_t stream = CurrentStream(rows);
  at::Tensor t_idx = at::empty_like(ids), t_pos = at::empty_like(ids), remap = at::empty_like(ids);
  size_t lwork = 0;
  const int bits = IndexBits(num_categories + 1);   // the padding id num_categories is a key too
  ::cuembed_transpose_fixed_hotness_remapped(nullptr, nullptr, static_cast<int>(nnz), 1, CUEMBED_I64, CUEMBED_F32, nullptr,
                                             nullptr, nullptr, nullptr, nullptr, &lwork, bits, 1, nullptr);
  at::Tensor work = at::empty({static_cast<int64_t>(lwork > 0 ? lwork : 1)}, ids.options().dtype(at::kByte));
  ::cuembed_transpose_fixed_hotness_remapped(Ptr(ids), nullptr, static_cast<int>(nnz), 1, CUEMBED_I64, CUEMBED_F32,
                                             MutPtr(t_idx), MutPtr(t_pos), nullptr, MutPtr(remap),
                                             static_cast<char*>(work.data_ptr()), &lwork, bits, 1, stream);
  // capacity + 1 rows: the run of the padding ids needs a row too; too many distinct ids -> the kernels write nothing
  ::cuembed_embedding_backward_bounded(Ptr(rows), elem, static_cast<int>(width), -1, static_cast<int>(nnz), Ptr(t_idx),
                                       Ptr(t_pos), Ptr(remap), CUEMBED_I64, nullptr, /*skip_grad_init=*/0,
                                       out_rows.data_ptr(), out_ids.data_ptr(), 1, nullptr,
                                       static_cast<int>(capacity + 1), nullptr, /*pad_to_capacity=*/1, stream);
  ::cuembed_exchange_finish_piece(static_cast<const int64_t*>(t_idx.data_ptr()),
                                  static_cast<const int64_t*>(remap.data_ptr()), nnz, capacity, num_categories, pad_lo,
                                  pad_len, static_cast<int64_t*>(out_ids.data_ptr()), out_rows.data_ptr(), elem,
                                  static_cast<int>(width), tail_ptr, static_cast<int64_t*>(flag.data_ptr()), count_ptr,
                                  stream);
}

// grad_kind: see GradKind.
at::Tensor cuemb_embedding_autograd_op(const at::Tensor& params, const at::Tensor& indices, const at::Tensor& offsets,
                                    const at::Tensor& weights, const int64_t grad_kind, const int64_t row_loads,
                                    const at::Tensor& sample_order, const at::Tensor& row_loads_device) {
  TORCH_CHECK(grad_kind >= kGradDense && grad_kind <= kGradSparsePadded, "cuembed_pyt: unknown grad_kind");
  return CuEmbEmbeddingNode::apply(params, indices, offsets,
                                   weights.defined() ? c10::optional<at::Tensor>(weights) : c10::nullopt, grad_kind,
                                   row_loads,
                                   sample_order.defined() ? c10::optional<at::Tensor>(sample_order) : c10::nullopt,
                                   row_loads_device.defined() ? c10::optional<at::Tensor>(row_loads_device) : c10::nullopt);
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
  // forward + backward of cuemb_embedding as one native autograd node (see CuEmbEmbeddingNode)
  m.def(
      "cuemb_embedding_step(Tensor params, Tensor indices, Tensor offsets, Tensor? weights, int grad_kind, int row_loads, "
      "Tensor? sample_order, Tensor? row_loads_device) -> Tensor");
  m.def(
      "cuembed_embedding_forward_hinted(Tensor params, Tensor indices, Tensor offsets, Tensor? weights, str mode, int "
      "row_loads, Tensor? sample_order, Tensor? row_loads_device) -> Tensor");
  m.def("cuembed_decide_row_loads(Tensor indices, int table_bytes, Tensor(a!) decision) -> ()");
  m.def("cuembed_bag_order_by_length(Tensor offsets, int max_length) -> Tensor");
  m.def(
      "cuembed_exchange_pack(Tensor ids, Tensor rows, Tensor? count, Tensor cuts, int slot_capacity, int input_capacity, "
      "int num_categories, Tensor(a!) send_ids, Tensor(b!) send_rows, Tensor(c!) range_starts, Tensor(d!) flag) -> ()");
  m.def(
      "cuembed_exchange_merge(Tensor ids, Tensor rows, int num_categories, int pad_lo, int pad_len, Tensor(a!) out_ids, "
      "Tensor(b!) out_rows, Tensor(c!)? tail, Tensor(d!) flag, Tensor(e!)? count) -> ()");
}

TORCH_LIBRARY_IMPL(cuembed_pyt, Autograd, m) {
  m.impl("cuemb_embedding_step", [](const at::Tensor& params, const at::Tensor& indices, const at::Tensor& offsets,
                                    const c10::optional<at::Tensor>& weights, int64_t grad_kind, int64_t row_loads,
                                    const c10::optional<at::Tensor>& sample_order,
                                    const c10::optional<at::Tensor>& row_loads_device) {
    return cuemb_embedding_autograd_op(params, indices, offsets, weights.has_value() ? *weights : at::Tensor(), grad_kind,
                                       row_loads, sample_order.has_value() ? *sample_order : at::Tensor(),
                                       row_loads_device.has_value() ? *row_loads_device : at::Tensor());
  });
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
  m.impl("cuembed_embedding_forward_hinted", cuembed_embedding_forward_hinted_op);
  m.impl("cuembed_bag_order_by_length", cuembed_bag_order_by_length_op);
  m.impl("cuembed_decide_row_loads", cuembed_decide_row_loads_op);
  m.impl("cuembed_exchange_pack", cuembed_exchange_pack_op);
  m.impl("cuembed_exchange_merge", cuembed_exchange_merge_op);
}
