"""PyTorch op surface: the reference's `cuembed_pyt` torch library, rebuilt on PyTorch-ROCm.

Same op names and schemas as examples/pytorch/cuembed_embedding.cu:169-183 (registered for the
CUDA dispatch key, which is what HIP tensors use on ROCm) and the same Python entry point as
examples/pytorch/cuembed_pyt.py:48-51:

    from cuembed_amd.cuembed_pyt import cuemb_embedding
    out = cuemb_embedding(weight, indices, offsets, per_sample_weights)   # like nn.EmbeddingBag(sum)

The ops live in a native extension, cuembed_amd/lib/libcuembed_pyt.so (TORCH_LIBRARY +
TORCH_LIBRARY_IMPL in cuembed_amd/csrc/torch_binding.cpp, built by cuembed_amd/build.py), loaded
here with torch.ops.load_library exactly like the reference loads its own (cuembed_pyt.py:8-10).
They run the HIP kernels on torch's current stream.  Relative to the reference binding (fp32 /
int64 / sum only, cuembed_embedding.cu:15-32) they also accept fp16 / bf16 tables, int32 indices
and offsets and mode="mean".

CUEMBED_PYT_BACKEND=python registers the same ops from Python instead (torch.library + ctypes
calls into the same HIP library): kept for A/B timing of the binding overhead
(tools/torch_op_step_probe.py), not a fallback -- a missing native library is an error.
"""
import os

import torch

from . import ops as _ops
from . import policy as _policy

BACKEND = os.environ.get("CUEMBED_PYT_BACKEND", "native")
NATIVE_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libcuembed_pyt.so")


def _load_native():
    from . import _lib as _clib
    _clib.lib()                                   # libcuembed_amd.so first: fail loudly if it is absent
    if not os.path.exists(NATIVE_LIB):
        raise _clib.CuembedLibraryError(
            "cuembed_amd: %s not found. Build it with `python -m cuembed_amd.build` (needs g++ and the "
            "torch headers); there is no fallback." % NATIVE_LIB)
    torch.ops.load_library(NATIVE_LIB)


def _define_python_ops():
    global _lib
    _lib = torch.library.Library("cuembed_pyt", "DEF")
    _lib.define("cuembed_extract_row_ids_from_csr(Tensor offsets, int nnz) ->Tensor")
    _lib.define("cuembed_transpose(Tensor rows, Tensor cols, Tensor weights) -> (Tensor, Tensor, Tensor)")
    _lib.define("cuembed_extract_row_ids_from_offsets(Tensor offsets, int nnz) -> Tensor")
    _lib.define("cuembed_transpose_bounded(Tensor rows, Tensor cols, Tensor weights, int num_categories)"
                " -> (Tensor, Tensor, Tensor)")
    _lib.define("cuembed_transpose_sample_ids(Tensor sample_ids, Tensor indices, Tensor weights, int num_categories)"
                " -> (Tensor, Tensor, Tensor)")
    _lib.define("cuembed_transpose_sample_blocks(Tensor sample_ids, Tensor indices, Tensor weights, int num_categories,"
                " int sample_blocks) -> (Tensor, Tensor, Tensor)")
    _lib.define("cuembed_transpose_fixed_hotness(Tensor indices, Tensor weights, int num_categories, bool compressed)"
                " -> (Tensor, Tensor, Tensor, Tensor)")
    _lib.define("cuembed_compute_compressed_grad_indices(Tensor transpose_indices) -> Tensor")
    _lib.define("cuembed_embedding_backward_compressed(Tensor y_grad, int num_unique, Tensor transpose_indices,"
                " Tensor transpose_sample_ids, Tensor transpose_remapped_indices, Tensor transpose_weights)"
                " -> (Tensor, Tensor)")
    _lib.define("cuembed_embedding_forward_fixed(Tensor params, Tensor indices, Tensor weights, str mode)"
                " -> Tensor")
    _lib.define("cuembed_embedding_weight_grad(Tensor params, Tensor indices, Tensor offsets, Tensor y_grad)"
                " -> Tensor")
    _lib.define("cuembed_embedding_forward(Tensor params, Tensor indices, Tensor offsets, Tensor weights,"
                " str mode) -> Tensor")
    _lib.define("cuembed_embedding_backward(Tensor y_grad, int num_categories, Tensor transpose_indices,"
                " Tensor transpose_sample_ids, Tensor transpose_weights) -> Tensor")
    _lib.define("cuembed_embedding_forward_hinted(Tensor params, Tensor indices, Tensor offsets, Tensor? weights,"
                " str mode, int row_loads, Tensor? sample_order, Tensor? row_loads_device) -> Tensor")
    _lib.define("cuembed_bag_order_by_length(Tensor offsets, int max_length) -> Tensor")
    _lib.define("cuembed_decide_row_loads(Tensor indices, int table_bytes, Tensor(a!) decision) -> ()")
    _lib.impl("cuembed_decide_row_loads", _decide_row_loads_impl, "CUDA")
    _lib.impl("cuembed_embedding_forward_hinted", _forward_hinted_impl, "CUDA")
    _lib.impl("cuembed_bag_order_by_length", _bag_order_impl, "CUDA")
    _lib.impl("cuembed_extract_row_ids_from_offsets", _extract_closed_impl, "CUDA")
    _lib.impl("cuembed_transpose_sample_ids", _transpose_sample_ids_impl, "CUDA")
    _lib.impl("cuembed_transpose_sample_blocks", _transpose_sample_blocks_impl, "CUDA")
    _lib.impl("cuembed_transpose_fixed_hotness", _transpose_fixed_impl, "CUDA")
    _lib.impl("cuembed_embedding_weight_grad", _weight_grad_impl, "CUDA")
    _lib.impl("cuembed_embedding_forward_fixed", _forward_fixed_impl, "CUDA")
    _lib.impl("cuembed_compute_compressed_grad_indices", _compress_impl, "CUDA")
    _lib.impl("cuembed_embedding_backward_compressed", _backward_compressed_impl, "CUDA")
    _lib.impl("cuembed_embedding_forward", _forward_impl, "CUDA")
    _lib.impl("cuembed_extract_row_ids_from_csr", _extract_impl, "CUDA")
    _lib.impl("cuembed_transpose", _transpose_impl, "CUDA")
    _lib.impl("cuembed_transpose_bounded", _transpose_bounded_impl, "CUDA")
    _lib.impl("cuembed_embedding_backward", _backward_impl, "CUDA")


# Extension ops beyond the reference's four (same list in torch_binding.cpp):
#   cuembed_transpose_bounded / _sample_ids   the transpose when the caller knows indices < num_categories
#                                             (and that `rows` are sample ids)
#   cuembed_transpose_sample_blocks           the transpose in blocks of samples, each sorted on its own (compressed
#                                             gradient only: uncoalesced, but every L2 gathers from 1 / blocks of
#                                             grad_y at a time -- C4 backward 0.258 -> 0.19 ms)
#   cuembed_transpose_fixed_hotness           row ids + transpose (+ dense ids) of a [batch, hotness] index
#                                             tensor in one call, sample ids never materialised
#   cuembed_compute_compressed_grad_indices,  compressed (sparse) table gradient: only the rows that were looked
#   cuembed_embedding_backward_compressed     up are materialised (293 MB instead of a zero-filled 5.12 GB at
#                                             the north-star shape)
#   cuembed_embedding_forward_fixed           fixed-hotness forward with every combine mode of the C++ API (the
#                                             reference binding only exposes CSR + sum, cuembed_embedding.cu:29-32)
#   cuembed_embedding_weight_grad             gradient w.r.t. the per-lookup weights

_FLOATS = (torch.float32, torch.float16, torch.bfloat16)
_INTS = (torch.int64, torch.int32)


def _require(cond, msg):
    if not cond:
        raise RuntimeError("cuembed_pyt: " + msg)


def _forward_hinted_impl(params, indices, offsets, weights, mode, row_loads, sample_order, row_loads_device=None):
    return _forward_impl(params, indices, offsets, weights, mode, row_loads, sample_order, row_loads_device)


def _decide_row_loads_impl(indices, table_bytes, decision):
    _ops.decide_row_loads(indices, int(table_bytes), decision)


def _bag_order_impl(offsets, max_length):
    return _ops.bag_order_by_length(offsets.contiguous(), max_length=int(max_length))


def _forward_impl(params, indices, offsets, weights, mode, row_loads=-1, sample_order=None, row_loads_device=None):
    _require(params.is_cuda and indices.is_cuda and offsets.is_cuda, "tensors must be on the GPU")
    _require(params.dtype in _FLOATS, "params must be float32 or float16")
    _require(indices.dtype in _INTS and offsets.dtype in _INTS, "indices/offsets must be int64 or int32")
    _require(mode in ("sum", "mean"), "mode must be 'sum' (or 'mean')")
    if weights is not None:
        _require(weights.dtype == params.dtype, "weights must have the dtype of params")
        weights = weights.contiguous()
    batch_size = offsets.numel() - 1
    return _ops.embedding_forward(params.contiguous(), indices.contiguous(), offsets.contiguous(), weights,
                                  batch_size=batch_size, num_hots=0, mode=mode,
                                  row_loads={-1: None, 0: "default", 1: "streaming"}[int(row_loads)],
                                  sample_order=sample_order, row_loads_device=row_loads_device)


def _extract_impl(offsets, nnz):
    _require(offsets.is_cuda, "offsets must be on the GPU")
    _require(offsets.dtype in _INTS, "offsets must be int64 or int32")
    # The reference passes `offsets[:-1]` and lets the kernel read one element past the slice
    # (cuembed_pyt.py:23 / cuembed_embedding.cu:60).  Here the end of the last bag is made
    # explicit from `nnz`, which is correct for the sliced AND the full offsets tensor.
    closed = torch.cat([offsets.reshape(-1), torch.tensor([nnz], dtype=offsets.dtype, device=offsets.device)])
    return _ops.extract_row_ids_from_csr(closed, nnz=nnz, dtype=offsets.dtype, batch_size=offsets.numel())


def _extract_closed_impl(offsets, nnz):
    _require(offsets.is_cuda and offsets.dtype in _INTS, "offsets must be int tensors on the GPU")
    return _ops.extract_row_ids_from_csr(offsets.contiguous(), nnz=nnz, dtype=offsets.dtype,
                                         batch_size=offsets.numel() - 1)


def _transpose_bounded_impl(rows, cols, weights, num_categories):
    return _transpose_impl(rows, cols, weights, num_categories)


def _transpose_sample_ids_impl(sample_ids, indices, weights, num_categories):
    return _transpose_impl(sample_ids, indices, weights, num_categories, num_rows=1 << 31)


def _transpose_sample_blocks_impl(sample_ids, indices, weights, num_categories, sample_blocks):
    return _transpose_impl(sample_ids, indices, weights, num_categories, num_rows=1 << 31, sample_blocks=sample_blocks)


def _transpose_fixed_impl(indices, weights, num_categories, compressed):
    _require(indices.is_cuda and indices.dim() == 2, "indices must be [batch, hotness] on the GPU")
    batch, hot = indices.shape
    w = None if weights is None else weights.contiguous().view(-1)
    t_idx, t_sid, t_w = _ops.transpose_fixed_hotness(indices.contiguous().view(-1), batch, hot, w,
                                                     num_categories=num_categories if num_categories > 0 else None)
    if t_w is None:
        t_w = torch.empty(0, dtype=torch.float32, device=indices.device)
    remap = _ops.compute_compressed_grad_indices(t_idx) if compressed else \
        torch.empty(0, dtype=indices.dtype, device=indices.device)
    return t_idx, t_sid, t_w, remap


def _transpose_impl(rows, cols, weights, num_categories=None, num_rows=None, sample_blocks=1):
    _require(rows.is_cuda and cols.is_cuda, "tensors must be on the GPU")
    _require(rows.dtype in _INTS and cols.dtype == rows.dtype, "rows/cols must both be int64 or int32")
    if weights is not None:
        _require(weights.dtype in _FLOATS, "weights must be float32 or float16")
        weights = weights.contiguous()
    t_rows, t_cols, t_w = _ops.transpose(rows.contiguous(), cols.contiguous(), weights,
                                         num_categories=num_categories, num_rows=num_rows, sample_blocks=sample_blocks)
    if t_w is None:  # the reference returns a 0-length float tensor (cuembed_embedding.cu:90-93)
        t_w = torch.empty(0, dtype=torch.float32, device=rows.device)
    return t_rows, t_cols, t_w


def _backward_impl(y_grad, num_categories, transpose_indices, transpose_sample_ids, transpose_weights):
    _require(y_grad.is_cuda and transpose_indices.is_cuda and transpose_sample_ids.is_cuda,
             "tensors must be on the GPU")
    _require(y_grad.dtype in _FLOATS, "y_grad must be float32 or float16")
    _require(transpose_indices.dtype in _INTS and transpose_sample_ids.dtype == transpose_indices.dtype,
             "transposed indices must be int64 or int32")
    if transpose_weights is not None:
        transpose_weights = transpose_weights.contiguous()
    width = y_grad.size(1)
    grad = torch.zeros((num_categories, width), dtype=y_grad.dtype, device=y_grad.device)
    _ops.embedding_backward(y_grad.contiguous(), num_categories, transpose_indices.contiguous(),
                            transpose_sample_ids.contiguous(), None, transpose_weights,
                            skip_grad_init=True, grad_embedding=grad)
    return grad


def _forward_fixed_impl(params, indices, weights, mode):
    _require(params.is_cuda and indices.is_cuda, "tensors must be on the GPU")
    _require(params.dtype in _FLOATS and indices.dtype in _INTS, "unsupported dtypes")
    _require(indices.dim() == 2, "indices must be [batch, hotness]")
    _require(mode in ("sum", "mean", "concat"), "mode must be 'sum', 'mean' or 'concat'")
    batch, hot = indices.shape
    if weights is not None:
        _require(weights.dtype == params.dtype and weights.shape == indices.shape,
                 "weights must match indices in shape and params in dtype")
        weights = weights.contiguous().view(-1)
    return _ops.embedding_forward(params.contiguous(), indices.contiguous().view(-1), None, weights,
                                  batch_size=batch, num_hots=hot, mode=mode)


def _weight_grad_impl(params, indices, offsets, y_grad):
    _require(params.is_cuda and indices.is_cuda and offsets.is_cuda and y_grad.is_cuda, "tensors must be on the GPU")
    return _ops.embedding_weight_grad(params.contiguous(), indices.contiguous(), y_grad.contiguous(),
                                      offsets=offsets.contiguous(), batch_size=offsets.numel() - 1, num_hots=0)


def _compress_impl(transpose_indices):
    _require(transpose_indices.is_cuda and transpose_indices.dtype in _INTS, "indices must be int tensors on the GPU")
    return _ops.compute_compressed_grad_indices(transpose_indices.contiguous())


def _backward_compressed_impl(y_grad, num_unique, transpose_indices, transpose_sample_ids,
                              transpose_remapped_indices, transpose_weights):
    _require(y_grad.is_cuda and y_grad.dtype in _FLOATS, "y_grad must be float32/float16 on the GPU")
    if transpose_weights is not None:
        transpose_weights = transpose_weights.contiguous()
    grad, inv = _ops.embedding_backward(y_grad.contiguous(), num_unique, transpose_indices.contiguous(),
                                        transpose_sample_ids.contiguous(),
                                        transpose_remapped_indices.contiguous(), transpose_weights,
                                        skip_grad_init=False)
    return grad, inv




if BACKEND == "python":
    _define_python_ops()
elif BACKEND == "native":
    _load_native()
else:
    raise ValueError("CUEMBED_PYT_BACKEND must be 'native' or 'python'")

cuembed_extract_row_ids_from_csr = torch.ops.cuembed_pyt.cuembed_extract_row_ids_from_csr
cuembed_transpose = torch.ops.cuembed_pyt.cuembed_transpose
cuembed_transpose_bounded = torch.ops.cuembed_pyt.cuembed_transpose_bounded
cuembed_transpose_sample_ids = torch.ops.cuembed_pyt.cuembed_transpose_sample_ids
cuembed_transpose_fixed_hotness = torch.ops.cuembed_pyt.cuembed_transpose_fixed_hotness
cuembed_embedding_forward = torch.ops.cuembed_pyt.cuembed_embedding_forward
cuembed_embedding_backward = torch.ops.cuembed_pyt.cuembed_embedding_backward


def cuembed_forward(params, idx, offsets, weights, hints=None):
    if hints is None:
        return cuembed_embedding_forward(params, idx, offsets, weights, mode="sum")
    return torch.ops.cuembed_pyt.cuembed_embedding_forward_hinted(params, idx, offsets, weights, "sum", hints[0], hints[1],
                                                                  hints[2] if len(hints) > 2 else None)


def _auto_hints(params, idx, offsets):
    """(row_loads, sample_order, row_loads_device) chosen by cuembed_amd.policy for this table and this batch -- both
    decided on the device, no read-back; never changes a result."""
    return -1, _policy.sample_order(offsets, idx.numel()), _policy.row_loads_device(params, idx)


def _narrow_for_index_work(idx, offsets, num_categories):
    """The reference's binding takes int64 indices (cuembed_embedding.cu:18-23).  When the table has fewer
    than 2^31 rows and the batch fewer than 2^31 lookups, the index work of the backward (row ids, radix
    sort, remap, scatter-add) runs on int32 copies: half the bytes through every sorting pass
    (C4 step 0.624 -> 0.597 ms for one 17 MB conversion; small batches are launch-bound and the two extra
    conversion launches cost more than they save -- B = 1024: 0.199 -> 0.228 ms -- hence the size gate)."""
    if idx.dtype == torch.int64 and num_categories < 2 ** 31 and (1 << 18) <= idx.numel() < 2 ** 31:
        return idx.to(torch.int32), (None if offsets is None else offsets.to(torch.int32))
    if offsets is not None and offsets.dtype != idx.dtype:
        # the forward takes indices and offsets of different integer types; the index work has ONE type for lookups and
        # sample ids, the one of the indices (offsets hold values <= nnz, which fits it)
        offsets = offsets.to(idx.dtype)
    return idx, offsets


def cuembed_backward(ctx, out_grad):
    idx, offsets, weights = ctx.saved_tensors
    nnz = idx.size(0)
    idx, offsets = _narrow_for_index_work(idx, offsets, ctx.num_categories)
    if getattr(ctx, "sparse_grad", False):
        return _sparse_backward(ctx, out_grad, idx, offsets, weights, nnz)
    # equivalent of nn.EmbeddingBag(include_last_offset=True).  (The reference slices offsets[:-1] for its
    # op, cuembed_pyt.py:23; the saved tensor already has the closing entry, so no copy is needed.)
    sample_ids = torch.ops.cuembed_pyt.cuembed_extract_row_ids_from_offsets(offsets, nnz)
    transpose_indices, transpose_sample_ids, transpose_weights = cuembed_transpose_sample_ids(
        sample_ids, idx, weights, ctx.num_categories)
    if transpose_weights.numel() == 0:  # forward ran without weights
        transpose_weights = None
    grad_embedding = cuembed_embedding_backward(out_grad, ctx.num_categories, transpose_indices,
                                                transpose_sample_ids, transpose_weights)
    return grad_embedding, None, None, None  # no grad for indices, offsets or weights


def _sparse_backward(ctx, out_grad, idx, offsets, weights, nnz):
    """Compressed gradient as a sparse COO tensor (like nn.EmbeddingBag(sparse=True)); True / "reference" / "blocked":
    coalesced, exactly num_unique ascending rows.  Reads num_unique back to the host (one sync), exactly like the
    reference's benchmark does (manual_benchmark.cu:392-394) -- except "padded", which never does."""
    width = out_grad.size(1)
    if weights is not None and weights.numel() != nnz:
        raise ValueError("weights must have one entry per index")
    if nnz == 0:
        return (torch.sparse_coo_tensor(torch.empty((1, 0), dtype=torch.int64, device=out_grad.device),
                                        torch.empty((0, width), dtype=out_grad.dtype, device=out_grad.device),
                                        size=(ctx.num_categories, width)), None, None, None)
    sample_ids = torch.ops.cuembed_pyt.cuembed_extract_row_ids_from_offsets(offsets, nnz)
    blocks = 1
    if ctx.sparse_grad == "padded":
        # min(lookups, rows) entries, the tail zero rows naming rows of the batch in turn: no read-back at any size
        t_idx, t_sid, t_w = cuembed_transpose_sample_ids(sample_ids, idx, weights, ctx.num_categories)
        remap = torch.ops.cuembed_pyt.cuembed_compute_compressed_grad_indices(t_idx)
        capacity = min(nnz, ctx.num_categories)
        rows = torch.empty((capacity, width), dtype=out_grad.dtype, device=out_grad.device)
        inv = torch.empty((capacity,), dtype=t_idx.dtype, device=out_grad.device)
        _ops.embedding_backward(out_grad.contiguous(), None, t_idx, t_sid, remap, t_w if t_w.numel() else None,
                                grad_embedding=rows, inverse_mapping=inv, pad_to_capacity=True)
        grad = torch.sparse_coo_tensor(inv.to(torch.int64).unsqueeze(0), rows, size=(ctx.num_categories, width),
                                       is_coalesced=False)
        return grad, None, None, None
    if ctx.sparse_grad not in (True, "reference"):
        # while a block of samples is scattered every L2 gathers from 1 / blocks of out_grad only
        blocks = _ops.recommended_sample_blocks(out_grad.dtype, width, out_grad.size(0), nnz)
        if ctx.sparse_grad == "blocked":
            # "blocked" promises the COALESCED tensor: never fall through to the uncoalesced path because the
            # recommendation (up to 64 blocks for very wide rows) exceeds what the coalescing remap handles
            blocks = min(blocks, _ops.MAX_COALESCED_BLOCKS)
    if blocks > 1 and ctx.sparse_grad == "blocked":
        # the COALESCED gradient from the blocked order: the same rows and ids as the fully sorted order gives
        # (ComputeCompressedGradIndicesBlocked + EmbeddingBackward(sample_blocks); EmbeddingBackward at C4 0.257 -> 0.232 ms)
        t_idx, t_sid, t_w = torch.ops.cuembed_pyt.cuembed_transpose_sample_blocks(sample_ids, idx, weights,
                                                                                  ctx.num_categories, blocks)
        if t_w.numel() == 0:
            t_w = None
        pairs, pair_rows, count = _ops.compute_compressed_grad_indices_blocked(t_idx, blocks)
        num_unique = int(count.item())
        rows, inv = _ops.embedding_backward(out_grad.contiguous(), num_unique, t_idx, t_sid, pairs, t_w,
                                            sample_blocks=blocks, block_row_ids=pair_rows)
        grad = torch.sparse_coo_tensor(inv.to(torch.int64).unsqueeze(0), rows, size=(ctx.num_categories, width),
                                       is_coalesced=True)
        return grad, None, None, None
    if blocks > 1:
        # one gradient row per (block of samples, table row): the fastest backward (C4: 0.257 -> 0.185 ms)
        t_idx, t_sid, t_w = torch.ops.cuembed_pyt.cuembed_transpose_sample_blocks(sample_ids, idx, weights,
                                                                                  ctx.num_categories, blocks)
    else:
        t_idx, t_sid, t_w = cuembed_transpose_sample_ids(sample_ids, idx, weights, ctx.num_categories)
    if t_w.numel() == 0:
        t_w = None
    remap = torch.ops.cuembed_pyt.cuembed_compute_compressed_grad_indices(t_idx)
    num_unique = int(remap[-1].item()) + 1
    rows, inv = torch.ops.cuembed_pyt.cuembed_embedding_backward_compressed(out_grad, num_unique, t_idx, t_sid,
                                                                            remap, t_w)
    grad = torch.sparse_coo_tensor(inv.to(torch.int64).unsqueeze(0), rows, size=(ctx.num_categories, width),
                                   is_coalesced=blocks == 1)
    return grad, None, None, None


class _CuEmbEmbedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, idx, offsets, weights=None, sparse_grad=False, hints=None):
        ctx.save_for_backward(idx, offsets, weights)
        ctx.num_categories = params.size(0)
        ctx.sparse_grad = sparse_grad
        # the weight gradient (an extension; the reference returns None) needs the table rows
        ctx.params_for_weight_grad = params.detach() if (weights is not None and weights.requires_grad) else None
        return cuembed_forward(params, idx, offsets, weights, hints)

    @staticmethod
    def backward(ctx, out_grad):
        if ctx.needs_input_grad[0]:
            grads = list(cuembed_backward(ctx, out_grad))
        else:
            grads = [None, None, None, None]
        if ctx.params_for_weight_grad is not None:
            idx, offsets, _ = ctx.saved_tensors
            grads[3] = torch.ops.cuembed_pyt.cuembed_embedding_weight_grad(
                ctx.params_for_weight_grad, idx, offsets, out_grad.to(ctx.params_for_weight_grad.dtype))
        return tuple(grads) + (None, None)


# (native CuEmbEmbeddingNode: GradKind)
_GRAD_KINDS = {False: 0, True: 2, "reference": 2, "fastest": 1, "uncoalesced": 3, "padded": 4}
_SPARSE_KINDS = (False, True, "reference", "fastest", "uncoalesced", "padded", "blocked")


def cuemb_embedding(params, idx, offsets, weights=None, sparse_grad=False, hints="auto"):
    """Sum-pooled embedding bag (offsets include the last offset), the reference's entry point
    (examples/pytorch/cuembed_pyt.py:48-51).  Differentiable w.r.t. params and -- an extension over the reference --
    w.r.t. the per-lookup weights.

    sparse_grad (extension; False = the reference's dense, table-sized gradient).  Every sparse kind densifies to the
    same gradient; they differ in which entries the COO tensor holds:
      True           params.grad is a COALESCED sparse COO tensor holding exactly the rows that were looked up, ascending
                     (grad._nnz() == number of distinct rows; grad.indices() / .values() work) -- the reference's fully
                     sorted order, on either backend and under torch.compile alike.  The row count is read back once
                     per step (after everything is enqueued), so the step cannot be captured into a HIP graph;
      "reference"    the same, by its older name;
      "blocked"      the same coalesced tensor computed from the sample-blocked order (a faster EmbeddingBackward for
                     more index work; through this op surface the two cancel at C4);
      "uncoalesced"  where the backward gains from scattering the batch in blocks of samples (C4: 0.56 -> 0.49 ms per
                     step) a row may appear once per block (is_coalesced=False; what torch.sparse consumers -- SGD,
                     SparseAdam, .coalesce(), .to_dense() -- take as it is); exactly one entry per (block, row);
      "padded"       one block, PADDED to min(lookups, rows) entries (zero rows naming rows of the batch in turn;
                     is_coalesced=False): the step never waits for the device and can be captured into a HIP graph --
                     the entry count of torch's own EmbeddingBag(sparse=True) gradient; a consumer that pays per entry
                     (torch.optim.SGD) is better served by True;
      "fastest"      whichever of the above is fastest for the shape (native backend: "padded" while the worst case
                     fits 64 MiB, else "uncoalesced"); the entry count is an implementation detail -- consume it with
                     .coalesce(), .to_dense() or an optimizer.
    hints="auto" lets cuembed_amd.policy pick the forward's scheduling options for this table and batch (non-temporal
    row loads, bag order); None = the library defaults.  Results never depend on hints.

    Outside torch.compile the step runs as ONE native autograd node (forward, and row ids -> transpose + remap ->
    scatter-add in the backward, one dispatcher hop each; libcuembed_pyt.so: CuEmbEmbeddingNode); under torch.compile,
    for the weight gradient and for "blocked" it runs as a Python autograd.Function over the same ops."""
    if not (isinstance(sparse_grad, bool) or (isinstance(sparse_grad, str) and sparse_grad in _SPARSE_KINDS)):
        raise ValueError("sparse_grad must be one of %r" % (_SPARSE_KINDS,))
    needs_grad = params.requires_grad or (weights is not None and weights.requires_grad)
    quiet = torch.compiler.is_compiling()
    chosen = _auto_hints(params, idx, offsets) if (hints == "auto" and not quiet) else None
    if chosen is not None and chosen[0] < 0 and chosen[1] is None and chosen[2] is None:
        chosen = None
    if not torch.is_grad_enabled() or not needs_grad:
        return cuembed_forward(params, idx, offsets, weights, chosen)
    native = (BACKEND == "native" and not quiet and sparse_grad in _GRAD_KINDS and
              not (weights is not None and weights.requires_grad))
    if native:
        row_loads, order, decision = chosen if chosen is not None else (-1, None, None)
        return torch.ops.cuembed_pyt.cuemb_embedding_step(params, idx, offsets, weights, _GRAD_KINDS[sparse_grad],
                                                          row_loads, order, decision)
    return _CuEmbEmbedding.apply(params, idx, offsets, weights, sparse_grad, chosen)


class _CuEmbFixed(torch.autograd.Function):
    """Fixed-hotness lookup: indices [B, H]; mode sum / mean -> [B, W], concat -> [B, H, W]."""

    @staticmethod
    def forward(ctx, params, idx, weights, mode):
        ctx.save_for_backward(idx, weights)
        ctx.num_categories = params.size(0)
        ctx.mode = mode
        return torch.ops.cuembed_pyt.cuembed_embedding_forward_fixed(params, idx, weights, mode)

    @staticmethod
    def backward(ctx, out_grad):
        idx, weights = ctx.saved_tensors
        batch, hot = idx.shape
        idx, _ = _narrow_for_index_work(idx, None, ctx.num_categories)
        if ctx.mode == "concat":   # every lookup has its own gradient row: sample id = position
            y = out_grad.reshape(batch * hot, -1)
            layout = idx.contiguous().view(batch * hot, 1)
        else:
            y = out_grad if ctx.mode == "sum" else out_grad * (1.0 / hot)
            layout = idx
        if ctx.mode == "mean" and weights is not None:      # out = sum(w x) / sum(w)
            y = out_grad / weights.sum(1, keepdim=True).to(out_grad.dtype)
        # row ids + transpose in one call; the sample ids are never materialised
        t_idx, t_sid, t_w, _ = cuembed_transpose_fixed_hotness(layout, weights, ctx.num_categories, False)
        if t_w.numel() == 0:
            t_w = None
        grad = cuembed_embedding_backward(y.contiguous(), ctx.num_categories, t_idx, t_sid, t_w)
        return grad, None, None, None


def cuemb_embedding_fixed(params, idx, weights=None, mode="sum"):
    """Fixed-hotness lookup (this library's extension of the reference's Python surface):
    idx is [batch, hotness]; weights (optional, sum/mean only) has the same shape."""
    if not torch.is_grad_enabled() or not params.requires_grad:
        return torch.ops.cuembed_pyt.cuembed_embedding_forward_fixed(params, idx, weights, mode)
    return _CuEmbFixed.apply(params, idx, weights, mode)


# Shape functions so that torch.compile can trace through the ops without running them.
@torch.library.register_fake("cuembed_pyt::cuembed_embedding_forward_fixed")
def _(params, indices, weights=None, mode="sum"):
    b, h = indices.shape
    shape = (b, h, params.shape[1]) if mode == "concat" else (b, params.shape[1])
    return torch.empty(shape, device=params.device, dtype=params.dtype)


@torch.library.register_fake("cuembed_pyt::cuembed_extract_row_ids_from_csr")
def _(offsets, nnz):
    return torch.empty((nnz,), device=offsets.device, dtype=offsets.dtype)


@torch.library.register_fake("cuembed_pyt::cuembed_extract_row_ids_from_offsets")
def _(offsets, nnz):
    return torch.empty((nnz,), device=offsets.device, dtype=offsets.dtype)


@torch.library.register_fake("cuembed_pyt::cuembed_transpose")
def _(rows, cols, weights=None):
    n = 0 if weights is None else cols.shape[0]
    return (torch.empty_like(cols), torch.empty_like(rows),
            torch.empty((n,), device=rows.device, dtype=torch.float32 if weights is None else weights.dtype))


@torch.library.register_fake("cuembed_pyt::cuembed_embedding_weight_grad")
def _(params, indices, offsets, y_grad):
    return torch.empty((indices.shape[0],), device=params.device, dtype=params.dtype)


@torch.library.register_fake("cuembed_pyt::cuembed_transpose_bounded")
def _(rows, cols, weights=None, num_categories=0):
    n = 0 if weights is None else cols.shape[0]
    return (torch.empty_like(cols), torch.empty_like(rows),
            torch.empty((n,), device=rows.device, dtype=torch.float32 if weights is None else weights.dtype))


@torch.library.register_fake("cuembed_pyt::cuembed_transpose_sample_ids")
def _(sample_ids, indices, weights=None, num_categories=0):
    n = 0 if weights is None else indices.shape[0]
    return (torch.empty_like(indices), torch.empty_like(sample_ids),
            torch.empty((n,), device=indices.device, dtype=torch.float32 if weights is None else weights.dtype))


@torch.library.register_fake("cuembed_pyt::cuembed_transpose_sample_blocks")
def _(sample_ids, indices, weights=None, num_categories=0, sample_blocks=1):
    n = 0 if weights is None else indices.shape[0]
    return (torch.empty_like(indices), torch.empty_like(sample_ids),
            torch.empty((n,), device=indices.device, dtype=torch.float32 if weights is None else weights.dtype))


@torch.library.register_fake("cuembed_pyt::cuembed_transpose_fixed_hotness")
def _(indices, weights=None, num_categories=0, compressed=False):
    nnz = indices.shape[0] * indices.shape[1]
    flat = dict(device=indices.device, dtype=indices.dtype)
    return (torch.empty((nnz,), **flat), torch.empty((nnz,), **flat),
            torch.empty((0 if weights is None else nnz,), device=indices.device,
                        dtype=torch.float32 if weights is None else weights.dtype),
            torch.empty((nnz if compressed else 0,), **flat))


@torch.library.register_fake("cuembed_pyt::cuembed_compute_compressed_grad_indices")
def _(transpose_indices):
    return torch.empty_like(transpose_indices)


@torch.library.register_fake("cuembed_pyt::cuembed_embedding_backward_compressed")
def _(y_grad, num_unique, transpose_indices, transpose_sample_ids, transpose_remapped_indices, transpose_weights=None):
    return (torch.empty((num_unique, y_grad.shape[1]), device=y_grad.device, dtype=y_grad.dtype),
            torch.empty((num_unique,), device=y_grad.device, dtype=transpose_indices.dtype))


@torch.library.register_fake("cuembed_pyt::cuembed_embedding_forward")
def _(params, idx, offsets, weights=None, mode="sum"):
    return torch.empty((offsets.shape[0] - 1, params.shape[1]), device=params.device, dtype=params.dtype)


@torch.library.register_fake("cuembed_pyt::cuembed_decide_row_loads")
def _(indices, table_bytes, decision):
    return None


@torch.library.register_fake("cuembed_pyt::cuembed_embedding_forward_hinted")
def _(params, idx, offsets, weights=None, mode="sum", row_loads=-1, sample_order=None, row_loads_device=None):
    return torch.empty((offsets.shape[0] - 1, params.shape[1]), device=params.device, dtype=params.dtype)


@torch.library.register_fake("cuembed_pyt::cuembed_bag_order_by_length")
def _(offsets, max_length=0):
    return torch.empty((offsets.shape[0] - 1,), device=offsets.device, dtype=torch.int32)


@torch.library.register_fake("cuembed_pyt::cuembed_embedding_backward")
def _(y_grad, num_categories, transpose_indices, transpose_sample_ids, transpose_weights=None):
    return torch.empty((num_categories, y_grad.shape[1]), device=y_grad.device, dtype=y_grad.dtype)
