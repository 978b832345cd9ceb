"""Host-side mirror of the reference's C++ host API on torch device tensors.

Each function validates its arguments the way the reference does (raising
instead of aborting, like the reference's torch binding does with AT_ASSERT),
then enqueues the HIP kernels on torch's current stream through the C ABI.
torch is used for device memory and streams only.

    embedding_forward              <-> cuembed::EmbeddingForward   (embedding_lookup.cuh:245-308)
    embedding_backward             <-> cuembed::EmbeddingBackward  (embedding_lookup.cuh:423-483)
    transpose                      <-> cuembed::Transpose          (index_transforms.cuh:224-250)
    compute_compressed_grad_indices<-> ComputeCompressedGradIndices(index_transforms.cuh:278-323)
    extract_row_ids_from_fixed/csr/for_concat <-> index_transforms.cuh:45-93
"""
import ctypes

import torch

from . import _lib

SUM, MEAN, CONCAT = 0, 1, 2
_MODES = {"sum": SUM, "mean": MEAN, "concat": CONCAT, SUM: SUM, MEAN: MEAN, CONCAT: CONCAT}
_ELEM = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
_INDEX = {torch.int32: 0, torch.int64: 1}


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _check_dev(name, t, device=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("cuembed_amd: %s must live on the GPU (no CPU fallback exists)" % name)
    if device is not None and t.device != device:
        raise RuntimeError("cuembed_amd: %s is on %s, expected %s" % (name, t.device, device))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)


def _elem_code(name, t):
    if t.dtype not in _ELEM:
        raise TypeError("%s must be float32, float16 or bfloat16, got %s" % (name, t.dtype))
    return _ELEM[t.dtype]


def _index_code(name, t):
    if t.dtype not in _INDEX:
        raise TypeError("%s must be int32 or int64, got %s" % (name, t.dtype))
    return _INDEX[t.dtype]


def set_forward_reduction_order(order):
    """"sequential" (default, bit-identical to the reference for every batch size) or "split"
    (small batches may split a sample's hotness loop over wavefronts; equal up to fp rounding)."""
    if order not in ("sequential", "split"):
        raise ValueError("order must be 'sequential' or 'split'")
    _lib.lib().cuembed_set_forward_reduction_order(1 if order == "split" else 0)


def get_forward_reduction_order():
    return "split" if _lib.lib().cuembed_get_forward_reduction_order() else "sequential"


_ORDERS = {None: -1, "sequential": 0, "split": 1}
_ROW_LOADS = {None: -1, "default": 0, "streaming": 1}


def set_forward_row_load_policy(policy):
    """Process-wide default of how embedding_forward (sum / mean) loads table rows: "default" (rows stay in L2 /
    Infinity Cache: right whenever rows are re-used) or "streaming" (non-temporal loads: for batches in which nearly
    every lookup hits a different row of a table far larger than the caches).  Never changes a result.  Prefer the
    per-call `row_loads=` argument of embedding_forward."""
    if policy not in ("default", "streaming"):
        raise ValueError("policy must be 'default' or 'streaming'")
    _lib.lib().cuembed_set_forward_row_load_policy(_ROW_LOADS[policy])


def set_forward_wide_load(mode="auto"):
    """Tuning / tests (never changes a result): whether small batches take the wide-load forward kernel (one sample per
    workgroup, a bag's rows requested at once, pooled in lookup order) -- "auto" (the launcher decides), "never", "always"
    (whenever the row shape allows it), "always2" / "always4" / ... (with that many samples per workgroup, as far as the row
    allows)."""
    codes = {"auto": 0, "never": 1, "always": 2, "always2": 3, "always4": 4, "always8": 5, "always16": 6}
    _lib.lib().cuembed_set_forward_wide_load(codes[mode])


def get_forward_row_load_policy():
    return "streaming" if _lib.lib().cuembed_get_forward_row_load_policy() else "default"


def forward_launch_shape(elem_dtype, index_dtype, embed_width, batch_size, num_hots, is_csr=False,
                         is_weighted=False, mode="sum"):
    """Launch shape the forward kernel would use (pure host arithmetic)."""
    out = (ctypes.c_int * 6)()
    _lib.lib().cuembed_forward_launch_shape(_ELEM[elem_dtype], _INDEX[index_dtype], embed_width,
                                            batch_size, num_hots, int(is_csr), int(is_weighted),
                                            _MODES[mode], out)
    return dict(elems_per_lane=out[0], lanes_per_row=out[1], samples_per_block=out[2],
                grid=out[3], lds_bytes=out[4], staged=out[5] == 1, wide_load=out[5] == 2)


def device_shape():
    """What the launch heuristics know about the current device (needs a GPU): compute units, XCDs, resident lanes
    per compute unit, L2 bytes per XCD."""
    out = (ctypes.c_int * 4)()
    _lib.lib().cuembed_device_shape(out)
    return dict(compute_units=out[0], xcds=out[1], lanes_per_cu=out[2], l2_bytes_per_xcd=out[3])


def backward_launch_shape(elem_dtype, index_dtype, embed_width, nnz, is_weighted=False, compute_units=0, xcds=0):
    """Launch shape EmbeddingBackward would use (pure host arithmetic when compute_units > 0 describes the device:
    e.g. compute_units=32, xcds=1 for a CPX partition; 0 = ask the current device)."""
    out = (ctypes.c_int * 8)()
    _lib.lib().cuembed_backward_launch_shape(_ELEM[elem_dtype], _INDEX[index_dtype], embed_width, int(nnz),
                                             int(is_weighted), int(compute_units), int(xcds), out)
    return dict(column_slices=out[0], lanes=out[1], segments_per_block=out[2], segment_len=out[3], nz_blocks=out[4],
                grid=out[5], lds_bytes=out[6], xcds=out[7])


def embedding_forward(params, indices, offsets=None, weights=None, batch_size=None, num_hots=0,
                      mode="sum", fp16_math=False, out=None, reduction_order=None, row_loads=None, sample_order=None,
                      row_loads_device=None):
    """out[s] = combine_j weights[s,j] * params[indices[s,j]].

    Fixed hotness: offsets=None, num_hots>0 (indices holds batch_size*num_hots ids).
    CSR: offsets[batch_size+1], num_hots=0.  mode: "sum" | "mean" | "concat".
    Returns [batch, width] (sum/mean) or [batch, num_hots, width] (concat).
    Per-call options (extension; None = the process-wide default): reduction_order "sequential" | "split"
    (set_forward_reduction_order), row_loads "default" | "streaming" (set_forward_row_load_policy);
    sample_order (CSR only): an int32 permutation of range(batch_size) on the device, the order in which the samples
    are handed to the wavefronts -- a scheduling hint that never changes a result (bag_order_by_length());
    row_loads_device: the 4-word int32 device tensor decide_row_loads() filled -- the kernels read the decision from it
    instead of row_loads (the host never sees it)."""
    if mode not in _MODES:
        raise ValueError("mode must be 'sum', 'mean' or 'concat'")
    if reduction_order not in _ORDERS or row_loads not in _ROW_LOADS:
        raise ValueError("reduction_order: None, 'sequential' or 'split'; row_loads: None, 'default' or 'streaming'")
    m = _MODES[mode]
    _check_dev("params", params)
    dev = params.device
    _check_dev("indices", indices, dev)
    if params.dim() != 2:
        raise ValueError("params must be [rows, width]")
    et = _elem_code("params", params)
    it = _index_code("indices", indices)
    # --- the reference's argument contract (embedding_lookup.cuh:261-267) ---
    if weights is not None and m == CONCAT:
        raise ValueError("concat does not take weights")
    if not ((offsets is not None and num_hots == 0) or (offsets is None and num_hots > 0)):
        raise ValueError("either CSR (offsets given, num_hots == 0) or fixed hotness "
                         "(offsets None, num_hots > 0)")
    if offsets is not None and m == CONCAT:
        raise ValueError("CSR layout does not support concat")
    width = params.shape[1]
    if (width * params.element_size()) % 4 != 0:
        raise ValueError("row size must be a multiple of 4 bytes")
    ot = 0
    if offsets is not None:
        _check_dev("offsets", offsets, dev)
        ot = _index_code("offsets", offsets)
        if batch_size is None:
            batch_size = offsets.numel() - 1
        if offsets.numel() < batch_size + 1:
            raise ValueError("offsets must hold batch_size + 1 entries")
    else:
        if batch_size is None:
            if indices.numel() % num_hots:
                raise ValueError("indices.numel() is not a multiple of num_hots")
            batch_size = indices.numel() // num_hots
        if indices.numel() < batch_size * num_hots:
            raise ValueError("indices must hold batch_size * num_hots entries")
    if weights is not None:
        _check_dev("weights", weights, dev)
        if weights.dtype != params.dtype:
            raise TypeError("weights must have the table's dtype")
        if weights.numel() < indices.numel():
            raise ValueError("weights must have one entry per index")
    if sample_order is not None:
        _check_dev("sample_order", sample_order, dev)
        if offsets is None:
            raise ValueError("sample_order is a hint for CSR batches (bags of different lengths)")
        if sample_order.dtype != torch.int32 or sample_order.numel() != batch_size or not sample_order.is_contiguous():
            raise ValueError("sample_order must be a contiguous int32 permutation of range(batch_size)")
    if row_loads_device is not None:
        _check_dev("row_loads_device", row_loads_device, dev)
        if row_loads_device.dtype != torch.int32 or row_loads_device.numel() < 4 or not row_loads_device.is_contiguous():
            raise ValueError("row_loads_device must be the contiguous 4-word int32 tensor decide_row_loads() fills")
    shape = (batch_size, num_hots, width) if m == CONCAT else (batch_size, width)
    if out is None:
        out = torch.empty(shape, dtype=params.dtype, device=dev)
    else:
        _check_dev("out", out, dev)
        if out.dtype != params.dtype or out.numel() != batch_size * (num_hots if m == CONCAT else 1) * width:
            raise ValueError("out has the wrong dtype or size")
    if batch_size > 0:
        with torch.cuda.device(params.device):   # the launch must happen on the tensors' device
            _lib.lib().cuembed_embedding_forward_device_hints(
                _ptr(params), et, width, _ptr(indices), it, _ptr(offsets), ot, _ptr(weights),
                batch_size, num_hots, m, int(bool(fp16_math)), _ptr(out), _ORDERS[reduction_order],
                _ROW_LOADS[row_loads], _ptr(sample_order), _ptr(row_loads_device), _stream(params))
    return out


def decide_row_loads(indices, table_bytes, decision=None, distinct_fraction=None):
    """The row-load policy of embedding_forward decided ON THE DEVICE from the batch's own indices
    (cuembed::DecideRowLoads): one launch, no read-back, capturable.  Returns `decision`, a 4-word int32 device tensor
    (word 0: 1 = non-temporal row loads, 0 = ordinary; allocate it once with new_row_loads_decision() and re-use it) to
    pass as embedding_forward(..., row_loads_device=).  Streaming is chosen when at least `distinct_fraction` (default
    0.998, i.e. at most 8 repeats per group of 4,096: the measured crossover) of an evenly strided sample of up to 65,536 lookups names distinct rows, the table has >= 1 GiB and the batch
    >= 2^18 lookups.  Never changes a result."""
    _check_dev("indices", indices)
    it = _index_code("indices", indices)
    if decision is None:
        decision = new_row_loads_decision(indices.device)
    _check_dev("decision", decision, indices.device)
    if decision.dtype != torch.int32 or decision.numel() < 4 or not decision.is_contiguous():
        raise ValueError("decision must be a contiguous int32 tensor of 4 words (new_row_loads_decision())")
    th = 0 if distinct_fraction is None else max(1, min(65536, int(round(float(distinct_fraction) * 65536))))
    with torch.cuda.device(indices.device):
        _lib.lib().cuembed_decide_row_loads(_ptr(indices.contiguous()), it, indices.numel(), int(table_bytes),
                                            _ptr(decision), th, _stream(indices))
    return decision


def new_row_loads_decision(device="cuda"):
    """The four zeroed device words decide_row_loads() works on (word 0 is the decision: "default" until decided)."""
    return torch.zeros((4,), dtype=torch.int32, device=device)


def bag_order_by_length(offsets, batch_size=None, max_length=None, workspace=None):
    """int32 permutation of range(batch_size): the samples of a CSR batch by DESCENDING bag length (ties in input
    order) -- what embedding_forward(..., sample_order=) wants for ragged bags: the two bags of a wavefront are alike,
    neighbouring wavefronts are alike, and the longest bags start first (cuembed::BagOrderByLength).  max_length, when
    the caller knows a bound on the bag length (longer bags rank as max_length), keeps the sort short: with a bound
    of at most 255 -- or max_length=-1, "bags of 255 lookups and more rank alike" -- and up to 131,072 samples it is
    two small launches (a stable counting sort, 7 us for 65,536 bags); None = unknown: a key kernel + the library's stable
    sort.  It only depends on the offsets."""
    _check_dev("offsets", offsets)
    ot = _index_code("offsets", offsets)
    if batch_size is None:
        batch_size = offsets.numel() - 1
    if batch_size < 0 or offsets.numel() < batch_size + 1:
        raise ValueError("offsets must hold batch_size + 1 entries")
    bound = 0 if max_length is None else int(max_length)     # (< 0: bags of 255 lookups and more rank alike)
    order = torch.empty((batch_size,), dtype=torch.int32, device=offsets.device)
    need = ctypes.c_size_t(0)
    _lib.lib().cuembed_bag_order_by_length(None, ot, batch_size, bound, None, None, ctypes.byref(need), None)
    if workspace is None:
        workspace = torch.empty((max(need.value, 1),), dtype=torch.uint8, device=offsets.device)
    elif workspace.numel() * workspace.element_size() < need.value:
        raise ValueError("workspace too small: need %d bytes" % need.value)
    lwork = ctypes.c_size_t(workspace.numel() * workspace.element_size())
    if batch_size > 0:
        with torch.cuda.device(offsets.device):
            _lib.lib().cuembed_bag_order_by_length(_ptr(offsets), ot, batch_size, bound, _ptr(order), _ptr(workspace),
                                                   ctypes.byref(lwork), _stream(offsets))
    return order


def embedding_weight_grad(params, indices, grad_y, offsets=None, batch_size=None, num_hots=0):
    """grad_weights[s, j] = dot(params[indices[s, j]], grad_y[s]) -- the gradient of a weighted
    sum-forward w.r.t. its per-lookup weights (extension).  Returns a tensor with one entry per lookup."""
    _check_dev("params", params)
    dev = params.device
    _check_dev("indices", indices, dev)
    _check_dev("grad_y", grad_y, dev)
    et = _elem_code("params", params)
    it = _index_code("indices", indices)
    if grad_y.dtype != params.dtype or grad_y.dim() != 2 or grad_y.shape[1] != params.shape[1]:
        raise ValueError("grad_y must be [batch, width] of the table's dtype")
    if not ((offsets is not None and num_hots == 0) or (offsets is None and num_hots > 0)):
        raise ValueError("either CSR (offsets given, num_hots == 0) or fixed hotness")
    ot = 0
    if offsets is not None:
        _check_dev("offsets", offsets, dev)
        ot = _index_code("offsets", offsets)
        if batch_size is None:
            batch_size = offsets.numel() - 1
    elif batch_size is None:
        batch_size = indices.numel() // num_hots
    if grad_y.shape[0] < batch_size:
        raise ValueError("grad_y must have one row per sample")
    out = torch.empty((indices.numel(),), dtype=params.dtype, device=dev)
    if batch_size > 0 and indices.numel() > 0:
        with torch.cuda.device(params.device):   # the launch must happen on the tensors' device
            _lib.lib().cuembed_embedding_weight_grad(_ptr(params), et, params.shape[1], _ptr(indices), it,
                                                     _ptr(offsets), ot, _ptr(grad_y), batch_size, num_hots,
                                                     _ptr(out), _stream(params))
    return out


def set_backward_tuning(segment_len=0, column_slices=0):
    """Launch-shape overrides of the backward kernels (tuning / tests; 0 = built-in heuristic).
    Results never depend on them."""
    _lib.lib().cuembed_set_backward_tuning(int(segment_len), int(column_slices))


def get_backward_tuning():
    out = (ctypes.c_int * 2)()
    _lib.lib().cuembed_get_backward_tuning(out)
    return dict(segment_len=out[0], column_slices=out[1])


def embedding_backward(grad_y, num_grad_embedding_rows, transpose_indices, transpose_sample_ids,
                       transpose_remapped_indices=None, transpose_weights=None,
                       skip_grad_init=False, grad_embedding=None, inverse_mapping=None, sample_blocks=1,
                       block_row_ids=None, reference_sums=False, pad_to_capacity=False):
    """Scatter-add grad_y rows into the table gradient from index-sorted COO lookups.

    Full gradient: transpose_remapped_indices=None, num_grad_embedding_rows = table rows.
    Compressed: remapped indices given, num_grad_embedding_rows = num_unique; also returns
    inverse_mapping[num_unique].  With skip_grad_init=True the caller's grad_embedding must
    already be zero.  Returns (grad_embedding, inverse_mapping or None).

    Extension (compressed only): num_grad_embedding_rows=None = "num_unique is only known on the device"
    (it is transpose_remapped_indices[-1] + 1).  grad_embedding and inverse_mapping must then be given
    with at least that many rows; rows past the last id are left untouched and no host read-back is
    needed before the call.  Capacity that always suffices: nnz; min(nnz, table rows) when the ids
    are those of a fully sorted COO or of compute_compressed_grad_indices_blocked (one id per distinct
    table row); min(nnz, sample_blocks * table rows) for the UNCOALESCED gradient that
    compute_compressed_grad_indices yields on a sample-blocked transpose (one id per (block, row)).

    Extension (compressed only): sample_blocks > 1 = the COO is transpose(..., sample_blocks=...) and
    (transpose_remapped_indices, block_row_ids) come from compute_compressed_grad_indices_blocked(...,
    sample_blocks): the blocks are scattered one after the other (every L2 gathers from 1 / sample_blocks of
    grad_y at a time) and the result has the reference's layout: num_unique ascending rows, the fully sorted
    order's inverse_mapping.

    pad_to_capacity=True (with num_grad_embedding_rows=None, one block, skip_grad_init=False): the rows from the
    device-side count up to the buffers' row count are zeroed and their inverse_mapping entries name rows of
    the batch, different ones in turn -- (inverse_mapping, grad_embedding) as a whole is then a valid uncoalesced COO gradient whose
    row count never has to be read back.

    Extension: reference_sums=True computes every `grad += grad_y * weight` in the gradient's own type, lookup by
    lookup in nz order, like the CPU reference (cuembed::EmbeddingBackwardReferenceSums): bit-identical to it for any
    fp16 / bf16 / fp32 data, slow for rows that are looked up very often.  Host-known row count, one block."""
    _check_dev("grad_y", grad_y)
    dev = grad_y.device
    if grad_y.dim() != 2:
        raise ValueError("grad_y must be [rows, width]")
    et = _elem_code("grad_y", grad_y)
    _check_dev("transpose_indices", transpose_indices, dev)
    _check_dev("transpose_sample_ids", transpose_sample_ids, dev)
    it = _index_code("transpose_indices", transpose_indices)
    if transpose_sample_ids.dtype != transpose_indices.dtype:
        raise TypeError("transpose_sample_ids must have the dtype of transpose_indices")
    nnz = transpose_indices.numel()
    if transpose_sample_ids.numel() != nnz:
        raise ValueError("transpose_sample_ids must have nnz entries")
    width = grad_y.shape[1]
    compressed = transpose_remapped_indices is not None
    unknown_rows = num_grad_embedding_rows is None
    if unknown_rows:
        if not compressed or grad_embedding is None or inverse_mapping is None:
            raise ValueError("num_grad_embedding_rows=None needs a compressed call with grad_embedding and "
                             "inverse_mapping buffers given")
        if grad_embedding.dim() != 2 or grad_embedding.shape[1] != width or \
                inverse_mapping.numel() < grad_embedding.shape[0]:
            raise ValueError("grad_embedding must be [capacity, width] and inverse_mapping hold capacity entries")
        num_grad_embedding_rows = grad_embedding.shape[0]
    if compressed:
        _check_dev("transpose_remapped_indices", transpose_remapped_indices, dev)
        if transpose_remapped_indices.dtype != transpose_indices.dtype or \
                transpose_remapped_indices.numel() != nnz:
            raise ValueError("transpose_remapped_indices must match transpose_indices")
    if transpose_weights is not None:
        _check_dev("transpose_weights", transpose_weights, dev)
        if transpose_weights.dtype != grad_y.dtype or transpose_weights.numel() != nnz:
            raise ValueError("transpose_weights must be nnz entries of grad_y's dtype")
    if grad_embedding is None:
        if skip_grad_init:
            grad_embedding = torch.zeros((num_grad_embedding_rows, width), dtype=grad_y.dtype, device=dev)
        else:
            grad_embedding = torch.empty((num_grad_embedding_rows, width), dtype=grad_y.dtype, device=dev)
    else:
        _check_dev("grad_embedding", grad_embedding, dev)
        if grad_embedding.dtype != grad_y.dtype or grad_embedding.numel() != num_grad_embedding_rows * width:
            raise ValueError("grad_embedding has the wrong dtype or size")
    if compressed:
        if inverse_mapping is None:
            inverse_mapping = torch.empty((num_grad_embedding_rows,), dtype=transpose_indices.dtype, device=dev)
        else:
            _check_dev("inverse_mapping", inverse_mapping, dev)
            if inverse_mapping.dtype != transpose_indices.dtype:
                raise TypeError("inverse_mapping must have the index dtype")
    else:
        inverse_mapping = None
    if sample_blocks > 1 and not compressed:
        raise ValueError("sample_blocks > 1 is for the compressed gradient only")
    if _check_coalesced_blocks(nnz, sample_blocks) > 1:
        if block_row_ids is None:
            raise ValueError("a sample-blocked compressed gradient needs block_row_ids "
                             "(compute_compressed_grad_indices_blocked)")
        _check_dev("block_row_ids", block_row_ids, dev)
        if block_row_ids.dtype != torch.int32:
            raise TypeError("block_row_ids must be int32 (uint32 bit patterns)")
    else:
        block_row_ids = None
    if reference_sums:
        if unknown_rows or sample_blocks > 1:
            raise ValueError("reference_sums needs a host-known row count and a fully sorted order")
        with torch.cuda.device(grad_y.device):
            _lib.lib().cuembed_embedding_backward_reference_sums(
                _ptr(grad_y), et, width, num_grad_embedding_rows, nnz, _ptr(transpose_indices), _ptr(transpose_sample_ids),
                _ptr(transpose_remapped_indices), it, _ptr(transpose_weights), int(bool(skip_grad_init)),
                _ptr(grad_embedding), _ptr(inverse_mapping), _stream(grad_y))
        return grad_embedding, inverse_mapping
    # num_unique on the device only: the buffers' row count is the capacity the kernels check against -- too small,
    # and nothing is written and the device's sticky overflow word is raised (capacity_overflowed()) -- never an overrun
    capacity = min(grad_embedding.shape[0], inverse_mapping.numel()) if unknown_rows else 0
    if pad_to_capacity and (not unknown_rows or skip_grad_init or sample_blocks > 1):
        raise ValueError("pad_to_capacity needs num_grad_embedding_rows=None, skip_grad_init=False and one block")
    with torch.cuda.device(grad_y.device):   # the launch must happen on the tensors' device
        _lib.lib().cuembed_embedding_backward_bounded(
            _ptr(grad_y), et, width, -1 if unknown_rows else num_grad_embedding_rows, nnz, _ptr(transpose_indices),
            _ptr(transpose_sample_ids), _ptr(transpose_remapped_indices), it, _ptr(transpose_weights),
            int(bool(skip_grad_init)), _ptr(grad_embedding), _ptr(inverse_mapping), int(sample_blocks),
            _ptr(block_row_ids), int(capacity), _ptr(_overflow_word(dev)) if unknown_rows else None,
            int(bool(pad_to_capacity)), _stream(grad_y))
    return grad_embedding, inverse_mapping


_OVERFLOW_WORDS = {}


def _overflow_word(device):
    """The device's sticky "a compressed gradient did not fit its buffers" word (torch owns the 4 bytes; the library
    keeps no state of its own)."""
    key = (device.type, torch.cuda.current_device() if device.index is None else device.index)
    w = _OVERFLOW_WORDS.get(key)
    if w is None:
        w = _OVERFLOW_WORDS[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return w


def capacity_overflowed(device=None, reset=False):
    """True if any embedding_backward(num_grad_embedding_rows=None, ...) on `device` since the last reset found more
    gradient rows on the device than its grad_embedding / inverse_mapping buffers hold (such a call writes nothing).
    Synchronises with the device (one 4-byte read-back): call it where a sync is affordable -- e.g. once per epoch."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    w = _OVERFLOW_WORDS.get((device.type, torch.cuda.current_device() if device.index is None else device.index))
    if w is None:
        return False
    hit = bool(w.item())
    if reset:
        w.zero_()
    return hit


def transpose_workspace_bytes(nnz, index_dtype, weight_dtype=None):
    """Phase 1 of the reference's two-phase call: scratch bytes Transpose needs."""
    lwork = ctypes.c_size_t(0)
    wt = 0 if weight_dtype is None else _ELEM[weight_dtype]
    fake = ctypes.c_void_p(256)  # only nullness of `weights` matters for the query
    _lib.lib().cuembed_transpose(None, None, None if weight_dtype is None else fake, nnz,
                                 _INDEX[index_dtype], wt, None, None, None, None,
                                 ctypes.byref(lwork), None)
    return lwork.value


def transpose(rows, cols, weights=None, workspace=None, num_categories=None, num_rows=None, sample_blocks=1,
              remapped=False):
    """Stable sort of (rows[i][, weights[i]]) by key cols[i] (callers pass rows = sample ids,
    cols = lookup indices).  Returns (sorted cols, rows carried along, weights carried along).
    Generic like the reference's: `cols` are ordered as signed numbers, `rows` may hold anything.
    num_categories (optional, this library's extension): every value in `cols` lies in
    [0, num_categories); the sort then skips the key digits that are always zero (same result,
    fewer passes).  num_rows (optional, extension): every value in `rows` lies in [0, num_rows)
    -- int64 rows below 2^32 then travel as 32 bits without the library reading them to find out.
    sample_blocks (optional, extension, CHANGES the result): > 1 cuts the (sample-major) input into that many
    consecutive blocks and transposes each on its own -- compressed-gradient path only, see
    recommended_sample_blocks().
    remapped=True (extension): a fourth result, what compute_compressed_grad_indices(sorted cols) returns, from the same
    call -- up to 4,096 lookups the whole index work is then ONE kernel launch, up to 229,376 one launch per radix pass."""
    _check_dev("rows", rows)
    dev = rows.device
    _check_dev("cols", cols, dev)
    it = _index_code("rows", rows)
    if cols.dtype != rows.dtype or cols.numel() != rows.numel():
        raise ValueError("rows and cols must have the same dtype and length")
    nnz = rows.numel()
    wt = 0
    t_w = None
    if weights is not None:
        _check_dev("weights", weights, dev)
        wt = _elem_code("weights", weights)
        if weights.numel() != nnz:
            raise ValueError("weights must have nnz entries")
        t_w = torch.empty_like(weights)
    t_rows = torch.empty_like(cols)
    t_cols = torch.empty_like(rows)
    need = transpose_workspace_bytes(nnz, rows.dtype, None if weights is None else weights.dtype)
    if workspace is None:
        workspace = torch.empty((max(need, 1),), dtype=torch.uint8, device=dev)
    elif workspace.numel() * workspace.element_size() < need:
        raise ValueError("workspace too small: need %d bytes" % need)
    lwork = ctypes.c_size_t(workspace.numel() * workspace.element_size())
    bits = 0 if not num_categories else max(1, int(num_categories - 1).bit_length())
    row_bits = 0 if not num_rows else max(1, int(num_rows - 1).bit_length())
    remap = torch.empty_like(cols) if remapped else None
    if nnz > 0:
        with torch.cuda.device(rows.device):   # the launch must happen on the tensors' device
            _lib.lib().cuembed_transpose_remapped(_ptr(rows), _ptr(cols), _ptr(weights), nnz, it, wt,
                                                  _ptr(t_rows), _ptr(t_cols), _ptr(t_w), _ptr(remap), _ptr(workspace),
                                                  ctypes.byref(lwork), bits, row_bits, int(sample_blocks),
                                                  _stream(rows))
    return (t_rows, t_cols, t_w, remap) if remapped else (t_rows, t_cols, t_w)


def transpose_sample_block_length(nnz, sample_blocks):
    """Lookups per block of transpose(..., sample_blocks=...): block k = lookups [k * L, (k + 1) * L)."""
    return int(_lib.lib().cuembed_transpose_sample_block_length(int(nnz), int(sample_blocks)))


def recommended_sample_blocks(grad_dtype, embed_width, batch_size, nnz, compute_units=0, xcds=0, l2_bytes_per_xcd=0):
    """How many sample blocks a transpose feeding a COMPRESSED EmbeddingBackward should use so that the part of
    grad_y one L2 gathers from fits it (cuembed::RecommendedSampleBlocks; 1 = nothing to gain).  For the current
    device, or (compute_units > 0) for a described one."""
    return int(_lib.lib().cuembed_recommended_sample_blocks_on(_ELEM[grad_dtype], int(embed_width), int(batch_size),
                                                               int(nnz), int(compute_units), int(xcds),
                                                               int(l2_bytes_per_xcd)))


def transpose_fixed_hotness(indices, batch_size, num_hots, weights=None, workspace=None, num_categories=None,
                            sample_blocks=1, remapped=False):
    """extract_row_ids_from_fixed + transpose in one call without materialising the sample ids
    (cuembed::TransposeFixedHotness, extension): returns (sorted indices, sample ids, weights).
    num_hots=1 is the concat layout.  sample_blocks and remapped (a fourth result: the compressed-gradient ids of the
    sorted indices) as in transpose()."""
    _check_dev("indices", indices)
    dev = indices.device
    it = _index_code("indices", indices)
    nnz = batch_size * num_hots
    if num_hots <= 0 or indices.numel() < nnz:
        raise ValueError("indices must hold batch_size * num_hots entries, num_hots > 0")
    wt = 0
    t_w = None
    if weights is not None:
        _check_dev("weights", weights, dev)
        wt = _elem_code("weights", weights)
        if weights.numel() < nnz:
            raise ValueError("weights must have one entry per index")
        t_w = torch.empty((nnz,), dtype=weights.dtype, device=dev)
    t_idx = torch.empty((nnz,), dtype=indices.dtype, device=dev)
    t_sid = torch.empty((nnz,), dtype=indices.dtype, device=dev)
    need = transpose_workspace_bytes(nnz, indices.dtype, None if weights is None else weights.dtype)
    if workspace is None:
        workspace = torch.empty((max(need, 1),), dtype=torch.uint8, device=dev)
    elif workspace.numel() * workspace.element_size() < need:
        raise ValueError("workspace too small: need %d bytes" % need)
    lwork = ctypes.c_size_t(workspace.numel() * workspace.element_size())
    bits = 0 if not num_categories else max(1, int(num_categories - 1).bit_length())
    remap = torch.empty((nnz,), dtype=indices.dtype, device=dev) if remapped else None
    if nnz > 0:
        with torch.cuda.device(dev):
            _lib.lib().cuembed_transpose_fixed_hotness_remapped(
                _ptr(indices), _ptr(weights), batch_size, num_hots, it, wt, _ptr(t_idx), _ptr(t_sid), _ptr(t_w),
                _ptr(remap), _ptr(workspace), ctypes.byref(lwork), bits, int(sample_blocks), _stream(indices))
    return (t_idx, t_sid, t_w, remap) if remapped else (t_idx, t_sid, t_w)


def compressed_grad_workspace_bytes(nnz, index_dtype):
    lwork = ctypes.c_size_t(0)
    _lib.lib().cuembed_compute_compressed_grad_indices(None, nnz, _INDEX[index_dtype], None, None,
                                                       ctypes.byref(lwork), None)
    return lwork.value


def compute_compressed_grad_indices(indices, workspace=None):
    """[4,4,7,8,8,8,18] -> [0,0,1,2,2,2,3] for index-sorted lookups."""
    _check_dev("indices", indices)
    it = _index_code("indices", indices)
    nnz = indices.numel()
    out = torch.empty_like(indices)
    need = compressed_grad_workspace_bytes(nnz, indices.dtype)
    if workspace is None:
        workspace = torch.empty((max(need, 1),), dtype=torch.uint8, device=indices.device)
    elif workspace.numel() * workspace.element_size() < need:
        raise ValueError("workspace too small: need %d bytes" % need)
    lwork = ctypes.c_size_t(workspace.numel() * workspace.element_size())
    if nnz > 0:
        with torch.cuda.device(indices.device):   # the launch must happen on the tensors' device
            _lib.lib().cuembed_compute_compressed_grad_indices(_ptr(indices), nnz, it, _ptr(out),
                                                               _ptr(workspace), ctypes.byref(lwork),
                                                               _stream(indices))
    return out


SHARED_ROW_BIT = 1 << 30   # detail::kSharedRowBit of compute_compressed_grad_indices_blocked
MAX_COALESCED_BLOCKS = 8    # detail::kMaxCoalescedBlocks


def _check_coalesced_blocks(nnz, sample_blocks):
    """The number of blocks the library cuts nnz lookups into.  The C++ entry points abort on what is rejected
    here; raise instead."""
    if sample_blocks <= 1 or nnz <= 0:
        return 1
    length = transpose_sample_block_length(nnz, sample_blocks)
    blocks = -(-nnz // length)
    if blocks > MAX_COALESCED_BLOCKS:
        raise ValueError("the coalesced compressed gradient supports at most %d sample blocks" % MAX_COALESCED_BLOCKS)
    if blocks > 1 and nnz >= (1 << 30):
        raise ValueError("a sample-blocked coalesced gradient needs nnz < 2^30")
    return blocks


def compressed_grad_blocked_workspace_bytes(nnz, index_dtype, sample_blocks):
    lwork = ctypes.c_size_t(0)
    _lib.lib().cuembed_compute_compressed_grad_indices_blocked(None, nnz, _INDEX[index_dtype], int(sample_blocks), None,
                                                               None, None, None, ctypes.byref(lwork), None)
    return lwork.value


def compute_compressed_grad_indices_blocked(indices, sample_blocks, workspace=None, num_unique=None,
                                            block_row_ids=None):
    """compute_compressed_grad_indices for the output of transpose(..., sample_blocks=...) (extension).
    Returns (remapped, block_row_ids, num_unique) -- what embedding_backward(..., sample_blocks=...) consumes:
    remapped[i] numbers the (block, table row) pair of lookup i; block_row_ids[pair] (int32 tensor holding uint32
    bit patterns, nnz entries of room) is the id the reference's fully sorted order assigns to that table row (its
    rank among all distinct rows of the batch), with SHARED_ROW_BIT set when the row also occurs in an earlier
    block; num_unique is a one-element int32 device tensor.  With one block (sample_blocks <= 1 or at most 131,072
    lookups) remapped is compute_compressed_grad_indices' and block_row_ids is the identity."""
    _check_dev("indices", indices)
    it = _index_code("indices", indices)
    nnz = indices.numel()
    blocks = _check_coalesced_blocks(nnz, sample_blocks)
    out = torch.empty_like(indices)
    need = compressed_grad_blocked_workspace_bytes(nnz, indices.dtype, sample_blocks)
    if workspace is None:
        workspace = torch.empty((max(need, 1),), dtype=torch.uint8, device=indices.device)
    elif workspace.numel() * workspace.element_size() < need:
        raise ValueError("workspace too small: need %d bytes" % need)
    if num_unique is None:
        num_unique = torch.zeros((1,), dtype=torch.int32, device=indices.device)
    else:
        _check_dev("num_unique", num_unique, indices.device)
        if num_unique.dtype != torch.int32 or num_unique.numel() < 1:
            raise ValueError("num_unique must be an int32 tensor with one element")
    if block_row_ids is None:
        block_row_ids = (torch.empty((max(nnz, 1),), dtype=torch.int32, device=indices.device) if blocks > 1 else
                         torch.arange(max(nnz, 1), dtype=torch.int32, device=indices.device))
    else:
        _check_dev("block_row_ids", block_row_ids, indices.device)
        if block_row_ids.dtype != torch.int32 or block_row_ids.numel() < nnz:
            raise ValueError("block_row_ids must be an int32 tensor with nnz entries")
    lwork = ctypes.c_size_t(workspace.numel() * workspace.element_size())
    if nnz > 0:
        with torch.cuda.device(indices.device):
            _lib.lib().cuembed_compute_compressed_grad_indices_blocked(
                _ptr(indices), nnz, it, int(sample_blocks), _ptr(out), _ptr(block_row_ids), _ptr(num_unique),
                _ptr(workspace), ctypes.byref(lwork), _stream(indices))
    return out, block_row_ids, num_unique


def extract_row_ids_from_fixed(batch_size, num_hots, dtype=torch.int32, device="cuda"):
    out = torch.empty((batch_size * num_hots,), dtype=dtype, device=device)
    _check_dev("row_ids", out)
    with torch.cuda.device(out.device):   # the launch must happen on the tensors' device
        _lib.lib().cuembed_extract_row_ids_from_fixed(batch_size, num_hots, _INDEX[dtype], _ptr(out), _stream(out))
    return out


def extract_row_ids_from_csr(offsets, nnz=None, dtype=None, batch_size=None):
    """row_ids[i] = b for i in [offsets[b], offsets[b+1]).  nnz=None reads offsets[-1] (host sync)."""
    _check_dev("offsets", offsets)
    ot = _index_code("offsets", offsets)
    if batch_size is None:
        batch_size = offsets.numel() - 1
    if nnz is None:
        nnz = int(offsets[batch_size].item())
    dtype = offsets.dtype if dtype is None else dtype
    out = torch.empty((nnz,), dtype=dtype, device=offsets.device)
    if batch_size > 0 and nnz > 0:
        with torch.cuda.device(offsets.device):   # the launch must happen on the tensors' device
            _lib.lib().cuembed_extract_row_ids_from_csr(_ptr(offsets), ot, batch_size, _INDEX[dtype],
                                                        _ptr(out), _stream(offsets))
    return out


def extract_row_ids_for_concat(nnz, dtype=torch.int32, device="cuda"):
    out = torch.empty((nnz,), dtype=dtype, device=device)
    _check_dev("row_ids", out)
    with torch.cuda.device(out.device):   # the launch must happen on the tensors' device
        _lib.lib().cuembed_extract_row_ids_for_concat(nnz, _INDEX[dtype], _ptr(out), _stream(out))
    return out


# The reference's C++ names, for code that reads like the reference.
EmbeddingForward = embedding_forward
EmbeddingBackward = embedding_backward
Transpose = transpose
ComputeCompressedGradIndices = compute_compressed_grad_indices
ExtractRowIdsFromFixed = extract_row_ids_from_fixed
ExtractRowIdsFromCSR = extract_row_ids_from_csr
ExtractRowIdsForConcat = extract_row_ids_for_concat


# ---- multi-GPU: the device-side halves of the sparse gradient exchange (cuembed_amd/distributed.py) -----------------
def exchange_pack_rows(ids, rows, count, cuts, slot_capacity, input_capacity, num_categories, send_ids, send_rows,
                       range_starts, flag):
    """cuembed::PackRowsByOwner (extension): the first `count` (1-element device tensor of ids' dtype, None = all)
    ascending ids[n] / rows[n, W] into the fixed slots of an equal-split all-to-all -- slot r of
    send_ids[world * slot_capacity] (int64) / send_rows[world * slot_capacity, W] gets the rows of the id range
    [cuts[r], cuts[r + 1]) (cuts: int64[world + 1] on the device), then the padding id num_categories.
    range_starts: int64[world + 1] scratch; flag: an int64 word, |= 1 when a range exceeds its slot or (with
    input_capacity > 0) count exceeds input_capacity.  Two launches, nothing read back."""
    _check_dev("rows", rows)
    dev = rows.device
    for name, t in (("ids", ids), ("cuts", cuts), ("send_ids", send_ids), ("send_rows", send_rows),
                    ("range_starts", range_starts), ("flag", flag)):
        _check_dev(name, t, dev)
    it = _index_code("ids", ids)
    et = _elem_code("rows", rows)
    world = cuts.numel() - 1
    if rows.dim() != 2 or rows.shape[0] != ids.numel() or not rows.is_contiguous() or not ids.is_contiguous():
        raise ValueError("rows must be a contiguous [n, width] tensor with one contiguous id per row")
    if cuts.dtype != torch.int64 or world < 1 or world > 1024 or not cuts.is_contiguous():
        raise ValueError("cuts must be world + 1 contiguous int64 words (world <= 1024)")
    if send_ids.dtype != torch.int64 or send_ids.numel() != world * slot_capacity or not send_ids.is_contiguous():
        raise ValueError("send_ids must be world * slot_capacity contiguous int64 words")
    if send_rows.dtype != rows.dtype or send_rows.numel() != world * slot_capacity * rows.shape[1] or \
            not send_rows.is_contiguous():
        raise ValueError("send_rows must be a contiguous [world * slot_capacity, width] tensor of rows' dtype")
    if range_starts.dtype != torch.int64 or range_starts.numel() < world + 1 or flag.dtype != torch.int64:
        raise ValueError("range_starts must hold world + 1 int64 words, flag one")
    if count is not None:
        _check_dev("count", count, dev)
        count = count.contiguous() if count.dtype == ids.dtype else count.to(ids.dtype)
    with torch.cuda.device(dev):
        _lib.lib().cuembed_exchange_pack_rows(_ptr(ids), it, _ptr(rows), et, ids.numel(), rows.shape[1], _ptr(count),
                                              _ptr(cuts), world, int(slot_capacity), int(input_capacity),
                                              int(num_categories), _ptr(send_ids), _ptr(send_rows), _ptr(range_starts),
                                              _ptr(flag), _stream(rows))


def exchange_merge(ids, rows, num_categories, pad_lo, pad_len, out_ids, out_rows, tail, flag, count=None):
    """The owner's fixed-capacity merge (extension): transpose_fixed_hotness(ids as n samples of hotness 1, remapped) +
    embedding_backward(device-side row count, pad_to_capacity) into out_ids[capacity + 1] (int64) /
    out_rows[capacity + 1, W] + cuembed::FinishOwnerPiece.  ids (int64) >= num_categories are padding and dropped.
    Afterwards: the first *count entries are the ascending distinct ids and their summed rows, every entry past them a
    zero row named pad_lo + i % pad_len; `tail` (int64[capacity + 2] or None) = the ids, min(count, capacity), the flag
    word; flag |= 1 when count > capacity (nothing was written then).  Nothing read back."""
    _check_dev("rows", rows)
    dev = rows.device
    for name, t in (("ids", ids), ("out_ids", out_ids), ("out_rows", out_rows), ("flag", flag)):
        _check_dev(name, t, dev)
    et = _elem_code("rows", rows)
    n = ids.numel()
    capacity = out_ids.numel() - 1
    if ids.dtype != torch.int64 or out_ids.dtype != torch.int64 or flag.dtype != torch.int64:
        raise TypeError("ids, out_ids and flag must be int64")
    if n < 1 or rows.dim() != 2 or rows.shape[0] != n or not rows.is_contiguous() or not ids.is_contiguous():
        raise ValueError("rows must be a contiguous [n, width] tensor with one contiguous id per row, n >= 1")
    if capacity < 1 or out_rows.dtype != rows.dtype or tuple(out_rows.shape) != (capacity + 1, rows.shape[1]) or \
            not out_rows.is_contiguous() or not out_ids.is_contiguous():
        raise ValueError("out_ids / out_rows must be contiguous with capacity + 1 entries / rows")
    if tail is not None and (tail.dtype != torch.int64 or tail.numel() != capacity + 2 or not tail.is_contiguous()):
        raise ValueError("tail must be capacity + 2 contiguous int64 words")
    if count is not None and (count.dtype != torch.int64 or not count.is_contiguous()):
        raise ValueError("count must be a contiguous int64 word")
    if pad_len < 1:
        raise ValueError("pad_len must be at least 1")
    t_ids, t_pos, _, remap = transpose_fixed_hotness(ids, n, 1, num_categories=num_categories + 1, remapped=True)
    embedding_backward(rows, None, t_ids, t_pos, remap, grad_embedding=out_rows, inverse_mapping=out_ids,
                       pad_to_capacity=True)
    with torch.cuda.device(dev):
        _lib.lib().cuembed_exchange_finish_piece(_ptr(t_ids), _ptr(remap), n, capacity, int(num_categories), int(pad_lo),
                                                 int(pad_len), _ptr(out_ids), _ptr(out_rows), et, rows.shape[1],
                                                 _ptr(tail), _ptr(flag), _ptr(count), _stream(rows))
