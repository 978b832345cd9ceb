"""What the torch layer decides FOR the caller (it sees the table and every batch; the C++ entry points see one call).

The library's fast paths are options that never change a result (include/cuembed_amd.h: RowLoadPolicy,
ForwardOptions::sample_order); a caller of the reference's Python surface (examples/pytorch/cuembed_pyt.py:48-51) should
not have to know them.  Two hints are chosen here, per table and per offsets tensor:

  row_loads     "streaming" (non-temporal table-row loads) pays when nearly every lookup of a batch hits a different row
                of a table far larger than the caches (C2 shape, uniform indices: 0.380 -> 0.358 ms) and costs a lot
                when rows are re-used (alpha = 1.15: 0.136 -> 0.222 ms).  Decided from the distinct fraction of a
                sample of the batch's indices (one `unique` + one 4-byte read-back), on the first call for a table and
                again every RECHECK_CALLS calls; conservative: streaming only when >= STREAMING_DISTINCT of the sample
                is distinct, the table is >= STREAMING_MIN_TABLE_BYTES and the batch has >= STREAMING_MIN_LOOKUPS
                lookups (smaller ones are latency-bound whatever the loads are, and the decision itself would cost
                more host time than their launch).
  sample_order  the samples of a ragged CSR batch by descending bag length (cuembed::BagOrderByLength; C3: 0.170 ->
                0.148 ms).  It depends on the offsets alone and costs about what it saves, so it is computed once per
                DISTINCT offsets tensor (same storage, same version counter, same length) and kept for the last few:
                a pipeline that re-uses its offsets (fixed bag layout, evaluation over a cached batch) gains, one that
                builds new offsets every step pays nothing -- the order is only prepared on the SECOND sight of a tensor.

Nothing here runs under torch.compile tracing or stream capture (both need a read-back-free path): the hints are then
"no hint".  `set_enabled(False)` turns the whole module into "no hint" (the tests that pin kernels do).
"""
import collections

import torch

RECHECK_CALLS = 256
SAMPLE = 65536
STREAMING_DISTINCT = 0.8
STREAMING_MIN_TABLE_BYTES = 1 << 30
STREAMING_MIN_LOOKUPS = 1 << 18
ORDER_MIN_LOOKUPS = 1 << 20
ORDER_MIN_BATCH = 1 << 14
_ORDER_KEEP = 4

_enabled = True
_tables = {}                                  # (ptr, shape, dtype) -> [calls, streaming]
_orders = collections.OrderedDict()           # (ptr, version, numel, dtype) -> [sightings, order tensor or None]


def set_enabled(flag):
    global _enabled
    _enabled = bool(flag)
    _tables.clear()
    _orders.clear()


def enabled():
    return _enabled


def _quiet():
    """True where a decision would need a read-back that the context forbids."""
    if not _enabled:
        return True
    if torch.compiler.is_compiling():
        return True
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def distinct_fraction(indices):
    """Distinct fraction of an evenly strided sample of up to SAMPLE indices (one device -> host read-back)."""
    flat = indices.reshape(-1)
    n = flat.numel()
    if n == 0:
        return 0.0
    sample = flat[:: max(1, n // SAMPLE)][:SAMPLE]
    return float(torch.unique(sample).numel()) / float(sample.numel())


def row_loads(params, indices):
    """-1 (no hint: the process-wide default) / 0 (default loads) / 1 (streaming) for embedding_forward."""
    # (the cheap gates first: a small batch is latency-bound whatever the loads are, and at 7 us per launch the 3 us the
    # capture query of _quiet() costs would be the largest part of this function's caller)
    if indices.numel() < STREAMING_MIN_LOOKUPS or params.numel() * params.element_size() < STREAMING_MIN_TABLE_BYTES:
        return -1
    if _quiet():
        return -1
    key = (params.data_ptr(), tuple(params.shape), params.dtype)
    state = _tables.get(key)
    if state is None:
        if len(_tables) > 64:
            _tables.clear()
        state = _tables[key] = [0, False]
    if state[0] % RECHECK_CALLS == 0:
        state[1] = distinct_fraction(indices) >= STREAMING_DISTINCT
    state[0] += 1
    return 1 if state[1] else 0


def sample_order(offsets, nnz, max_length=0):
    """The cached bag order of `offsets` for ForwardOptions::sample_order, or None (small batch, first sight, quiet)."""
    if offsets is None or nnz < ORDER_MIN_LOOKUPS or offsets.numel() - 1 < ORDER_MIN_BATCH or _quiet():
        return None
    key = (offsets.data_ptr(), offsets._version, offsets.numel(), offsets.dtype)
    entry = _orders.get(key)
    if entry is None:
        _orders[key] = [1, None]
        while len(_orders) > _ORDER_KEEP:
            _orders.popitem(last=False)
        return None
    _orders.move_to_end(key)
    entry[0] += 1
    if entry[1] is None:
        entry[1] = torch.ops.cuembed_pyt.cuembed_bag_order_by_length(offsets, int(max_length))
    return entry[1]
