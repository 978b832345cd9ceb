"""What the torch layer decides FOR the caller (it sees the table and every batch; the C++ entry points see one call).

The library's fast paths are options that never change a result (include/cuembed_amd.h: RowLoadPolicy,
ForwardOptions::sample_order); a caller of the reference's Python surface (examples/pytorch/cuembed_pyt.py:48-51) should
not have to know them.  Two hints are chosen here, and since round 6 BOTH ARE DECIDED ON THE DEVICE: no `torch.unique`,
no read-back, nothing that a HIP graph capture or a stream-ordered pipeline would have to stop for (C++ / C-ABI callers
have the same two calls: cuembed_decide_row_loads, cuembed_bag_order_by_length).

  row_loads     "streaming" (non-temporal table-row loads) pays when nearly every lookup of a batch hits a different row
                of a table far larger than the caches (C2 shape, uniform indices: 0.380 -> 0.358 ms) and costs a lot
                when rows are re-used (alpha = 1.15: 0.136 -> 0.222 ms).  cuembed::DecideRowLoads counts the distinct
                rows of an evenly strided sample of the batch exactly (one launch, ~5 us) and leaves the decision in
                four device words that the forward kernels read (ForwardOptions::row_loads_device); run for the first
                batch of a table and again every RECHECK_CALLS calls; in between the forward re-reads the last decision.
                Conservative: streaming only when >= 99.8 % of the sample is distinct inside its group of 4,096 (the
                measured crossover: at 96.9 % -- power-law exponent 0.75 -- streaming already costs 20 %; round 5's
                host-side rule, >= 80 % of a 65,536 sample, chose it there), the table is >= 1 GiB and the
                batch has >= 2^18 lookups (the gates are the library's: kStreamingMinTableBytes / kStreamingMinLookups).
  sample_order  the samples of a ragged CSR batch by descending bag length (cuembed::BagOrderByLength; C3: 0.170 ->
                0.148 ms).  With bag lengths clamped at 255 it is two small launches, 7 us together (a stable counting sort), cheap
                enough to be computed for EVERY batch from its own offsets -- no cache, nothing to go stale (round 5
                kept an order per offsets tensor, keyed by address and version; a pipeline that builds fresh offsets
                in a recycled allocation would have been handed another batch's order: still a valid permutation,
                but not the one it paid for).

Nothing here runs under torch.compile tracing: the hints are then "no hint".  Inside a HIP-graph capture the bag order
is computed (two captured launches) and the table's EXISTING row-load decision is read by the captured forward; no new
decision is taken there.  `set_enabled(False)` turns the whole
module into "no hint" (the tests that pin kernels do).
"""
import torch

RECHECK_CALLS = 256
STREAMING_MIN_TABLE_BYTES = 1 << 30          # = cuembed::kStreamingMinTableBytes (the kernel-side gate)
STREAMING_MIN_LOOKUPS = 1 << 18              # = cuembed::kStreamingMinLookups
ORDER_MIN_LOOKUPS = 1 << 20
ORDER_MIN_BATCH = 1 << 14

_enabled = True
_tables = {}                                  # (ptr, shape, dtype, device) -> [calls, decision words]


def set_enabled(flag):
    global _enabled
    _enabled = bool(flag)
    _tables.clear()


def enabled():
    return _enabled


def _quiet():
    return not _enabled or torch.compiler.is_compiling()


def row_loads_device(params, indices):
    """The table's row-load decision words (int32[4] on the device, word 0: 1 = streaming) for
    embedding_forward(..., row_loads_device=), or None (no hint: the process-wide default).  Decided on the device from
    `indices` on the first call for a table and every RECHECK_CALLS calls; never read back."""
    # (the cheap gates first: a small batch is latency-bound whatever the loads are, and at 7 us per launch even a
    # dictionary lookup shows)
    if indices.numel() < STREAMING_MIN_LOOKUPS or params.numel() * params.element_size() < STREAMING_MIN_TABLE_BYTES:
        return None
    if _quiet():
        return None
    key = (params.data_ptr(), tuple(params.shape), params.dtype, params.device)
    state = _tables.get(key)
    if state is None or state[0] % RECHECK_CALLS == 0:
        # (only here -- first sight of a table, and every RECHECK_CALLS calls -- is the capture query paid: ~3 us)
        if torch.cuda.is_current_stream_capturing():
            # inside a HIP-graph capture nothing is created or decided: a captured forward reads the table's existing
            # decision words, which the eager calls around the graph keep fresh; a table first seen here gets no hint
            return None if state is None else state[1]
        if state is None:
            if len(_tables) > 64:
                _tables.clear()
            state = _tables[key] = [0, torch.zeros((4,), dtype=torch.int32, device=params.device)]
        torch.ops.cuembed_pyt.cuembed_decide_row_loads(indices, params.numel() * params.element_size(), state[1])
    state[0] += 1
    return state[1]


def row_loads_decision(params):
    """(tests, diagnostics) the last decision for `params` read back: 1 streaming, 0 default, -1 none taken."""
    state = _tables.get((params.data_ptr(), tuple(params.shape), params.dtype, params.device))
    return -1 if state is None else int(state[1][0].item())


def sample_order(offsets, nnz):
    """The bag order of `offsets` for ForwardOptions::sample_order (lengths clamped at 255: two small launches up to
    131,072 samples, a one-pass sort above), or None (small batch, quiet)."""
    if offsets is None or nnz < ORDER_MIN_LOOKUPS or offsets.numel() - 1 < ORDER_MIN_BATCH or _quiet():
        return None
    return torch.ops.cuembed_pyt.cuembed_bag_order_by_length(offsets, -1)
