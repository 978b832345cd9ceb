"""Table-row caching hook for tables that do not live in this GPU's HBM (extension; the reference
lists an embedding cache as future work, README.md:111-112).

    table = torch.empty((rows, width), dtype=torch.float16).pin_memory()      # host-resident, read over PCIe
    cached = CachedHostTable(table, device="cuda:0", capacity_rows=1_000_000)
    cached.cache_most_frequent(sample_of_recent_indices)                       # the caller's policy
    out = cached.forward(indices, num_hots=64)                                 # == EmbeddingForward on the table

How it works (cuembed::TranslateIndicesForRowCache): EmbeddingForward addresses a row as
params + int64(index) * width, so an index of (cache_rows - params) / width + slot reaches row
`slot` of a device buffer through the table's own base pointer.  The indices of a batch are
rewritten with one small kernel (slot_of_row[] lookup) and the unmodified forward kernel then reads
cached rows from HBM and only the others over PCIe.  Results are bit-identical to running on the
table itself as long as the cached copies are current: there is no staleness guard, the caller must
call refresh() after updating table rows.  Lookup indices outside the table are passed through untouched."""
import ctypes

import torch

from . import _lib
from . import ops as _ops


class CachedHostTable:
    def __init__(self, host_table, device="cuda", capacity_rows=1 << 20):
        if host_table.is_cuda or not host_table.is_pinned() or host_table.dim() != 2 or not host_table.is_contiguous():
            raise ValueError("host_table must be a contiguous 2-D tensor in PINNED host memory (tensor.pin_memory())")
        if host_table.dtype not in _ops._ELEM:
            raise TypeError("table must be float32, float16 or bfloat16")
        self.table = host_table
        self.device = torch.device(device)
        if self.device.index is None:   # "cuda" -> the current device, so that it compares equal to tensor.device
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.rows, self.width = host_table.shape
        self.capacity = int(capacity_rows)
        row_bytes = self.width * host_table.element_size()
        if row_bytes % 4:
            raise ValueError("row size must be a multiple of 4 bytes")
        # the cache must differ from the table by a whole number of rows: over-allocate by one row
        # and start where (cache - table) % row_bytes == 0
        self._raw = torch.empty(((self.capacity + 1) * row_bytes,), dtype=torch.uint8, device=self.device)
        shift = (host_table.data_ptr() - self._raw.data_ptr()) % row_bytes
        self.cache = self._raw[shift:shift + self.capacity * row_bytes].view(host_table.dtype).view(self.capacity, self.width)
        assert (self.cache.data_ptr() - host_table.data_ptr()) % row_bytes == 0
        self.cache_row_offset = (self.cache.data_ptr() - host_table.data_ptr()) // row_bytes
        self.slot_of_row = torch.full((self.rows,), -1, dtype=torch.int32, device=self.device)
        self.cached_ids = torch.empty((0,), dtype=torch.int64, device=self.device)

    def cache_rows(self, row_ids):
        """Make `row_ids` (distinct table rows, at most capacity_rows) the cached set.  Raises on more ids
        than the cache holds and on ids outside the table (one host read-back; this is not the step path)."""
        row_ids = row_ids.to(device=self.device, dtype=torch.int64).reshape(-1)
        if row_ids.numel() > self.capacity:
            raise ValueError("%d rows do not fit a cache of %d rows" % (row_ids.numel(), self.capacity))
        if row_ids.numel() and (int(row_ids.min()) < 0 or int(row_ids.max()) >= self.rows):
            raise IndexError("row ids must lie in [0, %d)" % self.rows)
        self.slot_of_row.fill_(-1)
        self.slot_of_row[row_ids] = torch.arange(row_ids.numel(), dtype=torch.int32, device=self.device)
        self.cached_ids = row_ids
        self.refresh()

    def cache_most_frequent(self, indices):
        """Policy helper: cache the most frequently looked-up rows of a sample of indices."""
        counts = torch.bincount(indices.to(self.device).reshape(-1).long(), minlength=self.rows)
        k = min(self.capacity, int((counts > 0).sum().item()))
        self.cache_rows(torch.topk(counts, k).indices)

    def refresh(self):
        """Re-copy the cached rows from the host table (after the table was updated)."""
        if self.cached_ids.numel():
            ids = self.cached_ids.cpu()
            self.cache[: ids.numel()].copy_(self.table[ids].pin_memory(), non_blocking=True)

    def translate(self, indices):
        """indices (int32 / int64 device tensor) -> int64 indices that address cached rows in HBM."""
        _ops._check_dev("indices", indices, self.device)
        it = _ops._index_code("indices", indices)
        out = torch.empty(indices.shape, dtype=torch.int64, device=self.device)
        if indices.numel():
            with torch.cuda.device(self.device):
                _lib.lib().cuembed_translate_indices_for_row_cache(
                    ctypes.c_void_p(indices.data_ptr()), it, indices.numel(), ctypes.c_void_p(self.slot_of_row.data_ptr()),
                    self.rows, self.cache_row_offset, ctypes.c_void_p(out.data_ptr()), _ops._stream(indices))
        return out

    def forward(self, indices, offsets=None, weights=None, batch_size=None, num_hots=0, mode="sum", use_cache=True,
                out=None):
        """EmbeddingForward on the host-resident table (zero-copy over PCIe), cached rows from HBM.
        use_cache=False reads every row from the host table (for comparison)."""
        if mode not in _ops._MODES:
            raise ValueError("mode must be 'sum', 'mean' or 'concat'")
        m = _ops._MODES[mode]
        idx = self.translate(indices) if use_cache else indices
        _ops._check_dev("indices", idx, self.device)
        it = _ops._index_code("indices", idx)
        ot = 0
        if offsets is not None:
            _ops._check_dev("offsets", offsets, self.device)
            ot = _ops._index_code("offsets", offsets)
            if batch_size is None:
                batch_size = offsets.numel() - 1
        elif batch_size is None:
            batch_size = idx.numel() // num_hots
        if weights is not None:
            _ops._check_dev("weights", weights, self.device)
            if weights.dtype != self.table.dtype:
                raise TypeError("weights must have the table's dtype")
        shape = (batch_size, num_hots, self.width) if m == _ops.CONCAT else (batch_size, self.width)
        if out is None:
            out = torch.empty(shape, dtype=self.table.dtype, device=self.device)
        if batch_size > 0:
            with torch.cuda.device(self.device):
                _lib.lib().cuembed_embedding_forward(
                    ctypes.c_void_p(self.table.data_ptr()), _ops._ELEM[self.table.dtype], self.width, _ops._ptr(idx), it,
                    _ops._ptr(offsets), ot, _ops._ptr(weights), batch_size, num_hots, m, 0, _ops._ptr(out),
                    _ops._stream(idx))
        return out
