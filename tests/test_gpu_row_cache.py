"""Table-row caching hook (cuembed::TranslateIndicesForRowCache + cuembed_amd/row_cache.py): a table in
pinned host memory read zero-copy by the unmodified forward kernel, hot rows served from a device
buffer through translated int64 indices.  Results must be bit-identical to the forward on a device
copy of the table -- with and without the cache -- and the cache must really be what is read."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16], ids=["f32", "f16"])
@pytest.mark.parametrize("idx_dtype", [torch.int32, torch.int64], ids=["i32", "i64"])
def test_host_table_with_row_cache(oracle, dtype, idx_dtype):
    import cuembed_amd as ce
    from cuembed_amd.row_cache import CachedHostTable
    rows, W, B, H = 20000, 64, 1500, 12
    a = oracle.allocate_forward(rows, W, B, H, alpha=1.15, elem=np.float16 if dtype == torch.float16 else np.float32)
    host = torch.from_numpy(a["table"]).pin_memory()
    idx = torch.from_numpy(a["indices"]).cuda().to(idx_dtype)
    w = torch.from_numpy(a["weights"]).cuda()
    want = ce.embedding_forward(host.cuda(), idx, num_hots=H)
    want_w = ce.embedding_forward(host.cuda(), idx, None, w, num_hots=H, mode="mean")
    t = CachedHostTable(host, "cuda", capacity_rows=3000)
    # no cache at all: every row over PCIe
    assert torch.equal(t.forward(idx, num_hots=H, use_cache=False), want)
    # empty cache (all slots -1), then the 3000 most frequent rows
    assert torch.equal(t.forward(idx, num_hots=H), want)
    t.cache_most_frequent(idx)
    hit = (t.slot_of_row[idx.long()] >= 0).float().mean().item()
    assert 0.3 < hit < 1.0
    assert torch.equal(t.forward(idx, num_hots=H), want)
    assert torch.equal(t.forward(idx, None, w, num_hots=H, mode="mean"), want_w)
    tr = t.translate(idx)
    cached = t.slot_of_row[idx.long()] >= 0
    assert torch.equal(tr[~cached], idx.long()[~cached])
    assert torch.equal(tr[cached] - t.cache_row_offset, t.slot_of_row[idx.long()][cached].long())
    # CSR layout through the same hook
    off = torch.arange(0, B * H + 1, H, dtype=torch.int32, device="cuda")
    assert torch.equal(t.forward(idx, off, num_hots=0), want)
    # the device copies are what is read: corrupt them and the result must change; refresh() heals it
    t.cache.zero_()
    assert not torch.equal(t.forward(idx, num_hots=H), want)
    t.refresh()
    assert torch.equal(t.forward(idx, num_hots=H), want)
    torch.cuda.synchronize()
    assert ce._lib.lib().cuembed_peek_last_error() == 0


def test_row_cache_rejects_bad_ids_and_passes_foreign_indices_through():
    from cuembed_amd.row_cache import CachedHostTable
    host = torch.zeros((100, 8), dtype=torch.float32).pin_memory()
    t = CachedHostTable(host, "cuda", capacity_rows=4)
    with pytest.raises(ValueError):
        t.cache_rows(torch.arange(5))                      # more ids than the cache holds
    with pytest.raises(IndexError):
        t.cache_rows(torch.tensor([1, 100]))               # outside the table
    with pytest.raises(IndexError):
        t.cache_rows(torch.tensor([-1]))
    t.cache_rows(torch.tensor([7, 3]))
    # indices outside [0, rows) are never used to index slot_of_row: they come back unchanged
    idx = torch.tensor([3, 7, 5, 100, 1 << 40, -2], dtype=torch.int64, device="cuda")
    tr = t.translate(idx)
    assert tr.tolist() == [t.cache_row_offset + 1, t.cache_row_offset + 0, 5, 100, 1 << 40, -2]
