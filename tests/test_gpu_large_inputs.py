"""Inputs an order of magnitude past BASELINE.json's configurations -- the sizes a 288 GB device invites and where
32-bit offsets, tile counts and grid sizes would break first: 67 M - 134 M lookups through Transpose / the compressed
remap (nnz is `int` in the reference API, embedding_lookup.cuh:423-435, so 2^27 is well inside the contract), a
67 M-lookup forward whose output offsets pass 2^31 elements, and a backward over the same lookups.  Checked exactly:
against torch's stable sort, and on integer-valued data against index_add_ (any summation order is exact there)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ce():
    import cuembed_amd
    return cuembed_amd


@pytest.mark.parametrize("nnz,categories,index_dtype", [
    (1 << 26, 50_000_000, torch.int32),                # 16,384 sort tiles
    ((1 << 27) + 12345, (1 << 31) - 1, torch.int32),   # ragged last tile, all 31 key bits
    ((1 << 26) + 7, 1 << 40, torch.int64),             # 64-bit keys: five more radix passes
])
def test_transpose_and_remap_far_past_the_benchmark_sizes(ce, nnz, categories, index_dtype):
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(nnz % 1000)
    idx = torch.randint(0, categories, (nnz,), device=dev, dtype=torch.int64, generator=g).to(index_dtype)
    idx[torch.randint(0, nnz, (nnz // 50,), device=dev, generator=g)] = 12345    # one run of ~1.3 M - 2.7 M lookups
    rows = torch.arange(nnz, device=dev, dtype=index_dtype)
    t_idx, t_rows, _ = ce.transpose(rows, idx)
    want_idx, perm = torch.sort(idx, stable=True)
    assert torch.equal(t_idx, want_idx)
    assert torch.equal(t_rows, rows[perm])          # stable: ties keep the input order
    del perm, rows
    remap = ce.compute_compressed_grad_indices(t_idx)
    want = torch.unique_consecutive(want_idx, return_inverse=True)[1]
    assert torch.equal(remap.long(), want)


def test_forward_and_backward_over_67m_lookups(ce):
    """B = 2^20 samples x 64 lookups, W = 2560 fp16 -> the pooled output has 2.7e9 elements (5.4 GB): offsets beyond
    2^31 elements on the output side, 2^26 lookups on the input side.  Integer-valued table: sums are exact."""
    dev = torch.device("cuda")
    B, H, W, rows = 1 << 20, 64, 2560, 50_000
    g = torch.Generator(device=dev).manual_seed(3)
    table = torch.randint(-2, 3, (rows, W), device=dev, generator=g).to(torch.float16)
    idx = torch.randint(0, rows, (B, H), device=dev, dtype=torch.int32, generator=g)
    out = ce.embedding_forward(table, idx.reshape(-1), batch_size=B, num_hots=H)
    assert out.shape == (B, W) and out.numel() > (1 << 31)
    for lo in (0, B // 2 + 17, B - 4096):     # slices, so that the check itself stays small
        sl = slice(lo, lo + 4096)
        want = table[idx[sl].long()].float().sum(1)
        assert torch.equal(out[sl].float(), want)
    # column sums of everything: (histogram of the indices) @ table, in fp64
    hist = torch.zeros(rows, dtype=torch.float64, device=dev).index_add_(
        0, idx.reshape(-1).long(), torch.ones(B * H, dtype=torch.float64, device=dev))
    assert torch.equal(out.sum(0, dtype=torch.float64), (hist.unsqueeze(0) @ table.double()).squeeze(0))
    del out, hist

    # backward over the same 2^26 lookups into a dense gradient (narrower rows: grad_y of B x 64)
    Wg = 64
    gy = torch.randint(-1, 2, (B, Wg), device=dev, generator=g).float()
    t_idx, t_sid, _ = ce.transpose_fixed_hotness(idx.reshape(-1), B, H, num_categories=rows)
    grad, _ = ce.embedding_backward(gy, rows, t_idx, t_sid)
    want = torch.zeros((rows, Wg), device=dev).index_add_(0, idx.reshape(-1).long(), gy.repeat_interleave(H, 0))
    assert torch.equal(grad, want)
