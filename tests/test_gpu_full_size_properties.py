"""BASELINE.json configs 3 and 4 at FULL size, checked through size-independent properties
(the oracle would need minutes at these sizes): sortedness / stability / permutation of the
transpose, conservation of column sums, checksum of checksums on exactly representable data,
plus bit-exact oracle comparisons on slices that the oracle finishes in seconds."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ce():
    import cuembed_amd
    return cuembed_amd


def test_config3_fp32_weighted_csr_full_size(ce, oracle):
    """C3: fp32 weighted sum, CSR with hotness U[0,128] (mean 64), 10M x 128 table, batch 65536."""
    from cuembed_amd import harness
    rows, W, B, H = 10_000_000, 128, 65536, 128
    a = harness.allocate_forward(rows, W, B, H, alpha=1.15, is_csr=True, with_table=False,
                                 consume_table_draws=False)
    idx, off, w = (torch.from_numpy(a[k]).cuda() for k in ("indices", "offsets", "weights"))
    nnz = idx.numel()
    assert abs(nnz / B - 64) < 1 and set(np.unique(a["weights"]).tolist()) == {0.25, 0.5}
    g = torch.Generator(device="cuda").manual_seed(5)
    table = torch.randint(-2, 3, (rows, W), device="cuda", generator=g).float()   # integers: exact sums
    out = ce.embedding_forward(table, idx, off, w, num_hots=0)
    # (a) checksum of checksums: sum_s out[s,:] == (weighted histogram of indices) @ table
    hist = torch.zeros(rows, dtype=torch.float64, device="cuda").index_add_(0, idx.long(), w.double())
    assert torch.equal(out.double().sum(0), (hist.unsqueeze(0) @ table.double()).squeeze(0))
    # (b) empty bags give exact zeros; (c) first / last 200 samples bit-exact against the oracle
    lens = np.diff(a["offsets"])
    assert (lens == 0).any() and not out[torch.from_numpy(lens == 0).cuda()].any()
    for lo in (0, B - 200):
        o = a["offsets"][lo:lo + 201].astype(np.int64)
        sub_idx = a["indices"][o[0]:o[-1]]
        uniq, inv = np.unique(sub_idx, return_inverse=True)
        small = table[torch.from_numpy(uniq).cuda().long()].cpu().numpy()
        want = oracle.embedding_forward(small, inv.astype(np.int32), (o - o[0]).astype(np.int32),
                                        a["weights"][o[0]:o[-1]], num_hots=0)
        assert np.array_equal(out[lo:lo + 200].cpu().numpy().view(np.uint32), want.view(np.uint32))
    # (d) mean == sum / sum-of-weights
    mean = ce.embedding_forward(table, idx, off, w, num_hots=0, mode="mean")
    wsum = torch.zeros(B, device="cuda").index_add_(0, ce.extract_row_ids_from_csr(off, nnz=nnz).long(), w)
    ref = torch.where(wsum[:, None] > 0, out * (1.0 / wsum)[:, None], torch.zeros_like(out))
    assert torch.equal(mean, ref)


@pytest.mark.parametrize("idx_t", [torch.int32, torch.int64], ids=["i32", "i64"])
def test_config4_backward_compressed_full_size(ce, oracle, idx_t):
    """C4: transpose + compressed gradient at the C2 shape (10M x 256, batch 65536, hotness 64)."""
    from cuembed_amd import harness
    rows, W, B, H = 10_000_000, 256, 65536, 64
    idx_np = harness.generate_indices(rows, B, H, alpha=1.15)
    idx = torch.from_numpy(idx_np).cuda().to(idx_t)
    nnz = idx.numel()
    sid = ce.extract_row_ids_from_fixed(B, H, idx_t, "cuda")
    assert torch.equal(sid, torch.arange(nnz, device="cuda").div(H, rounding_mode="floor").to(idx_t))
    t_idx, t_sid, _ = ce.transpose(sid, idx, num_categories=rows)
    # sorted by index; stable (sample ids ascend inside a run); a permutation of the input pairs
    assert bool((t_idx[1:] >= t_idx[:-1]).all())
    same = t_idx[1:] == t_idx[:-1]
    assert bool((t_sid[1:][same] > t_sid[:-1][same]).all())
    ref_idx, order = torch.sort(idx, stable=True)
    assert torch.equal(t_idx, ref_idx) and torch.equal(t_sid, sid[order])
    t_idx_full, t_sid_full, _ = ce.transpose(sid, idx)                      # all key bits: same result
    assert torch.equal(t_idx_full, t_idx) and torch.equal(t_sid_full, t_sid)
    remap = ce.compute_compressed_grad_indices(t_idx)
    uniq = torch.unique(idx)
    nu = int(remap[-1].item()) + 1
    assert nu == uniq.numel() == 572029      # unique rows of the C2 index stream
    assert torch.equal(uniq.to(idx_t)[remap.long()], t_idx)                 # remap is the dense rank of the id
    # gradient: fp32 (exact for any run length) and fp16 on sparse +-1 data (exact below 2048)
    g = torch.Generator(device="cuda").manual_seed(9)
    gy = torch.randint(-3, 4, (B, W), device="cuda", generator=g).float()
    grad, inv = ce.embedding_backward(gy, nu, t_idx, t_sid, remap)
    assert torch.equal(inv, uniq.to(idx_t))
    assert torch.equal(grad.double().sum(0), gy.double().sum(0) * H)        # every sample is used H times
    hottest = int(torch.mode(t_idx).values)
    run = (t_idx == hottest).nonzero().squeeze(1)
    assert run.numel() > 60000                                              # one row owns a 65k-lookup run
    assert torch.equal(grad[int(remap[run[0]])], gy[t_sid[run].long()].sum(0))
    once = (torch.bincount(remap.long()) == 1).nonzero().squeeze(1)[:1000]   # rows looked up exactly once
    pos = torch.searchsorted(remap, once.to(remap.dtype))
    assert torch.equal(grad[once], gy[t_sid[pos].long()])
    gy16 = (torch.rand((B, W), device="cuda", generator=g) < 0.02).half() * \
        (torch.randint(0, 2, (B, W), device="cuda", generator=g).half() * 2 - 1)
    grad16, _ = ce.embedding_backward(gy16, nu, t_idx, t_sid, remap)
    ref16 = torch.zeros((nu, W), device="cuda").index_add_(0, remap.long(), gy16.float()[t_sid.long()])
    assert float(ref16.abs().max()) < 2048 and torch.equal(grad16.float(), ref16)
    # the extension entry point at full size: same transposed arrays without a sample-id array
    f_idx, f_sid, _ = ce.transpose_fixed_hotness(idx, B, H, num_categories=rows)
    assert torch.equal(f_idx, t_idx) and torch.equal(f_sid, t_sid)
    # dense gradient of a smaller table slice of the same lookups == scatter of the compressed one
    small_rows = 200_000
    keep = t_idx < small_rows
    dense, _ = ce.embedding_backward(gy, small_rows, t_idx[keep].contiguous(), t_sid[keep].contiguous())
    ref = torch.zeros((small_rows, W), device="cuda")
    sel = inv < small_rows
    ref[inv[sel].long()] = grad[sel.nonzero().squeeze(1)]
    assert torch.equal(dense, ref)


def _sub_coo(remap, t_idx, t_sid, dense_ids):
    """The lookups of the selected runs (dense ids, ascending) as host arrays with the runs renumbered 0..k-1."""
    d = dense_ids.to(remap.dtype)
    lo = torch.searchsorted(remap, d)
    hi = torch.searchsorted(remap, d + 1)
    lens = (hi - lo)
    pos = torch.repeat_interleave(lo, lens) + (torch.arange(int(lens.sum()), device=remap.device)
                                               - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens))
    new_remap = torch.repeat_interleave(torch.arange(d.numel(), device=remap.device), lens).to(torch.int32)
    return (t_idx[pos].cpu().numpy().astype(np.int32), t_sid[pos].cpu().numpy().astype(np.int32),
            new_remap.cpu().numpy(), lens.cpu().numpy())


def test_config4_backward_reference_grad_y_recipe_against_oracle(ce, oracle):
    """C4 at full size on the REFERENCE's own inputs: power-law indices of the manual_benchmark shape and
    grad_y from AllocateBackward's recipe (integers U{-10..10}, engine 654321,
    utils/src/embedding_allocation.cu:221-247), against the CPU oracle (embedding_lookup_cpu.hpp:96-144)
    run on a sample of the runs -- 3,000 random rows plus the 64 longest runs:
      fp32: every sampled row bit-exact (integer sums < 2^24 are exact in any order);
      fp16: rows whose run is at most 204 lookups long keep every partial sum below 2048, where the
            reference's per-lookup fp16 rounding is exact: bit-exact against the oracle; the longer runs
            are compared with the exact (fp64) sum within the documented bound of fp32 partial sums +
            16-bit flushes (tests/test_gpu_backward_tolerance.py::_hip_error_bound)."""
    from cuembed_amd import harness
    from test_gpu_backward_tolerance import EPS, TINY, _hip_error_bound   # noqa: F401
    rows, W, B, H = 10_000_000, 256, 65536, 64
    idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15)).cuda()
    t_idx, t_sid, _ = ce.transpose_fixed_hotness(idx, B, H, num_categories=rows)
    remap = ce.compute_compressed_grad_indices(t_idx)
    nu = int(remap[-1].item()) + 1
    counts = torch.bincount(remap.long(), minlength=nu)
    g = torch.Generator(device="cuda").manual_seed(21)
    sample = torch.randperm(nu, device="cuda", generator=g)[:3000]
    longest = torch.topk(counts, 64).indices
    chosen = torch.unique(torch.cat([sample, longest]))                      # ascending dense ids
    s_idx, s_sid, s_remap, lens = _sub_coo(remap, t_idx, t_sid, chosen)
    assert lens.max() > 60000 and (lens == 1).any() and lens.sum() == s_idx.shape[0]
    k = chosen.numel()
    for elem in (np.float32, np.float16):
        gy_host = harness.allocate_grad_y(B * W, elem).reshape(B, W)
        assert gy_host.min() == -10 and gy_host.max() == 10 and np.array_equal(gy_host, np.rint(gy_host))
        gy = torch.from_numpy(gy_host).cuda()
        grad, inv = ce.embedding_backward(gy, nu, t_idx, t_sid, remap)
        got = grad[chosen].float().cpu().numpy()
        want, want_inv = oracle.embedding_backward(gy_host, W, k, s_idx, s_sid, s_remap)
        assert np.array_equal(inv[chosen].cpu().numpy(), want_inv)
        if elem == np.float32:
            assert np.array_equal(got, want)                                # all 3,000+ rows, every element
            continue
        short = lens <= 204
        assert short.sum() > 2500 and (~short).sum() >= 58
        assert np.array_equal(got[short], want[short].astype(np.float32))   # the reference's arithmetic, bit for bit
        # the long runs: exact integer sums in fp64, and the bound of this library's arithmetic
        terms = gy_host.astype(np.float64)
        exact = np.zeros((k, W))
        np.add.at(exact, s_remap, terms[s_sid])
        scale = np.zeros((k, W))
        np.add.at(scale, s_remap, np.abs(terms[s_sid]))
        sq = np.zeros((k, W))
        np.add.at(sq, s_remap, terms[s_sid] ** 2)
        bound = _hip_error_bound("f16", exact, scale, np.sqrt(sq), lens)
        err = np.abs(got.astype(np.float64) - exact)
        assert np.all(err[~short] <= bound[~short]), float((err - bound)[~short].max())
        # for the record: where no partial sum leaves the integers fp16 holds exactly, the result IS the integer sum
        small = (~short) & (scale.max(axis=1) < 2048)
        assert np.array_equal(got[small].astype(np.float64), exact[small])


def test_config4_sample_blocks_same_gradient_at_full_size(ce):
    """C4 through the sample-block extension (Transpose(..., sample_blocks = recommended = 2)): every block is sorted
    and stable on its own, the compressed gradient has one row per (block, table row), and scattered into the
    table it is the reference-order gradient bit for bit (fp32 on integers; fp16 on sparse +-1 data)."""
    from cuembed_amd import harness
    rows, W, B, H = 10_000_000, 256, 65536, 64
    idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15)).cuda()
    nnz = B * H
    P = ce.recommended_sample_blocks(torch.float16, W, B, nnz)
    assert P == 2
    L = ce.transpose_sample_block_length(nnz, P)
    assert L == nnz // 2
    t_idx, t_sid, _ = ce.transpose_fixed_hotness(idx, B, H, num_categories=rows)
    b_idx, b_sid, _ = ce.transpose_fixed_hotness(idx, B, H, num_categories=rows, sample_blocks=P)
    sid = torch.arange(nnz, device="cuda", dtype=torch.int32).div(H, rounding_mode="floor")
    for k in range(P):
        ref_idx, order = torch.sort(idx[k * L:(k + 1) * L], stable=True)
        assert torch.equal(b_idx[k * L:(k + 1) * L], ref_idx) and torch.equal(b_sid[k * L:(k + 1) * L], sid[k * L:(k + 1) * L][order])
    remap, b_remap = ce.compute_compressed_grad_indices(t_idx), ce.compute_compressed_grad_indices(b_idx)
    nu, nub = int(remap[-1].item()) + 1, int(b_remap[-1].item()) + 1
    assert nu == 572029 and nu < nub <= 2 * nu
    g = torch.Generator(device="cuda").manual_seed(33)
    gy32 = torch.randint(-3, 4, (B, W), device="cuda", generator=g).float()
    gy16 = (torch.rand((B, W), device="cuda", generator=g) < 0.02).half() * \
        (torch.randint(0, 2, (B, W), device="cuda", generator=g).half() * 2 - 1)
    for gy in (gy32, gy16):
        grad, inv = ce.embedding_backward(gy, nu, t_idx, t_sid, remap)
        gradb, invb = ce.embedding_backward(gy, nub, b_idx, b_sid, b_remap)
        # scatter the uncoalesced rows onto the unique ids of the reference order and compare everything
        pos = torch.searchsorted(inv, invb)
        assert torch.equal(inv[pos], invb)
        merged = torch.zeros((nu, W), dtype=torch.float32, device="cuda").index_add_(0, pos, gradb.float())
        assert float(merged.abs().max()) < 2048 and torch.equal(merged, grad.float())
