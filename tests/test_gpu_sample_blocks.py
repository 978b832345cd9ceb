"""Transpose in sample blocks (extension: cuembed::Transpose / TransposeFixedHotness, `sample_blocks`): the
sample-major input is cut into consecutive blocks, each block is transposed on its own, the compressed backward
then yields an UNCOALESCED compressed gradient (one row per (block, table row)) whose scatter-add into the table
equals the reference's gradient.  Checked against the oracle applied block by block."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


@pytest.fixture(scope="module")
def ce():
    import cuembed_amd
    return cuembed_amd


def _blockwise(oracle, rows, cols, w, L):
    out = [[], [], []]
    for lo in range(0, cols.shape[0], L):
        r = oracle.transpose(rows[lo:lo + L], cols[lo:lo + L], None if w is None else w[lo:lo + L], stable=True)
        for k in range(3):
            out[k].append(r[k])
    return [np.concatenate(o) if o[0] is not None else None for o in out]


@pytest.mark.parametrize("idx", [np.int32, np.int64], ids=["i32", "i64"])
@pytest.mark.parametrize("nnz,blocks", [(140_000, 2), (200_003, 3), (1 << 20, 2), (1_000_000, 5), (300_000, 64), (4096 * 40, 7),
                                        (100_000, 4), (5000, 3)])
def test_transpose_in_sample_blocks_equals_blockwise_oracle(ce, oracle, idx, nnz, blocks):
    rng = np.random.default_rng(nnz + blocks)
    ncat = 50_000
    cols = (ncat * rng.random(nnz) ** 3).astype(idx)             # skewed: long runs
    rows = np.arange(nnz, dtype=idx) // 7                          # sample-major input
    w = rng.uniform(0, 1, nnz).astype(np.float32)
    L = ce.transpose_sample_block_length(nnz, blocks)
    assert L % 4096 == 0 and L >= nnz / min(blocks, 64) and (nnz > 131072 or L >= nnz)
    for weights in (None, w):
        want = _blockwise(oracle, rows, cols, weights, L)
        got = ce.transpose(dev(rows), dev(cols), dev(weights), num_categories=ncat, sample_blocks=blocks)
        assert np.array_equal(host(got[0]), want[0]) and np.array_equal(host(got[1]), want[1])
        if weights is not None:
            assert np.array_equal(host(got[2]), want[2])
    got = ce.transpose(dev(rows), dev(cols), sample_blocks=blocks)           # all key bits, device-side pass skipping
    assert np.array_equal(host(got[0]), want[0]) and np.array_equal(host(got[1]), want[1])
    if nnz % 8 == 0:                                                          # fixed hotness 8: sample id = position // 8
        want = _blockwise(oracle, np.arange(nnz, dtype=idx) // 8, cols, None, L)
        got = ce.transpose_fixed_hotness(dev(cols), nnz // 8, 8, num_categories=ncat, sample_blocks=blocks)
        assert np.array_equal(host(got[0]), want[0]) and np.array_equal(host(got[1]), want[1])


@pytest.mark.parametrize("elem", [(np.float32, torch.float32), (np.float16, torch.float16)], ids=["f32", "f16"])
@pytest.mark.parametrize("weighted", [False, True], ids=["plain", "weighted"])
def test_compressed_backward_on_sample_blocks_is_the_same_gradient(ce, oracle, elem, weighted):
    """B = 40,000, H = 32 (1.28 M lookups: the sliced backward path), W = 128: the uncoalesced compressed gradient,
    scattered into the table, equals the oracle's dense gradient (integer data: exact in any order)."""
    ncat, W, B, H = 30_000, 128, 40_000, 32
    a = oracle.allocate_forward(ncat, W, B, H, alpha=1.15, elem=elem[0])
    idx = a["indices"]
    w = a["weights"] if weighted else None                                   # 0.5 / 0.25
    gy = (np.mod(oracle.allocate_grad_y(B * W).reshape(B, W), 3) - 1).astype(elem[0])
    sid = oracle.extract_row_ids_from_fixed(B, H)
    ti, ts, tw = oracle.transpose(sid, idx, w)
    want, _ = oracle.embedding_backward(gy.astype(np.float32), W, ncat, ti, ts, None, None if tw is None else tw.astype(np.float32))
    assert np.abs(want).max() < 1024
    blocks = ce.recommended_sample_blocks(elem[1], W, B, B * H)
    assert blocks >= 2 or elem[0] == np.float16
    for P in sorted({2, 3, blocks}):
        t_idx, t_sid, t_w = ce.transpose_fixed_hotness(dev(idx), B, H, dev(w), num_categories=ncat, sample_blocks=P)
        remap = ce.compute_compressed_grad_indices(t_idx)
        nu = int(remap[-1].item()) + 1
        grad, inv = ce.embedding_backward(dev(gy), nu, t_idx, t_sid, remap, t_w)
        uniq = np.unique(idx).shape[0]
        assert uniq <= nu <= P * uniq
        dense = torch.zeros((ncat, W), dtype=torch.float32, device="cuda").index_add_(0, inv.long(), grad.float())
        assert np.array_equal(host(dense), want), P
        # ids ascend inside a block; a table row appears at most once per block
        inv_h = host(inv)
        drops = np.flatnonzero(np.diff(inv_h) <= 0)
        assert drops.shape[0] <= P - 1
    torch.cuda.synchronize()
    assert ce._lib.lib().cuembed_peek_last_error() == 0


def test_recommended_sample_blocks(ce):
    assert ce.recommended_sample_blocks(torch.float16, 256, 65536, 65536 * 64) == 2      # C4: 8.4 MB per L2 -> 2 blocks
    assert ce.recommended_sample_blocks(torch.float16, 256, 32768, 32768 * 64) == 1      # 4.2 MB: fits
    assert ce.recommended_sample_blocks(torch.float32, 128, 65536, 65536 * 64) == 2      # C3 rows: the same 512 bytes
    assert ce.recommended_sample_blocks(torch.float16, 256, 65536, 65536 * 8) == 1       # < 2^20 lookups: not sliced
    assert ce.recommended_sample_blocks(torch.float16, 256, 524288, 524288 * 64) == 16
    assert ce.recommended_sample_blocks(torch.float16, 3, 65536, 65536 * 64) == 1        # a recommendation never aborts
    assert ce.recommended_sample_blocks(torch.float32, 0, 0, 0) == 1
