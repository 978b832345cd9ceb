"""The REFERENCE's compressed gradient from a sample-blocked order (extension; blocked_order.hpp):
Transpose(sample_blocks) -> ComputeCompressedGradIndicesBlocked -> EmbeddingBackward(sample_blocks) must give
exactly what the reference order gives -- num_unique ascending rows, the same inverse_mapping, the same sums
(embedding_lookup.cuh:423-483, index_transforms.cuh:278-323) -- checked against the CPU oracle on the fully
sorted order."""
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


def _expected_blocked_remap(keys, L, bit):
    """rank of every key among the distinct keys of the whole array, | bit when an earlier block holds the key"""
    uniq, rank = np.unique(keys, return_inverse=True)
    want = rank.astype(np.int64)
    seen = np.zeros(uniq.shape[0], dtype=bool)
    for lo in range(0, keys.shape[0], L):
        blk = rank[lo:lo + L]
        want[lo:lo + L] |= np.where(seen[blk], bit, 0)
        seen[blk] = True
    return want, uniq.shape[0]


def _check_pairs(pair, keys, L):
    """pair numbers count up from 0: a new one where the key changes or a block begins"""
    n = keys.shape[0]
    head = np.ones(n, dtype=bool)
    head[1:] = keys[1:] != keys[:-1]
    head[np.arange(0, n, L)] = True
    assert np.array_equal(pair.astype(np.int64), np.cumsum(head) - 1)


@pytest.mark.parametrize("idx", [np.int32, np.int64], ids=["i32", "i64"])
@pytest.mark.parametrize("nnz,blocks,ncat", [(140_000, 2, 50_000), (200_003, 3, 50_000), (1 << 20, 2, 3_000_000),
                                            (1_000_000, 5, 700), (4096 * 40, 7, 50_000), (4096 * 64, 8, 10),
                                            (1_500_000, 4, 1 << 28), (100_000, 4, 50_000), (5000, 3, 100)])
def test_blocked_remap_is_the_rank_among_all_distinct_rows(ce, idx, nnz, blocks, ncat):
    rng = np.random.default_rng(nnz + blocks)
    cols = (ncat * rng.random(nnz) ** 3).astype(idx)             # skewed: long runs, many rows in several blocks
    rows = np.arange(nnz, dtype=idx) // 7
    L = ce.transpose_sample_block_length(nnz, blocks)
    t_idx, _, _ = ce.transpose(dev(rows), dev(cols), num_categories=ncat, sample_blocks=blocks)
    pair, table, nu = ce.compute_compressed_grad_indices_blocked(t_idx, blocks)
    want, uniq = _expected_blocked_remap(host(t_idx), L, ce.SHARED_ROW_BIT)
    if L >= nnz:                                                   # one block: exactly ComputeCompressedGradIndices
        assert np.array_equal(host(pair), host(ce.compute_compressed_grad_indices(t_idx)))
    assert int(nu.item()) == uniq
    _check_pairs(host(pair), host(t_idx), L)
    assert np.array_equal(host(table)[host(pair)].astype(np.int64), want)
    torch.cuda.synchronize()
    assert ce._lib.lib().cuembed_peek_last_error() == 0


@pytest.mark.parametrize("nnz,blocks,ncat,power", [
    (6_000_000, 3, 40_000_000, 1),      # > 4096 rank tiles (tile counts scanned by their own launch); ~1.9 M distinct
                                        # keys per block: more fences than the LDS level holds
    (9_000_001, 2, 5_000_000, 3),       # skewed, ragged last tile
    (5_500_000, 8, 2_000_000, 2),
], ids=["6M_uniform", "9M_skewed", "5.5M_8_blocks"])
def test_blocked_remap_beyond_the_benchmark_size(ce, nnz, blocks, ncat, power):
    """More (block, row) pairs than a finishing workgroup scans itself (4096 tiles of 1024 pairs = what C4 just
    fits) and more distinct keys per block than the two-level search keeps fences for."""
    rng = np.random.default_rng(nnz % 977)
    keys = (ncat * rng.random(nnz) ** power).astype(np.int32)
    L = ce.transpose_sample_block_length(nnz, blocks)
    blocked = np.concatenate([np.sort(keys[lo:lo + L]) for lo in range(0, nnz, L)])
    pair, table, nu = ce.compute_compressed_grad_indices_blocked(dev(blocked), blocks)
    want, uniq = _expected_blocked_remap(blocked, L, ce.SHARED_ROW_BIT)
    assert int(nu.item()) == uniq
    _check_pairs(host(pair), blocked, L)
    assert np.array_equal(host(table)[host(pair)].astype(np.int64), want)


def test_blocked_remap_extremes(ce):
    """every block holds the same single key; every key distinct; blocks with disjoint key ranges in both orders"""
    n = 4096 * 48
    L = ce.transpose_sample_block_length(n, 3)
    bit = ce.SHARED_ROW_BIT
    for name, keys in (("one key", np.full(n, 7, np.int32)),
                       ("all distinct, ascending blocks", np.arange(n, dtype=np.int32)),
                       ("all distinct, descending blocks",
                        np.concatenate([np.arange(n - (b + 1) * L, n - b * L, dtype=np.int32) for b in range(3)])),
                       ("negative keys", np.sort(np.random.default_rng(1).integers(-500, 500, n).astype(np.int32)
                                                 .reshape(3, -1), axis=1).reshape(-1))):
        blocked = np.concatenate([np.sort(keys[lo:lo + L]) for lo in range(0, n, L)])
        pair, table, nu = ce.compute_compressed_grad_indices_blocked(dev(blocked), 3)
        want, uniq = _expected_blocked_remap(blocked, L, bit)
        assert int(nu.item()) == uniq, name
        _check_pairs(host(pair), blocked, L)
        assert np.array_equal(host(table)[host(pair)].astype(np.int64), want), name


@pytest.mark.parametrize("elem", [(np.float32, torch.float32), (np.float16, torch.float16)], ids=["f32", "f16"])
@pytest.mark.parametrize("weighted", [False, True], ids=["plain", "weighted"])
@pytest.mark.parametrize("idx", [np.int32, np.int64], ids=["i32", "i64"])
def test_blocked_backward_equals_reference_order_compressed_gradient(ce, oracle, elem, weighted, idx):
    """B = 40,000, H = 32 (1.28 M lookups: the sliced backward path), W = 128, integer data (sums exact in any
    order): rows, their order and inverse_mapping equal the oracle's compressed gradient of the FULLY sorted order."""
    ncat, W, B, H = 30_000, 128, 40_000, 32
    a = oracle.allocate_forward(ncat, W, B, H, alpha=1.15, elem=elem[0], index=idx)
    ids = a["indices"]
    w = a["weights"] if weighted else None                                   # 0.5 / 0.25
    gy = (np.mod(oracle.allocate_grad_y(B * W).reshape(B, W), 3) - 1).astype(elem[0])
    sid = oracle.extract_row_ids_from_fixed(B, H).astype(idx)
    ti, ts, tw = oracle.transpose(sid, ids, w)
    o_remap = oracle.compute_compressed_grad_indices(ti)
    nu = int(o_remap[-1]) + 1
    want, want_inv = oracle.embedding_backward(gy.astype(np.float32), W, nu, ti, ts, o_remap,
                                               None if tw is None else tw.astype(np.float32))
    assert np.abs(want).max() < 512            # every partial sum is exact in fp16 too (multiples of 0.25)
    want = want.astype(elem[0])
    for P in (2, 3, 8):
        t_idx, t_sid, t_w = ce.transpose_fixed_hotness(dev(ids), B, H, dev(w), num_categories=ncat, sample_blocks=P)
        remap, table, nu_dev = ce.compute_compressed_grad_indices_blocked(t_idx, P)
        assert int(nu_dev.item()) == nu
        grad, inv = ce.embedding_backward(dev(gy), nu, t_idx, t_sid, remap, t_w, sample_blocks=P, block_row_ids=table)
        assert np.array_equal(host(inv), want_inv), P
        assert np.array_equal(host(grad), want), P
        # num_unique left on the device: over-allocated buffers, rows past the last id untouched
        cap = min(B * H, ncat)
        g2 = torch.full((cap, W), 3.0, dtype=elem[1], device="cuda")
        i2 = torch.full((cap,), -5, dtype=t_idx.dtype, device="cuda")
        ce.embedding_backward(dev(gy), None, t_idx, t_sid, remap, t_w, grad_embedding=g2, inverse_mapping=i2,
                              sample_blocks=P, block_row_ids=table)
        assert np.array_equal(host(g2[:nu]), want) and np.array_equal(host(i2[:nu]), want_inv)
        assert float(g2[nu:].min()) == 3.0 and int(i2[nu:].max()) == -5
        # skip_grad_init: the caller zeroed the buffer
        g3 = torch.zeros((nu, W), dtype=elem[1], device="cuda")
        ce.embedding_backward(dev(gy), nu, t_idx, t_sid, remap, t_w, skip_grad_init=True, grad_embedding=g3,
                              sample_blocks=P, block_row_ids=table)
        assert np.array_equal(host(g3), want)
    torch.cuda.synchronize()
    assert ce._lib.lib().cuembed_peek_last_error() == 0


@pytest.mark.parametrize("segment_len", [8, 24, 64, 128])
def test_blocked_backward_forced_segment_lengths_and_csr(ce, oracle, segment_len):
    """ragged CSR bags (a sample's lookups may straddle a block boundary), long runs that cross workgroups inside a
    block and rows that are a workgroup's edge row in one block and an interior row in another"""
    ncat, W, B, H = 2_000, 64, 30_000, 16
    a = oracle.allocate_forward(ncat, W, B, H, alpha=1.3, elem=np.float32, is_csr=True)
    ids, off = a["indices"], a["offsets"]
    nnz = int(off[-1])
    sid = oracle.extract_row_ids_from_csr(off)
    gy = (np.mod(oracle.allocate_grad_y(B * W).reshape(B, W), 5) - 2).astype(np.float32)
    ti, ts, _ = oracle.transpose(sid, ids, None)
    o_remap = oracle.compute_compressed_grad_indices(ti)
    nu = int(o_remap[-1]) + 1
    want, want_inv = oracle.embedding_backward(gy, W, nu, ti, ts, o_remap, None)
    ce.set_backward_tuning(segment_len=segment_len)
    try:
        for P in (2, 5):
            t_idx, t_sid, _ = ce.transpose(dev(sid), dev(ids), num_categories=ncat, sample_blocks=P)
            remap, table, _ = ce.compute_compressed_grad_indices_blocked(t_idx, P)
            grad, inv = ce.embedding_backward(dev(gy), nu, t_idx, t_sid, remap, sample_blocks=P, block_row_ids=table)
            assert np.array_equal(host(inv), want_inv) and np.array_equal(host(grad), want), (P, nnz)
    finally:
        ce.set_backward_tuning(0, 0)


def test_blocked_calls_reject_what_the_library_would_abort_on(ce):
    t = torch.zeros(4096 * 90, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        ce.compute_compressed_grad_indices_blocked(t, 64)                  # 45 blocks of two tiles
    gy = torch.zeros((8, 4), device="cuda")
    with pytest.raises(ValueError):
        ce.embedding_backward(gy, 10, t, t, None, sample_blocks=2)         # dense gradient
