"""cuembed::EmbeddingBackwardRunAware (extension): the few very long runs of a skewed batch are
summed chunk-major out of LDS (hot_row_kernels.hpp), everything else goes through the segmented
kernel.  Results must equal EmbeddingBackward's and the oracle's -- bit for bit on data whose partial
sums are exactly representable (grad_y in {-1,0,1}, weights 0.5 / 0.25), which makes the result
independent of where a run is cut into partial sums.

The hot path normally needs >= 2^20 lookups; `set_backward_tuning(hot_stride=...)` forces it on
small shapes (a run is hot when it contains two consecutive multiples of the stride)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ELEMS = [(np.float32, torch.float32), (np.float16, torch.float16)]


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return None if t is None else t.cpu().numpy()


@pytest.fixture(scope="module")
def ce():
    import cuembed_amd
    assert torch.cuda.is_available()
    return cuembed_amd


@pytest.fixture
def tuning(ce):
    yield ce.set_backward_tuning
    ce.set_backward_tuning(0, 0, 0)


def _coo(oracle, ncat, W, B, H, alpha, idx_t, plant=()):
    """Sorted COO of a fixed-hotness batch; plant = [(row, first_sample, num_samples)] makes `row`
    the lookup of column 0 for a range of samples (a run of exactly that length, sample ids ascending)."""
    a = oracle.allocate_forward(ncat, W, B, H, alpha=alpha, index=idx_t)
    idx = a["indices"].copy().reshape(B, H)
    for row, s0, n in plant:
        idx[idx == row] = (row + 1) % ncat            # the planted row appears nowhere else
        idx[s0:s0 + n, 0] = row
    idx = idx.reshape(-1)
    sid = oracle.extract_row_ids_from_fixed(B, H, idx_t)
    ti, ts, tw = oracle.transpose(sid, idx, a["weights"], stable=True)
    return ti, ts, tw


def _check(ce, oracle, elem, ti, ts, tw, B, W, ncat, compressed, weighted, skip_init=False):
    ints = oracle.allocate_grad_y(B * W).reshape(B, W)
    gy = (np.mod(ints, 3) - 1).astype(elem[0])
    w = tw.astype(elem[0]) if weighted else None
    remap = oracle.compute_compressed_grad_indices(ti) if compressed else None
    rows = int(remap[-1]) + 1 if compressed else ncat
    want, winv = oracle.embedding_backward(gy.astype(np.float32), W, rows, ti, ts, remap,
                                           None if w is None else w.astype(np.float32))
    # exactness premise in fp16: integers below 2048, or multiples of 0.25 below 512
    assert np.abs(want).max() < (512 if weighted else 2048)
    args = (dev(gy), rows, dev(ti), dev(ts), dev(remap), dev(w))
    plain, pinv = ce.embedding_backward(*args)
    buf = torch.zeros((rows, W), dtype=elem[1], device="cuda") if skip_init else \
        torch.full((rows, W), 55.0, dtype=elem[1], device="cuda")
    got, ginv = ce.embedding_backward(*args, skip_grad_init=skip_init, grad_embedding=buf, run_aware=True)
    assert np.array_equal(host(got).astype(np.float32), want)
    assert torch.equal(got, plain)
    if compressed:
        assert np.array_equal(host(ginv), winv) and np.array_equal(host(pinv), winv)
    torch.cuda.synchronize()
    assert ce._lib.lib().cuembed_peek_last_error() == 0


@pytest.mark.parametrize("elem", ELEMS, ids=["f32", "f16"])
@pytest.mark.parametrize("idx_t", [np.int32, np.int64], ids=["i32", "i64"])
@pytest.mark.parametrize("shape", [(64, 3000, 16, 300), (256, 2000, 8, 150), (8, 5000, 8, 200), (512, 1500, 6, 80)],
                         ids=lambda s: "w%d_b%d_h%d_c%d" % s)
def test_run_aware_equals_plain_and_oracle(ce, oracle, tuning, elem, idx_t, shape):
    """Power-law batches over few categories: dozens of runs of hundreds to thousands of lookups.
    Row widths cover 2 / 4 / 16 / 32 / 64 lanes per row (1 to 32 lookups per wavefront step)."""
    W, B, H, ncat = shape
    if W * np.dtype(elem[0]).itemsize < 16 or W * np.dtype(elem[0]).itemsize > 1024:
        pytest.skip("hot path needs rows of 16..1024 bytes")
    tuning(hot_stride=256)
    ti, ts, tw = _coo(oracle, ncat, W, B, H, 1.15, idx_t)
    assert np.bincount(ti.astype(np.int64)).max() > 1000
    for compressed, weighted, skip_init in [(True, False, False), (False, True, False), (True, True, True),
                                            (False, False, True)]:
        _check(ce, oracle, elem, ti, ts, tw, B, W, ncat, compressed, weighted, skip_init)


def test_more_hot_runs_than_slots_and_runs_at_both_ends(ce, oracle, tuning):
    """100 categories, every one a run of ~1000 lookups: with stride 256 all are hot, 64 get a
    slot, the rest must stay in the segmented kernel; the first run starts at position 0 and the
    last one ends at nnz."""
    tuning(hot_stride=256)
    W, B, H, ncat = 128, 6000, 16, 100
    ti, ts, tw = _coo(oracle, ncat, W, B, H, 0.0, np.int32)
    counts = np.bincount(ti)
    assert (counts > 600).sum() >= 90
    for elem in ELEMS:
        for compressed in (False, True):
            _check(ce, oracle, elem, ti, ts, tw, B, W, ncat, compressed, weighted=False)


@pytest.mark.parametrize("length", [511, 512, 513, 1023, 1024, 1025, 2048, 2049, 3000])
def test_planted_run_lengths_around_the_detection_stride(ce, oracle, tuning, length):
    """One planted run of an exact length among short runs (stride 512: hot when it spans two
    multiples of 512), at the start, in the middle and at the end of the sample range; nz-block
    boundaries fall inside, at the start and at the end of the run."""
    tuning(hot_stride=512)
    W, B, H, ncat = 64, 4000, 4, 50000
    for s0 in (0, 777, B - length):
        ti, ts, tw = _coo(oracle, ncat, W, B, H, 0.0, np.int32, plant=[(31337, s0, length)])
        assert np.bincount(ti)[31337] == length
        _check(ce, oracle, ELEMS[1], ti, ts, tw, B, W, ncat, compressed=True, weighted=True)
        _check(ce, oracle, ELEMS[0], ti, ts, tw, B, W, ncat, compressed=False, weighted=False)


def test_two_planted_runs_share_chunks_and_sample_gaps(ce, oracle, tuning):
    """Two hot runs whose sample ranges overlap partly, leaving chunks that contain lookups of
    one, both or neither of them."""
    tuning(hot_stride=256)
    W, B, H, ncat = 256, 3000, 4, 40000
    ti, ts, tw = _coo(oracle, ncat, W, B, H, 0.0, np.int32, plant=[(10, 100, 1500), (20000, 1200, 1700)])
    for elem in ELEMS:
        _check(ce, oracle, elem, ti, ts, tw, B, W, ncat, compressed=True, weighted=True)


def test_many_samples_several_fills_per_chunk(ce, oracle, tuning):
    """More samples than kHotMaxChunks LDS fills (1024 x 128 samples of 1 KiB rows): every
    workgroup loops over two fills and accumulates its partial rows across them."""
    tuning(hot_stride=4096)
    W, B, H, ncat = 512, 140000, 3, 200000
    ti, ts, tw = _coo(oracle, ncat, W, B, H, 0.0, np.int32, plant=[(5, 0, 139000), (77777, 60000, 70000)])
    _check(ce, oracle, ELEMS[1], ti, ts, tw, B, W, ncat, compressed=True, weighted=False)


def test_shapes_that_cannot_use_the_hot_path_fall_back(ce, oracle, tuning):
    """Rows that do not split into 16-byte lanes of one wavefront, concat-like calls (one lookup per
    grad_y row) and tiny inputs: the run-aware entry point must give EmbeddingBackward's result."""
    tuning(hot_stride=256)
    for (W, B, H, ncat) in [(36, 1023, 26, 200), (2042, 37, 9, 30), (64, 5, 3, 4)]:
        ti, ts, tw = _coo(oracle, ncat, W, B, H, 1.15, np.int32)
        _check(ce, oracle, ELEMS[0], ti, ts, tw, B, W, ncat, compressed=True, weighted=True)
    # concat: grad_y has one row per lookup
    W, B, H, ncat = 64, 400, 8, 20
    a = oracle.allocate_forward(ncat, W, B, H, alpha=1.15)
    nnz = B * H
    sid = oracle.extract_row_ids_for_concat(nnz)
    ti, ts, _ = oracle.transpose(sid, a["indices"])
    gy = (np.mod(oracle.allocate_grad_y(nnz * W).reshape(nnz, W), 3) - 1).astype(np.float32)
    want, _ = oracle.embedding_backward(gy, W, ncat, ti, ts)
    got, _ = ce.embedding_backward(dev(gy), ncat, dev(ti), dev(ts), run_aware=True)
    assert np.array_equal(host(got), want)
    assert ce.backward_workspace_bytes(torch.float32, torch.int32, 36, 1 << 22, 65536) == 256


def test_default_heuristics_at_a_million_lookups(ce, oracle):
    """No forced stride: >= 2^20 lookups switch the hot path on by themselves (stride 4096)."""
    W, B, H, ncat = 128, 16384, 64, 100_000
    ti, ts, tw = _coo(oracle, ncat, W, B, H, 1.15, np.int32)
    assert (np.bincount(ti) > 8192).sum() >= 3
    _check(ce, oracle, ELEMS[1], ti, ts, tw, B, W, ncat, compressed=True, weighted=False)
    _check(ce, oracle, ELEMS[0], ti, ts, tw, B, W, ncat, compressed=False, weighted=True)


def test_run_aware_bf16(ce, oracle, tuning):
    """bfloat16 gradients through the hot path (fp32 partial sums, bf16 flush atomics): equal to the plain
    entry point and, as fp32 values, to an fp32 oracle run on the same exactly-representable data."""
    tuning(hot_stride=256)
    W, B, H, ncat = 128, 3000, 12, 250
    ti, ts, tw = _coo(oracle, ncat, W, B, H, 1.15, np.int32)
    gy32 = (np.mod(oracle.allocate_grad_y(B * W).reshape(B, W), 3) - 1).astype(np.float32)
    gy = torch.from_numpy(gy32).cuda().bfloat16()
    remap = oracle.compute_compressed_grad_indices(ti)
    nu = int(remap[-1]) + 1
    want, winv = oracle.embedding_backward(gy32, W, nu, ti, ts, remap)
    assert np.abs(want).max() < 256                      # integers below 2^8 are exact in bf16
    plain, _ = ce.embedding_backward(gy, nu, dev(ti), dev(ts), dev(remap))
    got, ginv = ce.embedding_backward(gy, nu, dev(ti), dev(ts), dev(remap), run_aware=True)
    assert torch.equal(got, plain)
    assert np.array_equal(got.float().cpu().numpy(), want) and np.array_equal(host(ginv), winv)
