"""GPU parity: index transforms and EmbeddingBackward (HIP, through the C ABI) vs the CPU
oracle -- bit-exact.  Shapes follow tests/test_embedding_transpose.cu, test_embedding_backward.cu
and the test_embedding_against_cpu.cu sweep of the reference."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ELEMS = [(np.float32, torch.float32), (np.float16, torch.float16)]
IDXS = [(np.int32, torch.int32), (np.int64, torch.int64)]


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
    """cuembed::SetBackwardTuning for one test; the heuristics are restored afterwards."""
    yield ce.set_backward_tuning
    ce.set_backward_tuning(0, 0)


@pytest.fixture(scope="module")
def kats(golden_dir):
    with open(os.path.join(golden_dir, "reference_kats.json")) as f:
        return json.load(f)


# ---- row ids ------------------------------------------------------------------
@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
def test_extract_row_ids(ce, oracle, kats, idx):
    r = kats["readme_examples"]
    e = r["extract_fixed"]
    assert host(ce.extract_row_ids_from_fixed(e["batch_size"], e["num_hots"], idx[1])).tolist() == e["row_ids"]
    assert host(ce.extract_row_ids_for_concat(4, idx[1])).tolist() == r["extract_concat"]["row_ids"]
    for off_t in (np.int32, np.int64):
        off = np.array(r["extract_csr"]["offsets"], dtype=off_t)
        assert host(ce.extract_row_ids_from_csr(dev(off), dtype=idx[1])).tolist() == r["extract_csr"]["row_ids"]
    for B, H in [(1, 1), (3, 4), (1023, 26), (65536, 64), (5000, 1)]:
        assert np.array_equal(host(ce.extract_row_ids_from_fixed(B, H, idx[1])),
                              oracle.extract_row_ids_from_fixed(B, H, idx[0]))
    assert np.array_equal(host(ce.extract_row_ids_for_concat(100003, idx[1])),
                          oracle.extract_row_ids_for_concat(100003, idx[0]))
    rng = np.random.default_rng(2)
    for B, H in [(1, 5), (7, 0), (1023, 26), (1024, 3), (1025, 128), (70000, 9), (3000, 700)]:
        lens = rng.integers(0, H + 1, B)
        lens[rng.integers(0, B, max(1, B // 5))] = 0          # plenty of empty bags
        off = np.concatenate([[0], np.cumsum(lens)])
        for off_t in (np.int32, np.int64):
            got = host(ce.extract_row_ids_from_csr(dev(off.astype(off_t)), dtype=idx[1]))
            assert np.array_equal(got, oracle.extract_row_ids_from_csr(off.astype(off_t), idx[0])), (B, H)


# ---- transpose -----------------------------------------------------------------
@pytest.mark.parametrize("elem", ELEMS, ids=["f32", "f16"])
@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
def test_transpose_kat(ce, kats, elem, idx):
    k = kats["transpose"]
    for weighted in (False, True):
        w = dev(np.array(k["weights"], dtype=elem[0])) if weighted else None
        ti, ts, tw = ce.transpose(dev(np.array(k["sample_ids"], dtype=idx[0])),
                                  dev(np.array(k["indices"], dtype=idx[0])), w)
        assert host(ti).tolist() == k["transpose_indices"]
        assert host(ts).tolist() == k["transpose_sample_ids"]
        if weighted:
            assert host(tw).tolist() == k["transpose_weights"]
        else:
            assert tw is None


@pytest.mark.parametrize("elem", ELEMS, ids=["f32", "f16"])
@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
def test_transpose_stable_with_repeats_and_remap(ce, oracle, elem, idx):
    """Repeated indices inside a sample: the device contract is a STABLE sort (input order kept
    inside a run), which is what a stable radix sort gives (index_transforms.cuh:95-137)."""
    rng = np.random.default_rng(4)
    for nnz, ncat in [(1, 5), (257, 3), (10007, 50), (300000, 1000), (300000, 3_000_000_000 if idx[0] == np.int64 else 2_000_000_000)]:
        cols = rng.integers(0, ncat, nnz).astype(idx[0])
        rows = rng.integers(0, 1000, nnz).astype(idx[0])
        w = rng.uniform(0, 1, nnz).astype(elem[0])
        for weights in (None, w):
            ti, ts, tw = ce.transpose(dev(rows), dev(cols), dev(weights))
            oi, os_, ow = oracle.transpose(rows, cols, weights, stable=True)
            assert np.array_equal(host(ti), oi) and np.array_equal(host(ts), os_)
            if weights is not None:
                assert np.array_equal(host(tw).view(np.uint8), ow.view(np.uint8))
        remap = ce.compute_compressed_grad_indices(ti)
        assert np.array_equal(host(remap), oracle.compute_compressed_grad_indices(oi))


def test_compressed_readme_example(ce, kats):
    c = kats["readme_examples"]["compressed"]
    for dt in (np.int32, np.int64):
        got = host(ce.compute_compressed_grad_indices(dev(np.array(c["indices"], dtype=dt))))
        assert got.tolist() == c["remapped"]


def test_transpose_workspace_two_phase(ce):
    """work == NULL returns the scratch size; an undersized buffer is refused by the host layer."""
    n = 100000
    need = ce.transpose_workspace_bytes(n, torch.int64, torch.float32)
    assert need >= 2 * n * 4
    rows = torch.arange(n, device="cuda")
    with pytest.raises(ValueError):
        ce.transpose(rows, rows, torch.ones(n, device="cuda"), workspace=torch.empty(16, dtype=torch.uint8, device="cuda"))
    big = torch.empty(need + 1024, dtype=torch.uint8, device="cuda")
    ti, ts, tw = ce.transpose(rows, rows.flip(0).contiguous(), torch.ones(n, device="cuda"), workspace=big)
    assert torch.equal(ti, rows) and torch.equal(ts, rows.flip(0))


# ---- backward --------------------------------------------------------------------
@pytest.mark.parametrize("elem", ELEMS, ids=["f32", "f16"])
@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
@pytest.mark.parametrize("compressed", [False, True], ids=["full", "compressed"])
@pytest.mark.parametrize("skip_init", [False, True], ids=["init", "skipinit"])
def test_backward_kat(ce, kats, elem, idx, compressed, skip_init):
    k = kats["backward"]
    W = k["embed_width"]
    t_idx = dev(np.array(k["transpose_indices"], dtype=idx[0]))
    remap = dev(np.array(k["transpose_remapped_indices"], dtype=idx[0])) if compressed else None
    w = dev(np.array(k["transpose_weights"], dtype=elem[0]))
    rows = k["num_unique"] if compressed else k["num_categories"]
    pre = "compressed_grad_" if compressed else "grad_"
    for mode in ("sum", "concat"):
        sid = dev(np.array(k["transpose_sample_ids" + ("_concat" if mode == "concat" else "")], dtype=idx[0]))
        gy = dev(np.array(k["grad_y_" + mode], dtype=elem[0]).reshape(-1, W))
        for weighted in (False, True):
            buf = torch.zeros((rows, W), dtype=elem[1], device="cuda") if skip_init else \
                torch.full((rows, W), 77, dtype=elem[1], device="cuda")
            grad, inv = ce.embedding_backward(gy, rows, t_idx, sid, remap, w if weighted else None,
                                              skip_grad_init=skip_init, grad_embedding=buf)
            assert host(grad).ravel().tolist() == k[pre + mode + ("_weighted" if weighted else "")]
            if compressed:
                assert host(inv).tolist() == k["inverse_mapping"]
            else:
                assert inv is None


SWEEP_SHAPES = [(2, 3, 4), (4, 3, 4), (32, 1023, 26), (36, 1023, 26), (512, 3, 63), (512, 1023, 63),
                (514, 1023, 63)]


@pytest.mark.parametrize("elem", ELEMS, ids=["f32", "f16"])
@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
@pytest.mark.parametrize("shape", SWEEP_SHAPES, ids=lambda s: "w%d_b%d_h%d" % s)
def test_pipeline_sweep_against_oracle(ce, oracle, elem, idx, shape):
    """extract -> transpose -> (remap) -> backward, every stage bit-exact against the oracle on
    the reference's synthetic inputs (integer grad_y in [-10,10], weights 0.5/0.25:
    utils/src/embedding_allocation.cu:160-168, :234-237)."""
    W, B, H = shape
    if elem[0] == np.float16 and (W * 2) % 4:
        pytest.skip("row bytes not a multiple of 4")
    ncat = 20 * 1024
    for mode, csr, weighted, compressed in [("sum", False, False, False), ("sum", True, False, False),
                                             ("sum", False, True, False), ("sum", True, True, True),
                                             ("sum", False, False, True), ("concat", False, False, False),
                                             ("concat", False, False, True)]:
        a = oracle.allocate_forward(ncat, W, B, H, alpha=0.0, is_csr=csr, elem=elem[0], index=idx[0])
        indices, nnz = a["indices"], a["indices"].shape[0]
        if mode == "concat":
            o_sid = oracle.extract_row_ids_for_concat(nnz, idx[0])
            d_sid = ce.extract_row_ids_for_concat(nnz, idx[1])
        elif csr:
            o_sid = oracle.extract_row_ids_from_csr(a["offsets"], idx[0])
            d_sid = ce.extract_row_ids_from_csr(dev(a["offsets"]), nnz=nnz, dtype=idx[1])
        else:
            o_sid = oracle.extract_row_ids_from_fixed(B, H, idx[0])
            d_sid = ce.extract_row_ids_from_fixed(B, H, idx[1])
        assert np.array_equal(host(d_sid), o_sid)
        w = a["weights"] if weighted else None
        o_ti, o_ts, o_tw = oracle.transpose(o_sid, indices, w, stable=True)
        d_ti, d_ts, d_tw = ce.transpose(d_sid, dev(indices), dev(w))
        assert np.array_equal(host(d_ti), o_ti) and np.array_equal(host(d_ts), o_ts)
        if weighted:
            assert np.array_equal(host(d_tw), o_tw)
        o_remap = d_remap = None
        rows = ncat
        if compressed:
            o_remap = oracle.compute_compressed_grad_indices(o_ti)
            d_remap = ce.compute_compressed_grad_indices(d_ti)
            assert np.array_equal(host(d_remap), o_remap)
            rows = int(o_remap[-1]) + 1 if nnz else 0
        n_out = nnz if mode == "concat" else B
        gy = oracle.allocate_grad_y(n_out * W, elem[0]).reshape(n_out, W)
        o_grad, o_inv = oracle.embedding_backward(gy, W, rows, o_ti, o_ts, o_remap, o_tw)
        d_grad, d_inv = ce.embedding_backward(dev(gy), rows, d_ti, d_ts, d_remap, d_tw)
        assert np.array_equal(host(d_grad).view(np.uint8), o_grad.view(np.uint8)), (mode, csr, weighted, compressed)
        if compressed:
            assert np.array_equal(host(d_inv), o_inv)


def test_backward_long_runs_power_law(ce, oracle):
    """alpha = 1.15 over few categories: runs far longer than one nz-segment, so many segments
    share a row and combine through atomics.  Data is chosen so that every partial sum is exactly
    representable (fp32: integers/8 times 0.5|0.25; fp16: values in {-1,0,1}, runs < 2048), which
    makes the result independent of the order in which segments arrive."""
    W, B, H, ncat = 64, 2000, 16, 600
    a = oracle.allocate_forward(ncat, W, B, H, alpha=1.15)
    sid = oracle.extract_row_ids_from_fixed(B, H)
    ti, ts, tw = oracle.transpose(sid, a["indices"], a["weights"])
    assert 1000 < np.bincount(ti).max() < 2048
    ints = oracle.allocate_grad_y(B * W).reshape(B, W)
    gy32 = (ints / 8).astype(np.float32)
    for weights in (None, tw):
        want, _ = oracle.embedding_backward(gy32, W, ncat, ti, ts, None, weights)
        got, _ = ce.embedding_backward(dev(gy32), ncat, dev(ti), dev(ts), None, dev(weights))
        assert np.array_equal(host(got), want)
    gy16 = (np.mod(ints, 3) - 1).astype(np.float16)
    want, _ = oracle.embedding_backward(gy16.astype(np.float32), W, ncat, ti, ts)
    got, _ = ce.embedding_backward(dev(gy16), ncat, dev(ti), dev(ts))
    assert np.array_equal(host(got).astype(np.float32), want)


@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
def test_transpose_bounded_keys_extension(ce, oracle, idx):
    """num_categories bound (this library's extension): fewer radix passes, identical result."""
    rng = np.random.default_rng(8)
    for ncat in (2, 1000, 65536, 65537, 10_000_000):
        nnz = 200003
        cols = rng.integers(0, ncat, nnz).astype(idx[0])
        cols[0] = ncat - 1                                     # the largest admissible key is present
        rows = rng.integers(0, 5000, nnz).astype(idx[0])
        w = rng.uniform(0, 1, nnz).astype(np.float32)
        oi, os_, ow = oracle.transpose(rows, cols, w, stable=True)
        for weights in (None, w):
            ti, ts, tw = ce.transpose(dev(rows), dev(cols), dev(weights), num_categories=ncat)
            assert np.array_equal(host(ti), oi) and np.array_equal(host(ts), os_), ncat
            if weights is not None:
                assert np.array_equal(host(tw), ow)


@pytest.mark.parametrize("slices", ["2", "4"])
def test_backward_column_slices_small_shapes(ce, oracle, slices, tuning):
    """Forces the XCD column-slice mapping (normally only used for >= 1M lookups) on small, odd
    shapes: partial grids of 8-workgroup rounds, segments shorter than the unroll, long runs."""
    tuning(column_slices=int(slices))
    for (W, B, H, ncat, alpha) in [(128, 1023, 26, 20480, 0.0), (256, 300, 63, 500, 1.15), (64, 5, 3, 50, 0.0),
                                   (512, 2000, 16, 600, 1.15)]:
        for elem in ELEMS:
            a = oracle.allocate_forward(ncat, W, B, H, alpha=alpha, elem=elem[0])
            sid = oracle.extract_row_ids_from_fixed(B, H)
            ti, ts, tw = oracle.transpose(sid, a["indices"], a["weights"])
            ints = oracle.allocate_grad_y(B * W).reshape(B, W)
            gy = (np.mod(ints, 3) - 1).astype(elem[0])          # {-1,0,1}: exact for any run length here
            remap = oracle.compute_compressed_grad_indices(ti)
            nu = int(remap[-1]) + 1
            for weights in (None, tw):
                if weights is not None and elem[0] == np.float16 and alpha > 0:
                    continue                                     # 0.25-steps x long runs exceed fp16's 11 bits
                want, winv = oracle.embedding_backward(gy.astype(np.float32), W, nu, ti, ts, remap,
                                                       None if weights is None else weights.astype(np.float32))
                got, ginv = ce.embedding_backward(dev(gy), nu, dev(ti), dev(ts), dev(remap), dev(weights))
                assert np.array_equal(host(got).astype(np.float32), want), (W, B, H, elem[0], weights is not None)
                assert np.array_equal(host(ginv), winv)


@pytest.mark.parametrize("segment_len", [8, 12, 20, 40, 64, 256, 4096])
def test_backward_forced_segment_lengths(ce, oracle, segment_len, tuning):
    """The segment walk is a rolling window of 8 gathers over a uniform loop: forced lengths that are
    not multiples of 8 (rounded down by the launcher), a single batch per segment (8), segments
    longer than most runs and longer than the whole input (4096), with 1 and 4 column slices; the
    last workgroup is ragged (nnz is not a multiple of anything) and the grid has idle workgroups."""
    for slices in (1, 4):
        tuning(segment_len=segment_len, column_slices=slices)
        for (W, B, H, ncat, alpha) in [(256, 700, 37, 900, 1.15), (128, 123, 5, 40, 0.0), (64, 3001, 1, 7, 0.0)]:
            a = oracle.allocate_forward(ncat, W, B, H, alpha=alpha, elem=np.float32)
            sid = oracle.extract_row_ids_from_fixed(B, H)
            ti, ts, tw = oracle.transpose(sid, a["indices"], a["weights"])
            gy = (np.mod(oracle.allocate_grad_y(B * W).reshape(B, W), 3) - 1).astype(np.float32)
            remap = oracle.compute_compressed_grad_indices(ti)
            nu = int(remap[-1]) + 1
            for weights in (None, tw):
                want, winv = oracle.embedding_backward(gy, W, nu, ti, ts, remap, weights)
                got, ginv = ce.embedding_backward(dev(gy), nu, dev(ti), dev(ts), dev(remap), dev(weights))
                assert np.array_equal(host(got), want), (segment_len, slices, W, B, H, weights is not None)
                assert np.array_equal(host(ginv), winv)
            want, _ = oracle.embedding_backward(gy, W, ncat, ti, ts)          # dense gradient, memset path
            got, _ = ce.embedding_backward(dev(gy), ncat, dev(ti), dev(ts))
            assert np.array_equal(host(got), want), (segment_len, slices, W, B, H, "dense")


@pytest.mark.parametrize("W", [128, 256])
def test_backward_million_lookups_default_heuristics(ce, oracle, W):
    """nnz >= 2^20 takes the column-sliced path by default (fp32: W=128 four 128-byte slices,
    W=256 eight, one per XCD)."""
    B, H, ncat = 16384, 64, 100_000
    a = oracle.allocate_forward(ncat, W, B, H, alpha=1.15)
    sid = oracle.extract_row_ids_from_fixed(B, H)
    ti, ts, _ = oracle.transpose(sid, a["indices"])
    gy = oracle.allocate_grad_y(B * W).reshape(B, W)
    want, _ = oracle.embedding_backward(gy, W, ncat, ti, ts)
    got, _ = ce.embedding_backward(dev(gy), ncat, dev(ti), dev(ts))
    assert np.array_equal(host(got), want)


@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
def test_transpose_tile_boundaries_and_degenerate_keys(ce, oracle, idx):
    """The hand-written radix sort works on 4096-key tiles with 64-key wave rounds: sizes around
    those boundaries, all-equal keys (every key of a tile in one bin), already sorted and reverse
    sorted keys, the largest representable key, and remap on the results."""
    rng = np.random.default_rng(1)
    big = np.iinfo(idx[0]).max
    for nnz in (1, 63, 64, 65, 4095, 4096, 4097, 8191, 8192 + 17, 3 * 4096):
        for name, cols in [("equal", np.full(nnz, 7)), ("sorted", np.arange(nnz)),
                           ("reverse", np.arange(nnz)[::-1].copy()), ("two", rng.integers(0, 2, nnz)),
                           ("huge", rng.integers(big - 5, big, nnz, endpoint=True)),
                           ("random", rng.integers(0, 1 << 20, nnz))]:
            cols = cols.astype(idx[0])
            rows = rng.integers(0, 100, nnz).astype(idx[0])
            w = rng.uniform(0, 1, nnz).astype(np.float16)
            oi, os_, ow = oracle.transpose(rows, cols, w, stable=True)
            ti, ts, tw = ce.transpose(dev(rows), dev(cols), dev(w))
            assert np.array_equal(host(ti), oi) and np.array_equal(host(ts), os_), (nnz, name)
            assert np.array_equal(host(tw).view(np.uint16), ow.view(np.uint16)), (nnz, name)
            assert np.array_equal(host(ce.compute_compressed_grad_indices(ti)),
                                  oracle.compute_compressed_grad_indices(oi)), (nnz, name)


@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
@pytest.mark.parametrize("weights", [None, np.float32, np.float16], ids=["unweighted", "w32", "w16"])
def test_one_launch_index_work_of_small_batches(ce, oracle, idx, weights):
    """Up to 4,096 lookups the whole index work -- row ids, stable sort, remap -- is ONE launch of one 1024-thread
    workgroup (block_sort_kernels.hpp): sizes around its round (64), chunk (1024 x rounds) and range (4096) boundaries,
    and on into the one-launch-per-pass range; keys with one or all varying digits, negative keys, runs longer than a wavefront; through the
    reference-shaped call sequence, through transpose(remapped=True) and through transpose_fixed_hotness(remapped=True).
    Same results from the same calls on either side."""
    rng = np.random.default_rng(77)
    info = np.iinfo(idx[0])
    for nnz in (1, 2, 63, 64, 65, 1000, 1023, 1024, 1025, 2047, 3000, 4095, 4096, 4097, 5001, 8192, 12345, 16383, 16384,
                16385):
        kinds = [("random20", rng.integers(0, 1 << 20, nnz)), ("few", rng.integers(0, 3, nnz)),
                 ("full", rng.integers(info.min, info.max, nnz, endpoint=True)), ("equal", np.full(nnz, 12345))]
        for name, cols in kinds:
            cols = cols.astype(idx[0])
            hot = 1 if nnz % 7 else 7
            batch = nnz // hot
            n = batch * hot
            if n == 0:
                continue
            cols = cols[:n]
            w = None if weights is None else rng.uniform(0, 1, n).astype(weights)
            o_sid = oracle.extract_row_ids_from_fixed(batch, hot, idx[0])
            oi, os_, ow = oracle.transpose(o_sid, cols, w, stable=True)
            oremap = oracle.compute_compressed_grad_indices(oi)

            def check(got, what):
                assert np.array_equal(host(got[0]), oi) and np.array_equal(host(got[1]), os_), (nnz, name, what)
                if w is not None:
                    assert np.array_equal(host(got[2]).view(np.uint8), ow.view(np.uint8)), (nnz, name, what)
                if len(got) > 3:
                    assert np.array_equal(host(got[3]), oremap), (nnz, name, what)

            sid = ce.extract_row_ids_from_fixed(batch, hot, idx[1], "cuda")
            check(ce.transpose(sid, dev(cols), dev(w)), "transpose")
            check(ce.transpose(sid, dev(cols), dev(w), remapped=True), "transpose+remap")
            check(ce.transpose_fixed_hotness(dev(cols), batch, hot, dev(w), remapped=True), "fixed+remap")
            if name == "random20":
                check(ce.transpose_fixed_hotness(dev(cols), batch, hot, dev(w), num_categories=1 << 20, remapped=True),
                      "fixed+remap bounded")
    assert ce._lib.lib().cuembed_peek_last_error() == 0


@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
@pytest.mark.parametrize("weights", [None, np.float16], ids=["unweighted", "w16"])
def test_transpose_chained_and_tiled_paths(ce, oracle, idx, weights):
    """Between 4,097 and 229,376 lookups the sort runs ONE launch per pass: every scatter pass counts the histogram of
    the next working pass with global atomics, per (destination tile of 1,024 keys, next digit), aggregated over runs
    of equal neighbours.  Sizes around its tile and range boundaries (229,377 and up: the three-launch passes); hot
    keys (runs far longer than a wavefront, whose atomics must aggregate), two-valued and constant keys, keys using all
    bits incl. the sign; reference call sequence and the fused calls."""
    rng = np.random.default_rng(5)
    info = np.iinfo(idx[0])
    for nnz in (16385, 17407, 17408, 17409, 65536, 100003, 229376, 229377, 262144, 270001):
        hot_keys = np.where(rng.uniform(0, 1, nnz) < 0.4, 777, rng.integers(0, 10_000_000, nnz))
        kinds = [("random24", rng.integers(0, 10_000_000, nnz)), ("hot", hot_keys), ("two", rng.integers(0, 2, nnz) * 65536),
                 ("full", rng.integers(info.min, info.max, nnz, endpoint=True)), ("equal", np.full(nnz, 3))]
        if nnz not in (17408, 65536, 229376, 229377):
            kinds = kinds[:2] + kinds[3:4]
        for name, cols in kinds:
            cols = cols.astype(idx[0])
            hot = 1 if nnz % 3 else 3
            batch = nnz // hot
            n = batch * hot
            cols = cols[:n]
            w = None if weights is None else rng.uniform(0, 1, n).astype(weights)
            o_sid = oracle.extract_row_ids_from_fixed(batch, hot, idx[0])
            oi, os_, ow = oracle.transpose(o_sid, cols, w, stable=True)
            oremap = oracle.compute_compressed_grad_indices(oi)

            def check(got, what):
                assert np.array_equal(host(got[0]), oi) and np.array_equal(host(got[1]), os_), (nnz, name, what)
                if w is not None:
                    assert np.array_equal(host(got[2]).view(np.uint8), ow.view(np.uint8)), (nnz, name, what)
                if len(got) > 3:
                    assert np.array_equal(host(got[3]), oremap), (nnz, name, what)

            sid = ce.extract_row_ids_from_fixed(batch, hot, idx[1], "cuda")
            t = ce.transpose(sid, dev(cols), dev(w))
            check(t, "transpose")
            assert np.array_equal(host(ce.compute_compressed_grad_indices(t[0])), oremap), (nnz, name, "remap")
            check(ce.transpose_fixed_hotness(dev(cols), batch, hot, dev(w), remapped=True), "fixed+remap")
            if name in ("random24", "hot"):
                check(ce.transpose(sid, dev(cols), dev(w), num_categories=10_000_000, num_rows=batch, remapped=True),
                      "bounded+remap")
    assert ce._lib.lib().cuembed_peek_last_error() == 0


@pytest.mark.parametrize("nnz", [3 * 4096 + 123, 40000, 270000], ids=["one_workgroup", "chained", "tiled"])
@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
def test_transpose_skips_identity_passes_on_device(ce, oracle, idx, nnz):
    """Radix passes whose digit is the same for every key are skipped on the device and the
    remaining passes re-route their buffers: every combination of varying digits must still give
    the stable order in the caller's output arrays (odd and even numbers of working passes) -- in each of the three
    implementations of the sort (one workgroup, one launch per pass, three launches per pass)."""
    rng = np.random.default_rng(21)
    ndig = np.dtype(idx[0]).itemsize
    combos = [(0,), (1,), (2,), (3,), (0, 2), (1, 2), (1, 3), (0, 1, 2), (0, 1, 2, 3)]
    if ndig == 8:
        combos += [(5,), (0, 5), (2, 4, 6), (1, 2, 3, 4, 5, 6), (0, 1, 2, 3, 4, 5, 6, 7)]
    for combo in combos:
        cols = np.zeros(nnz, dtype=np.uint64)
        for d in range(ndig):
            hi = 128 if d == ndig - 1 else 256                 # keys stay non-negative
            if d in combo:
                cols |= rng.integers(0, hi, nnz).astype(np.uint64) << np.uint64(8 * d)
            else:
                cols |= np.uint64(int(rng.integers(0, hi))) << np.uint64(8 * d)   # constant digit
        cols = cols.astype(idx[0])
        rows = rng.integers(0, 1000, nnz).astype(idx[0])
        w = rng.uniform(0, 1, nnz).astype(np.float32)
        oi, os_, ow = oracle.transpose(rows, cols, w, stable=True)
        for weights in (None, w):
            ti, ts, tw = ce.transpose(dev(rows), dev(cols), dev(weights))
            assert np.array_equal(host(ti), oi) and np.array_equal(host(ts), os_), combo
            if weights is not None:
                assert np.array_equal(host(tw), ow), combo


def test_backward_compressed_zero_init_without_full_memset(ce, oracle):
    """Compressed gradient with skip_grad_init=False: only rows that can receive atomics and an
    over-allocated tail are zeroed by the library; every other row is overwritten.  The result must
    equal the oracle's (which memsets everything) on a garbage-filled, over-allocated buffer."""
    W, B, H, ncat = 128, 3000, 16, 700
    for elem in ELEMS:
        a = oracle.allocate_forward(ncat, W, B, H, alpha=1.15, elem=elem[0])
        sid = oracle.extract_row_ids_from_fixed(B, H)
        ti, ts, _ = oracle.transpose(sid, a["indices"])
        remap = oracle.compute_compressed_grad_indices(ti)
        nu = int(remap[-1]) + 1
        gy = (np.mod(oracle.allocate_grad_y(B * W).reshape(B, W), 3) - 1).astype(elem[0])
        for extra in (0, 1, 130):
            buf = torch.full((nu + extra, W), 123.0, dtype=elem[1], device="cuda")
            got, inv = ce.embedding_backward(dev(gy), nu + extra, dev(ti), dev(ts), dev(remap),
                                             skip_grad_init=False, grad_embedding=buf)
            want, winv = oracle.embedding_backward(gy.astype(np.float32), W, nu + extra, ti, ts, remap)
            assert np.array_equal(host(got).astype(np.float32), want), (elem[0], extra)
            assert np.array_equal(host(inv)[:nu], winv[:nu])
        # extension: num_unique unknown on the host (num_grad_embedding_rows=None -> -1 in the C ABI): worst-case
        # buffers, the first num_unique rows are the gradient, everything past the last id is left untouched
        cap = min(ti.shape[0], ncat)
        buf = torch.full((cap, W), 123.0, dtype=elem[1], device="cuda")
        ibuf = torch.full((cap,), -5, dtype=torch.int32, device="cuda")
        got, inv = ce.embedding_backward(dev(gy), None, dev(ti), dev(ts), dev(remap), grad_embedding=buf,
                                         inverse_mapping=ibuf)
        want, winv = oracle.embedding_backward(gy.astype(np.float32), W, nu, ti, ts, remap)
        assert np.array_equal(host(got[:nu]).astype(np.float32), want) and np.array_equal(host(inv[:nu]), winv)
        assert bool((got[nu:] == 123.0).all()) and bool((inv[nu:] == -5).all())
        with pytest.raises(ValueError):
            ce.embedding_backward(dev(gy), None, dev(ti), dev(ts), dev(remap))       # buffers are required
        with pytest.raises(ValueError):
            ce.embedding_backward(dev(gy), None, dev(ti), dev(ts), grad_embedding=buf, inverse_mapping=ibuf)  # dense


@pytest.mark.parametrize("blocks", [1, 2])
def test_backward_device_side_row_count_respects_the_buffer_capacity(ce, oracle, blocks):
    """num_grad_embedding_rows=None (num_unique stays on the device): buffers ONE row short of num_unique must give the
    sticky overflow flag and leave every byte alone -- the row behind the buffer included -- instead of an overrun;
    exactly num_unique rows must work and leave the flag down."""
    W, ncat = 64, 50000
    B, H = (3000, 16) if blocks == 1 else (40000, 8)          # (sample blocks need > 131,072 lookups)
    a = oracle.allocate_forward(ncat, W, B, H, alpha=1.05)
    gy = (np.mod(oracle.allocate_grad_y(B * W).reshape(B, W), 3) - 1).astype(np.float32)
    idx = dev(a["indices"])
    sid = ce.extract_row_ids_from_fixed(B, H, torch.int32, "cuda")
    if blocks == 1:
        ti, ts, _ = ce.transpose(sid, idx)
        remap, pair_rows = ce.compute_compressed_grad_indices(ti), None
    else:
        ti, ts, _ = ce.transpose(sid, idx, sample_blocks=blocks)
        remap, pair_rows, _ = ce.compute_compressed_grad_indices_blocked(ti, blocks)
    nu = int(torch.unique(idx).numel())
    ce.capacity_overflowed(reset=True)
    # the allocation is one row LONGER than what the call is told about: the guard row shows an overrun
    for cap, fits in ((nu, True), (nu - 1, False)):
        whole = torch.full((cap + 1, W), 77.0, device="cuda")
        iwhole = torch.full((cap + 1,), -9, dtype=torch.int32, device="cuda")
        got, inv = ce.embedding_backward(dev(gy), None, ti, ts, remap, grad_embedding=whole[:cap],
                                         inverse_mapping=iwhole[:cap], sample_blocks=blocks, block_row_ids=pair_rows)
        torch.cuda.synchronize()
        assert bool((whole[cap] == 77.0).all()) and int(iwhole[cap]) == -9, "wrote behind the buffer"
        assert ce.capacity_overflowed() == (not fits)
        if fits:
            o_sid = oracle.extract_row_ids_from_fixed(B, H)
            oti, ots, _ = oracle.transpose(o_sid, a["indices"])
            want, winv = oracle.embedding_backward(gy, W, nu, oti, ots, oracle.compute_compressed_grad_indices(oti))
            assert np.array_equal(host(inv), winv) and np.array_equal(host(got), want)
        elif blocks == 1:
            assert bool((whole == 77.0).all()) and bool((iwhole == -9).all()), "an over-full call must write nothing"
    assert ce.capacity_overflowed(reset=True) is True and ce.capacity_overflowed() is False      # sticky until reset


def test_backward_zeroes_a_large_over_allocated_tail(ce, oracle):
    """A host-known row count far above num_unique (an over-allocated min(nnz, rows) buffer): every row past the last
    id must read zero, also when the tail is odd-sized and starts off a 16-byte boundary."""
    W, B, H, ncat = 34, 2000, 8, 300          # 68-byte fp16 rows: the tail starts on every 4-byte phase of a 16-byte line
    a = oracle.allocate_forward(ncat, W, B, H, alpha=1.15, elem=np.float16)
    sid = oracle.extract_row_ids_from_fixed(B, H)
    ti, ts, _ = oracle.transpose(sid, a["indices"])
    remap = oracle.compute_compressed_grad_indices(ti)
    nu = int(remap[-1]) + 1
    gy = (np.mod(oracle.allocate_grad_y(B * W).reshape(B, W), 3) - 1).astype(np.float16)
    rows = nu + 70001
    buf = torch.full((rows, W), 5.0, dtype=torch.float16, device="cuda")
    got, inv = ce.embedding_backward(dev(gy), rows, dev(ti), dev(ts), dev(remap), skip_grad_init=False, grad_embedding=buf)
    want, winv = oracle.embedding_backward(gy.astype(np.float32), W, nu, ti, ts, remap)
    assert np.array_equal(host(got[:nu]).astype(np.float32), want) and np.array_equal(host(inv)[:nu], winv)
    assert bool((got[nu:] == 0).all())


@pytest.mark.parametrize("elem", ELEMS, ids=["f32", "f16"])
def test_very_wide_rows_forward_weight_grad_backward(ce, oracle, elem):
    """Rows at the API limit of 1024 lanes (embedding_lookup.cuh:179): 16 KiB rows, one sample or one
    nz-segment per 1024-thread workgroup, in every kernel family."""
    es = np.dtype(elem[0]).itemsize
    for W in (16 // es * 1024, 16 // es * 512 + 16 // es * 3, 2040, 2042):   # 16-, 16-, 16- and 4-byte lanes
        if (W * es) % 4:
            continue
        ncat, B, H = 300, 37, 9
        a = oracle.allocate_forward(ncat, W, B, H, alpha=1.15, elem=elem[0])
        want = oracle.embedding_forward(a["table"], a["indices"], None, a["weights"], num_hots=H)
        got = ce.embedding_forward(dev(a["table"]), dev(a["indices"]), None, dev(a["weights"]), num_hots=H)
        view = np.uint16 if es == 2 else np.uint32
        assert np.array_equal(host(got).view(view), want.view(view)), W
        sid = oracle.extract_row_ids_from_fixed(B, H)
        ti, ts, tw = oracle.transpose(sid, a["indices"], a["weights"])
        remap = oracle.compute_compressed_grad_indices(ti)
        nu = int(remap[-1]) + 1
        gy = (np.mod(oracle.allocate_grad_y(B * W), 3) - 1).reshape(B, W).astype(elem[0])
        want_g, want_inv = oracle.embedding_backward(gy.astype(np.float32), W, nu, ti, ts, remap, tw.astype(np.float32))
        got_g, got_inv = ce.embedding_backward(dev(gy), nu, dev(ti), dev(ts), dev(remap), dev(tw))
        assert np.array_equal(host(got_g).astype(np.float32), want_g), W
        assert np.array_equal(host(got_inv), want_inv)
        gw = ce.embedding_weight_grad(dev(a["table"]), dev(a["indices"]), dev(gy), num_hots=H)
        ref = np.einsum("nw,nw->n", a["table"][a["indices"]].astype(np.float64),
                        gy[np.repeat(np.arange(B), H)].astype(np.float64))
        tol = 2e-2 if es == 2 else 1e-4
        assert np.allclose(host(gw).astype(np.float64), ref, rtol=tol, atol=tol * np.abs(ref).max()), W


@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
def test_transpose_is_a_generic_coo_transpose(ce, oracle, idx):
    """The reference's Transpose is cub::DeviceRadixSort::SortPairs over all bits of a SIGNED key
    type with arbitrary IndexT payloads (index_transforms.cuh:108-136, :224-250): negative keys sort
    first, payloads are carried verbatim.  Sizes cover the single-tile sort, the folded scan and the
    three-launch pass; keys cover constant / varying sign digits; payloads cover the narrow (all
    values in [0, 2^32)) and the wide route (>= 2^32, negative), decided on the device."""
    rng = np.random.default_rng(31)
    info = np.iinfo(idx[0])
    key_sets = {
        "mixed_sign_small": lambda n: rng.integers(-1000, 1000, n),
        "mixed_sign_full": lambda n: rng.integers(info.min, info.max, n, endpoint=True),
        "all_negative": lambda n: rng.integers(info.min, -1, n, endpoint=True),
        "minus_one_and_zero": lambda n: rng.integers(-1, 0, n, endpoint=True),
        "extremes": lambda n: rng.choice(np.array([info.min, -1, 0, 1, info.max]), n),
    }
    row_sets = {"sample_ids": lambda n: rng.integers(0, 1 << 20, n)}
    if idx[0] == np.int64:
        row_sets["beyond_2^32"] = lambda n: rng.integers(0, 1 << 40, n)
        row_sets["exactly_2^32"] = lambda n: np.where(np.arange(n) == n // 2, 1 << 32, rng.integers(0, 100, n))
        row_sets["negative"] = lambda n: rng.integers(-5, 5, n)
        row_sets["int64_extremes"] = lambda n: rng.choice(np.array([info.min, -1, 0, (1 << 32) - 1, info.max]), n)
    else:
        row_sets["negative"] = lambda n: rng.integers(info.min, info.max, n, endpoint=True)
    for nnz in (5, 4096, 4097, 40000, 200003):
        for kname, kf in key_sets.items():
            for rname, rf in row_sets.items():
                cols = kf(nnz).astype(idx[0])
                rows = rf(nnz).astype(idx[0])
                w = rng.uniform(0, 1, nnz).astype(np.float32)
                oi, os_, ow = oracle.transpose(rows, cols, w, stable=True)
                for weights in (None, w):
                    ti, ts, tw = ce.transpose(dev(rows), dev(cols), dev(weights))
                    assert np.array_equal(host(ti), oi), (nnz, kname, rname)
                    assert np.array_equal(host(ts), os_), (nnz, kname, rname)
                    if weights is not None:
                        assert np.array_equal(host(tw), ow), (nnz, kname, rname)
    # the hints: bounded non-negative keys, rows known to be sample ids
    nnz = 150001
    cols = rng.integers(0, 10_000_000, nnz).astype(idx[0])
    rows = rng.integers(0, nnz, nnz).astype(idx[0])
    oi, os_, _ = oracle.transpose(rows, cols, None, stable=True)
    for kw in (dict(num_categories=10_000_000), dict(num_rows=nnz), dict(num_categories=10_000_000, num_rows=nnz)):
        ti, ts, _ = ce.transpose(dev(rows), dev(cols), **kw)
        assert np.array_equal(host(ti), oi) and np.array_equal(host(ts), os_), kw


def test_compressed_indices_many_tiles(ce, oracle):
    """More than kSelfSumTiles = 4096 tiles (16.7M lookups): the run-head scan takes the route with
    the single-workgroup prefix pass over the tile counts."""
    rng = np.random.default_rng(5)
    nnz = 4097 * 4096 + 77
    keys = np.sort(rng.integers(0, 3_000_000, nnz).astype(np.int32))
    got = host(ce.compute_compressed_grad_indices(dev(keys)))
    want = oracle.compute_compressed_grad_indices(keys)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
def test_transpose_fixed_hotness_without_sample_id_array(ce, oracle, idx):
    """cuembed::TransposeFixedHotness (extension) == ExtractRowIdsFromFixed + Transpose: the first
    radix pass derives the sample id of lookup i as i / num_hots (a multiply-shift; checked here for
    hotness values that are not powers of two, up to the largest positions) instead of loading it."""
    rng = np.random.default_rng(41)
    for B, H in [(1, 1), (3, 4), (100, 1), (157, 26), (65, 63), (1023, 7), (4096, 1), (9000, 5), (70001, 3),
                 (65536, 64), (1, 70000), (3, 99991)]:
        nnz = B * H
        cols = rng.integers(0, 1 << 22, nnz).astype(idx[0])
        w = rng.uniform(0, 1, nnz).astype(np.float32)
        sid = ce.extract_row_ids_from_fixed(B, H, idx[1])
        for weights in (None, w):
            for ncat in (None, 1 << 22):
                want = ce.transpose(sid, dev(cols), dev(weights), num_categories=ncat)
                got = ce.transpose_fixed_hotness(dev(cols), B, H, dev(weights), num_categories=ncat)
                assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), (B, H, ncat)
                if weights is not None:
                    assert torch.equal(got[2], want[2])
        if nnz <= 300000:
            oi, os_, _ = oracle.transpose(oracle.extract_row_ids_from_fixed(B, H, idx[0]), cols, None, stable=True)
            got = ce.transpose_fixed_hotness(dev(cols), B, H)
            assert np.array_equal(host(got[0]), oi) and np.array_equal(host(got[1]), os_), (B, H)


_KNOB_SCRIPT = r"""
import sys
import numpy as np
import torch
sys.path.insert(0, {root!r})
import cuembed_amd as ce

rng = np.random.default_rng(3)
for dt, tt in ((np.int32, torch.int32), (np.int64, torch.int64)):
    info = np.iinfo(dt)
    for nnz in (900, 1001, 5000, 19999, 20001, 70000, 300000):
        for name, cols in (("full", rng.integers(info.min, info.max, nnz, endpoint=True)),
                           ("rows", rng.integers(0, 5_000_000, nnz))):
            cols = cols.astype(dt)
            sid = np.arange(nnz, dtype=dt)
            order = np.argsort(cols, kind="stable")
            t = ce.transpose(torch.from_numpy(sid).cuda(), torch.from_numpy(cols).cuda(), None, remapped=True)
            assert np.array_equal(t[0].cpu().numpy(), cols[order]), (dt, nnz, name)
            assert np.array_equal(t[1].cpu().numpy(), sid[order]), (dt, nnz, name)
            heads = np.concatenate(([0], (cols[order][1:] != cols[order][:-1]).astype(np.int64)))
            assert np.array_equal(t[3].cpu().numpy(), np.cumsum(heads).astype(dt)), (dt, nnz, name)
assert ce._lib.lib().cuembed_peek_last_error() == 0
print("knobs ok")
"""


def test_sort_regime_knobs_give_the_same_order():
    """CUEMBED_BLOCK_SORT_MAX, CUEMBED_CHAINED_SORT_MAX and CUEMBED_SORT_HIGH_WORD_LAUNCHES move the limits between the
    sort's three implementations and bring back the launched passes over the high word of 64-bit keys (the code the
    default build only runs through RadixHighPassesKernel): in a process of its own with all three set, sizes on both
    sides of the moved limits, keys using every bit and keys that are table rows -- the stable order and the run-head
    ids of a plain numpy argsort."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CUEMBED_BLOCK_SORT_MAX="1000", CUEMBED_CHAINED_SORT_MAX="20000",
               CUEMBED_SORT_HIGH_WORD_LAUNCHES="1")
    r = subprocess.run([sys.executable, "-c", _KNOB_SCRIPT.format(root=root)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "knobs ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
def test_padded_gradient_names_different_rows_of_the_batch(ce, oracle, idx):
    """pad_to_capacity: the zero rows past the device-side count name rows OF THE BATCH, different ones in turn (entry i
    names the row of entry (i - unique) mod unique) -- one row for the whole tail would make a consumer that adds by row
    (torch's coalesce(), index_add_) serialise the whole tail on it.  Coalescing gives the reference's gradient."""
    rng = np.random.default_rng(12)
    B, H, W, ncat = 600, 20, 48, 5000
    indices = rng.integers(0, ncat, B * H).astype(idx[0])
    indices[: B * H // 2] = rng.integers(0, 40, B * H // 2)          # few distinct rows: a long tail to pad
    gy = rng.integers(-3, 4, (B, W)).astype(np.float32)
    ti, ts, _, remap = ce.transpose_fixed_hotness(dev(indices), B, H, None, num_categories=ncat, remapped=True)
    unique = int(remap[-1].item()) + 1
    capacity = min(B * H, ncat)
    grad = torch.full((capacity, W), 9.0, device="cuda")
    inv = torch.full((capacity,), -5, device="cuda", dtype=idx[1])
    ce.embedding_backward(dev(gy), None, ti, ts, remap, grad_embedding=grad, inverse_mapping=inv, pad_to_capacity=True)
    want_grad, want_inv = ce.embedding_backward(dev(gy), unique, ti, ts, remap)
    assert torch.equal(grad[:unique], want_grad) and torch.equal(inv[:unique], want_inv)
    assert bool((grad[unique:] == 0).all())
    names = want_inv[torch.arange(capacity - unique, device="cuda") % unique]
    assert torch.equal(inv[unique:], names)
    assert int(torch.bincount(inv.long()).max()) <= -(-capacity // unique)       # no row is named more often than that
    sparse = torch.sparse_coo_tensor(inv.long().unsqueeze(0), grad, (ncat, W)).coalesce()
    assert torch.equal(sparse._indices()[0], want_inv.long()) and torch.equal(sparse._values(), want_grad)
