"""GPU parity: EmbeddingForward (HIP, through the C ABI) vs the CPU oracle.

Bit-exact everywhere: the kernels accumulate in lookup order with unfused
operations, exactly like the reference's sequential host loop.  Shapes follow
the reference's own suites (tests/test_embedding_forward.cu KATs,
tests/test_embedding_against_cpu.cu:236-293 sweep)."""
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


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint16 if a.dtype == np.float16 else np.uint32)


@pytest.fixture(scope="module")
def ce():
    import cuembed_amd
    assert torch.cuda.is_available()
    return cuembed_amd


@pytest.fixture(scope="module")
def kats(golden_dir):
    with open(os.path.join(golden_dir, "reference_kats.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("elem", ELEMS, ids=["f32", "f16"])
@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
@pytest.mark.parametrize("csr", [False, True], ids=["fixed", "csr"])
def test_forward_kat(ce, kats, elem, idx, csr):
    k = kats["forward"]
    table = dev(np.array(k["embedding"], dtype=elem[0]).reshape(5, 4))
    indices = dev(np.array(k["indices"], dtype=idx[0]))
    weights = dev(np.array(k["weights"], dtype=elem[0]))
    for off_t in ([np.int32, np.int64] if csr else [None]):
        offsets = dev(np.array(k["offsets"], dtype=off_t)) if csr else None
        hots = 0 if csr else k["hotness"]
        cases = [("sum", None, "sum"), ("sum", weights, "sum_weighted"), ("mean", None, "mean")]
        if not csr:
            cases.append(("concat", None, "concat"))
        for mode, w, key in cases:
            for f16m in ([False, True] if elem[0] == np.float16 else [False]):
                out = ce.embedding_forward(table, indices, offsets, w, batch_size=2, num_hots=hots,
                                           mode=mode, fp16_math=f16m)
                assert out.cpu().numpy().ravel().tolist() == k[key], (mode, key, f16m)


# (width, batch, hot) x variants of tests/test_embedding_against_cpu.cu:236-293
SWEEP_SHAPES = [(2, 3, 4), (4, 3, 4), (32, 1023, 26), (36, 1023, 26), (512, 3, 63), (512, 1023, 63),
                (514, 1023, 63)]
VARIANTS = [("sum", False, False), ("sum", True, False), ("sum", False, True), ("sum", True, True),
            ("mean", False, False), ("mean", True, False), ("concat", False, False)]


@pytest.mark.parametrize("elem,fp16_math", [(ELEMS[0], False), (ELEMS[1], False), (ELEMS[1], True)],
                         ids=["f32", "f16", "f16-fp16math"])
@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
@pytest.mark.parametrize("shape", SWEEP_SHAPES, ids=lambda s: "w%d_b%d_h%d" % s)
def test_forward_sweep_against_oracle(ce, oracle, elem, fp16_math, idx, shape):
    W, B, H = shape
    if elem[0] == np.float16 and (W * 2) % 4:
        pytest.skip("row bytes not a multiple of 4")
    for mode, csr, weighted in VARIANTS:
        a = oracle.allocate_forward(20 * 1024, W, B, H, alpha=0.0, is_csr=csr, elem=elem[0], index=idx[0])
        offsets = a["offsets"] if csr else None
        w = a["weights"] if weighted else None
        want = oracle.embedding_forward(a["table"], a["indices"], offsets, w, batch_size=B,
                                        num_hots=0 if csr else H, mode=mode, fp16_math=fp16_math)
        for row_loads in (None, "streaming"):       # the row-load policy never changes a bit
            got = ce.embedding_forward(dev(a["table"]), dev(a["indices"]), dev(offsets), dev(w), batch_size=B,
                                       num_hots=0 if csr else H, mode=mode, fp16_math=fp16_math, row_loads=row_loads)
            got = got.cpu().numpy().reshape(want.shape)
            assert (bits(got) == bits(want)).all(), (mode, csr, weighted, row_loads)


def test_forward_row_load_policy_default_and_per_call(ce, oracle):
    """process-wide default (SetForwardRowLoadPolicy) vs the per-call option; both bit-identical to the oracle"""
    a = oracle.allocate_forward(50_000, 128, 4099, 37, alpha=0.0, elem=np.float16)
    want = oracle.embedding_forward(a["table"], a["indices"], num_hots=37)
    t, i = dev(a["table"]), dev(a["indices"])
    assert ce.get_forward_row_load_policy() == "default"
    try:
        ce.set_forward_row_load_policy("streaming")
        assert ce.get_forward_row_load_policy() == "streaming"
        for per_call in (None, "default", "streaming"):
            got = ce.embedding_forward(t, i, num_hots=37, row_loads=per_call).cpu().numpy()
            assert (bits(got) == bits(want)).all(), per_call
    finally:
        ce.set_forward_row_load_policy("default")
    with pytest.raises(ValueError):
        ce.embedding_forward(t, i, num_hots=37, row_loads="nt")


@pytest.mark.parametrize("off_t", [np.int32, np.int64], ids=["o32", "o64"])
def test_forward_weighted_mean_and_ragged_bags(ce, oracle, off_t):
    """Weighted mean (GPU combiner semantics, embedding_lookup_ops.cuh:259-285), empty bags,
    bags longer than the unroll, int64 offsets."""
    rng = np.random.default_rng(11)
    table = rng.uniform(-1, 1, (300, 64)).astype(np.float32)
    lens = np.array([0, 1, 7, 8, 9, 0, 0, 33, 64, 2, 0], dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(off_t)
    nnz = int(offsets[-1])
    indices = rng.integers(0, 300, nnz).astype(np.int64)
    weights = rng.uniform(0.1, 1, nnz).astype(np.float32)
    for mode, w in [("sum", None), ("sum", weights), ("mean", None), ("mean", weights)]:
        want = oracle.embedding_forward(table, indices, offsets, w, num_hots=0, mode=mode)
        got = ce.embedding_forward(dev(table), dev(indices), dev(offsets), dev(w), num_hots=0, mode=mode)
        assert (bits(got.cpu().numpy()) == bits(want)).all(), mode


def test_forward_unaligned_views_and_wide_rows(ce, oracle):
    """Base pointers that are only 4/8-byte aligned must fall back to narrower lanes;
    rows wider than one workgroup row (lanes_per_row > 256) use 1 sample per workgroup."""
    rng = np.random.default_rng(5)
    for W, shift in [(64, 1), (64, 2), (64, 0), (2048, 0), (4096, 0)]:
        table = rng.uniform(-1, 1, (100, W)).astype(np.float32)
        idx = rng.integers(0, 100, (17, 9)).astype(np.int32)
        flat = torch.zeros(100 * W + 4, dtype=torch.float32, device="cuda")
        view = flat[shift:shift + 100 * W].view(100, W)
        view.copy_(torch.from_numpy(table))
        want = oracle.embedding_forward(table, idx.ravel(), num_hots=9)
        got = ce.embedding_forward(view, dev(idx.ravel()), num_hots=9)
        assert (bits(got.cpu().numpy()) == bits(want)).all(), (W, shift)


def test_forward_large_hotness_not_staged(ce, oracle):
    """Fixed hotness too large for the LDS staging budget takes the global-index path."""
    shape = ce.forward_launch_shape(torch.float32, torch.int64, 8, 4, 5000, is_weighted=True)
    assert not shape["staged"]
    rng = np.random.default_rng(9)
    table = rng.integers(-3, 4, (64, 8)).astype(np.float32)
    idx = rng.integers(0, 64, (4, 5000)).astype(np.int64)
    w = rng.choice([0.5, 0.25], (4, 5000)).astype(np.float32)
    want = oracle.embedding_forward(table, idx.ravel(), None, w.ravel(), num_hots=5000)
    got = ce.embedding_forward(dev(table), dev(idx.ravel()), None, dev(w.ravel()), num_hots=5000)
    assert (bits(got.cpu().numpy()) == bits(want)).all()


def test_forward_argument_contract(ce):
    t = torch.zeros(4, 4, device="cuda")
    i = torch.zeros(4, dtype=torch.int64, device="cuda")
    o = torch.tensor([0, 2, 4], dtype=torch.int64, device="cuda")
    w = torch.ones(4, device="cuda")
    with pytest.raises(ValueError):
        ce.embedding_forward(t, i, None, w, num_hots=2, mode="concat")
    with pytest.raises(ValueError):
        ce.embedding_forward(t, i, o, None, num_hots=0, mode="concat")
    with pytest.raises(ValueError):
        ce.embedding_forward(t, i, o, None, num_hots=2, mode="sum")
    with pytest.raises(ValueError):
        ce.embedding_forward(t, i, None, None, num_hots=0, mode="sum")
    with pytest.raises(RuntimeError):
        ce.embedding_forward(t.cpu(), i, None, None, num_hots=2)
    with pytest.raises(TypeError):
        ce.embedding_forward(t.double(), i, None, None, num_hots=2)


def test_forward_full_size_c2_properties(ce, oracle):
    """BASELINE config 2 (fp16 sum, 10M x 256, batch 65536, hotness 64, alpha 1.15) at full
    size: (a) bit-exact against the oracle on the first and last 256 samples; (b) a checksum
    of checksums over ALL samples on an integer-valued table (exact in fp32/fp16):
    sum_s out[s,:] == histogram(indices) @ table."""
    rows, W, B, H = 10_000_000, 256, 65536, 64
    idx_np = oracle.generate_indices(rows, B, H, alpha=1.15)
    idx = torch.from_numpy(idx_np).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    table = (torch.rand((rows, W), device="cuda", generator=g) * 2 - 1).half()
    out = ce.embedding_forward(table, idx, num_hots=H)
    torch.cuda.synchronize()
    assert ce._lib.lib().cuembed_peek_last_error() == 0
    for lo in (0, B - 256):
        sub = idx_np[lo * H:(lo + 256) * H]
        uniq, inv = np.unique(sub, return_inverse=True)
        small = table[torch.from_numpy(uniq).cuda().long()].cpu().numpy()
        want = oracle.embedding_forward(small, inv.astype(np.int32), num_hots=H)
        assert (bits(out[lo:lo + 256].cpu().numpy()) == bits(want)).all()
    # (b) integer table in [-2, 2]: every partial sum is an integer < 2^11 -> exact everywhere
    table_i = torch.randint(-2, 3, (rows, W), device="cuda", generator=g).half()
    out_i = ce.embedding_forward(table_i, idx, num_hots=H)
    hist = torch.bincount(idx.long(), minlength=rows).double()
    want_sum = (hist.unsqueeze(0) @ table_i.double()).squeeze(0)
    assert torch.equal(out_i.double().sum(0), want_sum)
    # (c) concat then sum in lookup order == sum, on a slice (concat output is 64x larger)
    sub_idx = idx[:1024 * H]
    cat = ce.embedding_forward(table_i, sub_idx, num_hots=H, mode="concat")
    assert torch.equal(cat.float().sum(1).half(), out_i[:1024])


@pytest.mark.parametrize("elem,fp16_math,rtol", [(ELEMS[0], False, 1e-3), (ELEMS[1], False, 1e-2),
                                                  (ELEMS[1], True, 1e-2)], ids=["f32", "f16", "f16-fp16math"])
@pytest.mark.parametrize("W", [8, 32, 128, 256, 512, 1024])
def test_forward_split_hotness_small_batch(ce, oracle, elem, fp16_math, rtol, W):
    """ReductionOrder kAllowSplit: small batches split a sample's hotness loop over wavefronts
    (LDS partial rows + cross-lane folds).  Tolerance parity (north_star: 1e-3 fp32 / 1e-2 fp16
    relative) against the oracle, EXACT on integer-valued tables, and the default order must be
    unaffected afterwards."""
    rng = np.random.default_rng(W)
    B, H = 37, 61
    table = rng.uniform(-1, 1, (2000, W)).astype(elem[0])
    table_i = rng.integers(-3, 4, (2000, W)).astype(elem[0])
    idx = rng.integers(0, 2000, (B, H)).astype(np.int32)
    w = rng.choice([0.5, 0.25], (B, H)).astype(elem[0])
    lens = rng.integers(0, H + 1, B)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    idx_csr = rng.integers(0, 2000, int(off[-1])).astype(np.int32)
    assert ce.get_forward_reduction_order() == "sequential"
    ce.set_forward_reduction_order("split")
    try:
        for mode, weights in [("sum", None), ("sum", w), ("mean", None), ("mean", w)]:
            wf = None if weights is None else weights.ravel()
            want = oracle.embedding_forward(table, idx.ravel(), None, wf, num_hots=H, mode=mode,
                                            fp16_math=fp16_math).astype(np.float64)
            got = ce.embedding_forward(dev(table), dev(idx.ravel()), None, dev(wf), num_hots=H, mode=mode,
                                       fp16_math=fp16_math).cpu().numpy().astype(np.float64)
            scale = np.abs(want).max()
            assert np.abs(got - want).max() <= rtol * scale, (mode, weights is not None)
            want_i = oracle.embedding_forward(table_i, idx.ravel(), None, wf, num_hots=H, mode="sum")
            got_i = ce.embedding_forward(dev(table_i), dev(idx.ravel()), None, dev(wf), num_hots=H, mode="sum")
            assert (bits(got_i.cpu().numpy()) == bits(want_i)).all()
        want = oracle.embedding_forward(table_i, idx_csr, off, None, num_hots=0)      # ragged + empty bags
        got = ce.embedding_forward(dev(table_i), dev(idx_csr), dev(off), None, num_hots=0)
        assert (bits(got.cpu().numpy()) == bits(want)).all()
    finally:
        ce.set_forward_reduction_order("sequential")
    want = oracle.embedding_forward(table, idx.ravel(), num_hots=H, fp16_math=fp16_math)
    got = ce.embedding_forward(dev(table), dev(idx.ravel()), num_hots=H, fp16_math=fp16_math)
    assert (bits(got.cpu().numpy()) == bits(want)).all()


@pytest.mark.parametrize("elem", ELEMS, ids=["f32", "f16"])
@pytest.mark.parametrize("W,B", [(128, 40003), (32, 140000)], ids=["w128", "w32"])
def test_forward_csr_skewed_bags_large_batch(ce, oracle, elem, W, B):
    """Large CSR batches with deliberately skewed bag lengths (many empty, a few of several
    hundred lookups, lengths that are not multiples of the load batch): bit-exact."""
    rng = np.random.default_rng(B)
    lens = rng.integers(0, 33, B)
    lens[rng.integers(0, B, B // 7)] = 0
    lens[rng.integers(0, B, 50)] = rng.integers(100, 400, 50)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nnz = int(off[-1])
    table = rng.uniform(-1, 1, (3000, W)).astype(elem[0])
    idx = rng.integers(0, 3000, nnz).astype(np.int32)
    w = rng.uniform(0, 1, nnz).astype(elem[0])
    for mode, weights in [("sum", None), ("sum", w), ("mean", None), ("mean", w)]:
        want = oracle.embedding_forward(table, idx, off, weights, num_hots=0, mode=mode, threads=8)
        got = ce.embedding_forward(dev(table), dev(idx), dev(off), dev(weights), num_hots=0, mode=mode)
        assert (bits(got.cpu().numpy()) == bits(want)).all(), mode


@pytest.mark.parametrize("tdtype", [torch.float32, torch.float16, torch.bfloat16], ids=["f32", "f16", "bf16"])
@pytest.mark.parametrize("idx", IDXS, ids=["i32", "i64"])
def test_weight_gradient_all_types_and_layouts(ce, tdtype, idx):
    """EmbeddingWeightGrad (extension): grad_w[s, j] = <table[idx[s, j]], grad_y[s]>.  Small-integer
    data keep every partial sum exactly representable (|dot| <= 6 * W... capped below 2^8 for bf16),
    so the packed-dot / transposing-butterfly kernel must agree EXACTLY with a float64 dot product,
    whatever order it adds in; widths cover the 2-lane fallback, a masked 9-of-16-lane group, the
    32- and 64-lane groups and rows wider than a wavefront; fixed hotness (full and partial batches of
    8 lookups) and ragged CSR bags."""
    rng = np.random.default_rng(5)
    ncat = 700
    for W in (8, 36, 64, 256, 512, 1024):
        lim = 1 if tdtype == torch.bfloat16 else 2                # bf16 has 8 significant bits
        table = rng.integers(-lim, lim + 1, (ncat, W)).astype(np.float32)
        cases = [("fixed", 37, H) for H in (1, 7, 8, 19, 64)] + [("csr", 53, 21)]
        for layout, B, H in cases:
            gy = rng.integers(-1, 2, (B, W)).astype(np.float32)
            if layout == "fixed":
                offsets = None
                indices = rng.integers(0, ncat, B * H).astype(idx[0])
                sample_of = np.repeat(np.arange(B), H)
            else:
                lens = rng.integers(0, H + 1, B)
                lens[3] = 0
                offsets = np.concatenate([[0], np.cumsum(lens)]).astype(idx[0])
                indices = rng.integers(0, ncat, int(offsets[-1])).astype(idx[0])
                sample_of = np.repeat(np.arange(B), lens)
            want = np.einsum("nw,nw->n", table[indices].astype(np.float64), gy[sample_of].astype(np.float64))
            assert np.abs(want).max() < 256
            got = ce.embedding_weight_grad(dev(table).to(tdtype), dev(indices), dev(gy).to(tdtype),
                                           offsets=dev(offsets), num_hots=0 if layout == "csr" else H)
            assert got.dtype == tdtype and got.numel() == indices.size
            assert np.array_equal(got.float().cpu().numpy().astype(np.float64), want), (W, layout, H)


@pytest.mark.parametrize("dtype,index_dtype,weighted,mode", [
    (torch.float32, torch.int32, True, "sum"), (torch.float16, torch.int64, False, "mean"),
    (torch.float16, torch.int32, True, "sum"), (torch.float32, torch.int64, False, "sum")])
def test_forward_sample_order_is_a_scheduling_hint_only(oracle, dtype, index_dtype, weighted, mode):
    """ForwardOptions::sample_order (extension): whatever permutation the samples are handed to the wavefronts in,
    every output row holds the ORACLE's bits (not merely the default order's); bag_order_by_length() is the permutation
    by descending bag length."""
    import cuembed_amd as ce
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(11)
    B, rows, W = 3001, 5000, 128 if dtype == torch.float32 else 256
    lens = torch.randint(0, 129, (B,), device=dev, generator=g)
    lens[7] = 0
    lens[B - 1] = 700                     # one bag far longer than the rest
    off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(lens, 0)
    nnz = int(off[-1])
    off = off.to(index_dtype)
    idx = torch.randint(0, rows, (nnz,), device=dev, generator=g).to(index_dtype)
    table = torch.randn((rows, W), device=dev, generator=g).to(dtype)
    w = torch.rand((nnz,), device=dev, generator=g).to(dtype) if weighted else None
    want = ce.embedding_forward(table, idx, off, w, num_hots=0, mode=mode)
    want_oracle = oracle.embedding_forward(table.cpu().numpy(), idx.cpu().numpy(), off.cpu().numpy(),
                                           None if w is None else w.cpu().numpy(), num_hots=0, mode=mode)
    assert np.array_equal(want.cpu().numpy().view(np.uint8), want_oracle.view(np.uint8))
    by_length = ce.bag_order_by_length(off, max_length=700)
    assert by_length.dtype == torch.int32 and torch.equal(torch.sort(by_length).values,
                                                            torch.arange(B, device=dev, dtype=torch.int32))
    sorted_lens = lens[by_length.long()]
    assert bool((sorted_lens[1:] <= sorted_lens[:-1]).all()) and int(by_length[0]) == B - 1
    assert torch.equal(ce.bag_order_by_length(off), by_length)           # without the bound: all key bits, same result
    too_small = ce.bag_order_by_length(off, max_length=64)                # a bound that is too small costs balance only
    assert torch.equal(torch.sort(too_small).values, torch.arange(B, device=dev, dtype=torch.int32))
    clamped = torch.clamp(lens, max=64)[too_small.long()]
    assert bool((clamped[1:] <= clamped[:-1]).all())
    for order in (by_length, torch.randperm(B, device=dev, generator=g).int(),
                  torch.arange(B - 1, -1, -1, device=dev, dtype=torch.int32)):
        got = ce.embedding_forward(table, idx, off, w, num_hots=0, mode=mode, sample_order=order)
        assert np.array_equal(got.cpu().numpy().view(np.uint8), want_oracle.view(np.uint8))     # the oracle's bits
    with pytest.raises(ValueError):       # a hint for ragged bags only
        ce.embedding_forward(table, idx[:B * 2], num_hots=2, batch_size=B, sample_order=by_length)
    with pytest.raises(ValueError):
        ce.embedding_forward(table, idx, off, w, num_hots=0, sample_order=by_length.long())


@pytest.mark.parametrize("elem,fp16_math", [(ELEMS[0], False), (ELEMS[1], False), (ELEMS[1], True)],
                         ids=["f32", "f16", "f16math"])
@pytest.mark.parametrize("W", [1, 2, 6, 8, 32, 64, 100, 128, 256], ids=lambda w: "w%d" % w)
def test_forward_wide_load_small_batches_same_bits(ce, oracle, elem, fp16_math, W):
    """Small batches take GatherReduceWideLoadKernel (one sample per workgroup, a bag's rows requested at once, pooled in
    lookup order out of LDS): forced on ("always", and with several samples sharing a workgroup) wherever the row shape
    allows it and forced off ("never"), all must give the oracle's bits -- sum / mean, weighted, fixed hotness (also longer than one LDS chunk) and ragged CSR bags
    with empty ones, rows from 4 bytes to 1 KiB incl. widths whose lanes do not divide 256 (those stay sequential)."""
    if elem[0] == np.float16 and W % 2:
        pytest.skip("row bytes must be a multiple of 4")
    rng = np.random.default_rng(1000 + W)
    rows, B = 3000, 23
    table = rng.uniform(-1, 1, (rows, W)).astype(elem[0])
    try:
        for H in (1, 7, 33, 70, 300):
            idx = rng.integers(0, rows, (B, H)).astype(np.int32)
            w = rng.uniform(-1, 1, (B, H)).astype(elem[0])
            lens = rng.integers(0, 2 * H + 1, B)
            lens[3] = 0
            off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
            idx_csr = rng.integers(0, rows, int(off[-1])).astype(np.int64)
            w_csr = rng.uniform(-1, 1, int(off[-1])).astype(elem[0])
            for mode, weighted in (("sum", False), ("sum", True), ("mean", False), ("mean", True)):
                want = oracle.embedding_forward(table, idx.ravel(), None, w.ravel() if weighted else None, num_hots=H,
                                                mode=mode, fp16_math=fp16_math)
                want_csr = oracle.embedding_forward(table, idx_csr, off, w_csr if weighted else None, num_hots=0, mode=mode,
                                                    fp16_math=fp16_math)
                for force in ("always", "always2", "always4", "always16", "never", "auto"):   # 1 / 2 / 4 / 16 samples per workgroup
                    ce.set_forward_wide_load(force)
                    got = ce.embedding_forward(dev(table), dev(idx.ravel()), None, dev(w.ravel()) if weighted else None,
                                               num_hots=H, mode=mode, fp16_math=fp16_math)
                    assert (bits(got.cpu().numpy()) == bits(want)).all(), (H, mode, weighted, force)
                    got = ce.embedding_forward(dev(table), dev(idx_csr), dev(off), dev(w_csr) if weighted else None,
                                               num_hots=0, mode=mode, fp16_math=fp16_math)
                    assert (bits(got.cpu().numpy()) == bits(want_csr)).all(), (H, mode, weighted, force, "csr")
    finally:
        ce.set_forward_wide_load("auto")
    assert ce._lib.lib().cuembed_peek_last_error() == 0
