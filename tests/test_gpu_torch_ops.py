"""The torch op surface against nn.EmbeddingBag, reproducing the checks of the reference's
examples/pytorch/cuembed_test.py (fwd exact ==, bwd allclose, no-grad / frozen fast path,
non-contiguous inputs, torch.compile forward and backward)."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pyt():
    assert torch.cuda.is_available()
    from cuembed_amd import cuembed_pyt
    return cuembed_pyt


def make_bag(k, d, dtype=torch.float32):
    torch.manual_seed(0)
    return nn.EmbeddingBag(num_embeddings=k, embedding_dim=d, mode="sum", include_last_offset=True,
                           padding_idx=None, dtype=dtype).to("cuda")


def make_inputs(k, n):
    indices = (k * torch.rand([n], device="cuda")).to(torch.long).clamp_(max=k - 1)
    offsets = torch.arange(0, n + 1, device="cuda", dtype=torch.long)
    weights = torch.rand([n], device="cuda", dtype=torch.float32)
    return indices, offsets, weights


# cuembed_test.py:134-172 (n reduced from 2.88M for the first case to keep the suite quick)
@pytest.mark.parametrize("k,d,n", [(958, 128, 288000), (2048, 64, 104217)])
@pytest.mark.parametrize("weighted", [True, False])
def test_against_embedding_bag(pyt, k, d, n, weighted):
    bag = make_bag(k, d)
    indices, offsets, weights = make_inputs(k, n)
    w = weights if weighted else None
    res = pyt.cuemb_embedding(bag.weight, indices, offsets, w)
    ref = bag(indices, offsets, w)
    assert (res == ref).all()                                   # cuembed_test.py:23
    bag.weight.grad = None
    torch.mean(res).backward()
    grad_res = bag.weight.grad.clone()
    bag.weight.grad = None
    torch.mean(ref).backward()
    grad_ref = bag.weight.grad.clone()
    assert torch.allclose(grad_res, grad_ref)                   # cuembed_test.py:34


def test_multi_hot_bags_against_embedding_bag(pyt):
    """Bags with several lookups (the reference script only uses one index per bag)."""
    k, d, B = 5000, 96, 3000
    bag = make_bag(k, d)
    lens = torch.randint(0, 40, (B,), device="cuda")
    offsets = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), lens.cumsum(0)])
    n = int(offsets[-1])
    indices = torch.randint(0, k, (n,), device="cuda")
    weights = torch.rand(n, device="cuda")
    for w in (None, weights):
        res = pyt.cuemb_embedding(bag.weight, indices, offsets, w)
        ref = bag(indices, offsets, w)
        assert torch.allclose(res, ref, rtol=1e-5, atol=1e-5)
        bag.weight.grad = None
        (res * torch.arange(d, device="cuda")).sum().backward()
        g1 = bag.weight.grad.clone()
        bag.weight.grad = None
        (ref * torch.arange(d, device="cuda")).sum().backward()
        assert torch.allclose(g1, bag.weight.grad, rtol=1e-4, atol=1e-4)


def test_inference_fast_path(pyt):                               # cuembed_test.py:36-50
    bag = make_bag(2048, 64)
    indices, offsets, _ = make_inputs(2048, 10000)
    ref = bag(indices, offsets)
    with torch.no_grad():
        res_nograd = pyt.cuemb_embedding(bag.weight, indices, offsets)
    res_frozen = pyt.cuemb_embedding(bag.weight.detach(), indices, offsets)
    assert torch.allclose(res_nograd, ref) and torch.allclose(res_frozen, ref)
    assert not res_nograd.requires_grad and not res_frozen.requires_grad


def test_noncontiguous_inputs(pyt):                              # cuembed_test.py:52-73
    bag = make_bag(2048, 64)
    indices, offsets, _ = make_inputs(2048, 10000)
    weight = bag.weight
    d = weight.shape[1]
    w_nc = torch.cat([weight, weight], dim=1).detach()[:, :d]
    idx_nc = torch.stack([indices, indices], dim=1).reshape(-1)[::2]
    assert not w_nc.is_contiguous() and not idx_nc.is_contiguous()
    ref = bag(indices, offsets)
    with torch.no_grad():
        res = pyt.cuemb_embedding(w_nc, idx_nc, offsets)
    assert torch.allclose(res, ref)
    grad_mask = torch.ones(ref.shape[0], 2 * d, device=ref.device)[:, ::2]
    assert not grad_mask.is_contiguous()
    weight.grad = None
    (pyt.cuemb_embedding(weight, idx_nc, offsets) * grad_mask).sum().backward()
    grad_res = weight.grad.clone()
    weight.grad = None
    (bag(indices, offsets) * grad_mask).sum().backward()
    assert torch.allclose(grad_res, weight.grad)


def _compile(fn):
    """Default backend when a working inductor/triton is present, aot_eager otherwise (both trace
    through the ops' fake registrations, which is what the reference's compile tests exercise)."""
    def run(*a):
        try:
            return torch.compile(fn)(*a)
        except Exception:  # noqa: BLE001 - inductor toolchain may be absent on the test box
            torch._dynamo.reset()
            return torch.compile(fn, backend="aot_eager")(*a)
    return run


def test_compile_forward(pyt):                                   # cuembed_test.py:75-110
    k, d, batch = 15, 2, 256
    indices = torch.randint(0, k, (batch,), device="cuda", dtype=torch.long)
    offsets = torch.arange(0, batch + 1, device="cuda", dtype=torch.long)
    bag = make_bag(k, d)

    def fwd(weight, indices, offsets):
        return pyt.cuemb_embedding(weight, indices, offsets)

    with torch.no_grad():
        res = _compile(fwd)(bag.weight, indices, offsets)
        ref = bag(indices, offsets)
    assert res.shape == ref.shape and torch.allclose(res, ref)


def test_compile_backward(pyt):                                  # cuembed_test.py:112-131
    k, d, n = 958, 16, 4096
    bag = make_bag(k, d)
    indices = torch.randint(0, k, (n,), device="cuda", dtype=torch.long)
    offsets = torch.arange(0, n + 1, device="cuda", dtype=torch.long)

    def run(weight):
        return pyt.cuemb_embedding(weight, indices, offsets)

    w_ref = bag.weight.detach().clone().requires_grad_(True)
    run(w_ref).sum().backward()
    w_c = bag.weight.detach().clone().requires_grad_(True)
    _compile(run)(w_c).sum().backward()
    assert torch.allclose(w_ref.grad, w_c.grad, atol=1e-4)


def test_op_schemas_and_dtypes(pyt):
    """Schemas equal the reference's (cuembed_embedding.cu:169-183); fp16 / int32 / mean are
    accepted on top of the reference's fp32 / int64 / sum."""
    s = torch.ops.cuembed_pyt.cuembed_embedding_forward.default._schema
    assert [a.name for a in s.arguments] == ["params", "indices", "offsets", "weights", "mode"]
    s = torch.ops.cuembed_pyt.cuembed_embedding_backward.default._schema
    assert [a.name for a in s.arguments] == ["y_grad", "num_categories", "transpose_indices",
                                             "transpose_sample_ids", "transpose_weights"]
    k, d, B = 300, 32, 64
    table = torch.randn(k, d, device="cuda").half()
    idx = torch.randint(0, k, (B * 4,), device="cuda", dtype=torch.int32)
    off = torch.arange(0, B * 4 + 1, 4, device="cuda", dtype=torch.int32)
    out = pyt.cuembed_embedding_forward(table, idx, off, None, "mean")
    ref = table[idx.long()].float().view(B, 4, d).mean(1)
    assert torch.allclose(out.float(), ref, rtol=2e-3, atol=2e-3)
    with pytest.raises(RuntimeError):
        pyt.cuembed_embedding_forward(table.double(), idx, off, None, "sum")
    with pytest.raises(RuntimeError):
        pyt.cuembed_embedding_forward(table, idx, off, None, "max")


@pytest.mark.parametrize("weighted", [False, True])
def test_sparse_gradient_extension(pyt, weighted):
    """sparse_grad=True returns the compressed gradient as a COALESCED sparse COO tensor of exactly the looked-up rows
    (on either backend; "reference" is its older name) that densifies to the dense-path gradient (and to
    nn.EmbeddingBag's); "padded" / "fastest" are the opt-ins that never read the row count back."""
    k, d, B = 20000, 64, 2048
    bag = make_bag(k, d)
    lens = torch.randint(1, 20, (B,), device="cuda")
    offsets = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), lens.cumsum(0)])
    n = int(offsets[-1])
    indices = torch.randint(0, k, (n,), device="cuda")
    w = torch.rand(n, device="cuda") if weighted else None
    up = torch.randn(B, d, device="cuda")
    weight = bag.weight
    weight.grad = None
    (pyt.cuemb_embedding(weight, indices, offsets, w, sparse_grad=True) * up).sum().backward()
    g_sparse = weight.grad
    assert g_sparse.is_sparse
    assert g_sparse._nnz() == torch.unique(indices).numel()        # one entry per looked-up row
    ids = g_sparse._indices()[0]
    assert (ids[1:] > ids[:-1]).all()                                # ascending, no duplicates
    weight.grad = None
    (pyt.cuemb_embedding(weight, indices, offsets, w, sparse_grad="reference") * up).sum().backward()
    assert torch.equal(weight.grad._indices(), g_sparse._indices()) and torch.equal(weight.grad._values(), g_sparse._values())
    # the same contract from the Python autograd.Function (what torch.compile traces)
    t2 = weight.detach().clone().requires_grad_(True)
    (pyt._CuEmbEmbedding.apply(t2, indices, offsets, w, True, None) * up).sum().backward()
    assert t2.grad._nnz() == g_sparse._nnz() and torch.equal(t2.grad._indices(), g_sparse._indices())
    assert torch.equal(t2.grad._values(), g_sparse._values())
    # "padded" (and "fastest" at this size): the row count is never read back and the tensor is padded to min(nnz, rows)
    # entries (zero rows naming a row of the batch) -- the same gradient once coalesced; both backends
    for kind in ("padded", "fastest"):
        weight.grad = None
        (pyt.cuemb_embedding(weight, indices, offsets, w, sparse_grad=kind) * up).sum().backward()
        g_fast = weight.grad
        assert g_fast.is_sparse and not g_fast.is_coalesced() and g_fast._nnz() == min(n, k), kind
        merged = g_fast.coalesce()
        assert torch.equal(merged._indices(), g_sparse._indices()) and torch.equal(merged._values(), g_sparse._values())
    t2.grad = None
    (pyt._CuEmbEmbedding.apply(t2, indices, offsets, w, "padded", None) * up).sum().backward()
    assert t2.grad._nnz() == min(n, k)
    merged = t2.grad.coalesce()
    assert torch.equal(merged._indices(), g_sparse._indices()) and torch.equal(merged._values(), g_sparse._values())
    with pytest.raises(ValueError):
        pyt.cuemb_embedding(weight, indices, offsets, w, sparse_grad="sorted")
    weight.grad = None
    (pyt.cuemb_embedding(weight, indices, offsets, w) * up).sum().backward()
    g_dense = weight.grad.clone()
    assert torch.allclose(g_sparse.to_dense(), g_dense, rtol=1e-5, atol=1e-5)
    weight.grad = None
    (bag(indices, offsets, w) * up).sum().backward()
    assert torch.allclose(g_sparse.to_dense(), weight.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("ragged", [False, True], ids=["fixed", "ragged_weighted"])
def test_uncoalesced_sparse_gradient_in_sample_blocks(pyt, ragged):
    """sparse_grad="uncoalesced": the batch is transposed in the recommended number of sample blocks (2 here:
    B = 40,000 samples of 64 lookups, 512-byte rows), a table row may appear once per block in the sparse gradient,
    and the densified gradient equals the coalesced one bit for bit on integer data."""
    import cuembed_amd as ce
    k, d, B, H = 50000, 256, 40000, 64
    assert ce.recommended_sample_blocks(torch.float16, d, B, B * H) == 2
    torch.manual_seed(20240 + int(ragged))
    weight = torch.randint(-2, 3, (k, d), device="cuda").half().requires_grad_()
    if ragged:   # CSR bags of 32..96 lookups with weights 0.5 / 0.25: the blocks are cut by position, mid-bag if need be
        lens = torch.randint(32, 97, (B,), device="cuda")
        offsets = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), lens.cumsum(0)])
        n = int(offsets[-1])
        w = (torch.randint(0, 2, (n,), device="cuda").half() * 0.25 + 0.25)
    else:
        offsets = torch.arange(0, B * H + 1, H, device="cuda")
        n, w = B * H, None
    # Skew: the hottest row gets ~11 k lookups (several workgroups, both blocks).  Not more in the weighted case: its
    # partial sums are multiples of 0.25 and only exact in fp16 below 512 -- with rand ** 3 (69 k lookups, partial sums
    # of +-85 standard deviation per element) one run in twenty met a partial sum beyond that in SOME order of the
    # float atomics, and "bit for bit" then no longer holds between two correct summation orders.
    indices = (k * torch.rand(n, device="cuda") ** (2 if ragged else 3)).long()
    up = torch.randint(-1, 2, (B, d), device="cuda").half()
    grads = {}
    for kind in ("reference", True, "fastest", "uncoalesced", "blocked"):
        weight.grad = None
        (pyt.cuemb_embedding(weight, indices, offsets, w, sparse_grad=kind) * up).sum().backward()
        grads[kind] = weight.grad
    # "blocked": the coalesced gradient computed from the same sample-blocked order -- identical tensors
    assert torch.equal(grads["blocked"]._indices(), grads["reference"]._indices())
    assert torch.equal(grads["blocked"]._values(), grads["reference"]._values())
    ids = grads["reference"]._indices()[0]
    assert (ids[1:] > ids[:-1]).all()      # coalesced: ascending, no duplicates (autograd's accumulation drops the flag)
    n_unique = grads["reference"]._nnz()
    # exact in fp16: integers below 2048, or -- with weights 0.5 / 0.25 -- multiples of 0.25 below 512
    assert float(grads["reference"].to_dense().abs().max()) < (512 if ragged else 2048)
    assert grads[True].is_sparse and torch.equal(grads[True]._indices(), grads["reference"]._indices())
    assert torch.equal(grads[True]._values(), grads["reference"]._values())       # True = the coalesced tensor, always
    for kind in ("fastest", "uncoalesced"):     # the fastest order for this shape is the sample-blocked one
        assert not grads[kind].is_coalesced()
        assert n_unique < grads[kind]._nnz() <= 2 * n_unique
        assert torch.equal(grads[kind].to_dense(), grads["reference"].to_dense())
        assert torch.equal(grads[kind].coalesce()._values(), grads["reference"]._values())


@pytest.mark.parametrize("mode", ["sum", "mean", "concat"])
@pytest.mark.parametrize("weighted", [False, True])
def test_fixed_hotness_2d_indices_all_modes(pyt, mode, weighted):
    """cuemb_embedding_fixed: [B, H] index tensors with sum / mean / concat, forward and backward
    against plain torch indexing."""
    if mode == "concat" and weighted:
        pytest.skip("concat takes no weights (embedding_lookup.cuh:261)")
    k, d, B, H = 4000, 48, 700, 9
    torch.manual_seed(1)
    table = torch.randn(k, d, device="cuda", requires_grad=True)
    ref_table = table.detach().clone().requires_grad_(True)
    idx = torch.randint(0, k, (B, H), device="cuda")
    w = torch.rand(B, H, device="cuda") + 0.1 if weighted else None
    out = pyt.cuemb_embedding_fixed(table, idx, w, mode)
    rows = ref_table[idx]                                   # [B, H, d]
    if mode == "concat":
        ref = rows
    else:
        ref = (rows * w[..., None]).sum(1) if weighted else rows.sum(1)
        if mode == "mean":
            ref = ref / (w.sum(1, keepdim=True) if weighted else H)
    assert out.shape == ref.shape and torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
    up = torch.randn_like(ref)
    (out * up).sum().backward()
    (ref * up).sum().backward()
    assert torch.allclose(table.grad, ref_table.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("d", [8, 36, 128, 512])
def test_per_sample_weights_gradient_extension(pyt, d):
    """cuemb_embedding is differentiable w.r.t. the per-lookup weights (the reference returns None,
    cuembed_pyt.py:34-35): against nn.EmbeddingBag's per_sample_weights gradient."""
    k, B = 3000, 600
    bag = make_bag(k, d)
    lens = torch.randint(0, 30, (B,), device="cuda")
    offsets = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), lens.cumsum(0)])
    n = int(offsets[-1])
    indices = torch.randint(0, k, (n,), device="cuda")
    w1 = torch.rand(n, device="cuda", requires_grad=True)
    w2 = w1.detach().clone().requires_grad_(True)
    up = torch.randn(B, d, device="cuda")
    bag.weight.grad = None
    (pyt.cuemb_embedding(bag.weight, indices, offsets, w1) * up).sum().backward()
    g_table = bag.weight.grad.clone()
    bag.weight.grad = None
    (bag(indices, offsets, w2) * up).sum().backward()
    assert torch.allclose(w1.grad, w2.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(g_table, bag.weight.grad, rtol=1e-4, atol=1e-4)
    # frozen table, trainable weights only
    w3 = w1.detach().clone().requires_grad_(True)
    (pyt.cuemb_embedding(bag.weight.detach(), indices, offsets, w3) * up).sum().backward()
    assert torch.allclose(w3.grad, w2.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
@pytest.mark.parametrize("kind", [False, True, "fastest", "padded", "uncoalesced"])
def test_native_autograd_node_equals_the_python_function(pyt, kind, idx_dtype):
    """Outside torch.compile cuemb_embedding runs as ONE native autograd node (CuEmbEmbeddingNode: forward, and row ids ->
    transpose + remap -> scatter-add in the backward, the row count read back after everything is enqueued); the
    Python autograd.Function over the same ops must give the same bits -- small (one-workgroup sort), mid-size
    (chained sort), empty bags, weights, a non-contiguous incoming gradient."""
    torch.manual_seed(3)
    for k, d, B, hi in ((5000, 32, 300, 9), (200000, 64, 9000, 30)):
        lens = torch.randint(0, hi, (B,), device="cuda")
        offsets = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), lens.cumsum(0)]).to(idx_dtype)
        n = int(offsets[-1])
        indices = torch.randint(0, k, (n,), device="cuda").to(idx_dtype)
        for weighted in (False, True):
            w = torch.rand(n, device="cuda") if weighted else None
            table = torch.randn(k, d, device="cuda")
            up = torch.randint(-3, 4, (d, B), device="cuda").float().t()      # non-contiguous; integers: exact sums
            if weighted:
                w = torch.randint(1, 4, (n,), device="cuda").float() * 0.25
            t1 = table.clone().requires_grad_(True)
            y1 = pyt.cuemb_embedding(t1, indices, offsets, w, sparse_grad=kind, hints=None)
            assert "CuEmbEmbeddingNode" in y1.grad_fn.name()
            y1.backward(up)
            t2 = table.clone().requires_grad_(True)
            y2 = pyt._CuEmbEmbedding.apply(t2, indices, offsets, w, kind, None)
            y2.backward(up)
            assert torch.equal(y1, y2)
            if kind is False:
                assert torch.equal(t1.grad, t2.grad)
            else:
                g1, g2 = t1.grad, t2.grad
                if kind in ("fastest", "padded"):  # small batches: padded instead of read back (see test_sparse_gradient_extension)
                    assert g1.is_sparse and g1._nnz() == min(n, k)
                    g1 = g1.coalesce()
                    if kind == "padded":
                        assert g2._nnz() == min(n, k)
                    g2 = g2.coalesce()
                assert g1.is_sparse and g1._nnz() == torch.unique(indices).numel()      # (True: as delivered)
                ids = g1._indices()[0]
                assert bool((ids[1:] > ids[:-1]).all())                       # one block: ascending, no duplicates
                assert torch.equal(g1._indices(), g2._indices())
                assert torch.equal(g1._values(), g2._values())
    # frozen table: no node, no gradient
    y = pyt.cuemb_embedding(table, indices, offsets, None, sparse_grad=kind)
    assert y.grad_fn is None


def test_native_node_large_and_small_gradients_in_turn(pyt):
    """The native node reads the row count back first above 192 MB of worst-case gradient, enqueues everything and
    narrows afterwards below that (True) or pads and never reads it back below 64 MB ("fastest"): batches with few and
    with many distinct rows in turn, on all three paths."""
    torch.manual_seed(9)
    k, d, H = 300_000, 256, 50                               # 1 KiB rows
    table = torch.randn(k, d, device="cuda")
    for B, kind in ((4096, "fastest"), (2048, True), (1024, "fastest")):   # 200 MB / 100 MB / 50 MB worst case
        offsets = torch.arange(0, B * H + 1, H, device="cuda")
        up = torch.randint(-3, 4, (B, d), device="cuda").float()     # integers: sums are exact in any order of the atomics
        narrow = torch.randint(0, 1000, (B * H,), device="cuda")
        wide = torch.randint(0, k, (B * H,), device="cuda")
        t = table.clone().requires_grad_(True)
        for step, indices in enumerate((narrow, wide, narrow)):
            t.grad = None
            pyt.cuemb_embedding(t, indices, offsets, None, sparse_grad=kind, hints=None).backward(up)
            got = t.grad.coalesce() if B == 1024 else t.grad
            t2 = table.clone().requires_grad_(True)
            pyt._CuEmbEmbedding.apply(t2, indices, offsets, None, "reference", None).backward(up)
            assert got._nnz() == t2.grad._nnz() == torch.unique(indices).numel(), (B, step)
            assert torch.equal(got._indices(), t2.grad._indices()) and torch.equal(got._values(), t2.grad._values()), (B, step)


def test_native_step_replays_from_a_hip_graph(pyt):
    """Forward + backward of the op captured into a torch.cuda.CUDAGraph and replayed on NEW indices in the same buffers:
    the dense gradient and the padded sparse gradient (no read-back, no host decision inside the step) are capturable
    -- the way to run launch-bound batches (tools/torch_graph_step_probe.py: B = 1024, 0.095 -> 0.074 ms) -- and a
    gradient that needs its row count on the host says so instead of failing inside HIP."""
    torch.manual_seed(4)
    k, d, B, H = 50_000, 64, 512, 16
    table = torch.randn(k, d, device="cuda").requires_grad_(True)
    offsets = torch.arange(0, B * H + 1, H, device="cuda")
    up = torch.randint(-3, 4, (B, d), device="cuda").float()
    first = torch.randint(0, k, (B * H,), device="cuda")
    second = torch.randint(0, 2000, (B * H,), device="cuda")
    indices = first.clone()
    side = torch.cuda.Stream()
    for kind in (False, "padded", "fastest"):
        def step():
            out = pyt.cuemb_embedding(table, indices, offsets, None, sparse_grad=kind, hints=None)
            (g,) = torch.autograd.grad(out, table, up)
            return out, g
        with torch.cuda.stream(side):     # (warm-up and capture on one side stream: torch's capture rules)
            indices.copy_(first)
            step()
            torch.cuda.current_stream().synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                out_g, grad_g = step()
            indices.copy_(second)
            graph.replay()
            torch.cuda.current_stream().synchronize()
            got_out, got_grad = out_g.clone(), (grad_g.to_dense() if kind else grad_g.clone())
            want_out, want_grad = step()
            torch.cuda.current_stream().synchronize()
        assert torch.equal(got_out, want_out), kind
        assert torch.equal(got_grad, want_grad.to_dense() if kind else want_grad), kind
        if kind:
            assert grad_g._nnz() == B * H and not grad_g.is_coalesced()      # padded to min(lookups, rows) entries
    with torch.cuda.stream(side):
        graph = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="cannot be captured"):
            with torch.cuda.graph(graph, stream=side):
                out = pyt.cuemb_embedding(table, indices, offsets, None, sparse_grad=True, hints=None)
                torch.autograd.grad(out, table, up)
    torch.cuda.synchronize()


def test_policy_hints_never_change_a_result(pyt):
    """cuembed_amd.policy picks non-temporal row loads for a table whose batches are (nearly) all distinct rows and the
    bag order for ragged CSR batches -- both decided ON THE DEVICE (no torch.unique, no read-back: torch's sync debug
    mode is on while the hinted calls run); both are scheduling hints: same bits, and the decisions are the expected
    ones (uniform indices over a > 1 GiB table: streaming; a skewed batch: not)."""
    from cuembed_amd import policy
    policy.set_enabled(True)
    k, d, B = 2_200_000, 128, 20000          # 1.05 GiB of fp32
    table = torch.randn(k, d, device="cuda")
    lens = torch.randint(20, 100, (B,), device="cuda")
    offsets = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), lens.cumsum(0)])
    n = int(offsets[-1])
    assert n >= policy.ORDER_MIN_LOOKUPS
    uniform = torch.randint(0, k, (n,), device="cuda")
    skewed = (k * torch.rand(n, device="cuda") ** 8).long()
    plain = pyt.cuemb_embedding(table, uniform, offsets, None, hints=None)
    plain_skewed = pyt.cuemb_embedding(table, skewed, offsets, None, hints=None)
    pyt.cuemb_embedding(table, uniform, offsets, None)               # warm-up: allocations, the table's decision words
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        hinted = pyt.cuemb_embedding(table, uniform, offsets, None)      # hints="auto": decision + order, device-side
        t = table.clone().requires_grad_(True)
        y = pyt.cuemb_embedding(t, uniform, offsets, None, sparse_grad="padded")
        order = policy.sample_order(offsets, n)
        words = policy.row_loads_device(table, uniform)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    assert torch.equal(plain, hinted) and torch.equal(y, plain)
    assert policy.row_loads_decision(table) == 1
    assert words is not None and words.dtype == torch.int32 and words.tolist()[1:] == [0, 0, 0]
    assert order is not None and order.dtype == torch.int32
    assert torch.equal(torch.sort(order.long()).values, torch.arange(B, device="cuda"))
    assert bool((lens[order.long()][1:] <= lens[order.long()][:-1]).all())         # descending bag length
    # ... and inside a HIP-graph capture: the bag order is computed by captured launches, the captured forward reads the
    # table's existing decision words; replayed on other indices in the same buffer -- the plain result's bits
    idx_buf = uniform.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        pyt.cuemb_embedding(table, idx_buf, offsets, None)
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            out_g = pyt.cuemb_embedding(table, idx_buf, offsets, None)
        idx_buf.copy_(skewed)
        graph.replay()
        side.synchronize()
    assert torch.equal(out_g, plain_skewed)
    # a fresh offsets tensor gets ITS order (nothing is cached: nothing can go stale)
    lens2 = torch.flip(lens, [0])
    offsets2 = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), lens2.cumsum(0)])
    order2 = policy.sample_order(offsets2, n)
    assert bool((lens2[order2.long()][1:] <= lens2[order2.long()][:-1]).all()) and not torch.equal(order, order2)
    # another table, a skewed batch: default loads; same bits
    table2 = table.clone()
    assert torch.equal(pyt.cuemb_embedding(table2, skewed, offsets, None), plain_skewed)
    assert policy.row_loads_decision(table2) == 0
    assert policy.row_loads_device(table, uniform[: policy.STREAMING_MIN_LOOKUPS - 1]) is None   # small batches: no decision at all
    policy.set_enabled(False)
    assert policy.row_loads_device(table, uniform) is None and policy.sample_order(offsets, n) is None
    policy.set_enabled(True)


def test_ops_are_the_native_extension(pyt):
    """The ops come from cuembed_amd/lib/libcuembed_pyt.so (TORCH_LIBRARY / TORCH_LIBRARY_IMPL in
    cuembed_amd/csrc/torch_binding.cpp), like the reference's (cuembed_embedding.cu:169-190), not from
    Python registrations; the shared object is mapped into this process."""
    assert pyt.BACKEND == "native"
    with open("/proc/self/maps") as f:
        maps = f.read()
    assert "libcuembed_pyt.so" in maps and "libcuembed_amd.so" in maps
    for name in ("cuembed_embedding_forward", "cuembed_extract_row_ids_from_csr", "cuembed_transpose",
                 "cuembed_embedding_backward", "cuembed_transpose_fixed_hotness", "cuembed_transpose_sample_ids",
                 "cuembed_transpose_sample_blocks"):
        op = getattr(torch.ops.cuembed_pyt, name).default
        assert torch._C._dispatch_has_kernel_for_dispatch_key(op.name(), "CUDA")


def test_ops_validate_like_the_reference_binding(pyt):
    """AT_ASSERT in the reference (cuembed_embedding.cu:15-32) -> TORCH_CHECK here: Python exceptions,
    never an abort, for CPU tensors, wrong dtypes and unsupported modes."""
    table = torch.randn(50, 8, device="cuda")
    idx = torch.randint(0, 50, (12,), device="cuda")
    off = torch.arange(0, 13, 4, device="cuda")
    with pytest.raises(RuntimeError):
        torch.ops.cuembed_pyt.cuembed_embedding_forward(table, idx, off, None, "max")
    with pytest.raises(RuntimeError):
        torch.ops.cuembed_pyt.cuembed_embedding_forward(table.double(), idx, off, None, "sum")
    with pytest.raises(RuntimeError):
        torch.ops.cuembed_pyt.cuembed_embedding_forward(table, idx.float(), off, None, "sum")
    with pytest.raises(RuntimeError):
        torch.ops.cuembed_pyt.cuembed_embedding_forward(table, idx, off, torch.ones(12, device="cuda", dtype=torch.half), "sum")
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.cuembed_pyt.cuembed_embedding_forward(table.cpu(), idx.cpu(), off.cpu(), None, "sum")
    with pytest.raises(RuntimeError):
        torch.ops.cuembed_pyt.cuembed_transpose(idx, idx[:5], None)


@pytest.mark.parametrize("idx_dtype", [torch.int64, torch.int32])
def test_fused_index_ops(pyt, idx_dtype):
    """cuembed_transpose_fixed_hotness == extract + transpose (+ remap); cuembed_transpose_sample_ids ==
    cuembed_transpose_bounded."""
    B, H, k = 700, 9, 5000
    idx = torch.randint(0, k, (B, H), device="cuda", dtype=idx_dtype)
    w = torch.rand(B, H, device="cuda")
    sid = (torch.arange(B * H, device="cuda") // H).to(idx_dtype)
    for weights in (None, w):
        flat_w = None if weights is None else weights.reshape(-1)
        want = torch.ops.cuembed_pyt.cuembed_transpose_bounded(sid, idx.reshape(-1), flat_w, k)
        got = torch.ops.cuembed_pyt.cuembed_transpose_fixed_hotness(idx, weights, k, True)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
        assert torch.equal(got[3], torch.ops.cuembed_pyt.cuembed_compute_compressed_grad_indices(want[0]))
        got2 = torch.ops.cuembed_pyt.cuembed_transpose_sample_ids(sid, idx.reshape(-1), flat_w, k)
        assert all(torch.equal(a, b) for a, b in zip(got2, want))
        none = torch.ops.cuembed_pyt.cuembed_transpose_fixed_hotness(idx, weights, k, False)
        assert none[3].numel() == 0 and torch.equal(none[0], want[0])
    # the closed-offsets row-id op == the reference-shaped one on the slice
    off = torch.arange(0, B * H + 1, H, device="cuda", dtype=idx_dtype)
    assert torch.equal(torch.ops.cuembed_pyt.cuembed_extract_row_ids_from_offsets(off, B * H), sid)
    assert torch.equal(torch.ops.cuembed_pyt.cuembed_extract_row_ids_from_csr(off[:-1], B * H), sid)
    # stable order == torch's stable sort
    order = torch.sort(idx.reshape(-1), stable=True)
    assert torch.equal(want[0], order.values) and torch.equal(want[1], sid[order.indices])


def test_opcheck_schemas_and_fake_kernels(pyt):
    """torch.library.opcheck: the registered schema, the fake (meta) kernel and the real kernel of every op
    agree on output metadata -- what torch.compile relies on (reference: cuembed_pyt.py:55-77)."""
    B, H, k, d = 64, 4, 300, 16
    table = torch.randn(k, d, device="cuda")
    idx = torch.randint(0, k, (B * H,), device="cuda")
    off = torch.arange(0, B * H + 1, H, device="cuda")
    w = torch.rand(B * H, device="cuda")
    gy = torch.randn(B, d, device="cuda")
    sid = torch.arange(B * H, device="cuda") // H
    ops = torch.ops.cuembed_pyt
    t_idx, t_sid, t_w = ops.cuembed_transpose_bounded(sid, idx, w, k)
    remap = ops.cuembed_compute_compressed_grad_indices(t_idx)
    nu = int(remap[-1].item()) + 1
    checks = ["test_schema", "test_faketensor"]
    for op, args in [
            (ops.cuembed_embedding_forward, (table, idx, off, w, "sum")),
            (ops.cuembed_embedding_forward, (table, idx, off, None, "mean")),
            (ops.cuembed_extract_row_ids_from_csr, (off[:-1], B * H)),
            (ops.cuembed_extract_row_ids_from_offsets, (off, B * H)),
            (ops.cuembed_transpose, (sid, idx, w)),
            (ops.cuembed_transpose, (sid, idx, None)),
            (ops.cuembed_transpose_bounded, (sid, idx, None, k)),
            (ops.cuembed_transpose_sample_ids, (sid, idx, w, k)),
            (ops.cuembed_transpose_sample_blocks, (sid, idx, w, k, 2)),
            (ops.cuembed_transpose_fixed_hotness, (idx.view(B, H), w.view(B, H), k, True)),
            (ops.cuembed_transpose_fixed_hotness, (idx.view(B, H), None, k, False)),
            (ops.cuembed_compute_compressed_grad_indices, (t_idx,)),
            (ops.cuembed_embedding_backward, (gy, k, t_idx, t_sid, t_w)),
            (ops.cuembed_embedding_backward_compressed, (gy, nu, t_idx, t_sid, remap, None)),
            (ops.cuembed_embedding_forward_fixed, (table, idx.view(B, H), None, "concat")),
            (ops.cuembed_embedding_weight_grad, (table, idx, off, gy))]:
        torch.library.opcheck(op, args, test_utils=checks)


@pytest.mark.parametrize("kind", [False, True, "fastest"])
@pytest.mark.parametrize("dtypes", [(torch.int64, torch.int32), (torch.int32, torch.int64)], ids=["idx64_off32", "idx32_off64"])
def test_backward_with_indices_and_offsets_of_different_integer_types(pyt, dtypes, kind):
    """The forward takes indices and offsets of different integer types (a type code each); the backward's index work
    has one type for lookups and sample ids and follows the indices -- it used to build the sample ids in the type of the
    offsets and hand both arrays on under ONE type code (out-of-bounds accesses for int64 indices + int32 offsets, a wrong
    gradient the other way round).  Native node and Python function, against same-type inputs."""
    idx_t, off_t = dtypes
    torch.manual_seed(11)
    k, d, B = 3000, 32, 700
    lens = torch.randint(0, 12, (B,), device="cuda")
    offsets = torch.cat([torch.zeros(1, dtype=torch.long, device="cuda"), lens.cumsum(0)])
    n = int(offsets[-1])
    indices = torch.randint(0, k, (n,), device="cuda")
    table = torch.randn(k, d, device="cuda")
    up = torch.randint(-3, 4, (B, d), device="cuda").float()
    w = torch.randint(1, 4, (n,), device="cuda").float() * 0.5
    for weights in (None, w):
        want_t = table.clone().requires_grad_(True)
        want = pyt.cuemb_embedding(want_t, indices, offsets, weights, sparse_grad=kind, hints=None)
        want.backward(up)
        for apply in (lambda t: pyt.cuemb_embedding(t, indices.to(idx_t), offsets.to(off_t), weights, sparse_grad=kind, hints=None),
                      lambda t: pyt._CuEmbEmbedding.apply(t, indices.to(idx_t), offsets.to(off_t), weights, kind, None)):
            t = table.clone().requires_grad_(True)
            got = apply(t)
            got.backward(up)
            assert torch.equal(got, want)
            g, gw = t.grad, want_t.grad
            if kind is not False:
                g, gw = g.to_dense(), gw.to_dense()
            assert torch.equal(g, gw)


def test_backward_rejects_weights_of_the_wrong_length(pyt):
    """The forward reads the first nnz weights of a longer tensor; the backward moves exactly nnz of them with the
    lookups, so a different length is an error there (it used to be unchecked in the native node)."""
    k, d, B, H = 100, 16, 10, 4
    table = torch.randn(k, d, device="cuda", requires_grad=True)
    offsets = torch.arange(0, B * H + 1, H, device="cuda")
    indices = torch.randint(0, k, (B * H,), device="cuda")
    w = torch.rand(B * H + 3, device="cuda")
    out = pyt.cuemb_embedding(table, indices, offsets, w, hints=None)
    with pytest.raises((RuntimeError, ValueError)):
        out.sum().backward()
