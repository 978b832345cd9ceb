"""The scheduling hints decided on the device (hint_kernels.hpp): ForwardOptions::row_loads_device / DecideRowLoads and
the counting-sort BagOrderByLength (two small launches).  A hint never changes a result: every forward here is compared with the ORACLE, bit
for bit; the decisions and the order themselves are compared with their definitions computed in numpy.
(Reference: embedding_lookup.cuh:186-208 -- one launch rule whatever the data; these replace a caller's own statistics.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ce():
    import cuembed_amd
    assert torch.cuda.is_available()
    return cuembed_amd


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _offsets(lengths, dtype):
    return np.concatenate([[0], np.cumsum(lengths)]).astype(dtype)


@pytest.mark.parametrize("off_t", [np.int32, np.int64], ids=["o32", "o64"])
@pytest.mark.parametrize("batch", [1, 63, 1024, 1025, 5000, 65536, 131072])
def test_bag_order_counting_sort_is_the_stable_order_by_descending_length(ce, off_t, batch):
    """bound <= 255 and batch <= 131,072: the counting sort of hint_kernels.hpp (two small launches).  The permutation is DEFINED: samples by descending
    min(length, bound), ties in input order -- numpy's stable argsort of the negated clamped lengths; and it is what the
    general path (key kernel + the library's sort) gives for the same bound."""
    rng = np.random.default_rng(batch)
    cases = {
        "uniform_0_128": (rng.integers(0, 129, batch), 128),
        "all_equal": (np.full(batch, 17), 255),
        "two_lengths": (rng.choice([3, 200], batch), 200),
        "long_bags_clamped_at_255": (rng.integers(0, 400, batch), -1),
        "bound_below_the_longest": (rng.integers(0, 90, batch), 40),
        "mostly_empty": (rng.choice([0, 0, 0, 0, 9], batch), 9),
    }
    for name, (lengths, bound) in cases.items():
        off = dev(_offsets(lengths, off_t))
        got = ce.bag_order_by_length(off, max_length=bound).cpu().numpy()
        clamp = 255 if bound < 0 else bound
        want = np.argsort(-np.minimum(lengths, clamp), kind="stable").astype(np.int32)
        assert np.array_equal(got, want), (name, batch)
        if bound > 0 and lengths.max() <= bound:       # the general sort on the full lengths: the same permutation
            assert np.array_equal(ce.bag_order_by_length(off).cpu().numpy(), want), (name, batch, "general path")


def test_bag_order_beyond_the_counting_sort_limits_takes_the_general_sort(ce):
    rng = np.random.default_rng(5)
    batch = 131072 + 1000
    lengths = rng.integers(0, 60, batch)
    off = dev(_offsets(lengths, np.int64))
    want = np.argsort(-lengths, kind="stable").astype(np.int32)
    assert np.array_equal(ce.bag_order_by_length(off, max_length=59).cpu().numpy(), want)
    lengths = rng.integers(0, 3000, 20000)               # a bound beyond 255: general sort, unclamped
    off = dev(_offsets(lengths, np.int32))
    assert np.array_equal(ce.bag_order_by_length(off, max_length=2999).cpu().numpy(),
                          np.argsort(-lengths, kind="stable").astype(np.int32))


@pytest.mark.parametrize("elem", [np.float32, np.float16], ids=["f32", "f16"])
def test_forward_with_device_side_hints_has_the_oracles_bits(ce, oracle, elem):
    """CSR weighted forward with the one-launch bag order AND a row-load decision word of 0 and of 1: the oracle's bits."""
    rng = np.random.default_rng(8)
    rows, W, B = 20000, 128, 9000
    table = rng.standard_normal((rows, W)).astype(elem)
    lengths = rng.integers(0, 100, B)
    off = _offsets(lengths, np.int32)
    idx = rng.integers(0, rows, off[-1]).astype(np.int32)
    w = rng.random(off[-1]).astype(elem)
    want = oracle.embedding_forward(table, idx, off, w, batch_size=B, num_hots=0)
    order = ce.bag_order_by_length(dev(off), max_length=-1)
    bits = np.uint16 if elem == np.float16 else np.uint32
    for word in (0, 1):
        decision = ce.new_row_loads_decision()
        decision[0] = word
        got = ce.embedding_forward(dev(table), dev(idx), dev(off), dev(w), sample_order=order, row_loads_device=decision)
        assert np.array_equal(got.cpu().numpy().view(bits), want.view(bits)), word
    # fixed hotness through the staged kernel and a small batch through the wide-load kernel
    for Bf, H in ((4096, 32), (64, 64)):
        fidx = rng.integers(0, rows, Bf * H).astype(np.int32)
        fwant = oracle.embedding_forward(table, fidx, num_hots=H)
        for word in (0, 1):
            decision = ce.new_row_loads_decision()
            decision[0] = word
            got = ce.embedding_forward(dev(table), dev(fidx), num_hots=H, row_loads_device=decision)
            assert np.array_equal(got.cpu().numpy().view(bits), fwant.view(bits)), (Bf, H, word)


@pytest.mark.parametrize("idx_t", [torch.int32, torch.int64], ids=["i32", "i64"])
def test_row_load_decision_follows_the_sampled_distinct_fraction(ce, idx_t):
    """decision[0] = 1 exactly when >= 99.8 % (or the caller's fraction) of the strided sample -- 16 groups of 4,096 --
    is distinct inside its group, for a table of >= 1 GiB and a batch of >= 2^18 lookups; the kernel's own words are
    left at zero; the same words serve call after call."""
    g = torch.Generator(device="cuda").manual_seed(3)
    n, rows, big = 1 << 21, 10_000_000, 5 << 30
    uniform = torch.randint(0, rows, (n,), device="cuda", generator=g).to(idx_t)
    skewed = (rows * torch.rand(n, device="cuda", generator=g) ** 6).to(idx_t)
    few = torch.randint(0, 3000, (n,), device="cuda", generator=g).to(idx_t)

    def sampled_fraction(t):        # the definition, in numpy
        a = t.cpu().numpy()
        stride = n // 65536
        s = a[: 65536 * stride: stride]
        return float(np.mean([np.unique(s[grp::16]).size for grp in range(16)])) / 4096

    decision = ce.new_row_loads_decision()
    for name, t in (("uniform", uniform), ("skewed", skewed), ("few", few), ("uniform again", uniform)):
        ce.decide_row_loads(t, big, decision)
        frac = sampled_fraction(t)
        assert abs(frac - 0.998) > 0.001, "test data too close to the threshold"
        assert decision.tolist() == [1 if frac >= 0.998 else 0, 0, 0, 0], (name, frac)
    assert sampled_fraction(uniform) > 0.999 and sampled_fraction(skewed) < 0.9
    # a caller's own threshold
    frac = sampled_fraction(skewed)
    ce.decide_row_loads(skewed, big, decision, distinct_fraction=frac - 0.02)
    assert decision.tolist() == [1, 0, 0, 0]
    ce.decide_row_loads(skewed, big, decision, distinct_fraction=frac + 0.02)
    assert decision.tolist() == [0, 0, 0, 0]
    # the gates: a table that fits the caches, a batch that is latency-bound -- "default" without looking
    decision[0] = 1
    ce.decide_row_loads(uniform, 1 << 29, decision)
    assert decision.tolist() == [0, 0, 0, 0]
    decision[0] = 1
    ce.decide_row_loads(uniform[: (1 << 18) - 1], big, decision)
    assert decision.tolist() == [0, 0, 0, 0]
    ce.decide_row_loads(uniform[: 1 << 18], big, decision)         # the smallest batch that is looked at
    assert decision.tolist() == [1, 0, 0, 0]


def test_decision_and_forward_replay_from_a_hip_graph(ce, oracle):
    """Decision + forward captured into one graph and replayed on new indices in the same buffers: nothing in either
    needs the host."""
    rng = np.random.default_rng(2)
    rows, W, B, H = 4_000_000, 8, 8192, 32       # 2^18 lookups; 4,096 draws from 4M rows: 2 repeats (99.95 % distinct)
    table = rng.standard_normal((rows, W)).astype(np.float32)
    first = rng.integers(0, rows, B * H).astype(np.int32)
    second = rng.integers(0, 50, B * H).astype(np.int32)
    d_table, d_idx = dev(table), dev(first)
    out = torch.empty((B, W), device="cuda")
    decision = ce.new_row_loads_decision()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        def step():
            ce.decide_row_loads(d_idx, 5 << 30, decision)
            ce.embedding_forward(d_table, d_idx, num_hots=H, out=out, row_loads_device=decision)
        step()
        torch.cuda.current_stream().synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            step()
        for data, word in ((second, 0), (first, 1)):
            d_idx.copy_(dev(data))
            graph.replay()
            torch.cuda.current_stream().synchronize()
            assert int(decision[0]) == word
            want = oracle.embedding_forward(table, data, num_hots=H)
            assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
