"""bfloat16 tables (this library's extension; the reference lists bf16 as future work,
README.md:111): fp32 accumulation, one round-to-nearest-even at the output.  Bit-exact against
the oracle's bf16 path (which is itself checked against torch's bf16 conversion on CPU)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev_bf16(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).cuda().view(torch.bfloat16)


def host_bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def ce():
    import cuembed_amd
    return cuembed_amd


@pytest.mark.parametrize("shape", [(2, 3, 4), (32, 1023, 26), (36, 1023, 26), (256, 300, 63), (514, 129, 9)],
                         ids=lambda s: "w%d_b%d_h%d" % s)
@pytest.mark.parametrize("idx_t", [np.int32, np.int64], ids=["i32", "i64"])
def test_forward_bf16_against_oracle(ce, oracle, shape, idx_t):
    W, B, H = shape
    for mode, csr, weighted in [("sum", False, False), ("sum", True, True), ("mean", False, False),
                                ("sum", False, True), ("concat", False, False)]:
        a = oracle.allocate_forward(20 * 1024, W, B, H, alpha=0.0, is_csr=csr, index=idx_t)
        table = oracle.to_bf16_bits(a["table"])
        w = oracle.to_bf16_bits(a["weights"]) if weighted else None
        offsets = a["offsets"] if csr else None
        want = oracle.embedding_forward(table, a["indices"], offsets, w, batch_size=B, num_hots=0 if csr else H,
                                        mode=mode)
        got = ce.embedding_forward(dev_bf16(table), dev(a["indices"]), dev(offsets),
                                   None if w is None else dev_bf16(w), batch_size=B, num_hots=0 if csr else H,
                                   mode=mode)
        assert np.array_equal(host_bits(got).reshape(want.shape), want), (mode, csr, weighted)


def test_backward_and_transpose_bf16(ce, oracle):
    W, B, H, ncat = 128, 2000, 16, 600
    a = oracle.allocate_forward(ncat, W, B, H, alpha=1.15)
    sid = oracle.extract_row_ids_from_fixed(B, H)
    wbits = oracle.to_bf16_bits(a["weights"])
    ti, ts, tw = oracle.transpose(sid, a["indices"], wbits)
    d_ti, d_ts, d_tw = ce.transpose(dev(sid), dev(a["indices"]), dev_bf16(wbits), num_categories=ncat)
    assert np.array_equal(d_ti.cpu().numpy(), ti) and np.array_equal(d_ts.cpu().numpy(), ts)
    assert np.array_equal(host_bits(d_tw), tw)
    ints = oracle.allocate_grad_y(B * W).reshape(B, W)
    gy = oracle.to_bf16_bits((np.mod(ints, 3) - 1).astype(np.float32))     # {-1,0,1}: runs < 256 stay exact in bf16
    a2 = oracle.allocate_forward(20000, W, B, H, alpha=0.0)                 # short runs
    ti2, ts2, _ = oracle.transpose(sid, a2["indices"])
    remap = oracle.compute_compressed_grad_indices(ti2)
    nu = int(remap[-1]) + 1
    want, winv = oracle.embedding_backward(gy, W, nu, ti2, ts2, remap)
    got, ginv = ce.embedding_backward(dev_bf16(gy), nu, dev(ti2), dev(ts2), dev(remap))
    assert np.array_equal(host_bits(got), want) and np.array_equal(ginv.cpu().numpy(), winv)
    # long runs through the atomic path: compare as fp32 values against an fp32 oracle run
    want32, _ = oracle.embedding_backward(oracle.from_bf16_bits(gy), W, ncat, ti, ts)
    got_long, _ = ce.embedding_backward(dev_bf16(gy), ncat, dev(ti), dev(ts))
    assert np.abs(want32).max() < 256                                       # exactly representable in bf16
    assert np.array_equal(got_long.float().cpu().numpy(), want32)


def test_torch_op_bf16(ce):
    from cuembed_amd import cuembed_pyt as pyt
    k, d, B = 3000, 64, 500
    table = torch.randn(k, d, device="cuda").bfloat16().requires_grad_(True)
    idx = torch.randint(0, k, (B * 8,), device="cuda")
    off = torch.arange(0, B * 8 + 1, 8, device="cuda")
    out = pyt.cuemb_embedding(table, idx, off)
    ref = table.detach().float()[idx].view(B, 8, d)
    acc = torch.zeros(B, d, device="cuda")
    for j in range(8):
        acc = acc + ref[:, j]
    assert torch.equal(out.detach(), acc.bfloat16())
    out.float().sum().backward()
    counts = torch.bincount(idx, minlength=k).float()
    assert torch.equal(table.grad.float(), counts[:, None].expand(k, d).bfloat16().float())
