"""The entry points only enqueue stream-ordered work (no allocation, no host sync), so a whole
forward + transpose + remap + backward step can be captured into a HIP graph and replayed --
the intended way to run launch-bound small batches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_pipeline_under_hip_graph_capture(oracle):
    import cuembed_amd as ce
    ncat, W, B, H = 5000, 64, 512, 16
    a = oracle.allocate_forward(ncat, W, B, H, alpha=1.15)
    table = torch.from_numpy(a["table"]).cuda()
    idx = torch.from_numpy(a["indices"]).cuda()
    gy_np = oracle.allocate_grad_y(B * W).reshape(B, W)
    gy = torch.from_numpy(gy_np).cuda()
    nnz = B * H
    out = torch.empty((B, W), device="cuda")
    grad = torch.empty((ncat, W), device="cuda")
    work = torch.empty(max(ce.transpose_workspace_bytes(nnz, torch.int32), 1), dtype=torch.uint8, device="cuda")
    static = {}

    def step():
        ce.embedding_forward(table, idx, num_hots=H, out=out)
        sid = ce.extract_row_ids_from_fixed(B, H, torch.int32, "cuda")
        t_idx, t_sid, _ = ce.transpose(sid, idx, workspace=work, num_categories=ncat)
        ce.embedding_backward(gy, ncat, t_idx, t_sid, skip_grad_init=False, grad_embedding=grad)
        static.update(t_idx=t_idx, t_sid=t_sid)

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        step()                               # warm-up outside capture (library/module loading)
        torch.cuda.current_stream().synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
    out.zero_()
    grad.fill_(7)
    # new inputs in the SAME buffers, then replay
    b = oracle.allocate_forward(ncat, W, B, H, alpha=0.0)
    idx.copy_(torch.from_numpy(b["indices"]).cuda())
    g.replay()
    torch.cuda.synchronize()
    want = oracle.embedding_forward(a["table"], b["indices"], num_hots=H)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32))
    o_ti, o_ts, _ = oracle.transpose(oracle.extract_row_ids_from_fixed(B, H), b["indices"])
    assert np.array_equal(static["t_idx"].cpu().numpy(), o_ti)
    o_grad, _ = oracle.embedding_backward(gy_np, W, ncat, o_ti, o_ts)
    assert np.array_equal(grad.cpu().numpy(), o_grad)
