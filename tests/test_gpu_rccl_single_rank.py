"""BASELINE config 5 path (fwd + bwd with the gradient combined over RCCL) on ONE rank: the
collective code of cuembed_amd/distributed.py runs through the real RCCL backend ("nccl") on
HIP tensors, with the HIP kernels doing the compute.  The world-size-2 arithmetic is covered on
CPU/gloo in tests/test_distributed_gloo.py; multi-GPU boxes are not available to the test suite."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_fwd_bwd_allreduce_world1(oracle):
    import torch.distributed as dist
    import cuembed_amd as ce
    from cuembed_amd import distributed as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ncat, W, B, H = 4000, 128, 1000, 12
        a = oracle.allocate_forward(ncat, W, B, H, alpha=1.15, elem=np.float16)
        table = torch.from_numpy(a["table"]).cuda()
        idx_all = torch.from_numpy(a["indices"]).cuda()
        idx, _, b_loc = D.shard_fixed(idx_all, None, B, H, dist.get_rank(), dist.get_world_size())
        out = ce.embedding_forward(table, idx.contiguous(), batch_size=b_loc, num_hots=H)
        want = oracle.embedding_forward(a["table"], a["indices"], num_hots=H)
        assert np.array_equal(out.cpu().numpy().view(np.uint16), want.view(np.uint16))
        gy_np = oracle.allocate_grad_y(B * W, np.float16).reshape(B, W)
        gy = torch.from_numpy(gy_np).cuda()
        sid = ce.extract_row_ids_from_fixed(b_loc, H, torch.int32, "cuda")
        t_idx, t_sid, _ = ce.transpose(sid, idx.contiguous(), num_categories=ncat)
        dense, _ = ce.embedding_backward(gy, ncat, t_idx, t_sid)
        D.allreduce_dense_grad(dense)                                   # RCCL all-reduce
        o_ti, o_ts, _ = oracle.transpose(oracle.extract_row_ids_from_fixed(B, H), a["indices"])
        o_grad, _ = oracle.embedding_backward(gy_np, W, ncat, o_ti, o_ts)
        assert np.array_equal(dense.cpu().numpy().view(np.uint16), o_grad.view(np.uint16))
        remap = ce.compute_compressed_grad_indices(t_idx)
        nu = int(remap[-1].item()) + 1
        rows, inv = ce.embedding_backward(gy, nu, t_idx, t_sid, remap)
        for algorithm in ("allgather", "owner"):       # RCCL all-gather / all-to-all + all-gather
            ids, summed = D.allreduce_sparse_grad(rows, inv, ncat, algorithm=algorithm)
            rebuilt = torch.zeros((ncat, W), dtype=torch.float16, device="cuda")
            rebuilt[ids] = summed
            assert np.array_equal(rebuilt.cpu().numpy().view(np.uint16), o_grad.view(np.uint16)), algorithm
        # the fixed-capacity exchange over RCCL: after the warm-up (calibrate + one step: communicator set-up, allocator)
        # a step must not wait for the device ANYWHERE -- torch's sync debug mode raises on every synchronising call
        want_ids, want_rows = D.allreduce_sparse_grad(rows, inv, ncat, algorithm="owner")
        cap = min(t_idx.numel(), ncat)
        rows_cap = torch.zeros((cap, W), dtype=torch.float16, device="cuda")
        inv_cap = torch.zeros((cap,), dtype=torch.int32, device="cuda")
        count = remap[-1:] + 1
        ex = D.SparseGradExchange.calibrate(rows, inv, ncat)
        ex.start(rows, inv).wait()
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            for _ in range(3):
                ce.embedding_backward(gy, None, t_idx, t_sid, remap, grad_embedding=rows_cap, inverse_mapping=inv_cap)
                pending = ex.start(rows_cap, inv_cap, count=count)          # all-gather in flight (async_op)
                out2 = ce.embedding_forward(table, idx.contiguous(), batch_size=b_loc, num_hots=H)   # ... behind the next forward
                ids_all, rows_all, counts = pending.wait()
                ex.note_flags(pending)
                rebuilt = torch.zeros((ncat, W), dtype=torch.float32, device="cuda").index_add_(0, ids_all, rows_all.float())
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert not ex.overflowed()
        assert np.array_equal(rebuilt.half().cpu().numpy().view(np.uint16), o_grad.view(np.uint16))
        got_ids, got_rows = ex.compact(ids_all, rows_all, counts)
        assert torch.equal(got_ids, want_ids.long()) and torch.equal(got_rows, want_rows)
        assert torch.equal(out2, out)
    finally:
        dist.destroy_process_group()
