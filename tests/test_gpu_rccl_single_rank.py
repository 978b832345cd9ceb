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
        # the all-gathers on a second RCCL communicator (the next step's all-to-all does not queue behind them), steps
        # back to back with different gradients: every result is its own step's
        piped = D.SparseGradExchange.calibrate(rows, inv, ncat, gather_group=dist.new_group())
        piped.start(rows, inv).wait()
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            got, previous = [], None
            for step in range(4):
                pending = piped.start(rows * (step + 1), inv)
                if previous is not None:
                    ids_p, rows_p, _ = previous.wait()
                    piped.note_flags(previous)
                    got.append(torch.zeros((ncat, W), dtype=torch.float32, device="cuda").index_add_(0, ids_p, rows_p.float()))
                previous = pending
            ids_p, rows_p, _ = previous.wait()
            got.append(torch.zeros((ncat, W), dtype=torch.float32, device="cuda").index_add_(0, ids_p, rows_p.float()))
        finally:
            torch.cuda.set_sync_debug_mode("default")
        for step, g in enumerate(got):
            sent = (rows * (step + 1)).float()          # (what start() was handed, fp16 product included)
            assert torch.equal(g, torch.zeros_like(g).index_add_(0, inv.long(), sent)), step
        assert not piped.overflowed() and bool(got[0].any())
    finally:
        dist.destroy_process_group()


def test_fixed_exchange_of_an_uncoalesced_gradient_world1():
    """An UNCOALESCED gradient (sample blocks) in worst-case buffers with a device-side count through the fixed-capacity
    exchange over RCCL: the rank's own rows are merged first, natively, inside the same step -- without a host wait; and
    an exchange whose capacities do not fit: a sticky flag, a well-formed result, no overrun, no exception.
    Integer-valued gradients: every sum is exact, whatever the order."""
    import torch.distributed as dist
    import cuembed_amd as ce
    from cuembed_amd import distributed as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        ncat, W, B, H, blocks = 30_000, 64, 16384, 12, 2
        g = torch.Generator(device="cpu").manual_seed(5)
        idx = (torch.rand((B * H,), generator=g) ** 3 * ncat).to(torch.int32).clamp_(max=ncat - 1).cuda()
        gy = torch.randint(-3, 4, (B, W), generator=g).half().cuda()
        sid = ce.extract_row_ids_from_fixed(B, H, torch.int32, "cuda")
        want = torch.zeros((ncat, W), dtype=torch.float32, device="cuda")
        want.index_add_(0, idx.long(), gy.float()[sid.long()])
        b_ti, b_ts, _, b_remap = ce.transpose_fixed_hotness(idx, B, H, num_categories=ncat, sample_blocks=blocks,
                                                            remapped=True)
        assert ce.transpose_sample_block_length(B * H, blocks) < B * H            # really two blocks
        u_cap = B * H
        u_rows = torch.zeros((u_cap, W), dtype=torch.float16, device="cuda")
        u_inv = torch.zeros((u_cap,), dtype=torch.int32, device="cuda")
        u_count = b_remap[-1:] + 1
        ce.embedding_backward(gy, None, b_ti, b_ts, b_remap, grad_embedding=u_rows, inverse_mapping=u_inv)
        k = int(u_count.item())
        distinct = int(torch.unique(idx).numel())
        assert distinct < k <= blocks * distinct                                   # some rows are there once per block
        ex = D.SparseGradExchange.calibrate(u_rows, u_inv, ncat, count=u_count, coalesced=False)
        assert ex.input_capacity < u_cap
        ex.start(u_rows, u_inv, count=u_count, coalesced=False).wait()
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            for _ in range(2):
                pending = ex.start(u_rows, u_inv, count=u_count, coalesced=False)
                ids_all, rows_all, counts = pending.wait()
                ex.note_flags(pending)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert not ex.overflowed()
        assert int(counts.sum().item()) == distinct
        got_ids, got_rows = ex.compact(ids_all, rows_all, counts)
        assert bool((got_ids[1:] > got_ids[:-1]).all())
        rebuilt = torch.zeros((ncat, W), dtype=torch.float32, device="cuda")
        rebuilt[got_ids] = got_rows.float()
        assert torch.equal(rebuilt, want)
        # as it comes (zero rows with valid ids fill the slack) it is the same gradient
        as_is = torch.zeros((ncat, W), dtype=torch.float32, device="cuda").index_add_(0, ids_all, rows_all.float())
        assert torch.equal(as_is, want)
        # capacities that do not fit, one at a time and all together
        fits = dict(pair_capacity=ex.pair_capacity, piece_capacity=ex.piece_capacity, local_capacity=ex.local_capacity,
                    input_capacity=ex.input_capacity)
        for tight in ({"pair_capacity": distinct // 4}, {"piece_capacity": distinct // 4}, {"local_capacity": distinct // 4},
                      {"input_capacity": k // 2},
                      {"pair_capacity": 7, "piece_capacity": 5, "local_capacity": 3, "input_capacity": 11}):
            small = D.SparseGradExchange(ncat, W, torch.float16, torch.device("cuda"), **dict(fits, **tight))
            pending = small.start(u_rows, u_inv, count=u_count, coalesced=False, async_op=False)
            ids_s, rows_s, _ = pending.wait()
            assert small.overflowed(reset=True) and not small.overflowed(), tight
            assert bool(((ids_s >= 0) & (ids_s < ncat)).all()) and bool(torch.isfinite(rows_s.float()).all()), tight
    finally:
        dist.destroy_process_group()
