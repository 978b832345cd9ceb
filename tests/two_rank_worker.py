"""One rank of the BASELINE config-5 path (fp16 forward + backward, batch sharded over the ranks,
table replicated, gradients combined across ranks) with the HIP kernels doing the compute.

Started by tests/test_gpu_two_rank_config5.py as a fresh process per rank (nothing has touched
the GPU before this file runs).  The ranks share GPU 0 -- the test box has one -- so RCCL cannot
build a communicator (it refuses two ranks on one device); the exchange runs over gloo on host
copies, which cuembed_amd.distributed does by itself for device tensors on a gloo group.  What is
under test is everything except the wire: shard_* -> EmbeddingForward -> ExtractRowIds -> Transpose
-> ComputeCompressedGradIndices -> EmbeddingBackward -> allreduce_dense_grad /
allreduce_sparse_grad (with its GPU-side merge), against the unsharded HIP result and the oracle.

    python tests/two_rank_worker.py RANK WORLD PORT
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = port
    import numpy as np
    import torch
    import torch.distributed as dist
    import cuembed_amd as ce
    from cuembed_amd import distributed as D
    from oracle import oracle as O
    O.build(ref=False)

    assert torch.cuda.is_available(), "needs a GPU"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)

    def d(a):
        return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def check(name, ok):
        if not ok:
            raise AssertionError("rank %d: %s" % (rank, name))

    ncat, W, B, H = 20000, 128, 4099, 16          # B not divisible by the world size
    for csr in (False, True):
        a = O.allocate_forward(ncat, W, B, H, alpha=1.15, is_csr=csr, elem=np.float16)
        table = d(a["table"])
        lo, hi = D.shard_bounds(B, rank, world)
        ints = O.allocate_grad_y(B * W).reshape(B, W)
        gy_full = (np.mod(ints, 3) - 1).astype(np.float16)       # {-1,0,1}: every partial sum is exact in fp16
        weights = a["weights"] if csr else None                   # the CSR case is the weighted one (0.5 / 0.25)

        # ---- forward on this rank's shard: no collective -----------------------------------
        if csr:
            off, idx, w, b_loc = D.shard_csr(d(a["offsets"]), d(a["indices"]), d(weights), rank, world)
            out = ce.embedding_forward(table, idx.contiguous(), off.contiguous(), w.contiguous(), num_hots=0)
            sid = ce.extract_row_ids_from_csr(off.contiguous(), nnz=idx.numel(), dtype=torch.int32)
        else:
            idx, w, b_loc = D.shard_fixed(d(a["indices"]), None, B, H, rank, world)
            out = ce.embedding_forward(table, idx.contiguous(), batch_size=b_loc, num_hots=H)
            sid = ce.extract_row_ids_from_fixed(b_loc, H, torch.int32, dev)
        check("shard size", b_loc == hi - lo)
        want_out = O.embedding_forward(a["table"], a["indices"], a["offsets"] if csr else None, weights,
                                       batch_size=B, num_hots=0 if csr else H)
        check("forward shard == oracle rows [lo, hi)",
              np.array_equal(out.cpu().numpy().view(np.uint16), want_out[lo:hi].view(np.uint16)))

        # ---- backward on the shard, then the exchange -----------------------------------------
        gy = d(gy_full[lo:hi])
        t_idx, t_sid, t_w = ce.transpose(sid, idx.contiguous(), None if w is None else w.contiguous(),
                                         num_categories=ncat, num_rows=b_loc)
        dense, _ = ce.embedding_backward(gy, ncat, t_idx, t_sid, None, t_w)
        work = D.allreduce_dense_grad(dense, async_op=True)   # host-staged on gloo: a completed handle, not None
        assert work is not None and work.wait() is not False and work.is_completed()

        # the unsharded result: HIP on the whole batch (every rank computes it) and the oracle
        f_sid = (ce.extract_row_ids_from_csr(d(a["offsets"]), nnz=a["indices"].shape[0], dtype=torch.int32)
                 if csr else ce.extract_row_ids_from_fixed(B, H, torch.int32, dev))
        f_ti, f_ts, f_tw = ce.transpose(f_sid, d(a["indices"]), d(weights), num_categories=ncat)
        full_hip, _ = ce.embedding_backward(d(gy_full), ncat, f_ti, f_ts, None, f_tw)
        o_sid = O.extract_row_ids_from_csr(a["offsets"]) if csr else O.extract_row_ids_from_fixed(B, H)
        o_ti, o_ts, o_tw = O.transpose(o_sid, a["indices"], None if weights is None else weights.astype(np.float32))
        want, _ = O.embedding_backward(gy_full.astype(np.float32), W, ncat, o_ti, o_ts, None, o_tw)
        check("exactness premise", np.abs(want).max() < 512)
        check("dense all-reduce == unsharded HIP", torch.equal(dense, full_hip))
        check("dense all-reduce == oracle", np.array_equal(dense.float().cpu().numpy(), want))

        remap = ce.compute_compressed_grad_indices(t_idx)
        nu = int(remap[-1].item()) + 1
        rows, inv = ce.embedding_backward(gy, nu, t_idx, t_sid, remap, t_w)
        # the same without the read-back: worst-case buffers, num_unique stays on the device
        cap = min(t_idx.numel(), ncat)
        rows_cap = torch.full((cap, W), 333.0, dtype=torch.float16, device=dev)
        inv_cap = torch.full((cap,), 7, dtype=torch.int32, device=dev)
        ce.embedding_backward(gy, None, t_idx, t_sid, remap, t_w, grad_embedding=rows_cap, inverse_mapping=inv_cap)
        check("backward without num_unique on the host", torch.equal(rows_cap[:nu], rows) and torch.equal(inv_cap[:nu], inv)
              and bool((rows_cap[nu:] == 333.0).all()))
        for algorithm in ("allgather", "owner"):
            ids, summed = D.allreduce_sparse_grad(rows_cap, inv_cap, ncat, algorithm=algorithm, num_unique=remap[-1:] + 1)
            check(algorithm + ": exchange of worst-case buffers", bool((ids[1:] > ids[:-1]).all()))
            rebuilt = torch.zeros((ncat, W), dtype=torch.float16, device=dev)
            rebuilt[ids.long()] = summed
            check(algorithm + ": padded sparse exchange == dense all-reduce", torch.equal(rebuilt, dense))
        # the shard transposed in sample blocks (forced: the shard is small): an uncoalesced gradient through both exchanges
        b_idx, b_sid, b_w = ce.transpose(sid, idx.contiguous(), None if w is None else w.contiguous(), num_categories=ncat,
                                         num_rows=b_loc, sample_blocks=3)
        if idx.numel() > 131072:
            check("sample blocks really cut the shard", not torch.equal(b_idx, t_idx))
        b_remap = ce.compute_compressed_grad_indices(b_idx)
        nub = int(b_remap[-1].item()) + 1
        b_rows, b_inv = ce.embedding_backward(gy, nub, b_idx, b_sid, b_remap, b_w)
        for algorithm in ("allgather", "owner"):
            ids, summed = D.allreduce_sparse_grad(b_rows, b_inv, ncat, algorithm=algorithm, coalesced=False)
            check(algorithm + ": uncoalesced exchange, ids ascending and unique", bool((ids[1:] > ids[:-1]).all()))
            rebuilt = torch.zeros((ncat, W), dtype=torch.float16, device=dev)
            rebuilt[ids.long()] = summed
            check(algorithm + ": uncoalesced sparse exchange == dense all-reduce", torch.equal(rebuilt, dense))
        for algorithm in ("allgather", "owner"):
            ids, summed = D.allreduce_sparse_grad(rows, inv, ncat, algorithm=algorithm)
            check(algorithm + ": ids ascending and unique", bool((ids[1:] > ids[:-1]).all()))
            rebuilt = torch.zeros((ncat, W), dtype=torch.float16, device=dev)
            rebuilt[ids.long()] = summed
            check(algorithm + ": sparse exchange == dense all-reduce", torch.equal(rebuilt, dense))
        # ---- the fixed-capacity exchange (SparseGradExchange): sizes from a warm-up step, then device-side counts only
        want_ids, want_rows = D.allreduce_sparse_grad(rows, inv, ncat, algorithm="owner")
        ex = D.SparseGradExchange.calibrate(rows_cap, inv_cap, ncat, count=remap[-1:] + 1)
        for step in range(3):
            pending = ex.start(rows_cap, inv_cap, count=remap[-1:] + 1)
            ids_all, rows_all, counts = pending.wait()
            ex.note_flags(pending)
            # the result as it comes is a valid uncoalesced COO gradient: zero rows with valid ids fill the slack
            rebuilt = torch.zeros((ncat, W), dtype=torch.float32, device=dev).index_add_(0, ids_all, rows_all.float())
            check("fixed-capacity exchange, scatter-added as it comes == dense all-reduce", torch.equal(rebuilt.half(), dense))
            got_ids, got_rows = ex.compact(ids_all, rows_all, counts)
            check("fixed-capacity exchange == exact exchange (ids)", torch.equal(got_ids, want_ids.long()))
            check("fixed-capacity exchange == exact exchange (rows, same bits)", torch.equal(got_rows, want_rows))
        check("fixed-capacity exchange: no overflow", not ex.overflowed())
        ex_u = D.SparseGradExchange.calibrate(b_rows, b_inv, ncat, coalesced=False)
        pending = ex_u.start(b_rows, b_inv, coalesced=False, async_op=False)
        got_ids, got_rows = ex_u.compact(*pending.wait())
        rebuilt = torch.zeros((ncat, W), dtype=torch.float16, device=dev)
        rebuilt[got_ids] = got_rows
        check("fixed-capacity exchange of an uncoalesced gradient == dense all-reduce",
              torch.equal(rebuilt, dense) and bool((got_ids[1:] > got_ids[:-1]).all()) and not ex_u.overflowed())
        tight = D.SparseGradExchange(ncat, W, torch.float16, dev, pair_capacity=max((ex.pair_capacity - 16) // 3, 1),
                                     piece_capacity=ex.piece_capacity)
        pending = tight.start(rows, inv)
        ids_all, rows_all, counts = pending.wait()
        tight.note_flags(pending)
        check("slots too small: sticky flag on every rank, well-formed result",
              tight.overflowed(reset=True) and not tight.overflowed() and int(ids_all.min()) >= 0
              and int(ids_all.max()) < ncat and bool(torch.isfinite(rows_all.float()).all()))
        tight = D.SparseGradExchange(ncat, W, torch.float16, dev, pair_capacity=ex.pair_capacity, piece_capacity=3)
        pending = tight.start(rows, inv, async_op=False)
        ids_all, rows_all, counts = pending.wait()
        check("piece too small: flag, ids still valid", tight.overflowed() and int(ids_all.min()) >= 0 and int(ids_all.max()) < ncat)
        # every rank ended with the same gradient
        digest = torch.tensor([float(dense.float().abs().sum().item())], dtype=torch.float64)
        both = [torch.zeros_like(digest) for _ in range(world)]
        dist.all_gather(both, digest)
        check("ranks agree", all(float(b) == float(both[0]) for b in both))

    # an empty exchange (_merge_on_gpu on zero rows)
    e_rows = torch.empty((0, W), dtype=torch.float16, device=dev)
    e_ids = torch.empty((0,), dtype=torch.int32, device=dev)
    for algorithm in ("allgather", "owner"):
        ids, summed = D.allreduce_sparse_grad(e_rows, e_ids, ncat, algorithm=algorithm)
        check("empty exchange", ids.numel() == 0 and summed.shape[0] == 0)
    torch.cuda.synchronize()
    assert ce._lib.lib().cuembed_peek_last_error() == 0
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank, flush=True)


if __name__ == "__main__":
    main()
