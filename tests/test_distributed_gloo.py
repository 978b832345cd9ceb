"""world_size-2 test of the batch-shard data-parallel path on CPU (gloo).  The HIP kernels need
a GPU, so the per-rank compute is the CPU oracle here; what is under test is the sharding
arithmetic and the collectives of cuembed_amd/distributed.py:
  concat(shard forwards) == full forward;  all-reduce(shard backward grads) == full backward grad
(dense and sparse exchange), for fixed-hotness and CSR batches."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _no_host_reads:
    """Inside: reading a tensor's VALUE into Python raises -- what would be a device-to-host wait on a GPU
    (.item() / .tolist() / bool() / int() / float() / index(), nonzero and friends).  The steady state of the
    fixed-capacity exchange must not do any."""
    NAMES = ("item", "tolist", "__bool__", "__int__", "__float__", "__index__", "nonzero", "numpy", "__len__")

    def __enter__(self):
        self.saved = {n: getattr(torch.Tensor, n) for n in self.NAMES if n != "__len__"}

        def make(name):
            def trap(*a, **k):
                raise AssertionError("host read-back inside a steady-state exchange step: Tensor.%s" % name)
            return trap
        for n in self.saved:
            setattr(torch.Tensor, n, make(n))
        return self

    def __exit__(self, *exc):
        for n, f in self.saved.items():
            setattr(torch.Tensor, n, f)
        return False


def _worker(rank, world, port, csr, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from cuembed_amd import distributed as D
        from oracle import oracle as O
        ncat, W, B, H = 300, 16, 101, 7          # B not divisible by world
        a = O.allocate_forward(ncat, W, B, H, alpha=1.15, is_csr=csr)
        table = a["table"]
        idx_t = torch.from_numpy(a["indices"])
        w_t = torch.from_numpy(a["weights"])
        if csr:
            off, idx, w, b_loc = D.shard_csr(torch.from_numpy(a["offsets"]), idx_t, w_t, rank, world)
            out = O.embedding_forward(table, idx.numpy(), off.numpy(), w.numpy(), num_hots=0)
            sid = O.extract_row_ids_from_csr(off.numpy())
        else:
            idx, w, b_loc = D.shard_fixed(idx_t, w_t, B, H, rank, world)
            out = O.embedding_forward(table, idx.numpy(), None, w.numpy(), batch_size=b_loc, num_hots=H)
            sid = O.extract_row_ids_from_fixed(b_loc, H)
        lo, hi = D.shard_bounds(B, rank, world)
        assert b_loc == hi - lo and out.shape[0] == b_loc
        # forward: gather the shards (test-only) and compare with the unsharded result
        gathered = [None] * world
        dist.all_gather_object(gathered, out)
        full_out = O.embedding_forward(table, a["indices"], a["offsets"] if csr else None, a["weights"],
                                       batch_size=B, num_hots=0 if csr else H)
        assert np.array_equal(np.concatenate(gathered), full_out)
        # backward on the shard (integer grad_y -> exact in any summation order)
        gy_full = O.allocate_grad_y(B * W).reshape(B, W)
        gy = gy_full[lo:hi]
        t_idx, t_sid, t_w = O.transpose(sid, idx.numpy(), w.numpy())
        dense, _ = O.embedding_backward(gy, W, ncat, t_idx, t_sid, None, t_w)
        dense_t = torch.from_numpy(dense.copy())
        work = D.allreduce_dense_grad(dense_t, async_op=True)     # a handle on every path
        work.wait()
        full_sid = O.extract_row_ids_from_csr(a["offsets"]) if csr else O.extract_row_ids_from_fixed(B, H)
        f_idx, f_sid, f_w = O.transpose(full_sid, a["indices"], a["weights"])
        want, _ = O.embedding_backward(gy_full, W, ncat, f_idx, f_sid, None, f_w)
        assert np.array_equal(dense_t.numpy(), want)
        # sparse exchange of compressed gradients
        remap = O.compute_compressed_grad_indices(t_idx)
        nu = int(remap[-1]) + 1
        comp, inv = O.embedding_backward(gy, W, nu, t_idx, t_sid, remap, t_w)
        for algorithm in ("allgather", "owner", "auto"):
            ids, rows = D.allreduce_sparse_grad(torch.from_numpy(comp), torch.from_numpy(inv), ncat,
                                                algorithm=algorithm)
            assert np.all(np.diff(ids.numpy()) > 0), algorithm          # ascending, no duplicates
            rebuilt = np.zeros_like(want)
            rebuilt[ids.numpy()] = rows.numpy()
            assert np.array_equal(rebuilt, want), algorithm
        # buffers sized for the worst case with the count in a tensor (what a step without host read-backs hands over)
        pad = 5 + rank
        comp_pad = np.concatenate([comp, np.full((pad, W), 777.0, dtype=comp.dtype)])
        inv_pad = np.concatenate([inv, np.full((pad,), 3, dtype=inv.dtype)])           # garbage ids past the count
        for algorithm in ("allgather", "owner"):
            ids, rows = D.allreduce_sparse_grad(torch.from_numpy(comp_pad), torch.from_numpy(inv_pad), ncat,
                                                algorithm=algorithm, num_unique=torch.tensor([nu]))
            rebuilt = np.zeros_like(want)
            rebuilt[ids.numpy()] = rows.numpy()
            assert np.array_equal(rebuilt, want), ("padded", algorithm)
        # an UNCOALESCED compressed gradient (what a transpose in sample blocks leads to): ids ascend only inside a
        # block and a row may appear once per block -- here: the rank's lookups cut in two halves
        half = t_idx.shape[0] // 2
        parts_rows, parts_ids = [], []
        for lo_, hi_ in ((0, half), (half, t_idx.shape[0])):
            o = np.argsort(idx.numpy()[lo_:hi_], kind="stable")
            b_idx, b_sid = idx.numpy()[lo_:hi_][o], sid[lo_:hi_][o]
            b_w = None if w is None else w.numpy()[lo_:hi_][o]
            b_remap = O.compute_compressed_grad_indices(b_idx)
            c_, i_ = O.embedding_backward(gy, W, int(b_remap[-1]) + 1, b_idx, b_sid, b_remap, b_w)
            parts_rows.append(c_)
            parts_ids.append(i_)
        u_rows, u_ids = np.concatenate(parts_rows), np.concatenate(parts_ids)
        for algorithm in ("allgather", "owner"):
            ids, rows = D.allreduce_sparse_grad(torch.from_numpy(u_rows), torch.from_numpy(u_ids), ncat,
                                                algorithm=algorithm, coalesced=False)
            assert np.all(np.diff(ids.numpy()) > 0), ("uncoalesced", algorithm)
            rebuilt = np.zeros_like(want)
            rebuilt[ids.numpy()] = rows.numpy()
            assert np.array_equal(rebuilt, want), ("uncoalesced", algorithm)
        # ---- the fixed-capacity exchange: sizes fixed by a warm-up step, then NO host read-back in a step
        want_ids, want_rows = D.allreduce_sparse_grad(torch.from_numpy(comp), torch.from_numpy(inv), ncat, algorithm="owner")
        ex = D.SparseGradExchange.calibrate(torch.from_numpy(comp), torch.from_numpy(inv), ncat)
        want_dense = torch.from_numpy(want.copy())
        for step in range(3):      # (twice the same buffers in turn, then once more: the double-buffered result)
            with _no_host_reads():
                pending = ex.start(torch.from_numpy(comp_pad), torch.from_numpy(inv_pad), count=torch.tensor([nu]))
                ids_all, rows_all, counts = pending.wait()
                ex.note_flags(pending)
                # as it comes: a valid uncoalesced COO gradient (zero rows with valid ids fill the slack)
                rebuilt_t = torch.zeros_like(want_dense).index_add_(0, ids_all, rows_all)
            assert torch.equal(rebuilt_t, want_dense), ("fixed", step)
            assert ids_all.numel() == world * ex.piece_capacity and int(ids_all.min()) >= 0 and int(ids_all.max()) < ncat
            got_ids, got_rows = ex.compact(ids_all, rows_all, counts)
            assert torch.equal(got_ids, want_ids) and torch.equal(got_rows, want_rows), ("fixed compact", step)
        assert not ex.overflowed()
        # ... buffers sized for the worst case (four times the rows, garbage behind the count): only input_capacity rows
        # are looked at; a count beyond it raises the flag
        big_rows = np.concatenate([comp, np.full((3 * comp.shape[0] + 40, W), 555.0, dtype=comp.dtype)])
        big_inv = np.concatenate([inv, np.full((3 * inv.shape[0] + 40,), 1, dtype=inv.dtype)])
        assert big_inv.shape[0] > ex.input_capacity >= nu
        with _no_host_reads():
            pending = ex.start(torch.from_numpy(big_rows), torch.from_numpy(big_inv), count=torch.tensor([nu]), async_op=False)
        got_ids, got_rows = ex.compact(*pending.wait())
        assert torch.equal(got_ids, want_ids) and torch.equal(got_rows, want_rows) and not ex.overflowed()
        pending = ex.start(torch.from_numpy(big_rows), torch.from_numpy(big_inv), count=torch.tensor([ex.input_capacity + 1]),
                           async_op=False)
        assert ex.overflowed(reset=True)
        # ... an uncoalesced gradient: the rank's own rows are merged first, inside the same step
        ex_u = D.SparseGradExchange.calibrate(torch.from_numpy(u_rows), torch.from_numpy(u_ids), ncat, coalesced=False)
        with _no_host_reads():
            pending = ex_u.start(torch.from_numpy(u_rows), torch.from_numpy(u_ids), coalesced=False, async_op=False)
            ids_all, rows_all, counts = pending.wait()
        got_ids, got_rows = ex_u.compact(ids_all, rows_all, counts)
        assert torch.equal(got_ids, want_ids) and torch.equal(got_rows, want_rows), "fixed uncoalesced"
        assert not ex_u.overflowed()
        # ... capacities that do not fit: a flag on EVERY rank instead of an overrun, and a well-formed result
        small = D.SparseGradExchange(ncat, W, torch.float32, torch.device("cpu"), pair_capacity=max((ex.pair_capacity - 16) // 4, 1),
                                     piece_capacity=ex.piece_capacity)
        with _no_host_reads():
            pending = small.start(torch.from_numpy(comp), torch.from_numpy(inv))
            ids_all, rows_all, counts = pending.wait()
            small.note_flags(pending)
        assert small.overflowed(reset=True) and not small.overflowed()
        assert int(ids_all.min()) >= 0 and int(ids_all.max()) < ncat and bool(torch.isfinite(rows_all).all())
        small = D.SparseGradExchange(ncat, W, torch.float32, torch.device("cpu"), pair_capacity=ex.pair_capacity,
                                     piece_capacity=2)
        pending = small.start(torch.from_numpy(comp), torch.from_numpy(inv), async_op=False)
        ids_all, rows_all, counts = pending.wait()
        assert small.overflowed() and int(ids_all.min()) >= 0 and int(ids_all.max()) < ncat
        # ... pipelined: the all-gathers on a second group over the same ranks, three steps in flight back to back with
        # DIFFERENT gradients (scaled by the step) -- each result must be its own step's (two sets of piece and result
        # buffers used in turn; a result stays valid until the start() after the next one)
        piped = D.SparseGradExchange.calibrate(torch.from_numpy(comp), torch.from_numpy(inv), ncat,
                                               gather_group=dist.new_group())
        with _no_host_reads():
            results, previous = [], None
            for step in range(4):
                pending = piped.start(torch.from_numpy(comp_pad * (step + 1)), torch.from_numpy(inv_pad),
                                      count=torch.tensor([nu]))
                if previous is not None:
                    ids_p, rows_p, _ = previous.wait()
                    piped.note_flags(previous)
                    results.append(torch.zeros_like(want_dense).index_add_(0, ids_p, rows_p))
                previous = pending
            ids_p, rows_p, _ = previous.wait()
            results.append(torch.zeros_like(want_dense).index_add_(0, ids_p, rows_p))
        for step, got in enumerate(results):
            assert torch.equal(got, want_dense * (step + 1)), ("pipelined", step)
        assert not piped.overflowed()
        # ... and nothing to exchange at all
        pending = ex.start(torch.empty((0, W)), torch.empty((0,), dtype=torch.int64))
        ids_all, rows_all, counts = pending.wait()
        assert int(counts.sum()) == 0 and not bool(rows_all.any())
        m = D.exchange_model_ms(8, 572000, 2447501, 256, 2)
        assert 0.2 < m["all_to_all_ms"] < 0.3 and 1.0 < m["all_gather_ms"] < 1.1
        # nothing to exchange on any rank (a batch without lookups)
        for algorithm in ("allgather", "owner"):
            ids, rows = D.allreduce_sparse_grad(torch.empty((0, W)), torch.empty((0,), dtype=torch.int64), ncat,
                                                algorithm=algorithm)
            assert ids.numel() == 0 and rows.shape[0] == 0, algorithm
        ret[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_batch_shard_world3_owner_exchange(oracle):
    """Three ranks: uneven sample shards AND uneven owner ranges (300 ids over 3 owners is even,
    101 samples are not); a rank may own ids it has no lookups for."""
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), False, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok", 2: "ok"}


@pytest.mark.parametrize("csr", [False, True], ids=["fixed", "csr"])
def test_batch_shard_world2(oracle, csr):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), csr, ret), nprocs=world, join=True)
    assert dict(ret) == {0: "ok", 1: "ok"}


def test_shard_bounds_cover_batch():
    from cuembed_amd import distributed as D
    for B in (1, 7, 64, 65536, 524288 + 3):
        for G in (1, 2, 3, 4, 8):
            spans = [D.shard_bounds(B, r, G) for r in range(G)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(G - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
