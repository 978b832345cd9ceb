#!/usr/bin/env python3
"""Randomised differential test of the whole path against the CPU oracle (test infrastructure):
random table shapes, batch sizes, hotness, index distributions and types, fixed / CSR layouts,
weights -- forward (bit-exact), row-id extraction, Transpose (bit-exact, stable; with and without the
key / row bounds; the fused fixed-hotness variant; signed keys and arbitrary payloads), compressed-index
remap, EmbeddingBackward dense and compressed (exact on small-integer gradients), the compressed backward with
num_unique left on the device, Transpose in sample blocks + the uncoalesced compressed gradient it leads to, and the
reference's compressed gradient computed from that blocked order (ComputeCompressedGradIndicesBlocked +
EmbeddingBackward(sample_blocks)), and the device-side halves of the multi-GPU sparse gradient exchange against numpy.

    python tools/fuzz_parity.py [--seconds 300] [--seed 0]

Not part of pytest's default run (it is open-ended); `tests/test_gpu_fuzz_smoke.py` runs a few
iterations of it."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def dev(a):
    import numpy as np
    import torch
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def one_case(rng, ce, O, np, torch, verbose=False):
    elem = [np.float32, np.float16][rng.integers(0, 2)]
    idx_t = [np.int32, np.int64][rng.integers(0, 2)]
    lane_bytes = [4, 8, 16][rng.integers(0, 3)]
    es = np.dtype(elem).itemsize
    W = int(rng.integers(1, 40)) * (lane_bytes // es if lane_bytes >= es else 1)
    if (W * es) % 4:
        W *= 2
    scale = rng.integers(0, 5)
    B = int(rng.integers(1, [40, 700, 5000, 20000, 60000][scale]))
    H = int(rng.integers(1, [70, 40, 20, 9, 8][scale]))
    ncat = int(rng.integers(H + 1, [300, 5000, 100000, 3000000][rng.integers(0, 4)]))
    alpha = [0.0, 1.05, 1.15, 1.6][rng.integers(0, 4)]
    csr = bool(rng.integers(0, 2))
    weighted = bool(rng.integers(0, 2))
    mode = ["sum", "mean"][rng.integers(0, 2)] if not weighted or True else "sum"
    desc = dict(elem=elem.__name__, idx=idx_t.__name__, W=W, B=B, H=H, ncat=ncat, alpha=alpha, csr=csr,
                weighted=weighted, mode=mode)
    if verbose:
        print(desc, flush=True)
    a = O.allocate_forward(ncat, W, B, H, alpha=alpha, is_csr=csr, elem=elem, index=idx_t)
    indices, offsets = a["indices"], (a["offsets"] if csr else None)
    weights = a["weights"] if weighted else None
    nnz = indices.size
    # ---- forward
    if not (weighted and mode == "mean" and False):
        want = O.embedding_forward(a["table"], indices, offsets, weights, batch_size=B, num_hots=0 if csr else H,
                                   mode=mode)
        got = ce.embedding_forward(dev(a["table"]), dev(indices), dev(offsets), dev(weights), batch_size=B,
                                   num_hots=0 if csr else H, mode=mode)
        view = np.uint16 if es == 2 else np.uint32
        assert np.array_equal(got.cpu().numpy().view(view), want.view(view)), ("forward", desc)
        if csr and mode != "concat":   # the scheduling hint moves no result: a random order and the one by bag length
            lengths = np.diff(offsets.astype(np.int64))
            counting = ce.bag_order_by_length(dev(offsets), batch_size=B, max_length=-1)      # the counting sort (clamp at 255)
            assert np.array_equal(counting.cpu().numpy(), np.argsort(-np.minimum(lengths, 255), kind="stable")), ("bag order", desc)
            decision = ce.new_row_loads_decision()
            decision[0] = int(rng.integers(0, 2))                                            # the device-side row-load word
            for order in (torch.randperm(B, device="cuda").int(), ce.bag_order_by_length(dev(offsets), batch_size=B), counting):
                got = ce.embedding_forward(dev(a["table"]), dev(indices), dev(offsets), dev(weights), batch_size=B,
                                           num_hots=0, mode=mode, sample_order=order, row_loads_device=decision)
                assert np.array_equal(got.cpu().numpy().view(view), want.view(view)), ("forward, sample_order", desc)
        elif not csr:
            decision = ce.new_row_loads_decision()
            decision[0] = int(rng.integers(0, 2))
            got = ce.embedding_forward(dev(a["table"]), dev(indices), None, dev(weights), batch_size=B, num_hots=H, mode=mode,
                                       row_loads_device=decision)
            assert np.array_equal(got.cpu().numpy().view(view), want.view(view)), ("forward, row_loads_device", desc)
    if nnz == 0:
        return desc
    # ---- row ids, transpose, remap
    if csr:
        sid = O.extract_row_ids_from_csr(offsets, idx_t)
        d_sid = ce.extract_row_ids_from_csr(dev(offsets), nnz, dev(indices).dtype)
    else:
        sid = O.extract_row_ids_from_fixed(B, H, idx_t)
        d_sid = ce.extract_row_ids_from_fixed(B, H, dev(indices).dtype, "cuda")
    assert np.array_equal(d_sid.cpu().numpy(), sid), ("extract", desc)
    ti, ts, tw = O.transpose(sid, indices, weights, stable=True)
    bound = ncat if rng.integers(0, 2) else None
    rows_bound = nnz if rng.integers(0, 2) else None
    d_ti, d_ts, d_tw = ce.transpose(d_sid, dev(indices), dev(weights), num_categories=bound, num_rows=rows_bound)
    assert np.array_equal(d_ti.cpu().numpy(), ti) and np.array_equal(d_ts.cpu().numpy(), ts), ("transpose", desc)
    if not csr:   # the fused fixed-hotness entry point must give the same three arrays
        f_ti, f_ts, f_tw = ce.transpose_fixed_hotness(dev(indices), B, H, dev(weights), num_categories=bound)
        assert torch.equal(f_ti, d_ti) and torch.equal(f_ts, d_ts), ("transpose_fixed_hotness", desc)
        if weighted:
            assert torch.equal(f_tw, d_tw), ("transpose_fixed_hotness weights", desc)
    if rng.integers(0, 4) == 0:   # Transpose as a generic COO transpose: signed keys, arbitrary payloads
        info = np.iinfo(idx_t)
        g_cols = rng.integers(info.min, info.max, nnz, endpoint=True).astype(idx_t) if rng.integers(0, 2) else \
            rng.integers(-5, 5, nnz).astype(idx_t)
        g_rows = rng.integers(info.min, info.max, nnz, endpoint=True).astype(idx_t)
        o = O.transpose(g_rows, g_cols, None, stable=True)
        g = ce.transpose(dev(g_rows), dev(g_cols))
        assert np.array_equal(g[0].cpu().numpy(), o[0]) and np.array_equal(g[1].cpu().numpy(), o[1]), ("generic transpose", desc)
    if weighted:
        assert np.array_equal(d_tw.cpu().numpy().view(np.uint16 if es == 2 else np.uint32),
                              tw.view(np.uint16 if es == 2 else np.uint32)), ("transpose weights", desc)
    remap = O.compute_compressed_grad_indices(ti)
    d_remap = ce.compute_compressed_grad_indices(d_ti)
    assert np.array_equal(d_remap.cpu().numpy(), remap), ("remap", desc)
    nu = int(remap[-1]) + 1
    # the remapped ids from the transpose call itself (one launch up to 4,096 lookups, one per pass up to 229,376)
    if rng.integers(0, 2) == 0:
        r4 = ce.transpose(d_sid, dev(indices), dev(weights), num_categories=bound, num_rows=rows_bound, remapped=True)
        assert torch.equal(r4[0], d_ti) and torch.equal(r4[1], d_ts) and torch.equal(r4[3], d_remap), ("transpose+remap", desc)
        if not csr:
            f4 = ce.transpose_fixed_hotness(dev(indices), B, H, dev(weights), num_categories=bound, remapped=True)
            assert torch.equal(f4[0], d_ti) and torch.equal(f4[1], d_ts) and torch.equal(f4[3], d_remap), ("fixed+remap", desc)
    # the reference's GradT arithmetic on data that is NOT exactly representable: bit-identical to the oracle
    if rng.integers(0, 4) == 0 and nnz <= 400000:
        gyr = rng.uniform(-1, 1, (B, W)).astype(elem)
        wr = None if weights is None else rng.uniform(0, 1, nnz).astype(elem)
        twr = None if wr is None else O.transpose(sid, indices, wr, stable=True)[2]
        want_r, want_rinv = O.embedding_backward(gyr, W, nu, ti, ts, remap, twr)
        got_r, got_rinv = ce.embedding_backward(dev(gyr), nu, d_ti, d_ts, d_remap, dev(twr), reference_sums=True)
        assert np.array_equal(got_r.cpu().numpy().view(np.uint8), want_r.view(np.uint8)), ("reference sums", desc)
        assert np.array_equal(got_rinv.cpu().numpy(), want_rinv), ("reference sums: inverse mapping", desc)
    # ---- backward: small-integer grad_y and (if weighted) weights 0.5/0.25 keep sums exact as long
    # as runs are short enough for the element type
    counts = np.bincount(remap)
    max_run = int(counts.max())
    limit = 2048 if es == 2 else (1 << 22)
    gy = (np.mod(O.allocate_grad_y(B * W), 3) - 1).reshape(B, W).astype(elem)
    use_w = weighted and max_run * 4 < limit
    if max_run < limit:
        w32 = None if not use_w else tw.astype(np.float32)
        want_c, want_inv = O.embedding_backward(gy.astype(np.float32), W, nu, ti, ts, remap, w32)
        got_c, got_inv = ce.embedding_backward(dev(gy), nu, d_ti, d_ts, d_remap, d_tw if use_w else None)
        assert np.array_equal(got_c.float().cpu().numpy(), want_c), ("backward compressed", desc)
        assert np.array_equal(got_inv.cpu().numpy(), want_inv), ("inverse mapping", desc)
        # the same without num_unique on the host: worst-case buffers, rows past the last id untouched
        if rng.integers(0, 3) == 0:
            cap = min(nnz, ncat)
            buf = torch.full((cap, W), 77.0, dtype=got_c.dtype, device="cuda")
            ibuf = torch.full((cap,), -3, dtype=d_ti.dtype, device="cuda")
            ce.embedding_backward(dev(gy), None, d_ti, d_ts, d_remap, d_tw if use_w else None, grad_embedding=buf,
                                  inverse_mapping=ibuf)
            assert torch.equal(buf[:nu], got_c) and torch.equal(ibuf[:nu], got_inv), ("backward, num_unique on device", desc)
            assert bool((buf[nu:] == 77.0).all()) and bool((ibuf[nu:] == -3).all()), ("rows past the last id", desc)
            # ... padded: the tail is zero and names the batch's rows in turn; one row short: the flag, nothing written
            ce.capacity_overflowed(reset=True)
            ce.embedding_backward(dev(gy), None, d_ti, d_ts, d_remap, d_tw if use_w else None, grad_embedding=buf,
                                  inverse_mapping=ibuf, pad_to_capacity=True)
            assert torch.equal(buf[:nu], got_c) and torch.equal(ibuf[:nu], got_inv), ("padded gradient", desc)
            pad_names = got_inv[torch.arange(buf.shape[0] - nu, device=buf.device) % nu]
            assert bool((buf[nu:] == 0).all()) and torch.equal(ibuf[nu:], pad_names), ("padding", desc)
            assert not ce.capacity_overflowed()
            if nu > 1:
                short = torch.full((nu - 1, W), 55.0, dtype=got_c.dtype, device="cuda")
                ishort = torch.full((nu - 1,), -4, dtype=d_ti.dtype, device="cuda")
                ce.embedding_backward(dev(gy), None, d_ti, d_ts, d_remap, d_tw if use_w else None, grad_embedding=short,
                                      inverse_mapping=ishort)
                assert ce.capacity_overflowed(reset=True) and bool((short == 55.0).all()) and bool((ishort == -4).all()), \
                    ("capacity one row short", desc)
        want_d = None
        if ncat * W <= 40_000_000:
            want_d, _ = O.embedding_backward(gy.astype(np.float32), W, ncat, ti, ts, None, w32)
            got_d, _ = ce.embedding_backward(dev(gy), ncat, d_ti, d_ts, None, d_tw if use_w else None)
            assert np.array_equal(got_d.float().cpu().numpy(), want_d), ("backward dense", desc)
        # ---- transpose in sample blocks (extension): every block sorted on its own == the oracle block by block;
        # the uncoalesced compressed gradient, scattered into the table, is the dense gradient
        if nnz > 131072 and rng.integers(0, 2) == 0:
            P = int(rng.integers(2, 10))
            L = ce.transpose_sample_block_length(nnz, P)
            parts = [O.transpose(sid[lo:lo + L], indices[lo:lo + L], None if weights is None else weights[lo:lo + L],
                                 stable=True) for lo in range(0, nnz, L)]
            b_ti, b_ts, b_tw = ce.transpose(d_sid, dev(indices), dev(weights), num_categories=bound, num_rows=rows_bound,
                                            sample_blocks=P)
            assert np.array_equal(b_ti.cpu().numpy(), np.concatenate([q[0] for q in parts])), ("block transpose keys", P, desc)
            assert np.array_equal(b_ts.cpu().numpy(), np.concatenate([q[1] for q in parts])), ("block transpose rows", P, desc)
            if not csr:
                f = ce.transpose_fixed_hotness(dev(indices), B, H, dev(weights), num_categories=bound, sample_blocks=P)
                assert torch.equal(f[0], b_ti) and torch.equal(f[1], b_ts), ("block transpose_fixed_hotness", P, desc)
            if want_d is not None:
                b_remap = ce.compute_compressed_grad_indices(b_ti)
                nub = int(b_remap[-1].item()) + 1
                bc, binv = ce.embedding_backward(dev(gy), nub, b_ti, b_ts, b_remap, b_tw if use_w else None)
                dense = torch.zeros((ncat, W), dtype=torch.float32, device="cuda").index_add_(0, binv.long(), bc.float())
                assert np.array_equal(dense.cpu().numpy(), want_d), ("backward over sample blocks", P, desc)
            # the REFERENCE's compressed gradient from the blocked order (at most 8 blocks): pair numbers + pair -> row
            # table compose to the fully sorted order's ids; rows, their order and the inverse mapping are the oracle's
            if -(-nnz // L) <= 8:
                pairs, table, nud = ce.compute_compressed_grad_indices_blocked(b_ti, P)
                assert int(nud.item()) == nu, ("blocked remap: num_unique", P, desc)
                full = table[pairs.long()].cpu().numpy().astype(np.int64)
                keys = b_ti.cpu().numpy()
                uniq, rank = np.unique(keys, return_inverse=True)
                assert np.array_equal(full & (ce.SHARED_ROW_BIT - 1), rank), ("blocked remap: ranks", P, desc)
                seen = np.zeros(uniq.shape[0], dtype=bool)
                for lo in range(0, nnz, L):
                    blk = rank[lo:lo + L]
                    assert np.array_equal((full[lo:lo + L] & ce.SHARED_ROW_BIT) != 0, seen[blk]), ("blocked remap: flags", P, desc)
                    seen[blk] = True
                cc, cinv = ce.embedding_backward(dev(gy), nu, b_ti, b_ts, pairs, b_tw if use_w else None, sample_blocks=P,
                                                 block_row_ids=table)
                assert np.array_equal(cc.float().cpu().numpy(), want_c), ("blocked coalesced backward", P, desc)
                assert np.array_equal(cinv.cpu().numpy(), want_inv), ("blocked coalesced inverse mapping", P, desc)
    exchange_case(rng, ce, np, torch, elem, idx_t, W, ncat, desc)
    return desc


def exchange_case(rng, ce, np, torch, elem, idx_t, W, ncat, desc):
    """The device-side halves of the sparse gradient exchange (PackRowsByOwner, the owner's fixed-capacity merge with
    FinishOwnerPiece) against numpy: random owners, slot and piece capacities (fitting and not), device-side counts."""
    from cuembed_amd import ops
    world = int(rng.integers(1, 10))
    n = int(rng.integers(0, min(ncat, 30000) + 1))
    ids = np.sort(rng.choice(ncat, size=n, replace=False)).astype(idx_t)
    rows = rng.integers(-4, 5, size=(n, W)).astype(elem)
    count = None if rng.integers(0, 3) == 0 else int(rng.integers(0, n + 2))
    valid = n if count is None else min(count, n)
    slot = max(1, int(rng.integers(1, 2 * (n // world + 2))))
    cuts = np.array([ncat * r // world for r in range(world)] + [ncat], dtype=np.int64)
    send_ids = torch.full((world * slot,), -1, dtype=torch.int64, device="cuda")
    send_rows = torch.zeros((world * slot, W), dtype=torch.from_numpy(rows).dtype, device="cuda")
    starts = torch.zeros((world + 1,), dtype=torch.int64, device="cuda")
    flag = torch.zeros((1,), dtype=torch.int64, device="cuda")
    d_count = None if count is None else torch.tensor([count], dtype=torch.from_numpy(ids).dtype, device="cuda")
    ops.exchange_pack_rows(torch.from_numpy(ids).cuda(), torch.from_numpy(rows).cuda().reshape(n, W), d_count, dev(cuts),
                           slot, 0, ncat, send_ids, send_rows, starts, flag)
    pos = np.searchsorted(ids[:valid], cuts, side="left")
    want_ids = np.full((world * slot,), ncat, dtype=np.int64)
    want_rows = np.zeros((world * slot, W), dtype=np.float32)
    over = False
    for r in range(world):
        take = min(int(pos[r + 1] - pos[r]), slot)
        over = over or pos[r + 1] - pos[r] > slot
        want_ids[r * slot: r * slot + take] = ids[pos[r]: pos[r] + take]
        want_rows[r * slot: r * slot + take] = rows[pos[r]: pos[r] + take]
    what = dict(desc, world=world, n=n, count=count, slot=slot)
    assert np.array_equal(starts.cpu().numpy(), pos), ("exchange pack: range starts", what)
    assert np.array_equal(send_ids.cpu().numpy(), want_ids), ("exchange pack: ids", what)
    assert np.array_equal(send_rows.float().cpu().numpy(), want_rows), ("exchange pack: rows", what)   # (slack: still zero)
    assert int(flag.item()) == int(over), ("exchange pack: flag", what)
    # the owner's merge of what one rank "received": several copies of the slots, so that ids repeat
    copies = int(rng.integers(1, 4))
    got_ids = send_ids.repeat(copies)
    got_rows = send_rows.repeat(copies, 1)
    real = want_ids[want_ids < ncat]
    distinct = np.unique(real).shape[0]
    capacity = max(1, distinct + int(rng.integers(-2, 6)))
    out_ids = torch.zeros((capacity + 1,), dtype=torch.int64, device="cuda")
    out_rows = torch.zeros((capacity + 1, W), dtype=send_rows.dtype, device="cuda")
    tail = torch.zeros((capacity + 2,), dtype=torch.int64, device="cuda")
    cnt = torch.zeros((1,), dtype=torch.int64, device="cuda")
    flag.zero_()
    pad_lo, pad_len = int(cuts[world // 2]), max(1, int(cuts[world // 2 + 1] - cuts[world // 2]))
    ops.exchange_merge(got_ids, got_rows, ncat, pad_lo, pad_len, out_ids, out_rows, tail, flag, cnt)
    assert int(cnt.item()) == distinct, ("exchange merge: count", what)
    assert int(flag.item()) == int(distinct > capacity), ("exchange merge: flag", what)
    t = tail.cpu().numpy()
    assert t[capacity] == min(distinct, capacity) and t[capacity + 1] == int(distinct > capacity), ("exchange merge: tail", what)
    if distinct <= capacity:
        uniq, inverse = np.unique(real, return_inverse=True)
        sums = np.zeros((capacity + 1, W), dtype=np.float64)
        np.add.at(sums, inverse, want_rows[want_ids < ncat].astype(np.float64) * copies)
        e_ids = pad_lo + np.arange(capacity + 1, dtype=np.int64) % pad_len
        e_ids[:distinct] = uniq
        assert np.array_equal(out_ids.cpu().numpy(), e_ids), ("exchange merge: ids", what)
        assert np.array_equal(out_rows.float().cpu().numpy(), sums.astype(np.float32)), ("exchange merge: rows", what)
        assert np.array_equal(t[:capacity], e_ids[:capacity]), ("exchange merge: tail ids", what)


def run(seconds=60.0, seed=0, max_cases=None, verbose=False):
    import numpy as np
    import torch
    import cuembed_amd as ce
    from oracle import oracle as O
    rng = np.random.default_rng(seed)
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds and (max_cases is None or n < max_cases):
        one_case(rng, ce, O, np, torch, verbose)
        n += 1
    torch.cuda.synchronize()
    assert ce._lib.lib().cuembed_peek_last_error() == 0
    return n


if __name__ == "__main__":
    p = argparse.ArgumentParser()
    p.add_argument("--seconds", type=float, default=300)
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--verbose", action="store_true")
    a = p.parse_args()
    print("fuzz_parity: %d cases passed (seed %d)" % (run(a.seconds, a.seed, verbose=a.verbose), a.seed))
