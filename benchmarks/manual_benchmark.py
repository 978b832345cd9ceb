#!/usr/bin/env python3
"""Counterpart of the reference's benchmarks/manual_benchmark.cu: same flags, same synthetic
workload recipe, same timing protocol (one warm-up call; every iteration timed alone with HIP
events, a 1.02 GB cache-flushing reduction between iterations unless --clear_caches=false) and the
same "Application BW" formulas (manual_benchmark.cu:250-261, :340-354, :444-471), for forward,
transpose(+compressed remap) and backward.

    python benchmarks/manual_benchmark.py --num_categories 10000000 --embed_width 256 \
        --batch_size 65536 --alpha 1.15 --hotness 64 --half_embedding_type=true --iterations 100
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HBM_PEAK_GBPS = 8000.0


def str2bool(v):
    if isinstance(v, bool):
        return v
    return str(v).lower() in ("1", "true", "yes", "on")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--num_categories", type=int, default=1048576)
    p.add_argument("--embed_width", type=int, default=128)
    p.add_argument("--batch_size", type=int, default=1024)
    p.add_argument("--hotness", type=int, default=1)
    p.add_argument("--iterations", type=int, default=1)
    p.add_argument("--alpha", type=float, default=0.0)
    for name, default in [("use_int64_indices", False), ("check_result", False),
                          ("half_embedding_type", False), ("csr_input", False), ("weighted_sum", False),
                          ("fp16_math", False), ("compressed_grad", True), ("skip_grad_init", True),
                          ("forward_only", False), ("enable_csv", False), ("clear_caches", True)]:
        p.add_argument("--" + name, type=str2bool, nargs="?", const=True, default=default)
    p.add_argument("--one_call_index_work", type=str2bool, nargs="?", const=True, default=False,
                   help="also time this library's one-call form of the index work (fixed hotness only)")
    p.add_argument("--bounded_sort", type=str2bool, nargs="?", const=True, default=False,
                   help="tell Transpose that indices < num_categories (this library's extension)")
    p.add_argument("--device_table_fill", type=str2bool, nargs="?", const=True, default=None,
                   help="fill the table on the GPU instead of with the reference's host RNG "
                        "(default: automatically for tables > 64M elements; not with --check_result)")
    return p.parse_args()


def main():
    run(parse())


def run(a, table_cache=None, quiet=False):
    """Runs one benchmark point; returns [(name, total_ms, bw_l2, bw_dram), ...].
    table_cache: optional dict reused across calls to keep (rows, width, dtype) tables alive."""
    import numpy as np
    import torch
    import cuembed_amd as ce
    from cuembed_amd import harness

    dev = torch.device("cuda", 0)
    elem_np = np.float16 if a.half_embedding_type else np.float32
    elem_t = torch.float16 if a.half_embedding_type else torch.float32
    idx_np = np.int64 if a.use_int64_indices else np.int32
    idx_t = torch.int64 if a.use_int64_indices else torch.int32
    es, isz = (2 if a.half_embedding_type else 4), (8 if a.use_int64_indices else 4)
    W, B, H = a.embed_width, a.batch_size, a.hotness
    dev_fill = a.device_table_fill
    if dev_fill is None:
        dev_fill = (a.num_categories * W > (64 << 20)) and not a.check_result
    w = harness.allocate_forward(a.num_categories, W, B, H, alpha=a.alpha, is_csr=a.csr_input, elem=elem_np,
                                 index=idx_np, with_table=not dev_fill, consume_table_draws=not dev_fill)
    key = (a.num_categories, W, elem_t)
    if dev_fill and table_cache is not None and key in table_cache:
        table = table_cache[key]
    elif dev_fill:
        table = torch.empty((a.num_categories, W), dtype=elem_t, device=dev)
        table.uniform_(-1, 1)
        if table_cache is not None:
            for k in [k for k in table_cache if k != "flush"]:
                del table_cache[k]
            table_cache[key] = table
    else:
        table = torch.from_numpy(w["table"]).to(dev)
    indices = torch.from_numpy(w["indices"]).to(dev)
    offsets = torch.from_numpy(w["offsets"]).to(dev) if a.csr_input else None
    weights = torch.from_numpy(w["weights"]).to(dev) if a.weighted_sum else None
    nnz = indices.numel()
    hots = 0 if a.csr_input else H
    out = torch.empty((B, W), dtype=elem_t, device=dev)

    flush = None
    if a.clear_caches:
        if table_cache is not None and "flush" in table_cache:
            flush = table_cache["flush"]
        else:
            flush = torch.ones(256_000_000, dtype=torch.int32, device=dev)
            if table_cache is not None:
                table_cache["flush"] = flush
    sink = torch.zeros((), dtype=torch.int32, device=dev)

    def clear():
        if flush is not None:
            sink.add_(flush.max())

    def timed(fn):
        fn()                      # warm-up (manual_benchmark.cu:207)
        clear()
        total = 0.0
        if a.clear_caches:
            for _ in range(a.iterations):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); fn(); e.record(); e.synchronize()
                total += s.elapsed_time(e)
                clear()
        else:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.iterations):
                fn()
            e.record(); e.synchronize()
            total = s.elapsed_time(e)
        return total

    rows = []

    def report(name, ms, bw_l2, bw_dram, label):
        if not quiet:
            print("%s. Iterations: %d , Total time [ms]: %.2f , Avg [ms]: %.4f , %s"
                  % (name, a.iterations, ms, ms / a.iterations, label), flush=True)
        rows.append((name.lower().split()[-1], ms, bw_l2, bw_dram))

    # ---- forward ------------------------------------------------------------------
    fwd = lambda: ce.embedding_forward(table, indices, offsets, weights, batch_size=B, num_hots=hots,  # noqa: E731
                                       mode="sum", fp16_math=a.fp16_math, out=out)
    ms = timed(fwd)
    if a.csr_input:
        nbytes = es * (nnz - 1 + B) * W
    else:
        nbytes = es * B * (H + 1) * W
    bw = nbytes * a.iterations / 1e6 / ms
    report("Embedding forward", ms, bw, 0.0, "Application BW [GB/s]: %.2f (algorithmic bytes / time: cache hits "
           "count, NOT an HBM rate)" % bw)
    if a.check_result:
        from oracle import oracle as O   # checker only
        want = O.embedding_forward(w["table"], w["indices"], w["offsets"] if a.csr_input else None,
                                   w["weights"] if a.weighted_sum else None, batch_size=B, num_hots=hots,
                                   fp16_math=a.fp16_math, threads=O.max_threads())
        assert np.array_equal(out.cpu().numpy().view(np.uint8), want.view(np.uint8))
        print("Check result forward passed")

    if not a.forward_only:
        # ---- transpose (+ compressed remap) -----------------------------------------
        lw = max(ce.transpose_workspace_bytes(nnz, idx_t, elem_t if a.weighted_sum else None),
                 ce.compressed_grad_workspace_bytes(nnz, idx_t), 1)
        work = torch.empty(lw, dtype=torch.uint8, device=dev)
        state = {}

        def transpose():
            if a.csr_input:
                sid = ce.extract_row_ids_from_csr(offsets, nnz=nnz, dtype=idx_t)
            else:
                sid = ce.extract_row_ids_from_fixed(B, H, idx_t, dev)
            t_idx, t_sid, t_w = ce.transpose(sid, indices, weights, workspace=work,
                                             num_categories=a.num_categories if a.bounded_sort else None)
            remap = ce.compute_compressed_grad_indices(t_idx, workspace=work) if a.compressed_grad else None
            state.update(t_idx=t_idx, t_sid=t_sid, t_w=t_w, remap=remap)

        ms = timed(transpose)
        tb = nnz * isz + (nnz * 4 if a.csr_input else 0) + (nnz * es if a.weighted_sum else 0)
        tb += (3 if a.compressed_grad else 2) * nnz * isz + (nnz * es if a.weighted_sum else 0)
        bw = tb * a.iterations / 1e6 / ms
        report("Transpose", ms, 0.0, bw, "Application BW [GB/s]: %.2f" % bw)
        if getattr(a, "one_call_index_work", False) and not a.csr_input:
            # this library's one-call form of the same index work (extension): fixed-hotness row ids are never
            # materialised and the remapped ids come out of the transpose call (one launch up to 4,096 lookups)
            def transpose_one_call():
                ce.transpose_fixed_hotness(indices, B, H, weights, workspace=work,
                                           num_categories=a.num_categories if a.bounded_sort else None,
                                           remapped=bool(a.compressed_grad))

            ms1 = timed(transpose_one_call)
            report("Transpose_one_call", ms1, 0.0, tb * a.iterations / 1e6 / ms1,
                   "Application BW [GB/s]: %.2f (one call: TransposeFixedHotness(..., remapped))" % (tb * a.iterations / 1e6 / ms1))

        # ---- backward ------------------------------------------------------------------
        num_unique = int(state["remap"][-1].item()) + 1 if a.compressed_grad else 0
        grad_rows = num_unique if a.compressed_grad else a.num_categories
        gy = torch.from_numpy(harness.allocate_grad_y(B * W, elem_np).reshape(B, W)).to(dev)
        grad = torch.zeros((grad_rows, W), dtype=elem_t, device=dev)
        inv = torch.empty((grad_rows,), dtype=idx_t, device=dev) if a.compressed_grad else None
        bwd = lambda: ce.embedding_backward(gy, grad_rows, state["t_idx"], state["t_sid"], state["remap"],  # noqa: E731
                                            state["t_w"], skip_grad_init=a.skip_grad_init,
                                            grad_embedding=grad, inverse_mapping=inv)
        ms = timed(bwd)
        uniq = int(torch.unique_consecutive(state["t_idx"]).numel())
        dram = es * W * uniq + isz * nnz * 2 + (es * nnz if a.weighted_sum else 0) + es * W * B
        l2 = dram + es * W * nnz
        bd, bl = dram * a.iterations / 1e6 / ms, l2 * a.iterations / 1e6 / ms
        report("Backward", ms, bl, bd, "Application DRAM BW [GB/s]: %.2f , Application L2 BW [GB/s]: %.2f" % (bd, bl))
        if a.check_result:
            from oracle import oracle as O
            o_sid = O.extract_row_ids_from_csr(w["offsets"], idx_np) if a.csr_input else \
                O.extract_row_ids_from_fixed(B, H, idx_np)
            o_ti, o_ts, o_tw = O.transpose(o_sid, w["indices"], w["weights"] if a.weighted_sum else None)
            assert np.array_equal(state["t_idx"].cpu().numpy(), o_ti) and np.array_equal(state["t_sid"].cpu().numpy(), o_ts)
            print("Check results transpose passed")
            o_remap = O.compute_compressed_grad_indices(o_ti) if a.compressed_grad else None
            grad.zero_()
            bwd()
            o_grad, o_inv = O.embedding_backward(harness.allocate_grad_y(B * W, elem_np).reshape(B, W), W,
                                                 grad_rows, o_ti, o_ts, o_remap, o_tw)
            assert np.array_equal(grad.cpu().numpy().view(np.uint8), o_grad.view(np.uint8))
            print("Check result backward passed")

    if a.enable_csv:
        fname = "manual_benchmark_out.csv"
        new = not os.path.exists(fname)
        with open(fname, "a") as f:
            if new:
                f.write("num_categories,batch_size,hotness,alpha,embed_width,combine_mode,is_csr,is_weighted,"
                        "compressed_grad,skip_grad_init,name,iterations,elapsed_time_ms,avg_time_ms,"
                        "algo_bw_l2,algo_bw_dram\n")
            for name, ms, l2, dram in rows:
                f.write("%d,%d,%d,%g,%d,kSum,%d,%d,%d,%d,%s,%d ,%.2f ,%.4f ,%.2f,%.2f\n"
                        % (a.num_categories, B, H, a.alpha, W, a.csr_input, a.weighted_sum, a.compressed_grad,
                           a.skip_grad_init, name, a.iterations, ms, ms / a.iterations, l2, dram))
    return rows


if __name__ == "__main__":
    main()
