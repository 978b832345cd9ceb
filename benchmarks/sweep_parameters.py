#!/usr/bin/env python3
"""Counterpart of the reference's benchmarks/sweep_parameters.sh: the same 108-point grid
(alpha x num_categories x embed_width x batch_size x hotness, fp32 tables, int32 indices,
compressed gradient) run in ONE process (tables are reused between points), forward + transpose +
backward per point, results appended to a CSV with the reference's columns plus the time per call.  `transpose` is the
reference's call sequence (row ids, Transpose, ComputeCompressedGradIndices); `transpose_one_call` the same index work
through this library's one-call form (TransposeFixedHotness(..., remapped)).

    python benchmarks/sweep_parameters.py [--iterations 100] [--csv sweep.csv] [--order split]
"""
import argparse
import itertools
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ALPHAS = [0, 1.05, 1.15]
CATEGORIES = [1000000, 10000000]
WIDTHS = [32, 128]
BATCHES = [1024, 32768, 131072]
HOTNESS = [1, 16, 64]


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--iterations", type=int, default=100)   # the reference sweeps with 1000
    p.add_argument("--csv", default="sweep_parameters_out.csv")
    p.add_argument("--order", default="sequential", choices=["sequential", "split"])
    p.add_argument("--forward_only", action="store_true")
    p.add_argument("--clear_caches", default="true")
    o = p.parse_args()
    import cuembed_amd as ce
    import manual_benchmark as mb
    ce.set_forward_reduction_order(o.order)
    cache = {}
    new = not os.path.exists(o.csv)
    with open(o.csv, "a") as f:
        if new:
            f.write("num_categories,batch_size,hotness,alpha,embed_width,forward_order,name,iterations,"
                    "avg_time_ms,algo_bw_l2,algo_bw_dram\n")
        # tables are the expensive part: iterate so that (categories, width) changes slowest
        for cats, width, alpha, batch, hot in itertools.product(CATEGORIES, WIDTHS, ALPHAS, BATCHES, HOTNESS):
            a = types.SimpleNamespace(
                num_categories=cats, embed_width=width, batch_size=batch, hotness=hot,
                iterations=o.iterations, alpha=float(alpha), use_int64_indices=False, check_result=False,
                half_embedding_type=False, csr_input=False, weighted_sum=False, fp16_math=False,
                compressed_grad=True, skip_grad_init=True, forward_only=o.forward_only, enable_csv=False,
                clear_caches=mb.str2bool(o.clear_caches), device_table_fill=True, bounded_sort=False,
                one_call_index_work=True)
            rows = mb.run(a, table_cache=cache, quiet=True)
            for name, ms, l2, dram in rows:
                f.write("%d,%d,%d,%g,%d,%s,%s,%d,%.5f,%.2f,%.2f\n" % (cats, batch, hot, alpha, width, o.order,
                                                                      name, o.iterations, ms / o.iterations, l2, dram))
            f.flush()
            print("cats=%d w=%d alpha=%g B=%d H=%d : " % (cats, width, alpha, batch, hot) +
                  "  ".join("%s %.4f ms" % (n, ms / o.iterations) for n, ms, _, _ in rows), flush=True)


if __name__ == "__main__":
    main()
