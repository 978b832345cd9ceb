#!/usr/bin/env python3
"""Counterpart of the reference's benchmarks/sweep_parameters.sh: the same 108-point grid
(alpha x num_categories x embed_width x batch_size x hotness, fp32 tables, int32 indices,
compressed gradient) run in ONE process (tables are reused between points), forward + transpose +
backward per point, results appended to a CSV with the reference's columns plus the time per call.  `transpose` is the
reference's call sequence (row ids, Transpose, ComputeCompressedGradIndices); `transpose_one_call` the same index work
through this library's one-call form (TransposeFixedHotness(..., remapped)).

    python benchmarks/sweep_parameters.py [--iterations 100] [--csv sweep.csv] [--order split]
    python benchmarks/sweep_parameters.py --binary --repetitions 3 --iterations 30 --csv sweep.csv
        (the same grid through the C++ benchmark binary, like the reference's shell script: mean / min / median per
         kernel over 3 independent processes and the kernel's share of the step)
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


def sweep_with_binary(o):
    """The grid through the C++ benchmark (benchmarks/manual_benchmark, the counterpart of the binary the reference's
    sweep_parameters.sh drives): one process per point and repetition, host launch overhead of a C++ program instead of
    Python's (a point of 8 launches of ~4 us each is otherwise HOST-bound here: round 5's CSV showed 75 us for an index
    work that traces at 38 us on the device).  Every iteration is timed alone after a cache flush (reference protocol);
    per point and kernel: mean, min and median over the iterations of `--repetitions` independent processes, and
    share_of_step = this kernel's median / the sum of the three medians."""
    import csv
    import statistics
    import subprocess
    import tempfile
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "manual_benchmark")
    new = not os.path.exists(o.csv)
    with open(o.csv, "a") as f:
        if new:
            f.write("num_categories,batch_size,hotness,alpha,embed_width,name,iterations,repetitions,avg_time_ms,"
                    "min_time_ms,median_time_ms,median_spread_over_repetitions,share_of_step,algo_bw_l2,algo_bw_dram\n")
        for cats, width, alpha, batch, hot in itertools.product(CATEGORIES, WIDTHS, ALPHAS, BATCHES, HOTNESS):
            per = {}
            for rep in range(o.repetitions):
                with tempfile.TemporaryDirectory() as tmp:
                    cmd = [exe, "--num_categories", str(cats), "--embed_width", str(width), "--batch_size", str(batch),
                           "--alpha", str(alpha), "--hotness", str(hot), "--iterations", str(o.iterations),
                           "--compressed_grad=true", "--skip_grad_init=true", "--enable_csv=true",
                           "--clear_caches=" + o.clear_caches]
                    r = subprocess.run(cmd, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
                    if r.returncode != 0:
                        raise RuntimeError("manual_benchmark failed: %s\n%s" % (" ".join(cmd), r.stdout[-2000:]))
                    with open(os.path.join(tmp, "manual_benchmark_out.csv")) as g:
                        for row in csv.DictReader(g):
                            per.setdefault(row["name"], []).append(
                                (float(row["avg_time_ms"]), float(row["min_time_ms"]), float(row["median_time_ms"]),
                                 float(row["algo_bw_l2"]), float(row["algo_bw_dram"])))
            med = {n: statistics.median(x[2] for x in v) for n, v in per.items()}
            step = sum(med.values())
            for name in ("forward", "transpose", "backward"):
                v = per[name]
                f.write("%d,%d,%d,%g,%d,%s,%d,%d,%.5f,%.5f,%.5f,%.3f,%.3f,%.2f,%.2f\n" % (
                    cats, batch, hot, alpha, width, name, o.iterations, o.repetitions, statistics.mean(x[0] for x in v),
                    min(x[1] for x in v), med[name], max(x[2] for x in v) / min(x[2] for x in v), med[name] / step,
                    statistics.mean(x[3] for x in v), statistics.mean(x[4] for x in v)))
            f.flush()
            print("cats=%d w=%d alpha=%g B=%d H=%d : " % (cats, width, alpha, batch, hot) +
                  "  ".join("%s %.4f ms (%.0f %%)" % (n, med[n], 100 * med[n] / step) for n in ("forward", "transpose", "backward")),
                  flush=True)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--iterations", type=int, default=100)   # the reference sweeps with 1000
    p.add_argument("--csv", default="sweep_parameters_out.csv")
    p.add_argument("--order", default="sequential", choices=["sequential", "split"])
    p.add_argument("--forward_only", action="store_true")
    p.add_argument("--clear_caches", default="true")
    p.add_argument("--binary", action="store_true", help="drive benchmarks/manual_benchmark (C++), one process per point")
    p.add_argument("--repetitions", type=int, default=3, help="--binary: independent processes per point")
    o = p.parse_args()
    if o.binary:
        return sweep_with_binary(o)
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
