#!/usr/bin/env python3
"""How much of EmbeddingBackward's time at the C4 shape belongs to the few very long runs?

Times the compressed backward (a) on the full sorted COO and (b) on the COO with every run longer
than --threshold removed (the time a run-aware split would leave to the segmented kernel), and
prints the run-length statistics.  Uses only the public ops.

    python tools/hot_run_probe.py [--threshold 8192] [--alpha 1.15]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--threshold", type=int, nargs="+", default=[2048, 4096, 8192, 16384])
    p.add_argument("--stride", type=int, nargs="+", default=[-1, 1024, 2048, 4096, 8192, 16384],
                   help="hot-run detection strides to time the run-aware backward with (-1: hot path off)")
    p.add_argument("--alpha", type=float, default=1.15)
    p.add_argument("--batch", type=int, default=65536)
    p.add_argument("--hotness", type=int, default=64)
    p.add_argument("--width", type=int, default=256)
    p.add_argument("--rows", type=int, default=10_000_000)
    a = p.parse_args()
    import numpy as np
    import torch
    import cuembed_amd as ce
    from cuembed_amd import harness
    dev = torch.device("cuda", 0)
    B, H, W = a.batch, a.hotness, a.width
    idx = torch.from_numpy(harness.generate_indices(a.rows, B, H, alpha=a.alpha)).to(dev)
    sid = ce.extract_row_ids_from_fixed(B, H, torch.int32, dev)
    ti, ts, _ = ce.transpose(sid, idx, num_categories=a.rows)
    gy = torch.randint(-10, 11, (B, W), device=dev).to(torch.float16)

    def timed(ti, ts, n=30):
        remap = ce.compute_compressed_grad_indices(ti)
        nu = int(remap[-1].item()) + 1
        grad = torch.empty((nu, W), dtype=torch.float16, device=dev)
        inv = torch.empty((nu,), dtype=torch.int32, device=dev)
        for _ in range(5):
            ce.embedding_backward(gy, nu, ti, ts, remap, grad_embedding=grad, inverse_mapping=inv)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            ce.embedding_backward(gy, nu, ti, ts, remap, grad_embedding=grad, inverse_mapping=inv)
        e.record()
        e.synchronize()
        return s.elapsed_time(e) / n, nu

    full_ms, nu = timed(ti, ts)
    uniq, counts = torch.unique_consecutive(ti, return_counts=True)
    per_lookup = torch.repeat_interleave(counts, counts)
    out = {"nnz": int(ti.numel()), "unique_rows": nu, "full_ms": round(full_ms, 5), "longest_run": int(counts.max()),
           "thresholds": []}
    for t in a.threshold:
        keep = per_lookup <= t
        ms, _ = timed(ti[keep].contiguous(), ts[keep].contiguous())
        out["thresholds"].append({"threshold": t, "hot_runs": int((counts > t).sum()),
                                  "hot_lookups": int((~keep).sum()),
                                  "hot_lookup_fraction": round(float((~keep).float().mean()), 4),
                                  "backward_ms_without_hot_runs": round(ms, 5)})
    # What would perfect L2 locality of the gathers buy?  The same COO with every sample id folded into
    # the first 2048 grad_y rows (1 MB: always L2-resident): everything but the gather misses stays.
    folded = (ts % 2048).contiguous()
    ms, _ = timed(ti, folded)
    out["full_with_l2_resident_grad_y_ms"] = round(ms, 5)
    keep = per_lookup <= 8192
    ms, _ = timed(ti[keep].contiguous(), folded[keep].contiguous())
    out["cold_part_with_l2_resident_grad_y_ms"] = round(ms, 5)
    out["plain_by_column_slices"] = []
    for sl in (1, 2, 4, 8):
        ce.set_backward_tuning(column_slices=sl)
        ms, _ = timed(ti, ts)
        out["plain_by_column_slices"].append({"slices": sl, "ms": round(ms, 5)})
    ce.set_backward_tuning()
    # the run-aware entry point itself, per detection stride
    remap = ce.compute_compressed_grad_indices(ti)
    grad = torch.empty((nu, W), dtype=torch.float16, device=dev)
    inv = torch.empty((nu,), dtype=torch.int32, device=dev)
    ws = torch.empty((ce.backward_workspace_bytes(torch.float16, torch.int32, W, ti.numel(), B),), dtype=torch.uint8,
                     device=dev)
    want, _ = ce.embedding_backward(gy, nu, ti, ts, remap)
    out["run_aware"] = []
    for stride in a.stride:
        ce.set_backward_tuning(hot_stride=stride)

        def call():
            ce.embedding_backward(gy, nu, ti, ts, remap, grad_embedding=grad, inverse_mapping=inv, run_aware=True,
                                  workspace=ws)
        for _ in range(5):
            call()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30):
            call()
        e.record()
        e.synchronize()
        out["run_aware"].append({"hot_stride": stride, "ms": round(s.elapsed_time(e) / 30, 5),
                                 "max_abs_diff_vs_plain": float((grad.float() - want.float()).abs().max())})
    ce.set_backward_tuning()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
