#!/usr/bin/env python3
"""C3 (fp32 weighted CSR, 10M x 128, batch 65536, bags U[0,128]): what would perfectly balanced bags be worth?
The SAME lookups (indices, weights, table) are pooled (a) as the ragged bags of the recipe, (b) as constant bags of
nnz / B lookups (re-cut offsets: the perfectly balanced case, same memory traffic up to which rows share a sample),
(c) with the bags ordered by length (neighbouring lane groups get similar lengths: an upper bound on what any static
pairing can reach).  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuembed_amd as ce  # noqa: E402
from cuembed_amd import harness  # noqa: E402


def timed(fn, n=50, warm=5):
    for _ in range(warm):
        fn()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    z.record()
    z.synchronize()
    return round(a.elapsed_time(z) / n, 5)


def main():
    rows, W, B, H = 10_000_000, 128, 65536, 128
    dev = torch.device("cuda")
    out = {}
    for alpha in (1.15, 0.0):
        a = harness.allocate_forward(rows, W, 2 * B, H, alpha=alpha, is_csr=True, elem=np.float32, index=np.int32,
                                     with_table=False, consume_table_draws=False)
        off = a["offsets"].astype(np.int64)
        table = torch.empty((rows, W), dtype=torch.float32, device=dev).uniform_(-1, 1)
        res = torch.empty((B, W), dtype=torch.float32, device=dev)
        batches = []
        for t in range(2):
            lo, hi = off[t * B], off[(t + 1) * B]
            batches.append(dict(idx=torch.from_numpy(a["indices"][lo:hi].copy()).to(dev),
                                w=torch.from_numpy(a["weights"][lo:hi].copy()).to(dev),
                                off=torch.from_numpy((off[t * B:(t + 1) * B + 1] - lo).astype(np.int32)).to(dev),
                                nnz=int(hi - lo)))
        e = {"nnz": batches[0]["nnz"]}
        state = {"t": 0}

        def run(key):
            b = batches[state["t"] % 2]
            state["t"] += 1
            ce.embedding_forward(table, b["idx"], b[key], b["w"], batch_size=B, num_hots=0, out=res)

        e["ragged_ms"] = timed(lambda: run("off"))
        # the same batches, untouched, with the scheduling hint: samples handed out by descending bag length
        for b in batches:
            b["order"] = ce.bag_order_by_length(b["off"], max_length=H)

        def run_ordered():
            b = batches[state["t"] % 2]
            state["t"] += 1
            ce.embedding_forward(table, b["idx"], b["off"], b["w"], batch_size=B, num_hots=0, out=res,
                                 sample_order=b["order"])

        e["ragged_with_sample_order_ms"] = timed(run_ordered)
        e["bag_order_by_length_ms"] = timed(lambda: ce.bag_order_by_length(batches[0]["off"], max_length=H))
        for b in batches:      # constant bags over the same lookups: offsets = round(i * nnz / B)
            b["const"] = torch.from_numpy(np.round(np.arange(B + 1) * (b["nnz"] / B)).astype(np.int32)).to(dev)
        e["constant_bags_ms"] = timed(lambda: run("const"))
        for b in batches:      # bags sorted by length: a permutation of the samples (rows of the output), same bags
            o = b["off"].cpu().numpy().astype(np.int64)
            lens = np.diff(o)
            order = np.argsort(lens, kind="stable")
            new_off = np.concatenate([[0], np.cumsum(lens[order])])
            gather = np.concatenate([np.arange(o[s], o[s + 1]) for s in order]) if b["nnz"] else np.zeros(0, np.int64)
            b["idx_sorted"] = b["idx"][torch.from_numpy(gather).to(dev)]
            b["w_sorted"] = b["w"][torch.from_numpy(gather).to(dev)]
            b["off_sorted"] = torch.from_numpy(new_off.astype(np.int32)).to(dev)

        def run_sorted():
            b = batches[state["t"] % 2]
            state["t"] += 1
            ce.embedding_forward(table, b["idx_sorted"], b["off_sorted"], b["w_sorted"], batch_size=B, num_hots=0, out=res)

        e["bags_sorted_by_length_ms"] = timed(run_sorted)
        # (d) the two bags of a wavefront alike, wavefronts in random order: the sorted samples, PAIRS of them shuffled.
        #     Separates "the two halves of a wavefront wait for each other" from "neighbouring wavefronts differ".
        rng = np.random.default_rng(7)
        for b in batches:
            o = b["off"].cpu().numpy().astype(np.int64)
            lens = np.diff(o)
            order = np.argsort(lens, kind="stable").reshape(-1, 2)
            order = order[rng.permutation(order.shape[0])].reshape(-1)
            new_off = np.concatenate([[0], np.cumsum(lens[order])])
            gather = np.concatenate([np.arange(o[s], o[s + 1]) for s in order])
            b["idx_sorted"] = b["idx"][torch.from_numpy(gather).to(dev)]
            b["w_sorted"] = b["w"][torch.from_numpy(gather).to(dev)]
            b["off_sorted"] = torch.from_numpy(new_off.astype(np.int32)).to(dev)
        e["equal_pairs_random_order_ms"] = timed(run_sorted)
        out["alpha_%g" % alpha] = e
        del table
    print(json.dumps(out))


if __name__ == "__main__":
    main()
