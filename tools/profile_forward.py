#!/usr/bin/env python3
"""Launches EmbeddingForward a few times on one index pattern, for rocprofv3 passes.

    rocprofv3 --kernel-trace --stats -f csv -d OUT -- python3 tools/profile_forward.py --pattern powerlaw
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -f csv -d OUT -- python3 tools/profile_forward.py ...

patterns:  powerlaw (alpha 1.15, the C2 workload) | uniform (alpha 0) |
           unique   (a random permutation of B*H distinct rows: every row is read exactly once,
                     so the HBM read volume is known = B*H*row_bytes -- calibrates FETCH_SIZE)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--pattern", default="powerlaw", choices=["powerlaw", "uniform", "unique", "all"],
                   help="all = --iters launches of powerlaw, then of uniform, then of unique (one process, so "
                        "one rocprofv3 pass covers the three; tools/traffic_from_pmc.py splits them by order)")
    p.add_argument("--iters", type=int, default=10)
    p.add_argument("--rows", type=int, default=10_000_000)
    p.add_argument("--width", type=int, default=256)
    p.add_argument("--batch", type=int, default=65536)
    p.add_argument("--hotness", type=int, default=64)
    p.add_argument("--elem", default="f16", choices=["f16", "f32"])
    p.add_argument("--flush", action="store_true", help="1 GB cache-flush reduction between launches")
    a = p.parse_args()
    import numpy as np
    import torch
    import cuembed_amd as ce
    from cuembed_amd import harness
    dev = torch.device("cuda", 0)
    dt = torch.float16 if a.elem == "f16" else torch.float32
    table = torch.empty((a.rows, a.width), dtype=dt, device=dev)
    table.uniform_(-1, 1)
    nb = 2

    def make(pattern):
        if pattern == "unique":
            g = torch.Generator(device=dev).manual_seed(3)
            return [torch.randperm(a.rows, device=dev, generator=g)[: a.batch * a.hotness].to(torch.int32)
                    for _ in range(nb)]
        alpha = 1.15 if pattern == "powerlaw" else 0.0
        idx = harness.generate_indices(a.rows, nb * a.batch, a.hotness, alpha=alpha).reshape(nb, -1)
        return [torch.from_numpy(np.ascontiguousarray(idx[i])).to(dev) for i in range(nb)]

    patterns = ["powerlaw", "uniform", "unique"] if a.pattern == "all" else [a.pattern]
    sets = [make(pt) for pt in patterns]
    out = torch.empty((a.batch, a.width), dtype=dt, device=dev)
    flush = torch.ones(256_000_000, dtype=torch.int32, device=dev) if a.flush else None
    sink = torch.zeros((), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for batches in sets:
        for it in range(a.iters):
            if flush is not None:
                sink.add_(flush.max())
            ce.embedding_forward(table, batches[it % nb], num_hots=a.hotness, out=out)
        torch.cuda.synchronize()
    es = 2 if a.elem == "f16" else 4
    print("pattern=%s launches=%d algorithmic_bytes_per_launch=%d row_read_bytes=%d out_bytes=%d index_bytes=%d"
          % (a.pattern, a.iters, es * a.batch * (a.hotness + 1) * a.width,
             es * a.batch * a.hotness * a.width, es * a.batch * a.width, 4 * a.batch * a.hotness))


if __name__ == "__main__":
    main()
