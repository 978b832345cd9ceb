#!/usr/bin/env python3
"""Where does RowLoadPolicy::kStreaming start to pay?  EmbeddingForward at the C2 shape (10M x 256 fp16, 65,536 x 64) over
a grid of power-law exponents, both row-load policies, next to the statistics a decision could be taken from: the
distinct fraction of an evenly strided sample of 65,536 lookups (what cuembed_amd.policy reads back today) and the mean
distinct fraction inside 16 strided groups of 4,096 lookups (what one workgroup can count exactly in LDS: the
device-side decision of cuembed_decide_row_loads).  One JSON line per exponent."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cuembed_amd as ce
from cuembed_amd import harness

dev = torch.device("cuda", 0)
rows_list = [int(x) for x in os.environ.get("PROBE_ROWS", "10000000").split(",")]
B, H, W = 65536, 64, 256
out = torch.empty((B, W), dtype=torch.float16, device=dev)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    z.record()
    z.synchronize()
    return a.elapsed_time(z) / n


for rows in rows_list:
    table = torch.empty((rows, W), dtype=torch.float16, device=dev).uniform_(-1, 1)
    for alpha in (0.0, 0.25, 0.5, 0.75, 0.9, 0.99, 1.05, 1.1, 1.15):    # (alpha = 1 is singular in the generator)
        idx = harness.generate_indices(rows, 2 * B, H, alpha=alpha).reshape(2, -1)
        d = [torch.from_numpy(np.ascontiguousarray(idx[i])).to(dev) for i in range(2)]
        it = [0]

        def run(policy):
            it[0] += 1
            ce.embedding_forward(table, d[it[0] % 2], num_hots=H, out=out, row_loads=policy)

        ms_default = timed(lambda: run("default"))
        ms_stream = timed(lambda: run("streaming"))
        flat = idx[0]
        n = flat.size
        s64 = flat[:: n // 65536][:65536]
        groups = s64.reshape(4096, 16).T          # group g = every 16th element of the sample: strided, 4,096 each
        print(json.dumps({
            "rows": rows, "alpha": alpha, "default_ms": round(ms_default, 4), "streaming_ms": round(ms_stream, 4),
            "streaming_over_default": round(ms_stream / ms_default, 4),
            "batch_distinct_fraction": round(np.unique(flat).size / n, 4),
            "sample_65536_distinct_fraction": round(np.unique(s64).size / 65536, 4),
            "groups_of_4096_mean_distinct_fraction": round(float(np.mean([np.unique(g).size for g in groups])) / 4096, 4),
        }), flush=True)
    del table
