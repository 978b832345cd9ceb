#!/usr/bin/env python3
"""The torch op's training step at a launch-bound batch (B = 1024, H = 64, fp16, int64 indices, 10M x 256 table) eagerly
and as a replayed HIP graph (torch.cuda.CUDAGraph over forward + backward with the padded sparse gradient: no read-back,
no host decision inside the step).  One JSON line; the replayed step is checked against the eager one."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cuembed_amd import cuembed_pyt as P
from cuembed_amd import harness

dev = torch.device("cuda", 0)
rows, W, H = 10_000_000, 256, 64
res = {"backend": P.BACKEND}
table = torch.empty((rows, W), dtype=torch.float16, device=dev).uniform_(-1, 1).requires_grad_(True)
for B in (1024, 2048):   # (padded, read-back-free gradients: up to 64 MiB of worst-case gradient rows)
    idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15).astype(np.int64)).to(dev).view(-1)
    offsets = torch.arange(0, B * H + 1, H, dtype=torch.int64, device=dev)
    up = torch.randint(-2, 3, (B, W), device=dev).to(torch.float16)

    def step():
        out = P.cuemb_embedding(table, idx, offsets, None, sparse_grad="padded", hints=None)
        (g,) = torch.autograd.grad(out, table, up)
        return out, g

    def timed(fn, n=200):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return round((time.perf_counter() - t0) / n * 1e3, 4)

    # (everything on ONE side stream, as torch's capture rules ask: an autograd graph first built on the default stream
    # leaves an AccumulateGrad node there, and a capture that has to hop to it dies inside hipStreamEndCapture)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eager_ms = timed(step)
        out_e, g_e = step()
        dense_e = g_e.to_dense() if B == 1024 else None
        torch.cuda.current_stream().synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            out_g, g_g = step()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    same = bool(torch.equal(out_g, out_e))
    if dense_e is not None:
        same = same and bool(torch.equal(g_g.to_dense(), dense_e))
    res["B=%d" % B] = {"eager_ms": eager_ms, "graph_replay_ms": timed(graph.replay), "replay_equals_eager": same,
                        "gradient_entries": int(g_g._indices().shape[1])}
print(json.dumps(res))
