#!/usr/bin/env python3
"""Small-batch training step (forward + row ids + Transpose + dense backward): stream launches
from Python vs replay of one captured HIP graph.  Small batches are launch-bound."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cuembed_amd as ce
from cuembed_amd import harness

dev = torch.device("cuda", 0)
for (ncat, W, B, H) in [(1_000_000, 128, 1024, 1), (1_000_000, 128, 1024, 16), (1_000_000, 128, 1024, 64),
                        (1_000_000, 128, 32768, 16)]:
    table = torch.empty((ncat, W), device=dev).uniform_(-1, 1)
    idx = torch.from_numpy(harness.generate_indices(ncat, B, H, alpha=1.15)).to(dev)
    gy = torch.randn((B, W), device=dev)
    nnz = B * H
    out = torch.empty((B, W), device=dev)
    grad = torch.zeros((nnz, W), device=dev)          # compressed gradient, sized for the worst case
    inv = torch.empty((nnz,), dtype=torch.int32, device=dev)
    work = torch.empty(max(ce.transpose_workspace_bytes(nnz, torch.int32), 1), dtype=torch.uint8, device=dev)

    def step():
        ce.embedding_forward(table, idx, num_hots=H, out=out)
        sid = ce.extract_row_ids_from_fixed(B, H, torch.int32, dev)
        t_idx, t_sid, _ = ce.transpose(sid, idx, workspace=work, num_categories=ncat)
        remap = ce.compute_compressed_grad_indices(t_idx)
        ce.embedding_backward(gy, nnz, t_idx, t_sid, remap, grad_embedding=grad, inverse_mapping=inv)

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
        s.synchronize()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        s.synchronize()
        t_stream = (time.perf_counter() - t0) / n * 1e3
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
        for _ in range(3):
            g.replay()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            g.replay()
        s.synchronize()
        t_graph = (time.perf_counter() - t0) / n * 1e3
    print("B=%6d H=%3d: stream launches %.4f ms/step, graph replay %.4f ms/step" % (B, H, t_stream, t_graph))
