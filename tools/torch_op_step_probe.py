#!/usr/bin/env python3
"""End-to-end time of one training step through the torch op surface (cuemb_embedding forward +
autograd backward) at the C2 shape, int64 indices as the reference's binding requires;
dense gradient (the reference's behaviour) and sparse_grad=True (extension)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cuembed_amd import cuembed_pyt as P
from cuembed_amd import harness

dev = torch.device("cuda", 0)
rows, W, B, H = 10_000_000, 256, 65536, 64
for dtype in (torch.float16, torch.float32):
    table = torch.empty((rows, W), dtype=dtype, device=dev).uniform_(-1, 1).requires_grad_(True)
    idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15).astype(np.int64)).to(dev)
    offsets = torch.arange(0, B * H + 1, H, dtype=torch.int64, device=dev)
    up = torch.ones((B, W), dtype=dtype, device=dev)
    for sparse in (True, False):
        def step():
            table.grad = None
            out = P.cuemb_embedding(table, idx, offsets, None, sparse_grad=sparse)
            out.backward(up)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        n = 10 if sparse else 3
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        print("%s  sparse_grad=%-5s  %.3f ms per fwd+bwd step" % (str(dtype).split(".")[1], sparse,
                                                                  (time.perf_counter() - t0) / n * 1e3))
    del table
    torch.cuda.empty_cache()
