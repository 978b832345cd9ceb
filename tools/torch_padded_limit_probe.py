#!/usr/bin/env python3
"""Up to which worst-case gradient size should the torch op hand on a PADDED sparse gradient (min(lookups, rows) entries, zero
rows past the device-side count, no read-back) instead of reading the row count back?  fwd + bwd of cuemb_embedding (fp16, 10M x
256, hotness 64, int64 indices) at B = 2,048 ... 32,768 for sparse_grad True (always reads the count) and "fastest", under
the CUEMBED_PYT_PADDED_MB of the environment (read once per process: run it once per limit).  One line per (batch, kind)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cuembed_amd import cuembed_pyt as P
from cuembed_amd import harness

rows, W, H = 10_000_000, 256, 64
table = torch.empty((rows, W), dtype=torch.float16, device="cuda").uniform_(-1, 1).requires_grad_(True)
print("CUEMBED_PYT_PADDED_MB=%s" % os.environ.get("CUEMBED_PYT_PADDED_MB", "(default)"))
for B in (2048, 4096, 8192, 16384, 32768):
    idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15).astype(np.int64)).cuda().view(-1)
    offsets = torch.arange(0, B * H + 1, H, dtype=torch.int64, device="cuda")
    up = torch.randint(-2, 3, (B, W), device="cuda").to(torch.float16)
    for kind in (True, "fastest"):
        def step():
            table.grad = None
            P.cuemb_embedding(table, idx, offsets, None, sparse_grad=kind).backward(up)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            step()
        torch.cuda.synchronize()
        print("B=%d worst_case_MB=%d sparse_grad=%s step_ms=%.4f gradient_entries=%d" % (
            B, min(B * H, rows) * W * 2 >> 20, kind, (time.perf_counter() - t0) / 50 * 1e3, table.grad._nnz()), flush=True)
