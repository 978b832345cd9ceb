#!/usr/bin/env python3
"""Where does the host time of one small training step through cuemb_embedding go?  torch.profiler over 20 steps at
B = 1024, H = 64 (fp16, int64 indices, sparse gradient); prints the top CPU-side entries and the wall time per step."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cuembed_amd import cuembed_pyt as P
from cuembed_amd import harness

dev = torch.device("cuda", 0)
rows, W, H, B = 10_000_000, 256, 64, int(os.environ.get("B", 1024))
dtype = torch.float16 if os.environ.get("DTYPE", "f16") == "f16" else torch.float32
table = torch.empty((rows, W), dtype=dtype, device=dev).uniform_(-1, 1).requires_grad_(True)
idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15).astype(np.int64)).to(dev)
offsets = torch.arange(0, B * H + 1, H, dtype=torch.int64, device=dev)
up = torch.ones((B, W), dtype=dtype, device=dev)


def step():
    table.grad = None
    out = P.cuemb_embedding(table, idx, offsets, None, sparse_grad="fastest")
    out.backward(up)


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    step()
torch.cuda.synchronize()
print("wall per step: %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(20):
        step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=30, max_name_column_width=60))
