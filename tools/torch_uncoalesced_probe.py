#!/usr/bin/env python3
"""fwd + bwd through cuemb_embedding at the C4 shape: coalesced sparse gradient vs sparse_grad="uncoalesced"."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cuembed_amd import cuembed_pyt as pyt, harness

dev = torch.device("cuda", 0)
rows, W, B, H = 10_000_000, 256, 65536, 64
table = torch.empty((rows, W), dtype=torch.float16, device=dev).uniform_(-1, 1).requires_grad_()
idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15)).to(dev).long()
off = torch.arange(0, B * H + 1, H, device=dev)
up = torch.randn(B, W, device=dev).half()
for kind in (True, "uncoalesced"):
    def step():
        table.grad = None
        pyt.cuemb_embedding(table, idx, off, None, sparse_grad=kind).backward(up)
    for _ in range(3):
        step()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        step()
    z.record()
    z.synchronize()
    print(json.dumps({"sparse_grad": kind, "fwd_bwd_ms": round(a.elapsed_time(z) / 20, 4), "gradient_rows": table.grad._nnz()}))
