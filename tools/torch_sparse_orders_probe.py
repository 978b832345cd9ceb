#!/usr/bin/env python3
"""fwd + bwd through the torch op surface (cuemb_embedding, int64 indices, fp16, B = 65536, H = 64, 10M x 256 table):
the ways of producing the sparse gradient -- sparse_grad=True (the coalesced tensor of the reference order), "fastest" (the fastest order for the shape),
"reference" (fully sorted, coalesced), "blocked" (coalesced, from the sample-blocked order), "uncoalesced".  One JSON line."""
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
rows, W, H, B = 10_000_000, 256, 64, 65536
table = torch.empty((rows, W), dtype=torch.float16, device=dev).uniform_(-1, 1).requires_grad_(True)
idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15).astype(np.int64)).to(dev)
offsets = torch.arange(0, B * H + 1, H, dtype=torch.int64, device=dev)
up = torch.randint(-2, 3, (B, W), device=dev).to(torch.float16)
res = {"backend": P.BACKEND}
grads = {}
for kind in (True, "fastest", "blocked", "uncoalesced"):
    def step():
        table.grad = None
        P.cuemb_embedding(table, idx, offsets, None, sparse_grad=kind).backward(up)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    res["sparse_grad=%s" % kind] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
    g = table.grad
    grads[kind] = (g._indices().clone(), g._values().clone(), g.is_coalesced())
a, b = grads[True], grads["blocked"]
res["blocked_coalesced_same_ids"] = bool(torch.equal(a[0], b[0]))
res["blocked_coalesced_max_abs_diff"] = float((a[1].float() - b[1].float()).abs().max().item())
res["fastest_rows"] = int(grads["fastest"][0].shape[1])
res["reference_rows"] = int(a[0].shape[1])
print(json.dumps(res))
