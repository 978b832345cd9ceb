#!/usr/bin/env python3
"""What a consumer pays for the PADDED sparse gradient of the torch op (min(lookups, rows) entries, the tail zero): fwd + bwd of
cuemb_embedding alone, followed by grad.coalesce(), followed by torch.optim.SGD.step(), for sparse_grad "reference" (exactly
num_unique entries, count read back) and True (padded where the limit allows), under the CUEMBED_PYT_PADDED_MB of the
environment (0 = never pad).  fp16, 10M x 256, hotness 64, B = 1,024 / 2,048 / 4,096."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from cuembed_amd import cuembed_pyt as P
from cuembed_amd import harness
rows, W, H = 10_000_000, 256, 64
table = torch.empty((rows, W), dtype=torch.float16, device="cuda").uniform_(-1, 1).requires_grad_(True)
opt = torch.optim.SGD([table], lr=0.01)
print("CUEMBED_PYT_PADDED_MB=%s" % os.environ.get("CUEMBED_PYT_PADDED_MB", "(default)"))
for B in [int(b) for b in sys.argv[1:]] or (1024, 2048, 4096):
    idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15).astype(np.int64)).cuda().view(-1)
    offsets = torch.arange(0, B * H + 1, H, dtype=torch.int64, device="cuda")
    up = torch.randint(-2, 3, (B, W), device="cuda").to(torch.float16)
    for kind in (True, "fastest"):
        for consumer in ("none", "coalesce", "sgd"):
            def step():
                opt.zero_grad(set_to_none=True)
                P.cuemb_embedding(table, idx, offsets, None, sparse_grad=kind).backward(up)
                if consumer == "coalesce":
                    table.grad.coalesce()
                elif consumer == "sgd":
                    opt.step()
            for _ in range(10): step()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(40): step()
            torch.cuda.synchronize()
            print("B=%d kind=%s consumer=%s step_ms=%.4f entries=%d" % (B, kind, consumer, (time.perf_counter()-t0)/40*1e3, table.grad._nnz()), flush=True)
