#!/usr/bin/env python3
"""End-to-end time of one training step through the torch op surface (cuemb_embedding forward +
autograd backward), int64 indices as the reference's binding requires; dense gradient (the
reference's behaviour) and sparse_grad="fastest" (extension: compressed rows as a sparse COO tensor, the fastest order for the shape).

    python tools/torch_op_step_probe.py                       # native binding (libcuembed_pyt.so)
    CUEMBED_PYT_BACKEND=python python tools/torch_op_step_probe.py   # same ops registered from Python (ctypes)

Shapes: the C2/C4 shape (batch 65536) and a launch-bound one (batch 1024), table 10M x 256.
Prints one JSON line per (dtype, batch, gradient kind)."""
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
for dtype in (torch.float16, torch.float32):
    table = torch.empty((rows, W), dtype=dtype, device=dev).uniform_(-1, 1).requires_grad_(True)
    for B in (1024, 65536):
        idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15).astype(np.int64)).to(dev)
        offsets = torch.arange(0, B * H + 1, H, dtype=torch.int64, device=dev)
        up = torch.ones((B, W), dtype=dtype, device=dev)
        for kind in ("sparse", "dense", "fixed_layout_dense"):
            def step():
                table.grad = None
                if kind == "fixed_layout_dense":
                    out = P.cuemb_embedding_fixed(table, idx.view(B, H))
                else:
                    out = P.cuemb_embedding(table, idx, offsets, None, sparse_grad=("fastest" if kind == "sparse" else False))
                out.backward(up)
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            n = 20 if kind == "sparse" else 4
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / n * 1e3
            # forward only, no autograd: the binding's own overhead shows at batch 1024
            with torch.no_grad():
                for _ in range(5):
                    P.cuemb_embedding(table, idx, offsets)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    P.cuemb_embedding(table, idx, offsets)
                torch.cuda.synchronize()
                fwd_ms = (time.perf_counter() - t0) / 200 * 1e3
            print(json.dumps({"backend": P.BACKEND, "dtype": str(dtype).split(".")[1], "batch": B, "hotness": H,
                              "gradient": kind, "fwd_bwd_ms": round(ms, 4), "inference_fwd_ms": round(fwd_ms, 4)}),
                  flush=True)
        table.grad = None
    del table
    torch.cuda.empty_cache()
