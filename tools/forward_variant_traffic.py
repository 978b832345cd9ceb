#!/usr/bin/env python3
"""Launches a few forward-kernel variants (tools/forward_variants.hip) on the C2 batch, a handful of
times each, so that a rocprofv3 --pmc pass can attribute fabric bytes and L2 hits to each of them:
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/forward_variant_traffic.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cuembed_amd import harness

L = ctypes.CDLL(os.path.join(ROOT, "tools", "forward_variants.so"))
dev = torch.device("cuda", 0)
B, H, W = 65536, 64, 256
alpha = float(os.environ.get("ALPHA", "1.15"))
big = torch.empty((10_000_000, W), dtype=torch.float16, device=dev).uniform_(-1, 1)
idx = harness.generate_indices(10_000_000, 2 * B, H, alpha=alpha).reshape(2, -1)
idxs = [torch.from_numpy(np.ascontiguousarray(idx[i])).to(dev) for i in range(2)]
out = torch.empty((B, W), dtype=torch.float16, device=dev)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for v in [int(x) for x in os.environ.get("VARIANTS", "0,5,10,11").split(",")]:
    for t in range(6):
        L.variant_launch(v, ctypes.c_void_p(big.data_ptr()), W, B, ctypes.c_void_p(idxs[t % 2].data_ptr()), H,
                         ctypes.c_void_p(out.data_ptr()), 8, stream)
    torch.cuda.synchronize()
