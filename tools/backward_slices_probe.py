#!/usr/bin/env python3
"""EmbeddingBackward (compressed gradient, reference order) at the sweep grid's mid sizes -- 65 k to 786 k lookups -- with the
column slices forced to 1 / 2 / 4 / 8 (SetBackwardTuning) and as the heuristic picks them ("auto"): the data behind
ChooseColumnSlices' rule for fewer than 2^20 lookups.  us per call through the Python wrapper (~14 us of host time per call:
everything below that is the wrapper, not the kernel).  One line per shape."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cuembed_amd as ce
from cuembed_amd import harness

shapes = []
for B, H in [(4096, 16), (8192, 16), (16384, 8), (32768, 4), (131072, 1), (2048, 64), (4096, 32), (16384, 16), (4096, 64),
             (262144, 1), (8192, 32), (32768, 16), (8192, 64), (4096, 128), (2048, 256), (16384, 32), (65536, 8), (65536, 4),
             (49152, 16)]:
    for W, dt in [(128, torch.float32), (256, torch.float16), (256, torch.float32), (64, torch.float32)]:
        for alpha in (1.05, 0.0):
            shapes.append((10_000_000, B, H, W, dt, alpha))
print("rows,batch,hotness,lookups,width,dtype,row_bytes,alpha,auto_us,slices1_us,slices2_us,slices4_us,slices8_us,auto_slices")
for (rows, B, H, W, dt, alpha) in shapes:
    idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=alpha).astype(np.int32)).cuda()
    gy = torch.randint(-3, 4, (B, W), device="cuda").to(dt)
    ti, ts, _, rm = ce.transpose_fixed_hotness(idx.view(-1), B, H, None, num_categories=rows, remapped=True)
    nu = int(rm[-1].item()) + 1
    grad = torch.empty((nu, W), dtype=dt, device="cuda")
    inv = torch.empty((nu,), dtype=torch.int32, device="cuda")
    row_bytes = W * grad.element_size()
    out = []
    for slices in (0, 1, 2, 4, 8):
        if slices > 1 and row_bytes // slices < 64:
            out.append("")
            continue
        ce.set_backward_tuning(0, slices)

        def fn():
            ce.embedding_backward(gy, nu, ti, ts, rm, grad_embedding=grad, inverse_mapping=inv)
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
        out.append("%.2f" % ((time.perf_counter() - t0) / 100 * 1e6))
    ce.set_backward_tuning(0, 0)
    shape = ce.backward_launch_shape(dt, torch.int32, W, B * H)
    print("%d,%d,%d,%d,%d,%s,%d,%.2f,%s,%d" % (rows, B, H, B * H, W, str(dt)[6:], row_bytes, alpha, ",".join(out),
                                              shape["column_slices"]), flush=True)
