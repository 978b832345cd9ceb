#!/usr/bin/env python3
"""EmbeddingBackward at the C4 shape under the launch-shape overrides (cuembed::SetBackwardTuning):
segment length x column slices.  One JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cuembed_amd as ce
from cuembed_amd import harness

dev = torch.device("cuda", 0)
rows, W, B, H = 10_000_000, 256, 65536, 64
idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15)).to(dev)
ti, ts, _ = ce.transpose_fixed_hotness(idx, B, H, num_categories=rows)
remap = ce.compute_compressed_grad_indices(ti)
nu = int(remap[-1].item()) + 1
out = []
for dtype in (torch.float16, torch.float32):
    gy = torch.randint(-3, 4, (B, W), device=dev).to(dtype)
    grad = torch.empty((nu, W), dtype=dtype, device=dev)
    inv = torch.empty((nu,), dtype=torch.int32, device=dev)
    for seg in (0, 16, 32, 64, 128):
        for sl in (0, 1, 2, 4):
            ce.set_backward_tuning(segment_len=seg, column_slices=sl)
            for _ in range(3):
                ce.embedding_backward(gy, nu, ti, ts, remap, grad_embedding=grad, inverse_mapping=inv)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                ce.embedding_backward(gy, nu, ti, ts, remap, grad_embedding=grad, inverse_mapping=inv)
            e.record()
            e.synchronize()
            out.append({"dtype": str(dtype).split(".")[1], "segment_len": seg, "slices": sl,
                        "ms": round(s.elapsed_time(e) / 20, 4)})
ce.set_backward_tuning()
print(json.dumps(out))
