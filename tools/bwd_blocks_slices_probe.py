#!/usr/bin/env python3
"""EmbeddingBackward (compressed, C4) over sample blocks P x column slices S: the L2 of an XCD fronts
(batch / P) x (row bytes / S) of grad_y; fewer slices = the COO is staged fewer times and the per-lookup bookkeeping
is paid per 512 / S bytes, more blocks = more duplicate gradient rows."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cuembed_amd as ce
    from cuembed_amd import harness
    dev = torch.device("cuda", 0)
    rows, W, B, H = 10_000_000, 256, 65536, 64
    idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15)).to(dev)
    gy = torch.randint(-3, 4, (B, W), device=dev).half()
    out = {}
    for P in (1, 2, 4, 8, 16):
        ti, ts, _ = ce.transpose_fixed_hotness(idx, B, H, num_categories=rows, sample_blocks=P)
        remap = ce.compute_compressed_grad_indices(ti)
        nu = int(remap[-1].item()) + 1
        grad = torch.empty((nu, W), dtype=torch.float16, device=dev)
        inv = torch.empty((nu,), dtype=torch.int32, device=dev)
        for S in (1, 2, 4):
            ce.set_backward_tuning(column_slices=S)
            for _ in range(3):
                ce.embedding_backward(gy, nu, ti, ts, remap, grad_embedding=grad, inverse_mapping=inv)
            a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(30):
                ce.embedding_backward(gy, nu, ti, ts, remap, grad_embedding=grad, inverse_mapping=inv)
            z.record()
            z.synchronize()
            out["blocks_%d_slices_%d" % (P, S)] = {"ms": round(a.elapsed_time(z) / 30, 4), "rows": nu,
                                                    "MB_per_L2": round(B / P * 512 / S / 1e6, 2)}
    ce.set_backward_tuning()
    for k, v in out.items():
        print(k, json.dumps(v))


if __name__ == "__main__":
    main()
