#!/usr/bin/env python3
"""EmbeddingBackward (compressed) on a COO that was transposed in P blocks of samples -- each block of the
input sorted by row on its own, the sorted blocks concatenated -- against the fully sorted COO, C4 shape.
A row looked up from several blocks then appears once per block in the compressed gradient (an uncoalesced
sparse gradient), but while a block is being processed every L2 gathers from 1 / P of grad_y only.

    gpurun -- 'python tools/bwd_sample_blocks_probe.py'
"""
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
    sid = ce.extract_row_ids_from_fixed(B, H, torch.int32, dev)
    gy = torch.randint(-3, 4, (B, W), device=dev).half()
    nnz = B * H
    ref = None
    out = {}
    for P in (1, 2, 4, 8):
        tis, tss = [], []
        for p in range(P):
            lo, hi = nnz * p // P, nnz * (p + 1) // P
            ti, ts, _ = ce.transpose(sid[lo:hi].contiguous(), idx[lo:hi].contiguous(), num_categories=rows)
            tis.append(ti)
            tss.append(ts)
        ti, ts = torch.cat(tis), torch.cat(tss)
        remap = ce.compute_compressed_grad_indices(ti)
        nu = int(remap[-1].item()) + 1
        grad = torch.empty((nu, W), dtype=torch.float16, device=dev)
        inv = torch.empty((nu,), dtype=torch.int32, device=dev)
        for _ in range(3):
            ce.embedding_backward(gy, nu, ti, ts, remap, grad_embedding=grad, inverse_mapping=inv)
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(30):
            ce.embedding_backward(gy, nu, ti, ts, remap, grad_embedding=grad, inverse_mapping=inv)
        z.record()
        z.synchronize()
        dense = torch.zeros((200_000, W), dtype=torch.float32, device=dev)     # same gradient? (first 200k table rows)
        keep = inv < 200_000
        dense.index_add_(0, inv[keep].long(), grad[keep].float())
        if ref is None:
            ref = dense
        out["sample_blocks_%d" % P] = {"ms": round(a.elapsed_time(z) / 30, 4), "compressed_rows": nu,
                                       "equals_fully_sorted": bool(torch.equal(dense, ref))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
