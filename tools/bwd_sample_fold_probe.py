#!/usr/bin/env python3
"""Upper bound of any scheme that shrinks the set of grad_y rows an L2 gathers from: the SAME
kernel on the SAME sorted COO at the C4 shape, with the sample ids folded into the first B / f rows of grad_y
(f = 1: the real ids).  f = 2 is what a split of every XCD pair by sample half could reach at best for its
gathers (4.2 MB of column slice per 4 MiB L2) -- before paying for the second staging of the COO and for
combining the two halves of ~60 % of the rows through memory.

    gpurun -- 'python tools/bwd_sample_fold_probe.py'            # times
    rocprofv3 --pmc FETCH_SIZE ... -- python3 tools/bwd_sample_fold_probe.py --launches 3   # bytes per fold
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cuembed_amd as ce
    from cuembed_amd import harness
    launches = int(sys.argv[sys.argv.index("--launches") + 1]) if "--launches" in sys.argv else 30
    dev = torch.device("cuda", 0)
    rows, W, B, H = 10_000_000, 256, 65536, 64
    idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15)).to(dev)
    ti, ts, _ = ce.transpose_fixed_hotness(idx, B, H, num_categories=rows)
    remap = ce.compute_compressed_grad_indices(ti)
    nu = int(remap[-1].item()) + 1
    gy = torch.randint(-3, 4, (B, W), device=dev).half()
    grad = torch.empty((nu, W), dtype=torch.float16, device=dev)
    inv = torch.empty((nu,), dtype=torch.int32, device=dev)
    out = {}
    for fold in (1, 2, 4, 8, 32):
        sid = (ts % (B // fold)).contiguous()
        for _ in range(3):
            ce.embedding_backward(gy, nu, ti, sid, remap, grad_embedding=grad, inverse_mapping=inv)
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(launches):
            ce.embedding_backward(gy, nu, ti, sid, remap, grad_embedding=grad, inverse_mapping=inv)
        z.record()
        z.synchronize()
        out["grad_y_rows_gathered_from_%d" % (B // fold)] = {
            "ms": round(a.elapsed_time(z) / launches, 4), "column_slice_MB_per_L2": round(B // fold * 128 / 1e6, 2)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
