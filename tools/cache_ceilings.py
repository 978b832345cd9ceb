#!/usr/bin/env python3
"""Where does the gather saturate?  Runs EmbeddingForward (fp16, W=256, B=65536, H=64, uniform
random indices) over tables of growing size, so that the rows come from L2 (table << 4 MiB per
XCD), from the Infinity Cache (table << 256 MiB) or from HBM.  The resulting algorithmic GB/s are
the ceilings of THIS access pattern (random 512-byte rows) at each level of the memory hierarchy,
which is what the forward / backward kernels should be judged against when their working set
lives at that level."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    import cuembed_amd as ce
    dev = torch.device("cuda", 0)
    B, H, W = 65536, 64, 256
    out = torch.empty((B, W), dtype=torch.float16, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    nbytes = 2 * B * (H + 1) * W
    print("rows, table_MiB, ms, algorithmic_GBps")
    for rows in [512, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 1 << 20, 4 << 20, 10_000_000]:
        table = torch.empty((rows, W), dtype=torch.float16, device=dev).uniform_(-1, 1)
        idx = [torch.randint(0, rows, (B * H,), device=dev, dtype=torch.int32, generator=g) for _ in range(2)]
        for t in range(3):
            ce.embedding_forward(table, idx[t % 2], num_hots=H, out=out)
        n = 20
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for t in range(n):
            ce.embedding_forward(table, idx[t % 2], num_hots=H, out=out)
        z.record()
        z.synchronize()
        ms = a.elapsed_time(z) / n
        print("%9d, %9.2f, %.4f, %.0f" % (rows, rows * W * 2 / 2 ** 20, ms, nbytes / ms / 1e6), flush=True)
        del table


if __name__ == "__main__":
    main()
