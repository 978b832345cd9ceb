#!/usr/bin/env python3
"""EmbeddingForward on NARROW rows of a table far larger than the caches, uniform indices (the HBM-bound case), next to
what tools/row_read_ceiling reaches with loads only: fp32 W = 32 / 64 / 128 (128 / 256 / 512-byte rows), 10M rows, the
reference sweep's batch sizes and hotness, both row-load policies; time per call by HIP events, back to back.

    gpurun -- 'tools/row_read_ceiling > gpurun_out/ceil.csv; python tools/narrow_row_probe.py'
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import cuembed_amd as ce
    from cuembed_amd import harness
    dev = torch.device("cuda", 0)
    rows = 10_000_000
    for W in (32, 64, 128):
        table = torch.empty((rows, W), dtype=torch.float32, device=dev).uniform_(-1, 1)
        for B, H in ((131072, 16), (32768, 16), (131072, 64), (32768, 64)):
            idx = [torch.from_numpy(harness.generate_indices(rows, B, H, alpha=0.0, index=np.int32)).to(dev) for _ in range(2)]
            out = torch.empty((B, W), dtype=torch.float32, device=dev)
            res = {"rows": rows, "width": W, "row_bytes": 4 * W, "batch": B, "hotness": H, "alpha": 0}
            for policy in ("default", "streaming"):
                for t in range(3):
                    ce.embedding_forward(table, idx[t % 2], num_hots=H, out=out, row_loads=policy)
                n = 20
                a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for t in range(n):
                    ce.embedding_forward(table, idx[t % 2], num_hots=H, out=out, row_loads=policy)
                z.record()
                z.synchronize()
                ms = a.elapsed_time(z) / n
                alg = 4 * B * (H + 1) * W            # manual_benchmark.cu:256-260
                res[policy] = {"ms": round(ms, 5), "GBps": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / 8000, 4)}
            print(json.dumps(res), flush=True)
        del table
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
