#!/usr/bin/env python3
"""Small batches: the sequential forward kernel against the wide-load kernel (one sample per workgroup, a bag's rows requested
at once, pooled in lookup order out of LDS -- the same bits).  Both forced through set_forward_wide_load; times are HIP-event
times of 200 launches replayed from HIP graphs of 20 (us per launch, device-bound), every wide result is compared bit for bit with the sequential one.
CSV: the data behind ForwardWideLoadPays (kWideLoadBelowWaves, kWideLoadMinHotness)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cuembed_amd as ce
from cuembed_amd import harness

rows = 10_000_000
print("batch,hotness,layout,width,dtype,row_bytes,waves_sequential,sequential_us,wide_us,auto_us,same_bits")
tables = {}
for W, dt in ((32, torch.float32), (128, torch.float32), (256, torch.float16), (64, torch.float16)):
    tables[(W, dt)] = torch.empty((rows, W), dtype=dt, device="cuda").uniform_(-1, 1)
for B in (16, 256, 1024, 2048, 4096, 8192):
    for H in (8, 16, 32, 64, 256):
        for layout in ("fixed", "csr"):
            if layout == "csr" and H not in (16, 64):
                continue
            for (W, dt), table in tables.items():
                if B * H > 1 << 21:
                    continue
                idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.05).astype(np.int32)).cuda().view(-1)
                offsets = None
                if layout == "csr":
                    lens = torch.randint(0, 2 * H + 1, (B,), device="cuda")
                    offsets = torch.zeros(B + 1, dtype=torch.int32, device="cuda")
                    offsets[1:] = torch.cumsum(lens, 0)
                    nnz = int(offsets[-1].item())     # (the same power-law stream as the fixed layout, cut into ragged bags)
                    idx = torch.from_numpy(harness.generate_indices(rows, max(nnz, 1), 1, alpha=1.05).astype(np.int32)).cuda().view(-1)[:nnz]
                outs, times = {}, {}
                for mode in ("never", "always", "auto"):
                    ce.set_forward_wide_load(mode)
                    out = torch.empty((B, W), dtype=dt, device="cuda")

                    def fn():
                        ce.embedding_forward(table, idx, offsets, None, batch_size=B, num_hots=0 if offsets is not None else H,
                                             out=out)
                    for _ in range(3):
                        fn()
                    # (20 launches replayed from a HIP graph: device time, not the ~11 us of host time per wrapper call)
                    side = torch.cuda.Stream()
                    with torch.cuda.stream(side):
                        fn()
                        torch.cuda.current_stream().synchronize()
                        graph = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(graph, stream=side):
                            for _ in range(20):
                                fn()
                    for _ in range(3):
                        graph.replay()
                    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    a.record()
                    for _ in range(10):
                        graph.replay()
                    z.record()
                    z.synchronize()
                    times[mode] = a.elapsed_time(z) / 200 * 1e3
                    outs[mode] = out
                ce.set_forward_wide_load("auto")
                es = table.element_size()
                lanes = W * es // 16
                print("%d,%d,%s,%d,%s,%d,%d,%.2f,%.2f,%.2f,%s" % (
                    B, H, layout, W, str(dt)[6:], W * es, B * lanes // 64, times["never"], times["always"], times["auto"],
                    bool(torch.equal(outs["never"].view(torch.uint8), outs["always"].view(torch.uint8)))), flush=True)
