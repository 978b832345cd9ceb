#!/usr/bin/env python3
"""CSR bags at 1,024 - 32,768 samples: the sequential forward kernel against the wide-load kernel with 1 / 2 / 4 / 8 / 16 samples
per workgroup (forced through set_forward_wide_load) and as the launcher picks ("auto").  Ragged bags U[0, 2H], power-law
indices, device times from graph replays (us per launch); every result compared with the sequential kernel's bits."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cuembed_amd as ce
from cuembed_amd import harness

rows = 10_000_000
MODES = ("never", "always", "always2", "always4", "always8", "always16", "auto")
print("batch,mean_hotness,width,dtype,row_bytes," + ",".join(m + "_us" for m in MODES) + ",same_bits,auto_samples_per_workgroup")
tables = {}
for W, dt in ((8, torch.float32), (16, torch.float32), (32, torch.float32), (64, torch.float16), (64, torch.float32),
              (128, torch.float32), (256, torch.float16)):
    tables[(W, dt)] = torch.empty((rows, W), dtype=dt, device="cuda").uniform_(-1, 1)
for B in (1024, 2048, 4096, 8192, 16384, 32768):
    for H in (16, 64):
        lens = torch.randint(0, 2 * H + 1, (B,), device="cuda")
        offsets = torch.zeros(B + 1, dtype=torch.int32, device="cuda")
        offsets[1:] = torch.cumsum(lens, 0)
        nnz = int(offsets[-1].item())
        idx = torch.from_numpy(harness.generate_indices(rows, nnz, 1, alpha=1.05).astype(np.int32)).cuda().view(-1)[:nnz]
        for (W, dt), table in tables.items():
            outs, times = {}, {}
            for mode in MODES:
                ce.set_forward_wide_load(mode)
                out = torch.empty((B, W), dtype=dt, device="cuda")

                def fn():
                    ce.embedding_forward(table, idx, offsets, None, batch_size=B, num_hots=0, out=out)
                side = torch.cuda.Stream()
                with torch.cuda.stream(side):
                    fn()
                    torch.cuda.current_stream().synchronize()
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph, stream=side):
                        for _ in range(10):
                            fn()
                for _ in range(2):
                    graph.replay()
                a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                for _ in range(5):
                    graph.replay()
                z.record()
                z.synchronize()
                times[mode] = a.elapsed_time(z) / 50 * 1e3
                outs[mode] = out
            ce.set_forward_wide_load("auto")
            same = all(torch.equal(outs["never"].view(torch.uint8), outs[m].view(torch.uint8)) for m in MODES)
            shape = ce.forward_launch_shape(dt, torch.int32, W, B, 0, is_csr=True)
            print("%d,%d,%d,%s,%d,%s,%s,%d" % (B, H, W, str(dt)[6:], W * table.element_size(),
                                              ",".join("%.2f" % times[m] for m in MODES), same,
                                              shape["samples_per_block"] if shape["wide_load"] else 0), flush=True)
