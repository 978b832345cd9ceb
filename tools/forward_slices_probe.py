#!/usr/bin/env python3
"""EmbeddingForward's kernel with the rows cut into 1 / 2 / 4 / 8 column slices pinned to XCDs (ColumnSlice, the `column_slices`
argument the launcher leaves at 1) over table sizes from L2-resident to 5 GB: where does slicing pay?  512-byte rows (fp16
W = 256), batch 65,536 x hotness 64 and 8,192 x 64, uniform and power-law indices.  CSV; every sliced result is compared with
the unsliced one (same bits)."""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cuembed_amd import harness

L = ctypes.CDLL(os.path.join(ROOT, "tools", "forward_variants.so"))
VARIANTS = {1: 0, 2: 10, 4: 11, 8: 12}
W = 256
big = torch.empty((10_000_000, W), dtype=torch.float16, device="cuda").uniform_(-1, 1)
print("rows,table_MB,batch,hotness,alpha,slices1_ms,slices2_ms,slices4_ms,slices8_ms,same_bits")
for B, H in ((65536, 64), (8192, 64)):
    out = torch.empty((B, W), dtype=torch.float16, device="cuda")
    ref = torch.empty((B, W), dtype=torch.float16, device="cuda")
    for rows in (4096, 65536, 262144, 524288, 1_000_000, 2_000_000, 10_000_000):
        for alpha in (0.0, 1.15):
            idx = [torch.from_numpy(harness.generate_indices(rows, B, H, alpha=alpha, seed=s).astype(np.int32)).cuda()
                   if "seed" in harness.generate_indices.__code__.co_varnames else None for s in range(2)]
            if idx[0] is None:
                flat = torch.from_numpy(harness.generate_indices(rows, 2 * B, H, alpha=alpha).astype(np.int32)).cuda().view(2, -1)
                idx = [flat[0].contiguous(), flat[1].contiguous()]
            table = big[:rows]
            res, same = [], True
            for slices, vid in VARIANTS.items():
                def launch(t):
                    L.variant_launch(vid, ctypes.c_void_p(table.data_ptr()), W, B, ctypes.c_void_p(idx[t % 2].data_ptr()), H,
                                     ctypes.c_void_p((ref if slices == 1 else out).data_ptr()), 8, None)
                for t in range(6):
                    launch(t)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for t in range(40):
                    launch(t)
                torch.cuda.synchronize()
                res.append("%.4f" % ((time.perf_counter() - t0) / 40 * 1e3))
                if slices > 1:
                    same = same and bool(torch.equal(out, ref))
            print("%d,%.0f,%d,%d,%.2f,%s,%s" % (rows, rows * W * 2 / 1e6, B, H, alpha, ",".join(res), same), flush=True)
