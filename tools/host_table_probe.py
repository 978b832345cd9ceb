#!/usr/bin/env python3
"""Does a software hot-row cache pay when the table is NOT in HBM?  (on MI355X the
256 MiB Infinity Cache already is the hot-row cache of an HBM-resident table -- an LDS cache was
measured and rejected in round 1 -- so the question only makes sense for host-resident tables.)

10M x 256 fp16 table in pinned host memory, read zero-copy by the unmodified forward kernel over
PCIe; C2 batch (65536 x 64, alpha 1.15).  Times the forward with no cache and with the most frequent
rows of earlier batches cached in HBM (cuembed_amd/row_cache.py), for several capacities, and the
HBM-resident table for reference.  One JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cuembed_amd as ce
from cuembed_amd import harness
from cuembed_amd.row_cache import CachedHostTable

rows, W, B, H = 10_000_000, 256, 65536, 64
dev = torch.device("cuda", 0)
t0 = time.time()
host = torch.empty((rows, W), dtype=torch.float16).pin_memory()
host.view(torch.int16).random_(0, 15360)                # fp16 bit patterns of [0, 1): finite; values do not matter for timing
alloc_s = time.time() - t0
idx_all = harness.generate_indices(rows, 5 * B, H, alpha=1.15).reshape(5, B * H)
profile = torch.from_numpy(np.ascontiguousarray(idx_all[:4])).to(dev)       # 4 earlier batches decide what is cached
idx = torch.from_numpy(np.ascontiguousarray(idx_all[4])).to(dev)            # the timed batch is a new one


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / n


out = {"table": "10M x 256 fp16 in pinned host memory (zero-copy over PCIe)", "pinned_alloc_and_fill_s": round(alloc_s, 1),
       "batch": "65536 x 64, alpha 1.15", "results": []}
dev_table = host.to(dev)
ref = ce.embedding_forward(dev_table, idx, num_hots=H)
out["hbm_resident_ms"] = round(timed(lambda: ce.embedding_forward(dev_table, idx, num_hots=H), 20), 4)
del dev_table
row_bytes = W * 2
for cap in (0, 10_000, 100_000, 1_000_000, 4_000_000):
    t = CachedHostTable(host, dev, capacity_rows=max(cap, 1))
    if cap:
        t.cache_most_frequent(profile)
    use = cap > 0
    got = t.forward(idx, num_hots=H, use_cache=use)
    ms = timed(lambda: t.forward(idx, num_hots=H, use_cache=use), 3 if cap < 100_000 else 10)
    hit = float((t.slot_of_row[idx.long()] >= 0).float().mean().item()) if use else 0.0
    miss_bytes = (1.0 - hit) * B * H * row_bytes
    out["results"].append({"cached_rows": cap, "cache_MB": round(cap * row_bytes / 1e6, 1), "lookup_hit_rate": round(hit, 4),
                           "forward_ms": round(ms, 3), "bytes_over_pcie_MB": round(miss_bytes / 1e6, 1),
                           "pcie_GBps": round(miss_bytes / (ms * 1e-3) / 1e9, 1),
                           "bit_identical_to_hbm_table": bool(torch.equal(got, ref))})
    del t
print(json.dumps(out))
