#!/usr/bin/env python3
"""Upper bound of what sample-chunk locality could give the backward (experiment, results of
the modified runs are meaningless): sample ids are rewritten so that every XCD only ever
gathers grad_y rows of 'its' chunk.  C4 shape, run with CUEMBED_BWD_SLICES=1.

Outcome (round 1): 0.339 -> 0.169 ms if EVERY lookup were local.  In the real batch only the
lookups of runs >= ~1000 are (sorted by sample inside a run): 31 % at the granularity of a
workgroup's 1024 lookups, 50 % at 128.  A deterministic in-kernel assignment of nz blocks to XCDs
(64-block bundles, 4-probe vote, ballot ranking; no workspace) was built and is correct, but
measured 0.315 ms against 0.276 ms for the 4 column slices it has to replace, because the other
half of the lookups then gathers unsliced.  Not adopted."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cuembed_amd as ce
from cuembed_amd import harness

dev = torch.device("cuda", 0)
rows, W, B, H = 10_000_000, 256, 65536, 64
idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15)).to(dev)
sid = ce.extract_row_ids_from_fixed(B, H, torch.int32, dev)
ti, ts, _ = ce.transpose(sid, idx, None, num_categories=rows)
remap = ce.compute_compressed_grad_indices(ti)
nu = int(remap[-1].item()) + 1
gy = torch.randint(-10, 11, (B, W), device=dev).to(torch.float16)
g = torch.empty((nu, W), dtype=torch.float16, device=dev)
inv = torch.empty((nu,), dtype=torch.int32, device=dev)


def t(samples, n=20):
    for _ in range(3):
        ce.embedding_backward(gy, nu, ti, samples, remap, grad_embedding=g, inverse_mapping=inv)
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        ce.embedding_backward(gy, nu, ti, samples, remap, grad_embedding=g, inverse_mapping=inv)
    z.record()
    z.synchronize()
    return a.elapsed_time(z) / n


e = torch.arange(B * H, device=dev, dtype=torch.int64)
print("slices env = %s" % os.environ.get("CUEMBED_BWD_SLICES", "(default)"))
print("real sample ids                         %.4f ms" % t(ts))
for group_entries in (1024,):
    for chunk in (8192, 4096, 2048):
        xcd = (e // group_entries) % 8
        fake = ((ts.long() % chunk) + chunk * xcd).to(torch.int32)
        print("chunk of %5d samples per XCD           %.4f ms" % (chunk, t(fake)))
fake = (ts.long() % 1024).to(torch.int32)
print("all gathers inside 1024 samples (L2)    %.4f ms" % t(fake))
