#!/usr/bin/env python3
"""Does the index work of the backward (row ids + transpose + remap: latency- and launch-bound, 0.10 ms at C4) hide under
the forward (fabric-bound, 0.137 ms) when it runs on a second stream?  C4 shape through the C ABI wrappers; one JSON line:
serial = forward then index work on one stream, overlapped = index work on a side stream forked before the forward."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cuembed_amd import harness
from cuembed_amd import ops as ce

rows, W, B, H = 10_000_000, 256, 65536, 64
table = torch.empty((rows, W), dtype=torch.float16, device="cuda").uniform_(-1, 1)
idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15).astype(np.int32)).cuda()
out = torch.empty((B, W), dtype=torch.float16, device="cuda")
work = torch.empty(ce.transpose_workspace_bytes(B * H, torch.int32) + (1 << 20), dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream(priority=int(os.environ.get("SIDE_PRIORITY", "0")))
main = torch.cuda.current_stream()


def forward():
    ce.embedding_forward(table, idx, num_hots=H, out=out)


def index_work(blocks):
    return ce.transpose_fixed_hotness(idx.view(-1), B, H, None, num_categories=rows, sample_blocks=blocks, remapped=True)


def serial(blocks):
    forward()
    index_work(blocks)


def overlapped(blocks):
    fork = torch.cuda.Event()
    fork.record(main)
    forward()
    with torch.cuda.stream(side):
        side.wait_event(fork)
        index_work(blocks)
        join = torch.cuda.Event()
        join.record(side)
    main.wait_event(join)


def timed(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e3, 4)


res = {"forward_ms": timed(forward)}
for blocks in (1, 2):
    res["blocks_%d" % blocks] = {"index_work_ms": timed(lambda: index_work(blocks)),
                                "serial_ms": timed(lambda: serial(blocks)),
                                "overlapped_ms": timed(lambda: overlapped(blocks))}
print(json.dumps(res))
