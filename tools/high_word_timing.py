#!/usr/bin/env python3
"""Transpose of int64 keys through the reference signature (all 64 bits sorted): the time when the high word is constant
(every lookup index: the one-launch kernel returns at once) and when it varies (the kernel works through its ticket
queue), next to the launched passes (CUEMBED_SORT_HIGH_WORD_LAUNCHES=1 in the environment).  One JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cuembed_amd as ce


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    z.record()
    z.synchronize()
    return round(a.elapsed_time(z) / n, 4)


res = {"high_word_launches_env": os.environ.get("CUEMBED_SORT_HIGH_WORD_LAUNCHES", "0"),
       "workgroups_env": os.environ.get("CUEMBED_SORT_HIGH_WORD_WORKGROUPS", "")}
g = torch.Generator(device="cuda").manual_seed(5)
for n in (100_000, 1 << 20, 1 << 22):
    sid = torch.arange(n, device="cuda", dtype=torch.int64)
    low = torch.randint(0, 10_000_000, (n,), generator=g, device="cuda", dtype=torch.int64)
    wide = torch.randint(-(1 << 62), 1 << 62, (n,), generator=g, device="cuda", dtype=torch.int64)
    work = torch.empty(ce.transpose_workspace_bytes(n, torch.int64) + 1024, dtype=torch.uint8, device="cuda")
    res["n=%d" % n] = {"high_word_constant_ms": timed(lambda: ce.transpose(sid, low, workspace=work)),
                       "all_64_bits_vary_ms": timed(lambda: ce.transpose(sid, wide, workspace=work))}
print(json.dumps(res))
