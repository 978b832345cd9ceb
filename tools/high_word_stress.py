#!/usr/bin/env python3
"""Stress of the sort's persistent high-word kernels (grid-wide barriers with agent-scope fences): Transpose of int64 keys
that use all 64 bits, many iterations on fresh data, every result compared with torch's stable sort on the device.  A
stale line behind a barrier would show up as a rare wrong element, not as a crash.  Prints one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cuembed_amd as ce

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
sizes = [5000, 40000, 229376, 300000, 1_300_000, 4_194_304]
g = torch.Generator(device="cuda")
g.manual_seed(1)
res = {"iterations": {}, "mismatches": 0}
t_end = time.time() + seconds
it = 0
while time.time() < t_end:
    n = sizes[it % len(sizes)]
    it += 1
    keys = torch.randint(-(1 << 62), 1 << 62, (n,), generator=g, device="cuda", dtype=torch.int64)
    if it % 3 == 0:      # only some high digits vary: other combinations of skipped passes
        keys = (keys >> 40) << 40 | (keys & 0xffff)
    sid = torch.arange(n, device="cuda", dtype=torch.int64)
    t = ce.transpose(sid, keys, None, remapped=True)
    want_keys, order = torch.sort(keys, stable=True)
    ok = bool(torch.equal(t[0], want_keys)) and bool(torch.equal(t[1], order))
    heads = torch.ones(n, dtype=torch.int64, device="cuda")
    heads[1:] = (want_keys[1:] != want_keys[:-1]).to(torch.int64)
    ok = ok and bool(torch.equal(t[3], torch.cumsum(heads, 0) - 1))
    res["iterations"][str(n)] = res["iterations"].get(str(n), 0) + 1
    if not ok:
        res["mismatches"] += 1
res["last_error"] = int(ce._lib.lib().cuembed_peek_last_error())
print(json.dumps(res))
