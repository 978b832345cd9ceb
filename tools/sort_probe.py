#!/usr/bin/env python3
"""Transpose / remap at the C4 shape: correctness against torch's stable sort and times of the entry
points (reference API, bounded, fixed-hotness, int64, weighted), back-to-back HIP-event timing.

    gpurun -- 'python tools/sort_probe.py'
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import cuembed_amd as ce
    from cuembed_amd import harness
    dev = torch.device("cuda", 0)
    rows, B, H = 10_000_000, 65536, 64
    idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15)).to(dev)
    sid = ce.extract_row_ids_from_fixed(B, H, torch.int32, dev)
    order = torch.sort(idx.long(), stable=True)
    want_idx, want_sid = order.values.int(), sid[order.indices]
    w = torch.rand(B * H, device=dev).half()

    def timed(fn, n=30):
        for _ in range(3):
            fn()
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        z.record()
        z.synchronize()
        return a.elapsed_time(z) / n

    work = torch.empty(ce.transpose_workspace_bytes(B * H, torch.int64, torch.float16) + 1024, dtype=torch.uint8, device=dev)
    cases = {
        "transpose (reference API, int32)": lambda: ce.transpose(sid, idx, workspace=work),
        "transpose bounded (24 bits)": lambda: ce.transpose(sid, idx, workspace=work, num_categories=rows),
        "transpose_fixed_hotness bounded": lambda: ce.transpose_fixed_hotness(idx, B, H, workspace=work, num_categories=rows),
        "transpose weighted (fp16)": lambda: ce.transpose(sid, idx, w, workspace=work),
    }
    for name, fn in cases.items():
        got = fn()
        torch.cuda.synchronize()
        ok = torch.equal(got[0], want_idx) and torch.equal(got[1], want_sid)
        if got[2] is not None:
            ok = ok and torch.equal(got[2], w[order.indices])
        print("%-40s %.4f ms   %s" % (name, timed(fn), "exact" if ok else "MISMATCH"), flush=True)
    idx64, sid64 = idx.long(), sid.long()
    got = ce.transpose(sid64, idx64, workspace=work)
    ok = torch.equal(got[0], want_idx.long()) and torch.equal(got[1], want_sid.long())
    print("%-40s %.4f ms   %s" % ("transpose (reference API, int64)", timed(lambda: ce.transpose(sid64, idx64, workspace=work)),
                                   "exact" if ok else "MISMATCH"), flush=True)
    t_idx = want_idx.contiguous()
    remap = ce.compute_compressed_grad_indices(t_idx)
    want_remap = torch.cumsum(torch.cat([torch.zeros(1, device=dev, dtype=torch.int32), (t_idx[1:] != t_idx[:-1]).int()]), 0).int()
    print("%-40s %.4f ms   %s" % ("compute_compressed_grad_indices", timed(lambda: ce.compute_compressed_grad_indices(t_idx)),
                                   "exact" if torch.equal(remap, want_remap) else "MISMATCH"), flush=True)

    def both():
        s = ce.extract_row_ids_from_fixed(B, H, torch.int32, dev)
        t = ce.transpose(s, idx, workspace=work)
        ce.compute_compressed_grad_indices(t[0])

    def both_fused():
        t = ce.transpose_fixed_hotness(idx, B, H, workspace=work, num_categories=rows)
        ce.compute_compressed_grad_indices(t[0])

    print("%-40s %.4f ms" % ("row ids + transpose + remap (bench.py)", timed(both)), flush=True)
    print("%-40s %.4f ms" % ("fixed-hotness bounded + remap", timed(both_fused)), flush=True)
    assert ce._lib.lib().cuembed_peek_last_error() == 0


if __name__ == "__main__":
    main()
