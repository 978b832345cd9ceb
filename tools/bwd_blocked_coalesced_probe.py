#!/usr/bin/env python3
"""C4 shape: the reference's compressed gradient from a sample-blocked order (ComputeCompressedGradIndicesBlocked +
EmbeddingBackward(sample_blocks)) against the reference order and the uncoalesced extension.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuembed_amd as ce  # noqa: E402
from cuembed_amd import harness  # noqa: E402


def timed(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    z.record()
    z.synchronize()
    return round(a.elapsed_time(z) / n, 5)


def main():
    rows, W, B, H = 10_000_000, 256, 65536, 64
    dev = torch.device("cuda")
    idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15, index=np.int32)).to(dev)
    gy = torch.randint(-10, 11, (B, W), device=dev).to(torch.float16)
    out = {}
    work = torch.empty(max(ce.transpose_workspace_bytes(B * H, torch.int32), 1), dtype=torch.uint8, device=dev)
    st = {}

    def ref_index():
        st["ti"], st["ts"], _ = ce.transpose_fixed_hotness(idx, B, H, workspace=work, num_categories=rows)
        st["rm"] = ce.compute_compressed_grad_indices(st["ti"])

    out["index_reference_order_ms"] = timed(ref_index)
    nu = int(st["rm"][-1].item()) + 1
    grad = torch.empty((nu, W), dtype=torch.float16, device=dev)
    inv = torch.empty((nu,), dtype=torch.int32, device=dev)
    out["backward_reference_order_ms"] = timed(lambda: ce.embedding_backward(
        gy, nu, st["ti"], st["ts"], st["rm"], grad_embedding=grad, inverse_mapping=inv))
    out["num_unique"] = nu
    for P in [int(a) for a in sys.argv[1:]] or [2, 3, 4]:
        wsb = torch.empty(ce.compressed_grad_blocked_workspace_bytes(B * H, torch.int32, P), dtype=torch.uint8, device=dev)
        nud = torch.zeros(1, dtype=torch.int32, device=dev)
        tab = torch.empty(B * H, dtype=torch.int32, device=dev)

        def blk_sort():
            st["bi"], st["bs"], _ = ce.transpose_fixed_hotness(idx, B, H, workspace=work, num_categories=rows, sample_blocks=P)

        def blk_remap():
            st["brm"], st["tab"], _ = ce.compute_compressed_grad_indices_blocked(st["bi"], P, workspace=wsb, num_unique=nud,
                                                                                 block_row_ids=tab)

        def blk_remap_uncoalesced():
            st["urm"] = ce.compute_compressed_grad_indices(st["bi"])

        e = {}
        e["transpose_ms"] = timed(blk_sort)
        e["remap_blocked_ms"] = timed(blk_remap)
        e["remap_uncoalesced_ms"] = timed(blk_remap_uncoalesced)
        e["index_total_ms"] = timed(lambda: (blk_sort(), blk_remap()))
        assert int(nud.item()) == nu
        g2 = torch.empty((nu, W), dtype=torch.float16, device=dev)
        i2 = torch.empty((nu,), dtype=torch.int32, device=dev)
        e["backward_blocked_coalesced_ms"] = timed(lambda: ce.embedding_backward(
            gy, nu, st["bi"], st["bs"], st["brm"], grad_embedding=g2, inverse_mapping=i2, sample_blocks=P,
            block_row_ids=st["tab"]))
        e["same_rows_as_reference_order"] = bool(torch.equal(g2, grad)) and bool(torch.equal(i2, inv))
        nub = int(st["urm"][-1].item()) + 1
        g3 = torch.empty((nub, W), dtype=torch.float16, device=dev)
        i3 = torch.empty((nub,), dtype=torch.int32, device=dev)
        e["backward_uncoalesced_ms"] = timed(lambda: ce.embedding_backward(
            gy, nub, st["bi"], st["bs"], st["urm"], grad_embedding=g3, inverse_mapping=i3))
        e["uncoalesced_rows"] = nub
        out["blocks_%d" % P] = e
    print(json.dumps(out))


if __name__ == "__main__":
    main()
