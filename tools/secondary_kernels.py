#!/usr/bin/env python3
"""Times the entry points the headline benchmark does not cover, on the C2 shape
(10M x 256 fp16 table, batch 65536, hotness 64, alpha 1.15): mean, concat, fp16_math, bf16,
weighted sum, the weight gradient, and the dense / compressed / weighted backward.
Back-to-back launches over 4 distinct index batches; prints ms and GB/s of the bytes each
kernel has to touch (its own algorithmic formula, stated per line)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    import cuembed_amd as ce
    from cuembed_amd import harness

    dev = torch.device("cuda", 0)
    rows, W, B, H = 10_000_000, 256, 65536, 64
    nb = 4
    idx = harness.generate_indices(rows, nb * B, H, alpha=1.15).reshape(nb, -1)
    idxs = [torch.from_numpy(np.ascontiguousarray(idx[i])).to(dev) for i in range(nb)]
    table = torch.empty((rows, W), dtype=torch.float16, device=dev).uniform_(-1, 1)
    wts = [(torch.randint(0, 2, (B * H,), device=dev).to(torch.float16) * 0.25 + 0.25) for _ in range(nb)]
    gy = torch.randint(-10, 11, (B, W), device=dev).to(torch.float16)

    def timeit(fn, n=20):
        for t in range(3):
            fn(t)
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for t in range(n):
            fn(t)
        z.record()
        z.synchronize()
        return a.elapsed_time(z) / n

    def report(name, ms, nbytes, what):
        print("%-34s %8.4f ms  %8.0f GB/s   (%s)" % (name, ms, nbytes / ms / 1e6, what))

    es = 2
    fwd_bytes = es * B * (H + 1) * W
    out = torch.empty((B, W), dtype=torch.float16, device=dev)
    report("forward sum", timeit(lambda t: ce.embedding_forward(table, idxs[t % nb], num_hots=H, out=out)),
           fwd_bytes, "elem*B*(H+1)*W")
    report("forward mean", timeit(lambda t: ce.embedding_forward(table, idxs[t % nb], num_hots=H, mode="mean", out=out)),
           fwd_bytes, "same")
    report("forward sum fp16_math", timeit(lambda t: ce.embedding_forward(table, idxs[t % nb], num_hots=H,
                                                                         fp16_math=True, out=out)), fwd_bytes, "same")
    report("forward weighted sum", timeit(lambda t: ce.embedding_forward(table, idxs[t % nb], weights=wts[t % nb],
                                                                        num_hots=H, out=out)), fwd_bytes, "same")
    idx64 = [i.to(torch.int64) for i in idxs]
    report("forward sum, int64 indices", timeit(lambda t: ce.embedding_forward(table, idx64[t % nb], num_hots=H, out=out)),
           fwd_bytes, "same")
    del idx64
    tb = table.view(torch.bfloat16)   # same bits, read as bf16
    outb = torch.empty((B, W), dtype=torch.bfloat16, device=dev)
    report("forward sum bf16", timeit(lambda t: ce.embedding_forward(tb, idxs[t % nb], num_hots=H, out=outb)),
           fwd_bytes, "same")
    Hc = 16
    outc = torch.empty((B, Hc, W), dtype=torch.float16, device=dev)
    idxc = [i[:B * Hc].contiguous() for i in idxs]
    report("forward concat (hotness 16)", timeit(lambda t: ce.embedding_forward(table, idxc[t % nb], num_hots=Hc,
                                                                               mode="concat", out=outc)),
           es * B * 2 * Hc * W, "elem*B*2H*W")
    del outc
    report("weight gradient", timeit(lambda t: ce.embedding_weight_grad(table, idxs[t % nb], gy, num_hots=H)),
           fwd_bytes, "rows + grad_y, as forward")

    # backward variants on batch 0
    sid = ce.extract_row_ids_from_fixed(B, H, torch.int32, dev)
    ti, ts, tw = ce.transpose(sid, idxs[0], wts[0], num_categories=rows)
    remap = ce.compute_compressed_grad_indices(ti)
    nu = int(remap[-1].item()) + 1
    nnz = B * H
    gather = es * nnz * W
    g_c = torch.empty((nu, W), dtype=torch.float16, device=dev)
    inv = torch.empty((nu,), dtype=torch.int32, device=dev)
    report("backward compressed", timeit(lambda t: ce.embedding_backward(gy, nu, ti, ts, remap, grad_embedding=g_c,
                                                                         inverse_mapping=inv)),
           gather, "elem*nnz*W gathered; %d unique rows" % nu)
    report("backward compressed weighted", timeit(lambda t: ce.embedding_backward(gy, nu, ti, ts, remap, tw,
                                                                                  grad_embedding=g_c,
                                                                                  inverse_mapping=inv)),
           gather, "same")
    g_d = torch.zeros((rows, W), dtype=torch.float16, device=dev)
    report("backward dense, skip_grad_init", timeit(lambda t: ce.embedding_backward(gy, rows, ti, ts, skip_grad_init=True,
                                                                                   grad_embedding=g_d), n=10),
           gather, "same; output rows spread over 5.12 GB")
    report("backward dense incl. 5.12 GB memset", timeit(lambda t: ce.embedding_backward(gy, rows, ti, ts,
                                                                                         grad_embedding=g_d), n=5),
           gather + es * rows * W, "gathered + memset bytes")
    gy32 = gy.float()
    g32 = torch.empty((nu, W), dtype=torch.float32, device=dev)
    report("backward compressed fp32", timeit(lambda t: ce.embedding_backward(gy32, nu, ti, ts, remap, grad_embedding=g32,
                                                                              inverse_mapping=inv)),
           4 * nnz * W, "4*nnz*W gathered")


if __name__ == "__main__":
    main()
