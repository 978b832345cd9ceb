#!/usr/bin/env python3
"""EmbeddingBackwardReferenceSums on ONE run of L lookups (fp16, 256-wide rows): time per lookup of the rounding chain, for
sequential and for random sample ids, next to a COO of many short runs.  Where does the long-run path spend its time?"""
import sys, time
sys.path.insert(0, '.')
import torch
import cuembed_amd as ce

B, W = 65536, 256
gy = (torch.randn(B, W, device='cuda') * 3).half()


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n * 1e3


for L in (256, 300, 1024, 4096, 16384, 65528):
    for order in ("sequential", "random", "one sample (every gather an L2 hit)"):
        ti = torch.zeros(L, dtype=torch.int32, device='cuda')
        ts = (torch.arange(L, device='cuda') if order == "sequential" else
              torch.randint(0, B if order == "random" else 1, (L,), device='cuda')).int()
        g = torch.empty((1, W), dtype=torch.float16, device='cuda')
        ms = timed(lambda: ce.embedding_backward(gy, 1, ti, ts, grad_embedding=g, reference_sums=True))
        print("one run of %6d lookups, %s sample ids: %.4f ms = %.1f ns per lookup" % (L, order, ms, ms * 1e6 / L), flush=True)

# Is the lone workgroup of a long run held back by the chip's clocks?  (One workgroup on 256 compute units is a light load:
# the power management may not raise the shader clock for it.)  The same 65,528-lookup run while another stream keeps the
# chip busy with large GEMMs.
L = 65528
ti = torch.zeros(L, dtype=torch.int32, device='cuda')
ts = torch.arange(L, device='cuda').int()
g = torch.empty((1, W), dtype=torch.float16, device='cuda')
a = torch.randn(8192, 8192, device='cuda', dtype=torch.float16)
side = torch.cuda.Stream()
torch.cuda.synchronize()
with torch.cuda.stream(side):
    for _ in range(60):
        a @ a
start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
start.record()
for _ in range(3):
    ce.embedding_backward(gy, 1, ti, ts, grad_embedding=g, reference_sums=True)
stop.record()
torch.cuda.synchronize()
print("one run of %6d lookups while GEMMs run on another stream: %.4f ms = %.1f ns per lookup"
      % (L, start.elapsed_time(stop) / 3, start.elapsed_time(stop) / 3 * 1e6 / L), flush=True)
