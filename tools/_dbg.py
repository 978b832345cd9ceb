import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuembed_amd import cuembed_pyt as P
from cuembed_amd import harness
dev = torch.device("cuda", 0)
rows, W, H, B = 1_000_000, 256, 64, 65536
table = torch.empty((rows, W), dtype=torch.float16, device=dev).uniform_(-1, 1).requires_grad_(True)
idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15).astype(np.int64)).to(dev)
offsets = torch.arange(0, B * H + 1, H, dtype=torch.int64, device=dev)
up = torch.randint(-2, 3, (B, W), device=dev).to(torch.float16)
g={}
for kind in ("reference_order", True):
    table.grad=None
    P.cuemb_embedding(table, idx, offsets, None, sparse_grad=kind).backward(up)
    gg=table.grad
    g[kind]=(gg._indices().clone(), gg._values().clone(), gg.is_coalesced())
    print(kind, gg._indices().shape, gg._indices().dtype, gg.is_coalesced(), gg._indices()[0,:5], gg._indices()[0,-5:])
print(torch.equal(g["reference_order"][0], g[True][0]), (g["reference_order"][0]!=g[True][0]).sum())
