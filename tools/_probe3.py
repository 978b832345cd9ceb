import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import cuembed_amd as ce
from cuembed_amd import harness
rows, W, B, H = 10_000_000, 256, 65536, 64
dev = torch.device("cuda")
idx = torch.from_numpy(harness.generate_indices(rows, B, H, alpha=1.15, index=np.int32)).to(dev)
gy = torch.randint(-10, 11, (B, W), device=dev).to(torch.float16)
P=int(sys.argv[1]) if len(sys.argv)>1 else 2
bi, bs, _ = ce.transpose_fixed_hotness(idx, B, H, num_categories=rows, sample_blocks=P)
brm, tab, nud = ce.compute_compressed_grad_indices_blocked(bi, P)
nu=int(nud.item())
g2 = torch.empty((nu, W), dtype=torch.float16, device=dev)
i2 = torch.empty((nu,), dtype=torch.int32, device=dev)
for _ in range(30):
    ce.embedding_backward(gy, nu, bi, bs, brm, grad_embedding=g2, inverse_mapping=i2, sample_blocks=P, block_row_ids=tab)
torch.cuda.synchronize()
