import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import cuembed_amd as ce
from cuembed_amd import harness
dev = torch.device("cuda", 0)
rows, W, B = 10_000_000, 128, 65536
table = torch.empty((rows, W), device=dev).uniform_(-1, 1)
out = torch.empty((B, W), device=dev)
a = harness.allocate_forward(rows, W, B, 128, alpha=1.15, is_csr=True, with_table=False, consume_table_draws=False)
idx_csr = torch.from_numpy(a["indices"]).to(dev); off_csr = torch.from_numpy(a["offsets"]).to(dev)
nnz = idx_csr.numel()
fixed = torch.from_numpy(harness.generate_indices(rows, B, 64, alpha=1.15)).to(dev)
off_const = torch.arange(0, B * 64 + 1, 64, device=dev, dtype=torch.int32)
# same lookups as the CSR batch but re-cut into constant-length bags (nnz truncated to B*k)
k = nnz // B
off_recut = torch.arange(0, B * k + 1, k, device=dev, dtype=torch.int32)
# CSR lengths sorted ascending (same multiset of lengths, no variation inside a workgroup)
lens = np.diff(a["offsets"]); order = np.argsort(lens, kind="stable")
off_sorted = np.concatenate([[0], np.cumsum(lens[order])]).astype(np.int32)
idx_sorted = np.concatenate([a["indices"][a["offsets"][s]:a["offsets"][s+1]] for s in order]) if False else None
def t(fn, n=30):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n
print("fixed H=64 (LDS staged)      %.4f ms" % t(lambda: ce.embedding_forward(table, fixed, num_hots=64, out=out)))
print("CSR const 64 (same indices)  %.4f ms" % t(lambda: ce.embedding_forward(table, fixed, off_const, num_hots=0, out=out)))
print("CSR recut const %d           %.4f ms  (nnz %d)" % (k, t(lambda: ce.embedding_forward(table, idx_csr, off_recut, num_hots=0, out=out)), B*k))
print("CSR variable U[0,128]        %.4f ms  (nnz %d)" % (t(lambda: ce.embedding_forward(table, idx_csr, off_csr, num_hots=0, out=out)), nnz))
off_sorted_t = torch.from_numpy(off_sorted).to(dev)
print("CSR variable, lengths sorted %.4f ms" % t(lambda: ce.embedding_forward(table, idx_csr, off_sorted_t, num_hots=0, out=out)))
