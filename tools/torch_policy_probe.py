#!/usr/bin/env python3
"""What a caller of the reference's Python entry point gets WITHOUT knowing the extensions: cuemb_embedding(params, idx,
offsets, weights) with hints="auto" (cuembed_amd.policy) against hints=None, forward only, HIP events, back to back.
  * C3 (fp32 weighted sum, CSR bags U[0, 128], 10M x 128, batch 65,536): the bag order is computed for EVERY call from
    the offsets (one launch, device-side; the hints_auto time includes it);
  * C2 shape with uniform indices (the HBM-bound case): non-temporal row loads once a sample of the batch is >= 99.8 %
    distinct rows (decided on the device, read by the kernels);
  * C2 itself (alpha = 1.15): the policy must NOT pick streaming.
One JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cuembed_amd import cuembed_pyt as P
from cuembed_amd import harness, policy

dev = torch.device("cuda", 0)


def timed(fn, n=30):
    for _ in range(5):
        fn()
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    z.record()
    z.synchronize()
    return round(a.elapsed_time(z) / n, 5)


res = {}
rows = 10_000_000
# ---- C3
w = harness.allocate_forward(rows, 128, 65536, 128, alpha=1.15, is_csr=True, elem=np.float32, index=np.int32, with_table=False,
                             consume_table_draws=False)
table = torch.empty((rows, 128), dtype=torch.float32, device=dev).uniform_(-1, 1)
idx, off = torch.from_numpy(w["indices"]).to(dev), torch.from_numpy(w["offsets"]).to(dev)
wt = torch.from_numpy(w["weights"]).to(dev)
with torch.no_grad():
    plain = P.cuemb_embedding(table, idx, off, wt, hints=None)
    res["c3_hints_none_ms"] = timed(lambda: P.cuemb_embedding(table, idx, off, wt, hints=None))
    auto = P.cuemb_embedding(table, idx, off, wt)
    res["c3_hints_auto_ms"] = timed(lambda: P.cuemb_embedding(table, idx, off, wt))
    res["c3_same_bits"] = bool(torch.equal(plain, P.cuemb_embedding(table, idx, off, wt)))
    res["c3_order_computed_per_call"] = policy.sample_order(off, idx.numel()) is not None
    res["c3_bag_order_alone_ms"] = timed(lambda: policy.sample_order(off, idx.numel()))
del table
torch.cuda.empty_cache()
# ---- C2 shape, uniform and power-law indices
table = torch.empty((rows, 256), dtype=torch.float16, device=dev).uniform_(-1, 1)
off2 = torch.arange(0, 65536 * 64 + 1, 64, dtype=torch.int32, device=dev)
for name, alpha in (("alpha0", 0.0), ("alpha115", 1.15)):
    ids = torch.from_numpy(harness.generate_indices(rows, 65536, 64, alpha=alpha, index=np.int32)).to(dev)
    policy.set_enabled(True)      # (forget the table's last decision: each index stream is judged afresh)
    with torch.no_grad():
        res["c2_%s_hints_none_ms" % name] = timed(lambda: P.cuemb_embedding(table, ids, off2, None, hints=None))
        res["c2_%s_hints_auto_ms" % name] = timed(lambda: P.cuemb_embedding(table, ids, off2, None))
    res["c2_%s_policy_row_loads" % name] = policy.row_loads_decision(table)
print(json.dumps(res))
