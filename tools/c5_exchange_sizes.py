#!/usr/bin/env python3
"""Row counts of the sparse gradient exchange at BASELINE config 5 (10M x 256 fp16, 65,536 samples x 64 lookups per
GPU, alpha 1.15) for 2 / 4 / 8 GPUs, from the benchmark's own deterministic index stream -- CPU only, ~1 minute:

    python tools/c5_exchange_sizes.py > profiles/c5_exchange_sizes.json

per rank: distinct rows of its shard (= rows of its compressed gradient); per (rank, owner): rows a rank sends to one
owner of the equal row-id ranges; per owner: distinct rows after merging all ranks (= its piece of the all-gather).
bench.py turns them into expected milliseconds (cuembed_amd.distributed.exchange_model_ms)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cuembed_amd import harness  # noqa: E402

B, H, ROWS = 65536, 64, 10_000_000
idx = harness.generate_indices(ROWS, 8 * B, H, alpha=1.15).reshape(8, -1)
per = [np.unique(idx[r]) for r in range(8)]
out = {"table_rows": ROWS, "batch_per_gpu": B, "hotness": H, "alpha": 1.15,
       "rows_per_rank": [int(p.size) for p in per], "by_n_gpus": {}}
for n in (2, 4, 8):
    merged = np.unique(np.concatenate(per[:n]))
    cuts = [(ROWS * r) // n for r in range(n + 1)]
    pair = [[int(np.searchsorted(per[s], cuts[d + 1]) - np.searchsorted(per[s], cuts[d])) for d in range(n)]
            for s in range(n)]
    own = [int(np.searchsorted(merged, cuts[d + 1]) - np.searchsorted(merged, cuts[d])) for d in range(n)]
    out["by_n_gpus"][str(n)] = {
        "rows_per_rank_mean": float(np.mean([p.size for p in per[:n]])),
        "merged_rows_all_ranks": int(merged.size),
        "rows_per_rank_and_owner_max": max(max(p) for p in pair),
        "rows_per_rank_and_owner_mean": float(np.mean(pair)),
        "merged_rows_per_owner_max": max(own), "merged_rows_per_owner_min": min(own)}
print(json.dumps(out, indent=1))
