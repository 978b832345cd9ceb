#!/usr/bin/env python3
"""gpurun_out/protocol_study.jsonl (one bench.py line per run) -> a table per protocol."""
import json
import sys
from collections import defaultdict

rows = defaultdict(list)
for ln in open(sys.argv[1]):
    ln = ln.strip()
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    key = (d["steps"], d["warmup"], d.get("preroll_ms", 0.0))
    ex = d.get("extras", {})
    rows[key].append((d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["single_launch_event_ms"]["min"],
                      ex.get("alpha0_uniform_back_to_back", {}).get("ms"), ex.get("cold_cache_flush_between_iters", {}).get("ms"),
                      ex.get("transpose_and_remap_ms"), ex.get("backward_compressed_ms"), d["roofline"]["frac"]))
print("%-28s %-4s | C2 wall ms/step (each run)            | HIP-event ms | single min | alpha0 ms | cold ms | transpose | backward | frac"
      % ("protocol", "runs"))
for key in sorted(rows):
    v = rows[key]
    col = lambda i: " ".join("%.4f" % r[i] for r in v if r[i] is not None)   # noqa: E731
    mean = lambda i: sum(r[i] for r in v if r[i] is not None) / max(1, len([r for r in v if r[i] is not None]))  # noqa: E731
    print("steps=%-3d warmup=%-2d pre=%-5g %-4d | %-38s | %.4f       | %.4f     | %.4f    | %.4f  | %.4f    | %.4f   | %.3f"
          % (key[0], key[1], key[2], len(v), col(0), mean(1), mean(2), mean(3), mean(4), mean(5), mean(6), mean(7)))
