#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel trace / stats / counter collection) into a short text
summary suitable for committing under profiles/.

    python tools/rocprof_summary.py OUT_DIR [--match GatherReduce] > profiles/xxx.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else None
    groups = int(sys.argv[sys.argv.index("--groups") + 1]) if "--groups" in sys.argv else 0   # dispatch-order groups
    files = sorted(glob.glob(os.path.join(d, "**", "*.csv"), recursive=True))
    for f in files:
        base = os.path.basename(f)
        with open(f, newline="") as fh:
            rows = list(csv.DictReader(fh))
        if not rows:
            continue
        if base.endswith("kernel_stats.csv") or base.endswith("_stats.csv"):
            print("== %s" % base)
            for r in rows[:12]:
                print("  " + ", ".join("%s=%s" % (k, (v[:90] if isinstance(v, str) else v)) for k, v in r.items()))
        elif base.endswith("kernel_trace.csv"):
            agg = defaultdict(list)
            meta = {}
            for r in rows:
                name = r.get("Kernel_Name", "?")
                if match and match not in name:
                    continue
                agg[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                meta[name] = r
            print("== %s (durations in ns)" % base)
            for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                m = meta[name]
                print("  %s\n     calls=%d avg=%.1f min=%d max=%d  grid=%s wg=%s lds=%s vgpr=%s sgpr=%s scratch=%s"
                      % (name[:160], len(v), sum(v) / len(v), min(v), max(v), m.get("Grid_Size_X", m.get("Grid_Size")),
                         m.get("Workgroup_Size_X", m.get("Workgroup_Size")), m.get("LDS_Block_Size"),
                         m.get("VGPR_Count"), m.get("SGPR_Count"), m.get("Scratch_Size")))
        elif base.endswith("counter_collection.csv"):
            agg = defaultdict(lambda: defaultdict(list))
            if rows and "Dispatch_Id" in rows[0]:
                rows.sort(key=lambda r: int(r["Dispatch_Id"]))
            for r in rows:
                name = r.get("Kernel_Name", "?")
                if match and match not in name:
                    continue
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            print("== %s (per-dispatch counter values)" % base)
            for name, cs in agg.items():
                print("  %s" % name[:160])
                for c, v in sorted(cs.items()):
                    print("     %-28s dispatches=%d avg=%.4g min=%.4g max=%.4g" % (c, len(v), sum(v) / len(v), min(v), max(v)))
                    if groups and len(v) % groups == 0 and len(v) > groups:
                        n = len(v) // groups
                        print("     %-28s   by dispatch order, %d groups of %d: %s"
                              % ("", groups, n, "  ".join("%.5g" % (sum(v[g * n:(g + 1) * n]) / n) for g in range(groups))))


if __name__ == "__main__":
    main()
