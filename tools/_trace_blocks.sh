#!/bin/bash
# usage: _trace_blocks.sh TAG  -> per-launch durations of the blocked scatter kernel
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$1 -o t -- python3 $GRAFT_REPO_ROOT/tools/_probe3.py 2 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/prof_$1/**/t_kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
seq=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in rows)
d=[e-s for s,e,n in seq if "ScatterAddKernel" in n]
d=d[10:]
print("$1", "block0", sum(d[0::2])/len(d[0::2]), "block1", sum(d[1::2])/len(d[1::2]))
PY
