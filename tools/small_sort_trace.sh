#!/bin/bash
# Index work (row ids + Transpose + remap) of the reference's sweep grid at its small and mid-size points, through the
# C++ harness: (1) the harness's own back-to-back timing, (2) per-kernel durations from a rocprofv3 kernel trace, so that
# launch gaps and kernel bodies can be told apart.  -> gpurun_out/small_sort_<tag>.txt
#     gpurun -- 'bash tools/small_sort_trace.sh <tag>'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
TAG=${1:-run}
OUT=$O/small_sort_$TAG.txt
SHAPES=${SHAPES:-"1024:1 1024:16 1024:64 32768:1 131072:1 32768:16 131072:16"}
EXTRA=${EXTRA:-}
mkdir -p "$O"
: > "$OUT"
cd /tmp && export TMPDIR=/tmp
for s in $SHAPES; do
  B=${s%%:*}; H=${s##*:}
  ARGS="--num_categories 10000000 --embed_width 32 --batch_size $B --hotness $H --alpha 1.05 $EXTRA"
  echo "== B=$B H=$H pairs=$((B*H)) $EXTRA" >> "$OUT"
  "$R/benchmarks/manual_benchmark" $ARGS --iterations 200 --clear_caches=false 2>&1 | grep -E "Transpose" >> "$OUT"
  "$R/benchmarks/manual_benchmark" $ARGS --iterations 50 2>&1 | grep -E "Transpose" | sed 's/^/flushed: /' >> "$OUT"
  rm -rf "$O/ss_prof"
  rocprofv3 --kernel-trace --output-format csv -d "$O/ss_prof" -- "$R/benchmarks/manual_benchmark" $ARGS --iterations 20 --clear_caches=false > "$O/ss_prof.log" 2>&1
  python3 "$R/tools/rocprof_summary.py" "$O/ss_prof" 2>/dev/null | grep -A1 -E "Radix|RunHead|FillQuotient|BlockSort|ExpandCsr" | grep -E "calls=|^  [_a-zA-Z]" | cut -c1-140 >> "$OUT"
  rm -rf "$O/ss_prof"
done
cat "$OUT"
