#!/bin/bash
# C2/C4 pipeline through the C++ harness + per-kernel durations (rocprofv3 kernel trace) -> gpurun_out/quick_*.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
C2="--num_categories 10000000 --embed_width 256 --batch_size 65536 --alpha 1.15 --hotness 64 --half_embedding_type=true"
cd "$R"
{
  for ex in "" "--bounded_sort --fused_row_ids" "--use_int64_indices" "--clear_caches=false" "--clear_caches=false --bounded_sort --fused_row_ids"; do
    echo "== C2/C4 $ex"; benchmarks/manual_benchmark $C2 --iterations 30 $ex 2>&1 | grep -E "Iterations"
  done
} > "$O/quick_manual_benchmark.txt"
cat "$O/quick_manual_benchmark.txt"
cd /tmp && export TMPDIR=/tmp
rm -rf "$O/quick_prof"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/quick_prof" -- "$R/benchmarks/manual_benchmark" $C2 --iterations 10 --clear_caches=false > "$O/quick_prof.log" 2>&1
cd "$R"
python tools/rocprof_summary.py "$O/quick_prof" > "$O/quick_pipeline_kernel_trace.txt" 2>/dev/null
rm -rf "$O/quick_prof"
grep -A1 -E "Radix|RunHead|Segmented|GatherReduce|FillQuotient|ZeroShared" "$O/quick_pipeline_kernel_trace.txt" | grep -E "calls=|^  [_a-zA-Z]" | cut -c1-150
