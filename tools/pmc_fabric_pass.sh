#!/bin/bash
# L2 <-> fabric interface counters (read requests, outstanding level, credit stalls, latency) of the
# C2 / C4 pipeline (C++ benchmark: forward alpha = 1.15, transpose, backward) and of the forward at
# alpha = 0.  Counters only, two passes of four TCC counters each.
#     gpurun --timeout 900 -- 'bash tools/pmc_fabric_pass.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_fabric
rm -rf "$O"; mkdir -p "$O"
C2="--num_categories 10000000 --embed_width 256 --batch_size 65536 --hotness 64 --half_embedding_type=true --iterations 5 --clear_caches=false"
cd /tmp && export TMPDIR=/tmp
for a in 1.15 0; do
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum --output-format csv -d "$O/ea_$a" -- "$R/benchmarks/manual_benchmark" $C2 --alpha $a > /dev/null 2>&1
  rocprofv3 --pmc TCC_READ_REQ_LATENCY_sum TCC_READ_REQ_sum TCC_EA0_WRREQ_STALL_sum TCC_BUSY_sum --output-format csv -d "$O/lat_$a" -- "$R/benchmarks/manual_benchmark" $C2 --alpha $a > /dev/null 2>&1
done
cd "$R"
{
  for a in 1.15 0; do for p in ea lat; do
    echo "#### alpha=$a $p"
    python tools/rocprof_summary.py "$O/${p}_$a" 2>/dev/null | grep -A5 "GatherReduceKernel\|SegmentedScatterAdd" | cut -c1-150
  done; done
} > "$R/gpurun_out/pmc_fabric.txt"
rm -rf "$O"
wc -l "$R/gpurun_out/pmc_fabric.txt"
