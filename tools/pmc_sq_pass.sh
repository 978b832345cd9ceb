#!/bin/bash
# One rocprofv3 counter pass (SQ issue / wait counters) over the C4 pipeline of the C++ benchmark.
#     gpurun --timeout 600 -- 'bash tools/pmc_sq_pass.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_sq
rm -rf "$O"; mkdir -p "$O"
C2="--num_categories 10000000 --embed_width 256 --batch_size 65536 --alpha 1.15 --hotness 64 --half_embedding_type=true --iterations 5 --clear_caches=false"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d "$O/sq" -- "$R/benchmarks/manual_benchmark" $C2 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$O/sq2" -- "$R/benchmarks/manual_benchmark" $C2 > /dev/null 2>&1
cd "$R"
{
  for p in sq sq2; do
    echo "#### $p"
    python tools/rocprof_summary.py "$O/$p" 2>/dev/null
  done
} > "$R/gpurun_out/pmc_sq.txt"
rm -rf "$O"
wc -l "$R/gpurun_out/pmc_sq.txt"
