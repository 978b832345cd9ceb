#!/bin/bash
# rocprofv3 counter passes over the whole C4 pipeline (forward, transpose, remap, backward) of the
# C++ benchmark: kernel trace, FETCH_SIZE, WRITE_SIZE and L2 hit/miss, each in its OWN pass
# (counters never together with other trace domains).  Writes gpurun_out/pmc_pipeline.txt.
#     gpurun --timeout 1200 -- 'bash tools/pmc_pipeline.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_pipeline
rm -rf "$O"; mkdir -p "$O"
C2="--num_categories 10000000 --embed_width 256 --batch_size 65536 --alpha 1.15 --hotness 64 --half_embedding_type=true --iterations 5 --clear_caches=false"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -- "$R/benchmarks/manual_benchmark" $C2 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -- "$R/benchmarks/manual_benchmark" $C2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -- "$R/benchmarks/manual_benchmark" $C2 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$O/tcc" -- "$R/benchmarks/manual_benchmark" $C2 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d "$O/sq" -- "$R/benchmarks/manual_benchmark" $C2 > /dev/null 2>&1
cd "$R"
{
  for p in trace fetch write tcc sq; do
    echo "#### $p"
    python tools/rocprof_summary.py "$O/$p" 2>/dev/null
  done
} > "$R/gpurun_out/pmc_pipeline.txt"
rm -rf "$O"
wc -l "$R/gpurun_out/pmc_pipeline.txt"
