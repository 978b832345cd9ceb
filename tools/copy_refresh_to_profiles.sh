#!/bin/bash
# copies gpurun_out/refresh/* into profiles/ under the round-4 names
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/refresh
cp $S/bench_c2_line.json profiles/r04_bench_c2_line.json
cp $S/bench_c2_line_200_steps.json profiles/r04_bench_c2_line_200_steps.json
cp $S/bench_c2_two_ranks_sharing_one_gpu.json profiles/r04_bench_c2_two_ranks_sharing_one_gpu.json
cp $S/bench_c2_kernel_trace_stats.txt profiles/r04_bench_c2_kernel_trace_stats.txt
cp $S/pipeline_c2_kernel_trace_stats.txt profiles/r04_pipeline_c2_kernel_trace_stats.txt
cp $S/manual_benchmark_c2_c3.txt profiles/r04_manual_benchmark_c2_c3.txt
cp $S/secondary_kernels.txt profiles/r04_secondary_kernels_c2.txt
cp $S/train_step_none.json profiles/r04_train_step_1gpu.jsonl
cp $S/torch_op_step_probe_native_binding.jsonl profiles/r04_torch_op_step_probe_native_binding.jsonl
cp $S/host_table_probe.json profiles/r04_host_table_row_cache_probe.json
cp $S/sweep_parameters_fwd_transpose_bwd.csv profiles/r04_sweep_parameters_fwd_transpose_bwd.csv
cp $S/pmc_passes.txt profiles/r04_pmc_passes_forward_pipeline_c3.txt
cp $S/pmc_sq_pipeline.txt profiles/r04_pmc_sq_counters_pipeline_c4.txt
cp $S/torch_sparse_orders_probe.json profiles/r04_torch_sparse_orders_probe.json
cp $S/bwd_blocked_coalesced_probe.json profiles/r04_bwd_blocked_coalesced_probe_c4.json
cp $S/c3_balance_probe.json profiles/r04_c3_balance_probe.json
cp $S/traffic_rows.txt profiles/r04_traffic_rows.txt
cp $S/traffic_c2.json profiles/traffic_c2.json
cp $S/traffic_c3.json profiles/traffic_c3.json
ls -la profiles | grep r03
