#!/bin/bash
# copies gpurun_out/refresh/* into profiles/ under the round-5 names
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/refresh
cp $S/bench_c2_line.json profiles/r05_bench_c2_line.json
cp $S/bench_c2_line_200_steps.json profiles/r05_bench_c2_line_200_steps.json
cp $S/bench_c2_two_ranks_sharing_one_gpu.json profiles/r05_bench_c2_two_ranks_sharing_one_gpu.json
cp $S/bench_c2_kernel_trace_stats.txt profiles/r05_bench_c2_kernel_trace_stats.txt
cp $S/pipeline_c2_kernel_trace_stats.txt profiles/r05_pipeline_c2_kernel_trace_stats.txt
cp $S/manual_benchmark_c2_c3.txt profiles/r05_manual_benchmark_c2_c3.txt
cp $S/secondary_kernels.txt profiles/r05_secondary_kernels_c2.txt
cp $S/train_step_none.json profiles/r05_train_step_1gpu.jsonl
cp $S/torch_op_step_probe_native_binding.jsonl profiles/r05_torch_op_step_probe_native_binding.jsonl
cp $S/host_table_probe.json profiles/r05_host_table_row_cache_probe.json
cp $S/sweep_parameters_fwd_transpose_bwd.csv profiles/r05_sweep_parameters_fwd_transpose_bwd.csv
cp $S/pmc_passes.txt profiles/r05_pmc_passes_forward_pipeline_c3.txt
cp $S/pmc_sq_pipeline.txt profiles/r05_pmc_sq_counters_pipeline_c4.txt
cp $S/torch_sparse_orders_probe.json profiles/r05_torch_sparse_orders_probe.json
cp $S/bwd_blocked_coalesced_probe.json profiles/r05_bwd_blocked_coalesced_probe_c4.json
cp $S/c3_balance_probe.json profiles/r05_c3_balance_probe.json
cp $S/traffic_rows.txt profiles/r05_traffic_rows.txt
cp $S/torch_step_profile_b1024.txt profiles/r05_torch_step_profile_b1024.txt
cp $S/torch_policy_probe.json profiles/r05_torch_policy_probe.json
cp $S/small_sort_reference_sequence.txt profiles/r05_small_sort_reference_sequence.txt
cp $S/small_sort_one_call.txt profiles/r05_small_sort_one_call.txt
cp $S/headline_pattern_loads_only_ceiling.csv profiles/r05_headline_pattern_loads_only_ceiling.csv
cp $S/row_read_ceiling.csv profiles/r05_row_read_ceiling.csv
cp $S/narrow_row_probe.jsonl profiles/r05_narrow_row_probe.jsonl
cp $S/torch_graph_step_probe.json profiles/r05_torch_graph_step_probe.json
cp $S/forward_parts_probe.csv profiles/r05_forward_parts_probe.csv
cp $S/traffic_c2.json profiles/traffic_c2.json
cp $S/traffic_c3.json profiles/traffic_c3.json
ls -la profiles | grep r05
