#!/bin/bash
# copies gpurun_out/refresh/* into profiles/ under this round's names (ROUND=r06 by default)
set -e
cd "$(dirname "$0")/.."
S=gpurun_out/refresh
R=${ROUND:-r06}
cp $S/bench_c2_line.json profiles/${R}_bench_c2_line.json
cp $S/bench_c2_line_200_steps.json profiles/${R}_bench_c2_line_200_steps.json
cp $S/bench_c2_two_ranks_sharing_one_gpu.json profiles/${R}_bench_c2_two_ranks_sharing_one_gpu.json
cp $S/bench_c2_kernel_trace_stats.txt profiles/${R}_bench_c2_kernel_trace_stats.txt
cp $S/pipeline_c2_kernel_trace_stats.txt profiles/${R}_pipeline_c2_kernel_trace_stats.txt
cp $S/manual_benchmark_c2_c3.txt profiles/${R}_manual_benchmark_c2_c3.txt
cp $S/secondary_kernels.txt profiles/${R}_secondary_kernels_c2.txt
cp $S/train_step_none.json profiles/${R}_train_step_1gpu.jsonl
cp $S/torch_op_step_probe_native_binding.jsonl profiles/${R}_torch_op_step_probe_native_binding.jsonl
cp $S/host_table_probe.json profiles/${R}_host_table_row_cache_probe.json
cp $S/sweep_parameters_fwd_transpose_bwd.csv profiles/${R}_sweep_parameters_fwd_transpose_bwd.csv
cp $S/pmc_passes.txt profiles/${R}_pmc_passes_forward_pipeline_c3.txt
cp $S/pmc_sq_pipeline.txt profiles/${R}_pmc_sq_counters_pipeline_c4.txt
cp $S/torch_sparse_orders_probe.json profiles/${R}_torch_sparse_orders_probe.json
cp $S/bwd_blocked_coalesced_probe.json profiles/${R}_bwd_blocked_coalesced_probe_c4.json
cp $S/c3_balance_probe.json profiles/${R}_c3_balance_probe.json
cp $S/traffic_rows.txt profiles/${R}_traffic_rows.txt
cp $S/torch_step_profile_b1024.txt profiles/${R}_torch_step_profile_b1024.txt
cp $S/torch_policy_probe.json profiles/${R}_torch_policy_probe.json
cp $S/small_sort_reference_sequence.txt profiles/${R}_small_sort_reference_sequence.txt
cp $S/small_sort_one_call.txt profiles/${R}_small_sort_one_call.txt
cp $S/headline_pattern_loads_only_ceiling.csv profiles/${R}_headline_pattern_loads_only_ceiling.csv
cp $S/row_read_ceiling.csv profiles/${R}_row_read_ceiling.csv
cp $S/narrow_row_probe.jsonl profiles/${R}_narrow_row_probe.jsonl
cp $S/torch_graph_step_probe.json profiles/${R}_torch_graph_step_probe.json
cp $S/forward_parts_probe.csv profiles/${R}_forward_parts_probe.csv
cp $S/traffic_c2.json profiles/traffic_c2.json
cp $S/traffic_c3.json profiles/traffic_c3.json
for f in high_word_timing.jsonl high_word_stress.json row_loads_crossover.jsonl reference_sums_run_probe.txt exchange_device_time.json; do
  [ -f $S/$f ] && cp $S/$f profiles/${R}_$f
done
if [ -s $S/exchange_one_rank_rccl_kernel_trace_stats.txt ]; then
  { cat <<'EOT'
# rocprofv3 --kernel-trace --stats of tools/exchange_device_time.py: ONE rank over real RCCL (world size 1), the C4 gradient of
# one GPU (572,029 compressed rows of 512 bytes): SparseGradExchange.start() + wait() (13 steps), the exact-size
# allreduce_sparse_grad() (12 steps) and the pack / merge halves alone (13 each) in one process.  With one rank the
# collectives are RCCL self-copies (rcclGenericKernel, copyBuffer: not what a link costs); everything else is the on-device
# work of a step that DESIGN section 6's link model does not contain: OwnerRangeStartsKernel + PackRowsByOwnerKernel (pack),
# three radix passes + SegmentedScatterAddKernel + FinishOwnerPieceKernel (the owner's merge).
EOT
    cat $S/exchange_one_rank_rccl_kernel_trace_stats.txt; } > profiles/${R}_exchange_one_rank_rccl_kernel_trace_stats.txt
fi
[ -f $S/sweep_parameters_cpp_binary.csv ] && cp $S/sweep_parameters_cpp_binary.csv profiles/${R}_sweep_parameters_cpp_binary_min_median_share.csv
[ -f $S/bench_c2_eight_ranks_sharing_one_gpu.json ] && cp $S/bench_c2_eight_ranks_sharing_one_gpu.json profiles/${R}_bench_c2_eight_ranks_sharing_one_gpu.json
ls -la profiles | grep ${R}
