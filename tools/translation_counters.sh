#!/bin/bash
# Address-translation counters for the headline's random 512-byte gathers over a 5.12 GB table:
# C2 at alpha 1.15, C2 at alpha 0, a batch of distinct rows, and the loads-only probe.  Each counter set is its OWN
# rocprofv3 run (no other trace domain); the program goes directly after `--`.
#     gpurun --timeout 1500 -- 'bash tools/translation_counters.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/translation
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$O/list_avail.txt" 2>&1
grep -i -E "utcl|tlb|xnack|transl|TCP_TA|TA_ADDR_STALL|TCP_PENDING|TCP_TCC_READ_REQ_LATENCY|TCP_TCP_LATENCY|TCC_EA0_RDREQ_LEVEL|TCC_EA0_RDREQ\b|MALL|TCC_TAG_STALL" "$O/list_avail.txt" | sort -u > "$O/candidates.txt"
PF="python3 $R/tools/profile_forward.py --pattern all --iters 6"
PL="$R/tools/row_read_ceiling --c2 1.15"
SETS=(
 "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum"
 "TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum"
 "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum"
 "TCP_UTCL1_STALL_LRU_INFLIGHT_sum TCP_UTCL1_STALL_LFIFO_FULL_sum"
 "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_MISSFIFO_FULL_sum"
 "TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum"
 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"
 "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum"
 "TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
 "TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum"
 "TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_DRAM_sum"
 "TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum"
 "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum GRBM_GUI_ACTIVE"
)
i=0
for s in "${SETS[@]}"; do
  d="$O/fwd_$i"
  timeout 600 rocprofv3 --pmc $s --output-format csv -d "$d" -- $PF > "$O/fwd_$i.log" 2>&1
  echo "#### forward (6 launches each: alpha 1.15, alpha 0, distinct rows): $s" >> "$O/translation_counters.txt"
  python3 "$R/tools/rocprof_summary.py" "$d" --groups 3 --match GatherReduce >> "$O/translation_counters.txt" 2>/dev/null || tail -3 "$O/fwd_$i.log" >> "$O/translation_counters.txt"
  rm -rf "$d"
  i=$((i+1))
done
i=0
for s in "${SETS[@]:0:7}"; do
  d="$O/lo_$i"
  timeout 600 rocprofv3 --pmc $s --output-format csv -d "$d" -- $PL > "$O/lo_$i.log" 2>&1
  echo "#### loads-only probe, headline pattern alpha 1.15: $s" >> "$O/translation_counters.txt"
  python3 "$R/tools/rocprof_summary.py" "$d" >> "$O/translation_counters.txt" 2>/dev/null || tail -3 "$O/lo_$i.log" >> "$O/translation_counters.txt"
  rm -rf "$d"
  i=$((i+1))
done
ls -la "$O"
