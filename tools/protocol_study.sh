#!/bin/bash
# the driver runs `bench.py --gpus 1 --steps 20 --warmup 5` in a fresh process, the builder's
# committed lines were `--steps 200 --warmup 20`.  Same box, alternating, five times each, plus the disclosed
# pre-roll at a few lengths; one JSON line per run -> gpurun_out/protocol_study.jsonl, table by
# tools/protocol_study_table.py.
#     gpurun --timeout 2400 -- 'bash tools/protocol_study.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p "$O"; : > "$O/protocol_study.jsonl"
cd "$R"
run() { python bench.py --no-cpu-baseline --no-c3 "$@" 2>> "$O/protocol_study.err" | tail -n 1 >> "$O/protocol_study.jsonl"; }
for i in 1 2 3 4 5; do
  run --steps 20 --warmup 5
  run --steps 200 --warmup 20
done
for pre in 50 200 1000; do
  for i in 1 2 3; do run --steps 20 --warmup 5 --preroll-ms $pre; done
done
python tools/protocol_study_table.py "$O/protocol_study.jsonl" > "$O/protocol_study_table.txt"
cat "$O/protocol_study_table.txt"
