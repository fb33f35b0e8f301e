#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for cfg in "1 2" "0 2" "1 1" "0 1" "1 3"; do
  set -- $cfg
  echo "== GRX_SOURCE_LEVEL=$1 GRX_PACE_DEPTH=$2" >> gpurun_out/cb_bench.log
  for w in kron lj; do
  GRX_SOURCE_LEVEL=$1 GRX_PACE_DEPTH=$2 timeout 300 python bench.py --workload $w --only bfs --no-cpu-baseline --steps 40 --warmup 5 2>&1 | grep "^{" | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w DO', d['value'], d['ms_per_step'], d['config']['enact_ms_last'], d['config']['kernel_launch_groups_per_step'])" >> gpurun_out/cb_bench.log 2>&1
  done
done
cat gpurun_out/cb_bench.log
