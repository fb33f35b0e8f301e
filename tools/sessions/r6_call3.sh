#!/bin/bash
# round 6, call 3: per-rank kernel times of the partitioned search, P ranks in lockstep in one process (tools/part_sim.py);
# then the full-size C5' tests with 2 and 8 ranks sharing the GPU
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
export PART_SIM_JSON=gpurun_out/r6_c3_part_sim.jsonl; rm -f $PART_SIM_JSON
timeout 600 python tools/part_sim.py lj 1 2 8 > gpurun_out/r6_c3_part_sim_lj.txt 2>&1; el "lj rc $?"
cat gpurun_out/r6_c3_part_sim_lj.txt | grep -v "^$" | tail -40
timeout 900 python tools/part_sim.py twitter 1 8 > gpurun_out/r6_c3_part_sim_twitter.txt 2>&1; el "twitter rc $?"
cat gpurun_out/r6_c3_part_sim_twitter.txt | tail -30
(timeout 1500 python -m pytest tests/test_distributed.py -m gpu -q -x --durations=4 -k "c5_twitter" > gpurun_out/r6_c3_pytest_c5.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c3_pytest_c5.log); el pytest-c5
tail -8 gpurun_out/r6_c3_pytest_c5.log
