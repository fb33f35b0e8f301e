#!/bin/bash
# round 6, call 4: hub-first in-rows of a partition: bottom-up levels of P ranks again (tools/part_sim.py), partition tests
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
export PART_SIM_JSON=gpurun_out/r6_c4_part_sim.jsonl; rm -f $PART_SIM_JSON
PART_SIM_DIR=optimized timeout 600 python tools/part_sim.py lj 2 8 > gpurun_out/r6_c4_part_sim_lj.txt 2>&1; el "lj rc $?"
grep -v "^$" gpurun_out/r6_c4_part_sim_lj.txt | tail -12
PART_SIM_DIR=optimized GRX_PREP_TIMING=1 timeout 900 python tools/part_sim.py twitter 8 > gpurun_out/r6_c4_part_sim_twitter.txt 2>&1; el "twitter rc $?"
grep -v "grx prep\] [a-oq-z ]" gpurun_out/r6_c4_part_sim_twitter.txt | tail -12
(timeout 900 python -m pytest tests/test_distributed.py -m gpu -q -x --durations=4 -k "partitioned_engine or two_ranks_real" > gpurun_out/r6_c4_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c4_pytest.log); el pytest
tail -6 gpurun_out/r6_c4_pytest.log
