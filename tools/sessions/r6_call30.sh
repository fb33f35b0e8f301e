#!/bin/bash
# round 6, call 30: margins of the timing assertion "one rank == the engine" (three runs, values printed)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 600 python -m pytest tests/test_distributed.py -m gpu -q -x -s -k "one_rank_partition_is_the_engine" 2>&1 | grep "one rank ==\|passed\|failed"; done > gpurun_out/r6_c30_margins.txt
cat gpurun_out/r6_c30_margins.txt
