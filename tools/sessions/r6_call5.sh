#!/bin/bash
# round 6, call 5: which kernel of a partitioned bottom-up level takes 0.3 ms per rank at P = 8 on the twitter stand-in
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
PART_SIM_DIR=optimized PART_SIM_REPS=1 bash tools/kt_stats.sh tw8 python tools/part_sim.py twitter 8 > gpurun_out/r6_c5_kt_twitter8.txt 2>&1
cat gpurun_out/r6_c5_kt_twitter8.txt
