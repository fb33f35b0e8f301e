#!/bin/bash
# round 6, call 6: partition grids by the density of the whole graph (a rank's rows alone looked road-like: 256 workgroups, first
# bottom-up body) -- tools/part_sim.py again
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
export PART_SIM_JSON=gpurun_out/r6_c6_part_sim.jsonl; rm -f $PART_SIM_JSON
timeout 600 python tools/part_sim.py lj 8 > gpurun_out/r6_c6_part_sim_lj.txt 2>&1; el "lj rc $?"
grep -v "^$" gpurun_out/r6_c6_part_sim_lj.txt | tail -8
timeout 900 python tools/part_sim.py twitter 2 8 > gpurun_out/r6_c6_part_sim_twitter.txt 2>&1; el "twitter rc $?"
tail -14 gpurun_out/r6_c6_part_sim_twitter.txt
