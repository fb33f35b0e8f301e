#!/bin/bash
# round 6, call 13: bucket width of the near-far schedule now that a bucket change costs no launch (GRX_NF_DELTA_SCALE)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/road_ab.py w 2 "-" "GRX_NF_DELTA_SCALE=0.25" "GRX_NF_DELTA_SCALE=0.5" "GRX_NF_DELTA_SCALE=0.75" "GRX_NF_DELTA_SCALE=1.5" "GRX_NF_DELTA_SCALE=2" "GRX_NF_DELTA_SCALE=4" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_c13_delta.txt
cat gpurun_out/r6_c13_delta.txt
