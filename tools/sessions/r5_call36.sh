#!/bin/bash
# round 5, call 36 (the last GPU minutes of the round): the two plan rules of the last session -- no heavy tile in the many-levels
# body (GRX_MID_TILE_E), early levels binned from 2^20 edges (GRX_BIN_EARLY_DIV) -- A/B-ed on eight multi_source sources inside one
# process with the kernel sequence of every search, then the evidence of these sources (tools/final_round_r5b.sh)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 200 bash tools/ms_trace.sh ab > gpurun_out/r5c36_ms_trace.txt 2>&1
cut -c1-330 gpurun_out/r5c36_ms_trace.txt
bash tools/final_round_r5b.sh
