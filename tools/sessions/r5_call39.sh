#!/bin/bash
# round 5, call 39 (the round's last GPU seconds): kernel sequences of DIRECTION-OPTIMISING searches from eight multi_source sources
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 120 bash tools/ms_trace.sh do MS_DIR=do > gpurun_out/r5c39_ms_trace_do.txt 2>&1
cut -c1-300 gpurun_out/r5c39_ms_trace_do.txt
