#!/bin/bash
# round 6, call 15: latency of every dependent step of a many-levels level on the road stand-in (fine-timers build), the 4-phase clocks
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
(echo "### libgrx_timers.so (4 phases, no forced waits)"; GRX_LIB_PATH=$PWD/gunrock_amd/libgrx_timers.so GRX_MID_DEBUG=1 timeout 600 python tools/mid_phases.py both 2>&1 | grep -v amdgpu.ids
 echo "### libgrx_fine.so (a full wait behind every sub-phase)"; GRX_LIB_PATH=$PWD/gunrock_amd/libgrx_fine.so GRX_MID_DEBUG=1 timeout 600 python tools/mid_phases.py both 2>&1 | grep -v amdgpu.ids) > gpurun_out/r6_c15_mid_phases.txt
cat gpurun_out/r6_c15_mid_phases.txt
