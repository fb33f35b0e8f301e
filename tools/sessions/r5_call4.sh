#!/bin/bash
# round 5, call 4: kernel timeline of the forward search on the current sources + SQ counter groups of its kernels
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
PROF_GROUPS_FILTER='^SQ_WAVES|^SQ_INSTS_LDS' bash tools/profile.sh r5fwd python tools/fwd_loop.py lj 12 > gpurun_out/r5c4_prof.log 2>&1
(python tools/timeline.py gpurun_out/prof_r5fwd/kt -3; python tools/timeline.py gpurun_out/prof_r5fwd/kt -2) > gpurun_out/r5c4_timeline.txt 2>&1
cp gpurun_out/prof_r5fwd/summary.md gpurun_out/r5c4_summary.md
cp gpurun_out/prof_r5fwd/kt/p_kernel_stats.csv gpurun_out/r5c4_kernel_stats.csv 2>/dev/null
rm -rf gpurun_out/prof_r5fwd; el "profile"
cat gpurun_out/r5c4_timeline.txt; cat gpurun_out/r5c4_summary.md | cut -c1-300 | head -70
