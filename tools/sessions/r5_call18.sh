#!/bin/bash
# round 5, call 18: every counter group of tools/profile.sh on the PageRank run of the kron stand-in (what bounds pr_pull_xcd_kernel?)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/profile.sh r5pr python tools/ab_pr5.py kron > gpurun_out/r5c18_prof.log 2>&1
cp gpurun_out/prof_r5pr/summary.md gpurun_out/r5c18_pr_summary.md; rm -rf gpurun_out/prof_r5pr
grep -n "pr_pull_xcd" -A 60 gpurun_out/r5c18_pr_summary.md | head -110
