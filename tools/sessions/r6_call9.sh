#!/bin/bash
# round 6, call 9: where the weighted road search spends its time, per launch group (tools/road_prof.py)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/road_prof.py w > gpurun_out/r6_c9_road_prof_w.txt 2>&1; echo "rc $?"
grep -v amdgpu.ids gpurun_out/r6_c9_road_prof_w.txt
timeout 600 python tools/road_prof.py unit > gpurun_out/r6_c9_road_prof_unit.txt 2>&1; echo "rc $?"
grep -v amdgpu.ids gpurun_out/r6_c9_road_prof_unit.txt
