#!/bin/bash
# round 6, call 43: 64 resident workgroups for the BFS policies only (near-far capped at 32), on the final many-levels body
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" _w64 "" _w64; do
  echo "### libgrx$v.so"
  GRX_LIB_PATH=$PWD/gunrock_amd/libgrx$v.so timeout 900 python tools/road_ab.py both 3 "-" 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r6_c43_w64.txt 2>&1
cat gpurun_out/r6_c43_w64.txt
