#!/bin/bash
# round 6, call 14: 64 resident workgroups on the home XCD (two per CU) against 32; bucket width 64 (new default) around its neighbours
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" _w64; do
  echo "### libgrx$v.so"
  GRX_LIB_PATH=$PWD/gunrock_amd/libgrx$v.so timeout 900 python tools/road_ab.py both 3 "-" "GRX_NF_DELTA_SCALE=0.8" "GRX_NF_DELTA_SCALE=1.25" 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r6_c14_road_ab.txt 2>&1
cat gpurun_out/r6_c14_road_ab.txt
