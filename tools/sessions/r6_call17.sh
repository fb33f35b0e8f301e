#!/bin/bash
# round 6, call 17: thread-mapped blocks with 64 resident workgroups; rows of <= 8 thread-mapped; none (TM off) -- same box, same run
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" _w64 _tm8 _tm0; do
  echo "### libgrx$v.so"
  GRX_LIB_PATH=$PWD/gunrock_amd/libgrx$v.so timeout 900 python tools/road_ab.py both 3 "-" 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r6_c17_road_ab.txt 2>&1
cat gpurun_out/r6_c17_road_ab.txt
