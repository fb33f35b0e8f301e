#!/bin/bash
# round 6, call 11: which of the body's changes cost the weighted road search 10 ms: guards on dead item groups (libgrx_nokmax.so: off)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/road_ab.py both 3 "-" > gpurun_out/r6_c11_road_ab.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6_c11_road_ab.txt
GRX_LIB_PATH=$PWD/gunrock_amd/libgrx_nokmax.so timeout 900 python tools/road_ab.py both 3 "-" > gpurun_out/r6_c11_road_ab_nokmax.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6_c11_road_ab_nokmax.txt
