#!/bin/bash
# round 5, call 3: sweep stream with the lean interior body (ring of 2 = default, ring of 3 = libgrx_ring3.so), 256 sweep items
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
for lib in "" ring3; do
  if [ -n "$lib" ]; then export GRX_LIB_PATH=$PWD/gunrock_amd/libgrx_$lib.so; else unset GRX_LIB_PATH; fi
  timeout 170 python tools/ab_r5.py lj 20 bfs 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5c3_ab_lj.log
  for lv in 1 2; do GRX_BIN_DEBUG=$lv timeout 90 python tools/bin_debug.py lj 2>&1 | grep -v amdgpu.ids | grep -A1 "^claim" | cut -c1-330 >> gpurun_out/r5c3_ab_lj.log; done
  el "ab lj ${lib:-new}"
done
unset GRX_LIB_PATH
(timeout 420 python -m pytest -q -x -m gpu tests/test_bfs_gpu.py tests/test_target_matrix_gpu.py::test_full_size_lj_bfs_depths_equal_the_oracle_array \
   --deselect tests/test_bfs_gpu.py::test_full_size_twitter_standin_properties > gpurun_out/r5c3_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c3_pytest.log)
el "pytest"
cut -c1-330 gpurun_out/r5c3_ab_lj.log; tail -4 gpurun_out/r5c3_pytest.log
