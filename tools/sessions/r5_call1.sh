#!/bin/bash
# round 5, call 1: A/B of two library builds -- libgrx_r4.so (the round-4 sources) against the working tree: scatter without the
# loop-top vmcnt(0) / hand-issued ticket atomic, sweep with a ring of candidate loads, exact launch schedule for a repeated source --
# phase clocks of the new scatter / sweep, and the GPU tests of the touched paths.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
for lib in r4 ""; do
  if [ -n "$lib" ]; then export GRX_LIB_PATH=$PWD/gunrock_amd/libgrx_$lib.so; else unset GRX_LIB_PATH; fi
  timeout 170 python tools/ab_r5.py lj 20 bfs,do,ssspw,multi 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5c1_ab_lj.log
  el "ab lj ${lib:-new}"
done
unset GRX_LIB_PATH
for lv in 1 2; do GRX_BIN_DEBUG=$lv timeout 90 python tools/bin_debug.py lj 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> gpurun_out/r5c1_bin_debug_lj.log; done
el "bin_debug"
(timeout 420 python -m pytest -q -x -m gpu tests/test_bfs_gpu.py tests/test_relax_gpu.py tests/test_target_matrix_gpu.py \
   --deselect tests/test_bfs_gpu.py::test_full_size_twitter_standin_properties --durations=8 > gpurun_out/r5c1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c1_pytest.log)
el "pytest"
cat gpurun_out/r5c1_ab_lj.log; cat gpurun_out/r5c1_bin_debug_lj.log | cut -c1-260; tail -15 gpurun_out/r5c1_pytest.log
