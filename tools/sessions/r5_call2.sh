#!/bin/bash
# round 5, call 2: the working tree after the one-lane atomics were given opaque addresses (call 1: the hand-issued asm atomics
# faulted in the DBG and relax builds -- the compiler may copy the asm's destination register before the atomic has returned):
# A/B lines, phase clocks of scatter / sweep, tests of the binned paths
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 170 python tools/ab_r5.py lj 20 bfs,do,ssspw 2>&1 | grep -v amdgpu.ids > gpurun_out/r5c2_ab_lj.log
el "ab lj"
for lv in 1 2; do GRX_BIN_DEBUG=$lv timeout 90 python tools/bin_debug.py lj 2>&1 | grep -v amdgpu.ids | cut -c1-330 >> gpurun_out/r5c2_bin_debug_lj.log; done
el "bin_debug"
(timeout 420 python -m pytest -q -x -m gpu tests/test_relax_gpu.py tests/test_target_matrix_gpu.py tests/test_bfs_gpu.py \
   --deselect tests/test_bfs_gpu.py::test_full_size_twitter_standin_properties --durations=8 > gpurun_out/r5c2_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c2_pytest.log)
el "pytest"
cat gpurun_out/r5c2_ab_lj.log; cat gpurun_out/r5c2_bin_debug_lj.log | cut -c1-300; tail -15 gpurun_out/r5c2_pytest.log
