#!/bin/bash
# round 5, call 5: uniform bins (no granule table in the scatter), wave sums on the DPP path, lane-0 broadcasts by v_readlane
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
timeout 170 python tools/ab_r5.py lj 20 bfs,do,ssspw 2>&1 | grep -v amdgpu.ids > gpurun_out/r5c5_ab_lj.log
for lv in 1 2; do GRX_BIN_DEBUG=$lv timeout 90 python tools/bin_debug.py lj 2>&1 | grep -v amdgpu.ids | grep -A1 "^scatter\|^claim" | cut -c1-330 >> gpurun_out/r5c5_ab_lj.log; done
el "ab lj"
(timeout 600 python -m pytest -q -x -m gpu tests --deselect tests/test_bfs_gpu.py::test_full_size_twitter_standin_properties \
   --deselect tests/test_distributed.py::test_c5_twitter_standin_two_ranks_one_gpu --deselect tests/test_distributed.py::test_c5_twitter_standin_eight_ranks_one_gpu \
   --durations=5 > gpurun_out/r5c5_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c5_pytest.log)
el "pytest"
cut -c1-330 gpurun_out/r5c5_ab_lj.log; tail -12 gpurun_out/r5c5_pytest.log
