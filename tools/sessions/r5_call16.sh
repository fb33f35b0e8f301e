#!/bin/bash
# round 5, call 17: PageRank with the row blocks of the XCD-blocked layout sorted by source (position stream), against unsorted and the round-4 build
# forward BFS on the deep stand-in with the 2^21 binning threshold
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
{
for g in kron lj; do
  GRX_LIB_PATH=$PWD/gunrock_amd/libgrx_r4.so timeout 120 python tools/ab_pr5.py $g 2>&1 | grep -v amdgpu.ids
  timeout 120 python tools/ab_pr5.py $g "" 8 7 6 4 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r5c17_pr.log 2>&1
el pr
(timeout 300 python -m pytest -q -x -m gpu tests/test_pr_gpu.py tests/test_target_matrix_gpu.py tests/test_sort_gpu.py tests/test_distributed_pr.py > gpurun_out/r5c17_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c17_pytest.log)
el pytest
bash tools/kt_last.sh fwd_deep reset_seed -- python tools/fwd_loop.py deep 10 fwd > gpurun_out/r5c17_deep.log 2>&1
cat gpurun_out/r5c17_pr.log; tail -5 gpurun_out/r5c17_pytest.log; head -3 gpurun_out/r5c17_deep.log
