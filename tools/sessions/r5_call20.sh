#!/bin/bash
# round 5, call 20: PageRank hot head: smaller sizes (more resident workgroups)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
{
for g in kron lj; do
  timeout 200 python tools/ab_pr5.py $g "" hot=0 hot=512 hot=1024 hot=2048 hot=3072 hot=4096 hot=5120 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r5c20_pr.log 2>&1
el pr
(timeout 300 python -m pytest -q -x -m gpu tests/test_pr_gpu.py tests/test_target_matrix_gpu.py tests/test_distributed_pr.py > gpurun_out/r5c20_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c20_pytest.log)
el pytest
cat gpurun_out/r5c20_pr.log; tail -5 gpurun_out/r5c20_pytest.log
