#!/bin/bash
# round 5, call 12: the head of the level behind a fat level run by the last workgroup of the sweep (exact schedule)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
{
bash tools/kt_fat.sh default
bash tools/kt_fat.sh heads_kept GRX_SWEEP_PLANS=0
KT_GRAPH=kron bash tools/kt_fat.sh kron_default
} > gpurun_out/r5c12_kt.log 2>&1
el kt
timeout 170 python tools/ab_r5.py lj 20 bfs,do,multi 2>&1 | grep -v amdgpu.ids > gpurun_out/r5c12_ab_lj.log
el "ab lj"
(timeout 400 python -m pytest -q -x -m gpu tests/test_bfs_gpu.py tests/test_target_matrix_gpu.py tests/test_fuzz_gpu.py tests/test_sssp_gpu.py --deselect tests/test_bfs_gpu.py::test_full_size_twitter_standin_properties > gpurun_out/r5c12_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c12_pytest.log)
el "pytest"
cut -c1-400 gpurun_out/r5c12_kt.log; cut -c1-330 gpurun_out/r5c12_ab_lj.log; tail -5 gpurun_out/r5c12_pytest.log
