#!/bin/bash
# round 5, call 9: label stores of the sweep behind the emission's loads (with the frontier stores)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
{
bash tools/kt_fat.sh default
bash tools/kt_fat.sh r4cut GRX_BIN_UNIFORM=0
KT_GRAPH=kron bash tools/kt_fat.sh kron_default
KT_GRAPH=kron bash tools/kt_fat.sh kron_r4cut GRX_BIN_UNIFORM=0
} > gpurun_out/r5c9_kt.log 2>&1
el kt
timeout 170 python tools/ab_r5.py lj 20 bfs,do 2>&1 | grep -v amdgpu.ids > gpurun_out/r5c9_ab_lj.log
for lv in 1 2; do GRX_BIN_DEBUG=$lv timeout 90 python tools/bin_debug.py lj 2>&1 | grep -v amdgpu.ids | grep -A1 "^scatter\|^claim" | cut -c1-330 >> gpurun_out/r5c9_ab_lj.log; done
el "ab lj"
(timeout 300 python -m pytest -q -x -m gpu tests/test_bfs_gpu.py tests/test_target_matrix_gpu.py tests/test_relax_gpu.py tests/test_fuzz_gpu.py --deselect tests/test_bfs_gpu.py::test_full_size_twitter_standin_properties > gpurun_out/r5c9_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c9_pytest.log)
el "pytest"
cut -c1-400 gpurun_out/r5c9_kt.log; cut -c1-330 gpurun_out/r5c9_ab_lj.log; tail -5 gpurun_out/r5c9_pytest.log
