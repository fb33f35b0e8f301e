#!/bin/bash
# round 5, call 11: granule table as one-add deltas (balanced cut, default again) against uniform bins; relax path tests
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
{
bash tools/kt_fat.sh default_balanced_cut
bash tools/kt_fat.sh uniform16 GRX_BIN_UNIFORM=1
KT_GRAPH=kron bash tools/kt_fat.sh kron_default
KT_GRAPH=kron bash tools/kt_fat.sh kron_uniform GRX_BIN_UNIFORM=1
} > gpurun_out/r5c11_kt.log 2>&1
el kt
timeout 170 python tools/ab_r5.py lj 20 bfs,do,ssspw 2>&1 | grep -v amdgpu.ids > gpurun_out/r5c11_ab_lj.log
el "ab lj"
(timeout 300 python -m pytest -q -x -m gpu tests/test_bfs_gpu.py tests/test_target_matrix_gpu.py tests/test_relax_gpu.py tests/test_sssp_gpu.py --deselect tests/test_bfs_gpu.py::test_full_size_twitter_standin_properties > gpurun_out/r5c11_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c11_pytest.log)
el "pytest"
cut -c1-400 gpurun_out/r5c11_kt.log; cut -c1-330 gpurun_out/r5c11_ab_lj.log; tail -5 gpurun_out/r5c11_pytest.log
