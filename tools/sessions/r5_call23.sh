#!/bin/bash
# round 5, call 23: reset + seed + level 0 of a forward search in ONE launch (bfs_fwd_start_kernel) and the last group of a repeated
# search as its head alone -- parity first, then kernel sequences and step times with each switch off
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 400 python -m pytest -q -x -m gpu tests/test_bfs_gpu.py tests/test_target_matrix_gpu.py tests/test_fuzz_gpu.py -k "not twitter and not sssp and not pr" > gpurun_out/r5c23_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c23_pytest.log)
el "pytest: $(tail -3 gpurun_out/r5c23_pytest.log | tr '\n' ' ')"
{
bash tools/kt_fat.sh default
bash tools/kt_fat.sh two_launches GRX_FWD_START=0
bash tools/kt_fat.sh last_group_full GRX_LAST_HEAD_ONLY=0
KT_GRAPH=kron bash tools/kt_fat.sh kron_default
KT_GRAPH=kron bash tools/kt_fat.sh kron_two_launches GRX_FWD_START=0
for v in "" "GRX_FWD_START=0" "GRX_LAST_HEAD_ONLY=0" "GRX_FWD_START=0 GRX_LAST_HEAD_ONLY=0" ""; do
  echo "== lj $v"; env $v GRX_KEEP=GRX_FWD_START,GRX_LAST_HEAD_ONLY timeout 100 python tools/ab_r5.py lj 20 bfs 2>&1 | grep -v amdgpu.ids | grep "^fwd" | head -1
done
echo "== kron"; timeout 100 python tools/ab_r5.py kron 10 bfs 2>&1 | grep -v amdgpu.ids | grep "^fwd" | head -1
echo "== kron GRX_FWD_START=0"; GRX_KEEP=GRX_FWD_START GRX_FWD_START=0 timeout 100 python tools/ab_r5.py kron 10 bfs 2>&1 | grep -v amdgpu.ids | grep "^fwd" | head -1
} > gpurun_out/r5c23_ab.log 2>&1
el "ab"
cut -c1-420 gpurun_out/r5c23_ab.log; tail -5 gpurun_out/r5c23_pytest.log
