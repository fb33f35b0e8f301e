#!/bin/bash
# round 5, call 24: direction-optimising search -- the tiny levels take over right behind the first top-down level that follows
# the bottom-up ones (they declined while ctrl.bu_R was still set); forward: last group = head alone, with forced mispredictions
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 500 python -m pytest -q -x -m gpu tests/test_bfs_gpu.py tests/test_target_matrix_gpu.py tests/test_fuzz_gpu.py tests/test_cli.py -k "not twitter and not sssp and not pr" > gpurun_out/r5c24_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c24_pytest.log)
el "pytest: $(tail -3 gpurun_out/r5c24_pytest.log | tr '\n' ' ')"
{
KT_DIR=do bash tools/kt_fat.sh do_lj
KT_DIR=do KT_GRAPH=kron bash tools/kt_fat.sh do_kron
KT_DIR=do KT_GRAPH=deep bash tools/kt_fat.sh do_deep
KT_GRAPH=deep bash tools/kt_fat.sh fwd_deep
bash tools/kt_fat.sh fwd_lj
echo "== lj"; timeout 100 python tools/ab_r5.py lj 20 bfs,do 2>&1 | grep -v amdgpu.ids | grep "^fwd default  \|^DO"
echo "== kron"; timeout 100 python tools/ab_r5.py kron 10 do 2>&1 | grep -v amdgpu.ids | grep "^DO"
} > gpurun_out/r5c24_ab.log 2>&1
el "ab"
cut -c1-420 gpurun_out/r5c24_ab.log; tail -5 gpurun_out/r5c24_pytest.log
