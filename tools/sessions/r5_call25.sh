#!/bin/bash
# round 5, call 25: PageRank -- number of source blocks of the XCD-blocked layout (8 = one per XCD, 4, 2, 1) on the LJ and kron
# stand-ins; parity of the test suite with 2 blocks; the repeated direction-optimising search (head alone in its last group)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
{
for g in lj kron; do for xb in 8 4 2 1; do
  GRX_PR_XB=$xb timeout 120 python tools/ab_pr5.py $g "" "hot=12288" 2>&1 | grep -v amdgpu.ids | grep "^lib" | sed "s/^lib libgrx.so/xb=$xb/"
done; done
} > gpurun_out/r5c25_pr_xb.log 2>&1
el "pr xb"
(GRX_PR_XB=2 timeout 300 python -m pytest -q -x -m gpu tests/test_pr_gpu.py -k "not kron_c4" > gpurun_out/r5c25_pytest_pr.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c25_pytest_pr.log)
el "pytest pr (2 blocks): $(tail -3 gpurun_out/r5c25_pytest_pr.log | tr '\n' ' ')"
(timeout 300 python -m pytest -q -x -m gpu tests/test_bfs_gpu.py > gpurun_out/r5c25_pytest_bfs.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c25_pytest_bfs.log)
el "pytest bfs: $(tail -3 gpurun_out/r5c25_pytest_bfs.log | tr '\n' ' ')"
{
KT_DIR=do bash tools/kt_fat.sh do_lj
echo "== lj"; timeout 100 python tools/ab_r5.py lj 20 do 2>&1 | grep -v amdgpu.ids | grep "^DO"
} > gpurun_out/r5c25_do.log 2>&1
timeout 200 bash tools/bench_refalg.sh > gpurun_out/r5c25_refalg.log 2>&1; el "refalg"
cut -c1-300 gpurun_out/r5c25_pr_xb.log; cut -c1-400 gpurun_out/r5c25_do.log; cat gpurun_out/refalg_times.txt
