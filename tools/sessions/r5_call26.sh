#!/bin/bash
# round 5, call 26: parity after the schedule changes (head-alone last group only where the previous search ended in a head;
# direction-optimising tail; PageRank with 4 source blocks on sparse graphs) + the reference's k-core / PPR on both operator sets
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 700 python -m pytest -q -m gpu tests/test_bfs_gpu.py tests/test_pr_gpu.py tests/test_target_matrix_gpu.py tests/test_sssp_gpu.py tests/test_fuzz_gpu.py -k "not twitter" > gpurun_out/r5c26_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c26_pytest.log)
el "pytest: $(tail -3 gpurun_out/r5c26_pytest.log | tr '\n' ' ')"
timeout 300 bash tools/bench_refalg.sh > gpurun_out/r5c26_refalg.log 2>&1; el "refalg"
{
timeout 100 python tools/ab_pr5.py lj "" 2>&1 | grep "^lib"
timeout 100 python tools/ab_r5.py lj 20 bfs,do 2>&1 | grep -v amdgpu.ids | grep "^fwd default  \|^DO"
KT_DIR=do bash tools/kt_fat.sh do_lj
bash tools/kt_fat.sh fwd_lj
} > gpurun_out/r5c26_ab.log 2>&1
el "ab"
tail -12 gpurun_out/r5c26_pytest.log | cut -c1-200; cat gpurun_out/refalg_times.txt; cut -c1-330 gpurun_out/r5c26_ab.log
