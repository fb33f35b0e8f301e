#!/bin/bash
# Round 3, GPU call 3: phase timelines of the second scatter / sweep on the two fat levels, the PR + SSSP tests the -x stop
# of call 2 skipped, the generic operators with the phased expansion.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
for L in 1 2; do
  GRX_BIN_DEBUG=$L timeout 200 python tools/bin_debug.py lj > gpurun_out/c3_bin_debug_l$L.log 2>&1
  GRX_BIN_DEBUG=$L GRX_SW2_ITEMS=512 timeout 200 python tools/bin_debug.py lj > gpurun_out/c3_bin_debug_l${L}_items512.log 2>&1
done
timeout 900 python -m pytest tests/test_pr_gpu.py tests/test_sssp_gpu.py -x -q -m gpu > gpurun_out/c3_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c3_pytest.log
timeout 600 bash tools/bench_generic.sh > gpurun_out/c3_generic.log 2>&1
tail -3 gpurun_out/c3_pytest.log; cut -c1-600 gpurun_out/c3_bin_debug_l1.log | grep -v amdgpu; cat gpurun_out/generic_bfs.log
