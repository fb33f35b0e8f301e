#!/bin/bash
# Round 3, GPU call 7: column-index warming in the second scatter (A/B), generic operators with the bounded owner search.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/ab_r3.py lj 20 > gpurun_out/c7_ab_lj.log 2>&1
timeout 300 python tools/ab_r3.py kron 10 > gpurun_out/c7_ab_kron.log 2>&1
timeout 400 python tools/ab_r3.py twitter 5 > gpurun_out/c7_ab_twitter.log 2>&1
for L in 1 2; do GRX_BIN_DEBUG=$L timeout 200 python tools/bin_debug.py lj > gpurun_out/c7_bin_debug_l$L.log 2>&1; done
timeout 600 bash tools/bench_generic.sh > gpurun_out/c7_generic.log 2>&1
timeout 300 bin/test_operators > gpurun_out/c7_test_operators.log 2>&1; echo "test_operators rc $?" >> gpurun_out/c7_test_operators.log
grep -hv amdgpu gpurun_out/c7_ab_lj.log gpurun_out/c7_ab_kron.log gpurun_out/c7_ab_twitter.log | cut -c1-300; cat gpurun_out/generic_bfs.log; tail -3 gpurun_out/c7_test_operators.log
