#!/bin/bash
# Round 3, GPU call 4: dynamic (per-XCD ticket) unit hand-out in the second scatter, sweep parts per bin; full pytest.
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/ab_r3.py lj 20 > gpurun_out/c4_ab_lj.log 2>&1
for L in 1 2; do GRX_BIN_DEBUG=$L timeout 200 python tools/bin_debug.py lj > gpurun_out/c4_bin_debug_l$L.log 2>&1; done
timeout 300 python tools/ab_r3.py kron 10 > gpurun_out/c4_ab_kron.log 2>&1
timeout 400 python tools/ab_r3.py twitter 5 > gpurun_out/c4_ab_twitter.log 2>&1
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/c4_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c4_pytest.log
tail -4 gpurun_out/c4_pytest.log; grep -v amdgpu gpurun_out/c4_ab_lj.log | cut -c1-330
