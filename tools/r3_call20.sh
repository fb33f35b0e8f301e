#!/bin/bash
# call 20: launch-group hint for the binned kernels, lighter Python wrapper: parity, then the A/B of the forward and DO searches
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bfs_gpu.py tests/test_mid_gpu.py -x -q -m gpu > gpurun_out/c20_pytest.log 2>&1; echo "pytest rc $?"
tail -2 gpurun_out/c20_pytest.log
for g in lj twitter; do
  timeout 600 python tools/ab_bu.py $g 20 2>&1 | grep -v amdgpu.ids > gpurun_out/c20_ab_bu_$g.log; echo "ab $g rc $?"
  grep -v "^source" gpurun_out/c20_ab_bu_$g.log
done
GRX_BIN_HINT=0 timeout 300 python tools/ab_r3.py lj 20 2>&1 | grep "default\|workload" > gpurun_out/c20_ab_fwd_nohint_lj.log
timeout 300 python tools/ab_r3.py lj 20 2>&1 | grep "default\|workload" > gpurun_out/c20_ab_fwd_hint_lj.log
cat gpurun_out/c20_ab_fwd_nohint_lj.log gpurun_out/c20_ab_fwd_hint_lj.log
timeout 300 python tools/ab_r3.py kron 20 2>&1 | grep "default\|workload" > gpurun_out/c20_ab_fwd_hint_kron.log; cat gpurun_out/c20_ab_fwd_hint_kron.log
