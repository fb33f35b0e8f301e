#!/bin/bash
# call 14: second bottom-up body -- parity tests first, then the A/B on the three scale-free stand-ins
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bfs_gpu.py -x -q -m gpu > gpurun_out/c14_pytest_bfs.log 2>&1; echo "pytest bfs rc $?" 
tail -3 gpurun_out/c14_pytest_bfs.log
for g in lj kron twitter; do
  timeout 600 python tools/ab_bu.py $g 20 > gpurun_out/c14_ab_bu_$g.log 2>&1; echo "ab $g rc $?"
  cat gpurun_out/c14_ab_bu_$g.log
done
