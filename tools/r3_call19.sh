#!/bin/bash
# call 19: second bottom-up body with workgroup-fetched slot words, no row offsets in the round, 16-byte index loads in the deferred pass
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bfs_gpu.py -x -q -m gpu > gpurun_out/c19_pytest_bfs.log 2>&1; echo "pytest bfs rc $?"
tail -2 gpurun_out/c19_pytest_bfs.log
for g in lj kron twitter; do
  timeout 600 python tools/ab_bu.py $g 20 2>&1 | grep -v amdgpu.ids > gpurun_out/c19_ab_bu_$g.log; echo "ab $g rc $?"
  cat gpurun_out/c19_ab_bu_$g.log
done
for l in 1 2 3; do GRX_BU_DEBUG=$l timeout 300 python tools/bu_debug.py lj 2>&1 | grep -v amdgpu.ids; done > gpurun_out/c19_bu_debug_lj.log
cat gpurun_out/c19_bu_debug_lj.log
