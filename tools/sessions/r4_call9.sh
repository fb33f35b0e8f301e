#!/bin/bash
# round 4, call 9: binned relaxation of weighted SSSP -- parity tests, then the A/B on the LJ / kron stand-ins
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_relax_gpu.py tests/test_sssp_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r4c9_pytest.log; cat gpurun_out/r4c9_pytest.log
for g in lj kron; do
  timeout 600 python tools/ab_relax.py $g > gpurun_out/r4c9_ab_relax_$g.log 2>&1; echo "rc $?"; cat gpurun_out/r4c9_ab_relax_$g.log | cut -c1-900
done
