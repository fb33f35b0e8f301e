#!/bin/bash
# round 4, call 12: binned relaxation with its defaults -- SSSP suites + fuzz, then LJ / kron timings
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_relax_gpu.py tests/test_sssp_gpu.py tests/test_mid_gpu.py tests/test_target_matrix_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -8 > gpurun_out/r4c12_pytest.log; cat gpurun_out/r4c12_pytest.log
for g in lj kron; do
  timeout 600 python tools/ab_relax.py $g "" GRX_RBIN_PARTS=128 GRX_RBIN_PARTS=512 2>&1 | grep -v amdgpu.ids | cut -c1-700 > gpurun_out/r4c12_ab_relax_$g.log; cat gpurun_out/r4c12_ab_relax_$g.log
done
