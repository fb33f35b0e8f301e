#!/bin/bash
# round 4, call 17: software-pipelined relax scatter + lazy emission in the relax sweep -- tests, A/B (GRX_RBIN_PIPE=0 = the BFS scatter with values)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_relax_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/r4c17_pytest.log; cat gpurun_out/r4c17_pytest.log
for g in lj kron; do
  timeout 600 python tools/ab_relax.py $g GRX_RBIN_PIPE=0 "" GRX_RBIN_PIPE=0 "" 2>&1 | grep -v amdgpu.ids | cut -c1-420 > gpurun_out/r4c17_ab_pipe_$g.log; cat gpurun_out/r4c17_ab_pipe_$g.log
done
