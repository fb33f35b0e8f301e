#!/bin/bash
# round 4, call 14: sub-counters in the relax scatter (tests + A/B), then the default bench run (rehearsal of the final one)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_relax_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -4 > gpurun_out/r4c14_pytest.log; cat gpurun_out/r4c14_pytest.log
for g in lj kron; do
  timeout 600 python tools/ab_relax.py $g "" GRX_RBIN_SUB=0 "" GRX_RBIN_SUB=0 2>&1 | grep -v amdgpu.ids | cut -c1-420 > gpurun_out/r4c14_ab_sub_$g.log; cat gpurun_out/r4c14_ab_sub_$g.log
done
S=$(date +%s); timeout 900 python bench.py > gpurun_out/r4c14_bench.log 2> gpurun_out/r4c14_bench.err; echo "rc $? seconds $(( $(date +%s) - S ))"; tail -1 gpurun_out/r4c14_bench.log | cut -c1-4200
cp gpurun_out/bench_detail.json gpurun_out/r4c14_bench_detail.json
