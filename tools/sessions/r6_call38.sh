#!/bin/bash
# round 6, call 38: relax scatter with the values stored from registers (79 KB of LDS, 64 VGPRs: two workgroups per CU), GRX_RSCATTER_DIRECT
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
(for g in lj kron; do timeout 600 python tools/ab_ssspw.py $g 7 "-" "GRX_RSCATTER_DIRECT=1" "-" "GRX_RSCATTER_DIRECT=1" 2>&1 | grep -v amdgpu.ids; done) > gpurun_out/r6_c38_rscatter_direct.txt
cat gpurun_out/r6_c38_rscatter_direct.txt
(GRX_RSCATTER_DIRECT=1 timeout 600 python -m pytest tests/test_relax_gpu.py tests/test_sssp_gpu.py -m gpu -q -x -k "not road" > gpurun_out/r6_c38_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c38_pytest.log); tail -3 gpurun_out/r6_c38_pytest.log
