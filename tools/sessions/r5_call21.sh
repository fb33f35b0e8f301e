#!/bin/bash
# round 5, call 21: weighted SSSP with uniform relax bins (no granule table in the value scatter) against the balanced cut
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for g in lj kron; do for u in 0 1; do
  echo "== $g GRX_RBIN_UNIFORM=$u"; GRX_RBIN_UNIFORM=$u timeout 200 python tools/ab_r5.py $g 20 ssspw 2>&1 | grep -v amdgpu.ids | grep sssp
done; done
GRX_RBIN_UNIFORM=1 bash tools/kt_last.sh sssp_w_lj_uniform fill_f32 -- python tools/sssp_loop.py lj 5 | head -8
} > gpurun_out/r5c21_sssp.log 2>&1
(GRX_RBIN_UNIFORM=1 timeout 300 python -m pytest -q -x -m gpu tests/test_relax_gpu.py tests/test_target_matrix_gpu.py -k "sssp or relax or binned" > gpurun_out/r5c21_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c21_pytest.log)
cut -c1-250 gpurun_out/r5c21_sssp.log; tail -4 gpurun_out/r5c21_pytest.log
