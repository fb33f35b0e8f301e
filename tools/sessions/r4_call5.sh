#!/bin/bash
# round 4, call 5: block kernel with coalesced write-back (sweep), sort digit width, device-side PR partition, pair stores
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_pr_gpu.py tests/test_sort_gpu.py tests/test_block_gpu.py tests/test_target_matrix_gpu.py::test_full_size_lj_pr -x -q > gpurun_out/r4c5_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r4c5_pytest.log)
tail -4 gpurun_out/r4c5_pytest.log
(timeout 600 python tools/ab_block.py 4894 3 > gpurun_out/r4c5_ab_block.log 2>&1; echo "rc $?" >> gpurun_out/r4c5_ab_block.log)
cat gpurun_out/r4c5_ab_block.log | cut -c1-250
for b in 9 8 7 6; do echo "== GRX_SORT_BITS=$b"; GRX_SORT_BITS=$b timeout 200 python tools/prep_timing.py lj kron 2>&1 | grep -v amdgpu.ids | grep "transpose\|XCD-blocked\|partition\|first call"; done > gpurun_out/r4c5_prep_timing.log 2>&1
cat gpurun_out/r4c5_prep_timing.log
(timeout 300 python tools/ab_r4.py lj 20 > gpurun_out/r4c5_ab_lj.log 2>&1; echo "rc $?" >> gpurun_out/r4c5_ab_lj.log)
(timeout 300 python tools/ab_r4.py kron 10 > gpurun_out/r4c5_ab_kron.log 2>&1; echo "rc $?" >> gpurun_out/r4c5_ab_kron.log)
grep -v amdgpu.ids gpurun_out/r4c5_ab_lj.log | cut -c1-330; grep -v amdgpu.ids gpurun_out/r4c5_ab_kron.log | cut -c1-330
