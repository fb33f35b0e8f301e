#!/bin/bash
# round 4, call 6: the radix sort with LDS-staged output (tests + step timers at three digit widths)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_sort_gpu.py tests/test_pr_gpu.py tests/test_target_matrix_gpu.py::test_full_size_lj_pr -x -q > gpurun_out/r4c6_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r4c6_pytest.log)
tail -4 gpurun_out/r4c6_pytest.log
for b in 8 7 9; do echo "== GRX_SORT_BITS=$b"; GRX_SORT_BITS=$b timeout 200 python tools/prep_timing.py lj kron 2>&1 | grep -v amdgpu.ids | grep "transpose\|xcd layout\|XCD-blocked\|partition\|first call\|symmetry"; done > gpurun_out/r4c6_prep_timing.log 2>&1
cat gpurun_out/r4c6_prep_timing.log
