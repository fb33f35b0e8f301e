#!/bin/bash
# round 4, call 7: preprocessing after the tile-wise expand / aggregated histogram / device ranking / transpose-based
# symmetry handling: tests, step timers, DO A/B (CSR vs hubs-first transpose on symmetric graphs)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_sort_gpu.py tests/test_pr_gpu.py tests/test_bfs_gpu.py tests/test_target_matrix_gpu.py::test_full_size_lj_pr tests/test_fuzz_gpu.py -x -q > gpurun_out/r4c7_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r4c7_pytest.log)
tail -4 gpurun_out/r4c7_pytest.log
timeout 300 python tools/prep_timing.py lj kron 2>&1 | grep -v amdgpu.ids > gpurun_out/r4c7_prep_timing.log
cat gpurun_out/r4c7_prep_timing.log
for g in kron lj; do for v in 0 1; do echo "== $g GRX_BU_SYMMETRIC_CSR=$v"; GRX_BU_SYMMETRIC_CSR=$v timeout 200 python tools/run_algo.py bfs $g 12 8 merge_path optimized 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-260; done; done > gpurun_out/r4c7_do_csr_vs_transpose.log 2>&1
cat gpurun_out/r4c7_do_csr_vs_transpose.log
