#!/bin/bash
# round 4, call 4: block path with interleaved local numbering (sweep), preprocessing step times, PR layout A/B on LJ,
# quick regression of the suites the preprocessing changes touch
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python tools/ab_block.py 4894 3 > gpurun_out/r4c4_ab_block.log 2>&1; echo "rc $?" >> gpurun_out/r4c4_ab_block.log)
cat gpurun_out/r4c4_ab_block.log | cut -c1-250
(timeout 300 python tools/prep_timing.py lj kron > gpurun_out/r4c4_prep_timing.log 2>&1; echo "rc $?" >> gpurun_out/r4c4_prep_timing.log)
grep -v amdgpu.ids gpurun_out/r4c4_prep_timing.log
(timeout 200 python tools/ab_pr.py lj > gpurun_out/r4c4_ab_pr_lj.log 2>&1; echo "rc $?" >> gpurun_out/r4c4_ab_pr_lj.log)
grep -v amdgpu.ids gpurun_out/r4c4_ab_pr_lj.log
(timeout 900 python -m pytest tests/test_pr_gpu.py tests/test_sort_gpu.py tests/test_block_gpu.py tests/test_bfs_gpu.py -x -q > gpurun_out/r4c4_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r4c4_pytest.log)
tail -4 gpurun_out/r4c4_pytest.log
