#!/bin/bash
# round 4, call 8: sub-counters per bin (A/B on LJ, kron, twitter) + the binned-level tests
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -x -q > gpurun_out/r4c8_pytest.log 2>&1; echo "rc $?" >> gpurun_out/r4c8_pytest.log)
tail -3 gpurun_out/r4c8_pytest.log
for g in lj kron twitter; do (timeout 400 python tools/ab_r4.py $g 15 > gpurun_out/r4c8_ab_$g.log 2>&1; echo "rc $?" >> gpurun_out/r4c8_ab_$g.log); grep -v amdgpu.ids gpurun_out/r4c8_ab_$g.log | cut -c1-330; done
