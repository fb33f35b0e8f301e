#!/bin/bash
# round 6, call 42: STICKY levels of the many-levels body (a level whose workgroups each appended <= one block is expanded where it was appended;
# column indices loaded in front of the exchange): road stand-in, parity tests
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
ROAD_AB_CHECK=1 timeout 900 python tools/road_ab.py both 3 "-" "GRX_MID_STICKY=0" "-" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_c42_road_ab.txt
cat gpurun_out/r6_c42_road_ab.txt
(timeout 900 python -m pytest tests/test_sssp_gpu.py tests/test_mid_gpu.py tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --durations=3 -k "not twitter" > gpurun_out/r6_c42_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c42_pytest.log)
tail -4 gpurun_out/r6_c42_pytest.log
