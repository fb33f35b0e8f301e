#!/bin/bash
# round 6, call 16: thread-mapped blocks in the many-levels body (rows of <= 4 out-edges): road stand-in, sub-phase clocks, parity tests
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
ROAD_AB_CHECK=1 timeout 900 python tools/road_ab.py both 3 "-" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_c16_road_ab.txt
cat gpurun_out/r6_c16_road_ab.txt
GRX_LIB_PATH=$PWD/gunrock_amd/libgrx_fine.so GRX_MID_DEBUG=1 timeout 600 python tools/mid_phases.py both 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_c16_mid_phases.txt
cat gpurun_out/r6_c16_mid_phases.txt
(timeout 900 python -m pytest tests/test_sssp_gpu.py tests/test_mid_gpu.py tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --durations=5 -k "not full_size" > gpurun_out/r6_c16_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c16_pytest.log)
tail -5 gpurun_out/r6_c16_pytest.log
