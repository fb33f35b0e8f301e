#!/bin/bash
# round 6, call 10: many-levels body with direct appends / one LDS atomic per wave / dead item groups skipped, near-far inside it
# with carried labels (one atomic per relaxed edge); parity tests, then the road stand-in, weighted and unit
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 900 python -m pytest tests/test_sssp_gpu.py tests/test_mid_gpu.py tests/test_relax_gpu.py tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --durations=5 -k "not full_size" > gpurun_out/r6_c10_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c10_pytest.log); el pytest
tail -12 gpurun_out/r6_c10_pytest.log
ROAD_AB_CHECK=1 timeout 900 python tools/road_ab.py both 3 "-" "GRX_NF_FOLD=0" > gpurun_out/r6_c10_road_ab.txt 2>&1; el "road_ab rc $?"
grep -v amdgpu.ids gpurun_out/r6_c10_road_ab.txt
GRX_LIB_PATH=$PWD/gunrock_amd/libgrx_nocarry.so timeout 900 python tools/road_ab.py w 3 "-" > gpurun_out/r6_c10_road_ab_nocarry.txt 2>&1; el "road_ab nocarry rc $?"
grep -v amdgpu.ids gpurun_out/r6_c10_road_ab_nocarry.txt
