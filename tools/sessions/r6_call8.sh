#!/bin/bash
# round 6, call 8: the next bucket of a near-far search pulled out of the pile INSIDE the many-levels launch (GRX_NF_FOLD):
# parity tests, then the road stand-in with the fold off / on
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 600 python -m pytest tests/test_sssp_gpu.py tests/test_mid_gpu.py tests/test_relax_gpu.py -m gpu -q -x --durations=5 -k "not full_size" > gpurun_out/r6_c8_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c8_pytest.log); el pytest
tail -12 gpurun_out/r6_c8_pytest.log
ROAD_AB_CHECK=1 timeout 900 python tools/road_ab.py w 3 "-" "GRX_NF_FOLD=0" "GRX_NF_FOLD=4096" "GRX_NF_FOLD=524288" > gpurun_out/r6_c8_road_ab.txt 2>&1; el "road_ab rc $?"
cat gpurun_out/r6_c8_road_ab.txt | grep -v amdgpu.ids
