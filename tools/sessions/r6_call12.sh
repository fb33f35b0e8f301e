#!/bin/bash
# round 6, call 12: second build of the many-levels chunk for blocks of <= S x 256 out-edges: S = 3 (default), 2, 4, none
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in "" _one _s2 _s4; do
  echo "### libgrx$v.so"
  GRX_LIB_PATH=$PWD/gunrock_amd/libgrx$v.so timeout 900 python tools/road_ab.py both 3 "-" 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r6_c12_road_ab.txt 2>&1
cat gpurun_out/r6_c12_road_ab.txt
(timeout 900 python -m pytest tests/test_sssp_gpu.py tests/test_mid_gpu.py tests/test_bfs_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --durations=5 -k "not full_size" > gpurun_out/r6_c12_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c12_pytest.log)
tail -5 gpurun_out/r6_c12_pytest.log
