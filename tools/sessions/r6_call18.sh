#!/bin/bash
# round 6, call 18: many-levels body as it goes in (thread-mapped rows <= 8, two barriers per level less): road stand-in, parity incl. full-size road tests
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python tools/road_ab.py both 3 "-" "GRX_NF_FOLD=0" 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_c18_road_ab.txt
cat gpurun_out/r6_c18_road_ab.txt
(timeout 1500 python -m pytest tests/test_sssp_gpu.py tests/test_mid_gpu.py tests/test_bfs_gpu.py tests/test_fuzz_gpu.py tests/test_target_matrix_gpu.py tests/test_relax_gpu.py tests/test_block_gpu.py -m gpu -q -x --durations=5 > gpurun_out/r6_c18_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c18_pytest.log)
tail -9 gpurun_out/r6_c18_pytest.log
