#!/bin/bash
# round 4, call 23: sweeps of EXISTING knobs on the final sources (sweep geometry of the binned BFS levels; threshold / parts of the
# binned relaxation), and the unit-weight SSSP whose inner search returns on the published end (one blocking wait instead of two)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 150 python tools/ab_r4b.py lj 15 unit,knobs 2>&1 | grep -v amdgpu.ids > gpurun_out/r4c23_ab_lj.log
(timeout 100 python -m pytest -q -x -m gpu tests/test_sssp_gpu.py::test_pattern_graph_equals_bfs_depths tests/test_sssp_gpu.py::test_goldens \
   tests/test_target_matrix_gpu.py::test_uniform_weights_other_than_one tests/test_target_matrix_gpu.py::test_full_size_lj_sssp \
   tests/test_sssp_gpu.py::test_full_size_road_standin_unit_weights > gpurun_out/r4c23_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4c23_pytest.log)
cat gpurun_out/r4c23_ab_lj.log; tail -3 gpurun_out/r4c23_pytest.log
