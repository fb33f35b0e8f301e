#!/bin/bash
# round 5, call 30: the concurrent-contexts tests repeated on the committed sources (a failure was seen once WITH the rejected
# mode-4 change built in), then the BFS suite
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
for i in 1 2 3 4 5 6; do
  timeout 200 python -m pytest -q -x -m gpu tests/test_bfs_gpu.py -k "two_contexts or race_on_a_fresh" 2>&1 | tail -1
done > gpurun_out/r5c30_repeat.log 2>&1
el "repeat"
(timeout 600 python -m pytest -q -m gpu tests/test_bfs_gpu.py tests/test_target_matrix_gpu.py tests/test_fuzz_gpu.py -k "not twitter" > gpurun_out/r5c30_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c30_pytest.log)
el "pytest: $(tail -3 gpurun_out/r5c30_pytest.log | tr '\n' ' ')"
cat gpurun_out/r5c30_repeat.log; tail -4 gpurun_out/r5c30_pytest.log | cut -c1-200
