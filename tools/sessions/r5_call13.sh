#!/bin/bash
# round 5, call 13: the whole GPU suite (without the three 530 M-edge cases) after the robustness changes -- build lock, RAII scratch,
# retryable state words, forward fallback of a direction-optimising search -- with the new tests; A/B line
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 900 python -m pytest -q -x -m gpu tests --deselect tests/test_bfs_gpu.py::test_full_size_twitter_standin_properties \
   --deselect tests/test_distributed.py::test_c5_twitter_standin_two_ranks_one_gpu --deselect tests/test_distributed.py::test_c5_twitter_standin_eight_ranks_one_gpu \
   --durations=8 > gpurun_out/r5c13_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c13_pytest.log)
el "pytest"
timeout 170 python tools/ab_r5.py lj 20 bfs,do,ssspw 2>&1 | grep -v amdgpu.ids | cut -c1-330 > gpurun_out/r5c13_ab_lj.log
el "ab"
tail -16 gpurun_out/r5c13_pytest.log; cat gpurun_out/r5c13_ab_lj.log
