#!/bin/bash
# round 6, call 31: the many-levels tests with the near-far refill configurations added
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_mid_gpu.py -m gpu -q -x --durations=5 > gpurun_out/r6_c31_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c31_pytest.log)
tail -12 gpurun_out/r6_c31_pytest.log
