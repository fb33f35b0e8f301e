#!/bin/bash
# round 6, call 32: the GPU suite once more on the final tree (flakiness check), smoke
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r6_c32_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c32_pytest.log)
tail -9 gpurun_out/r6_c32_pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
