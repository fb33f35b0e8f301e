#!/bin/bash
# round 6, call 7: the whole GPU suite + smoke on the tree of commit "hub-first in-rows" (baseline of the second session)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 > gpurun_out/r6_c7_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6_c7_pytest.log); el pytest
tail -20 gpurun_out/r6_c7_pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6_c7_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r6_c7_smoke.log; el smoke
tail -3 gpurun_out/r6_c7_smoke.log
