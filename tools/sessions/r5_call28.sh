#!/bin/bash
# round 5, call 28: Beamer's alpha (top-down -> bottom-up when frontier edges > unexplored edges / alpha) on single- and multi-source
# direction-optimising searches (LJ, deep, twitter stand-ins); PageRank hot head with 4 source blocks; CLI tests on the new operators
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 600 python -m pytest -q -m gpu tests/test_cli.py > gpurun_out/r5c28_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c28_pytest.log)
el "pytest: $(tail -3 gpurun_out/r5c28_pytest.log | tr '\n' ' ')"
{
for a in 14 28 56 112; do
  echo "== GRX_DO_ALPHA=$a"
  GRX_DO_ALPHA=$a timeout 300 python bench.py --only bfs_do,multi,bfs_deep,c5 --no-cpu-baseline --steps 10 > /tmp/b.log 2>/tmp/b.err
  python - <<'PY'
import json
d = json.loads(open("/tmp/b.log").read().split("\n")[0])
s = d["config"]["sections"]
print("  bfs_do", s.get("bfs_do", {}).get("ms"), "| multi do_mteps", s.get("multi_source", {}).get("do_mteps"), "fwd", s.get("multi_source", {}).get("forward_mteps"),
      "| deep do_ms", s.get("bfs_deep", {}).get("do_ms"), "| c5 do_ms", s.get("c5_1gpu", {}).get("do_ms"), "viol", s.get("multi_source", {}).get("violations"))
PY
done
} > gpurun_out/r5c28_alpha.log 2>&1
el "alpha"
{
for h in 1536 3072 6144; do timeout 100 python tools/ab_pr5.py lj "hot=$h" 2>&1 | grep "^lib"; done
} > gpurun_out/r5c28_pr_hot.log 2>&1
el "pr hot"
tail -4 gpurun_out/r5c28_pytest.log | cut -c1-200; cat gpurun_out/r5c28_alpha.log; cut -c1-200 gpurun_out/r5c28_pr_hot.log
