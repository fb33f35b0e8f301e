#!/bin/bash
# round 5, call 29: thin claim-per-edge levels append the next level's chunk map themselves (ctrl.mode 4) -- parity, then kernel
# sequences and step times with the switch off (GRX_MAP_MAX_EDGES=0) and at two thresholds
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 700 python -m pytest -q -x -m gpu tests/test_bfs_gpu.py tests/test_target_matrix_gpu.py tests/test_fuzz_gpu.py tests/test_sssp_gpu.py -k "not twitter and not pr" > gpurun_out/r5c29_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c29_pytest.log)
el "pytest: $(tail -3 gpurun_out/r5c29_pytest.log | tr '\n' ' ')"
{
bash tools/kt_fat.sh lj_default
bash tools/kt_fat.sh lj_off GRX_MAP_MAX_EDGES=0
KT_GRAPH=deep bash tools/kt_fat.sh deep_default
KT_GRAPH=deep bash tools/kt_fat.sh deep_off GRX_MAP_MAX_EDGES=0
KT_GRAPH=deep bash tools/kt_fat.sh deep_2M GRX_MAP_MAX_EDGES=2097152
for v in "GRX_MAP_MAX_EDGES=1048576" "GRX_MAP_MAX_EDGES=0" "GRX_MAP_MAX_EDGES=2097152" "GRX_MAP_MAX_EDGES=262144"; do
  echo "== $v"
  env $v GRX_KEEP=GRX_MAP_MAX_EDGES timeout 150 python tools/ab_r5.py lj 20 bfs,multi 2>&1 | grep -v amdgpu.ids | grep "^fwd default  \|other sources"
  env $v timeout 200 python bench.py --only bfs,multi,bfs_deep --no-cpu-baseline --steps 10 > /tmp/b.log 2>/tmp/b.err
  python - <<'PY'
import json
d = json.loads(open("/tmp/b.log").read().split("\n")[0])
s = d["config"]["sections"]
print("  bench: fwd ms", d["ms_per_step"], "| multi fwd", s.get("multi_source", {}).get("forward_mteps"), "cold", s.get("multi_source", {}).get("forward_cold_mteps"),
      "| deep fwd_ms", s.get("bfs_deep", {}).get("fwd_ms"), "thin", s.get("bfs_deep", {}).get("us_per_thin_level"))
PY
done
} > gpurun_out/r5c29_ab.log 2>&1
el "ab"
tail -4 gpurun_out/r5c29_pytest.log | cut -c1-200; cut -c1-400 gpurun_out/r5c29_ab.log
