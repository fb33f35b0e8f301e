#!/bin/bash
# round 5, call 33: thin levels on ~one workgroup per three chunks (thin_workgroups) -- parity, kernel sequences, divisors 2 / 3 / 4 / off
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$(date +%s); el() { echo "[$(( $(date +%s) - T0 )) s] $*"; }
(timeout 700 python -m pytest -q -x -m gpu tests/test_bfs_gpu.py tests/test_target_matrix_gpu.py tests/test_fuzz_gpu.py tests/test_sssp_gpu.py -k "not twitter and not pr" > gpurun_out/r5c33_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r5c33_pytest.log)
el "pytest: $(tail -3 gpurun_out/r5c33_pytest.log | tr '\n' ' ')"
{
for g in lj deep; do
  for k in 3 0 2 4 6; do KT_GRAPH=$g bash tools/kt_fat.sh ${g}_div$k GRX_THIN_CHUNKS_PER_WG=$k; done
done
KT_DIR=do KT_GRAPH=deep bash tools/kt_fat.sh do_deep_div3
KT_DIR=do KT_GRAPH=deep bash tools/kt_fat.sh do_deep_off GRX_THIN_CHUNKS_PER_WG=0
KT_DIR=do bash tools/kt_fat.sh do_lj_div3
for v in "GRX_THIN_CHUNKS_PER_WG=3" "GRX_THIN_CHUNKS_PER_WG=0"; do
  echo "== $v"
  env $v timeout 200 python bench.py --only bfs,bfs_do,multi,bfs_deep --no-cpu-baseline --steps 10 > /tmp/b.log 2>/tmp/b.err
  python - <<'PY'
import json
d = json.loads(open("/tmp/b.log").read().split("\n")[0])
s = d["config"]["sections"]
print("  bench: fwd ms", d["ms_per_step"], "| do", s.get("bfs_do", {}).get("ms"), "| multi fwd", s.get("multi_source", {}).get("forward_mteps"), "do", s.get("multi_source", {}).get("do_mteps"),
      "| deep fwd_ms", s.get("bfs_deep", {}).get("fwd_ms"), "do_ms", s.get("bfs_deep", {}).get("do_ms"), "thin", s.get("bfs_deep", {}).get("us_per_thin_level"))
PY
done
} > gpurun_out/r5c33_ab.log 2>&1
el "ab"
tail -4 gpurun_out/r5c33_pytest.log | cut -c1-200; cut -c1-400 gpurun_out/r5c33_ab.log
