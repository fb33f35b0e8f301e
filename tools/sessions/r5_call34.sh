#!/bin/bash
# round 5, call 34: thin_workgroups() second rule (1 chunk per workgroup up to 768 chunks, 2.5 from 1280) against off
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for g in lj deep kron; do
  for k in 1 0; do KT_GRAPH=$g bash tools/kt_fat.sh ${g}_thin$k GRX_THIN_CHUNKS_PER_WG=$k; done
done
for g in lj deep; do for k in 1 0; do KT_DIR=do KT_GRAPH=$g bash tools/kt_fat.sh do_${g}_thin$k GRX_THIN_CHUNKS_PER_WG=$k; done; done
for v in "GRX_THIN_CHUNKS_PER_WG=1" "GRX_THIN_CHUNKS_PER_WG=0" "GRX_THIN_CHUNKS_PER_WG=1"; do
  echo "== $v"
  env $v timeout 200 python bench.py --only bfs,bfs_do,multi,bfs_deep,c5 --no-cpu-baseline --steps 10 > /tmp/b.log 2>/tmp/b.err
  python - <<'PY'
import json
d = json.loads(open("/tmp/b.log").read().split("\n")[0])
s = d["config"]["sections"]
print("  bench: fwd ms", d["ms_per_step"], "| do", s.get("bfs_do", {}).get("ms"), "| multi fwd", s.get("multi_source", {}).get("forward_mteps"), "do", s.get("multi_source", {}).get("do_mteps"),
      "| deep fwd_ms", s.get("bfs_deep", {}).get("fwd_ms"), "do_ms", s.get("bfs_deep", {}).get("do_ms"), "thin", s.get("bfs_deep", {}).get("us_per_thin_level"), "| c5 fwd", s.get("c5_1gpu", {}).get("fwd_ms"), "do", s.get("c5_1gpu", {}).get("do_ms"))
PY
done
} > gpurun_out/r5c34_ab.log 2>&1
cut -c1-400 gpurun_out/r5c34_ab.log
