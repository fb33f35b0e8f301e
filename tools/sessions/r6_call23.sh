#!/bin/bash
# round 6, call 23: what the host does in the 40 ms in front of a generic block_mapped BFS run (HIP API trace)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
SRC=$(python - <<'PY'
import numpy as np, gunrock_amd as gr
from bench import WORKLOADS
wl = WORKLOADS["lj"]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
csr.write_binary("/tmp/lj.csr")
print(int(np.argmax(np.diff(csr.row_offsets))))
PY
)
cd /tmp
for lb in block_mapped merge_path; do
  rm -rf /tmp/hp_$lb
  timeout 200 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/hp_$lb -o p -- $GRAFT_REPO_ROOT/bin/bfs_generic --market /tmp/lj.csr --src $SRC -n 3 --advance_load_balance $lb 2>&1 | grep -i "elapsed"
  echo "== $lb: HIP API stats"; f=$(find /tmp/hp_$lb -name "*hip_api_stats.csv" | head -1); head -14 "$f" | cut -c1-150
  python - "$lb" <<'PY'
import csv, glob, sys
lb = sys.argv[1]
f = glob.glob("/tmp/hp_%s/**/*hip_api_trace.csv" % lb, recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print("   long HIP calls (> 2 ms):")
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 2.0:
        print("   t=%9.2f ms  %7.2f ms  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d, r["Function"]))
PY
done
