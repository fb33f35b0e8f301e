#!/bin/bash
# round 6, call 34: per-launch durations of the generic advance kernels on the LJ stand-in (warp / thread / block mapped, merge path)
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
for lb in warp_mapped thread_mapped block_mapped merge_path; do
  rm -rf /tmp/kt_$lb
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$lb -o p -- $GRAFT_REPO_ROOT/bin/bfs_generic --market /tmp/lj.csr --src $SRC -n 1 --advance_load_balance $lb > /dev/null 2>&1
  python - "$lb" <<'PY'
import csv, glob, sys
lb = sys.argv[1]
f = glob.glob("/tmp/kt_%s/**/*kernel_trace.csv" % lb, recursive=True)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); prev = None; out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    short = "ADV" if "advance::" in n and "kernel<" in n and "degrees" not in n else ("deg" if "degrees" in n else ("scan" if "scan_" in n else ("copy" if "copyBuffer" in n else ("fill" if "fill" in n else n[:12]))))
    gap = (s - prev) / 1e3 if prev else 0
    out.append("%s%s %.0f" % ("[gap %.0f] " % gap if gap > 100 else "", short, (e - s) / 1e3))
    prev = e
print("== %s span %.0f us: %s" % (lb, (prev - t0) / 1e3, " | ".join(out)))
PY
done > $GRAFT_REPO_ROOT/gpurun_out/r6_c34_generic_timeline.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r6_c34_generic_timeline.txt | cut -c1-1800
