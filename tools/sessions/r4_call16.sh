#!/bin/bash
# round 4, call 16: kernel statistics of the generic BFS with merge_path and NO filter (9 ms against the reference's 5.4 ms)
mkdir -p gpurun_out
python - <<'PY'
import numpy as np, gunrock_amd as gr
from bench import WORKLOADS
wl = WORKLOADS["lj"]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
csr.write_binary("/tmp/lj.csr")
open("/tmp/lj.src", "w").write(str(int(np.argmax(np.diff(csr.row_offsets)))))
PY
SRC=$(cat /tmp/lj.src)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pg; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -o p -- $R/bin/bfs_generic --market /tmp/lj.csr --src $SRC -n 3 --advance_load_balance merge_path > $R/gpurun_out/r4c16_run.log 2>&1
python - <<'PY' > $R/gpurun_out/r4c16_kernel_stats.md
import csv, glob
f = glob.glob("/tmp/pg/**/p_kernel_stats.csv", recursive=True)
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in csv.DictReader(open(f[0])):
    print("| %s | %s | %.3f | %.2f | %s |" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
t = glob.glob("/tmp/pg/**/p_kernel_trace.csv", recursive=True)
rows = sorted(csv.DictReader(open(t[0])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print("\nlast run, per dispatch: start us (relative), duration us, kernel")
last = [i for i, r in enumerate(rows) if "fill_kernel" in r["Kernel_Name"] or "sequence" in r["Kernel_Name"]]
for r in rows[-90:]:
    print("%10.1f %8.1f %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:90]))
PY
cat $R/gpurun_out/r4c16_kernel_stats.md | cut -c1-220
