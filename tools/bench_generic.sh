#!/bin/bash
# The GENERIC operator API (include/gunrock/framework/operators/**: the templates a user's own algorithm runs on)
# on the C2' LJ stand-in, beside the fused engine and -- when oracle/_ref travels with the tree -- the reference's
# own GPU path on the same arrays.  Outputs gpurun_out/generic_*.  Usage: tools/bench_generic.sh
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SRC=$(python - <<'PY'
import numpy as np, gunrock_amd as gr
from bench import WORKLOADS
wl = WORKLOADS["lj"]
props, csr = gr.generate(wl["kind"], wl["V"], wl["entries"], wl["a"], wl["b"], wl["c"], seed=42)
csr.write_binary("/tmp/lj.csr")
print(int(np.argmax(np.diff(csr.row_offsets))))
PY
)
echo "source $SRC" > gpurun_out/generic_bfs.log
for cfg in "block_mapped" "merge_path" "thread_mapped" "warp_mapped" "bucketing" "merge_path --enable_filter --filter_algorithm compact" "merge_path --enable_filter --filter_algorithm predicated"; do
  echo "== bfs_generic --advance_load_balance $cfg" >> gpurun_out/generic_bfs.log
  timeout 120 bin/bfs_generic --market /tmp/lj.csr --src $SRC -n 5 --advance_load_balance $cfg 2>&1 | grep -i "elapsed\|error" >> gpurun_out/generic_bfs.log
done
echo "== bfs (fused engine, same CLI)" >> gpurun_out/generic_bfs.log
timeout 120 bin/bfs --market /tmp/lj.csr --src $SRC -n 5 --advance_load_balance merge_path --enable_filter --filter_algorithm compact 2>&1 | grep -i "elapsed\|error" >> gpurun_out/generic_bfs.log
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_generic
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_generic -o p -- bin/bfs_generic --market /tmp/lj.csr --src $SRC -n 3 --advance_load_balance merge_path --enable_filter --filter_algorithm compact > gpurun_out/generic_prof.log 2>&1
python - <<'PY' > gpurun_out/generic_kernel_stats.md
import csv, glob
f = glob.glob("gpurun_out/prof_generic/**/p_kernel_stats.csv", recursive=True)
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in csv.DictReader(open(f[0])):
    print("| %s | %s | %.3f | %.2f | %s |" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf gpurun_out/prof_generic /tmp/lj.csr
cat gpurun_out/generic_bfs.log; head -14 gpurun_out/generic_kernel_stats.md
