#!/bin/bash
# round 6, call 22: block_mapped / bucketing at 38-41 ms in call 21 (4.5 / 3.3 in round 5): repeat, with a kernel trace
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
for cfg in block_mapped bucketing warp_mapped block_mapped; do
  echo "== $cfg"; timeout 120 bin/bfs_generic --market /tmp/lj.csr --src $SRC -n 5 --advance_load_balance $cfg 2>&1 | grep -i "elapsed\|error"
done
KT_LOG_LINES=2 bash tools/kt_stats.sh blk bin/bfs_generic --market /tmp/lj.csr --src $SRC -n 3 --advance_load_balance block_mapped
KT_LOG_LINES=2 bash tools/kt_stats.sh wrp bin/bfs_generic --market /tmp/lj.csr --src $SRC -n 3 --advance_load_balance warp_mapped
