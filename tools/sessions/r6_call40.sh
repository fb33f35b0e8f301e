#!/bin/bash
# round 6, call 40: what a k-core iteration on the generic operators is made of (HIP API + kernel statistics of the reference's driver on our operators)
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY'
import numpy as np, gunrock_amd as gr
_, c = gr.generate("rmat_sym", 1 << 17, 2000000, seed=11)
ro = np.asarray(c.row_offsets); ci = np.asarray(c.column_indices)
rows = np.repeat(np.arange(len(ro) - 1), np.diff(ro))
with open("/tmp/refalg.mtx", "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate pattern general\n%d %d %d\n" % (len(ro) - 1, len(ro) - 1, len(ci)))
    np.savetxt(f, np.stack([rows + 1, ci + 1], 1), fmt="%d")
PY
cd /tmp; rm -rf /tmp/hp_kcore
timeout 300 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/hp_kcore -o p -- $GRAFT_REPO_ROOT/tests/dropin/_build/refalg_kcore /tmp/refalg.mtx 2>&1 | grep -i "elapsed\|errors"
f=$(find /tmp/hp_kcore -name "*hip_api_stats.csv" | head -1); echo "== HIP API"; head -12 "$f" | cut -d, -f1-4
f=$(find /tmp/hp_kcore -name "*kernel_stats.csv" | head -1); echo "== kernels"; head -16 "$f" | cut -d, -f1-4 | cut -c1-140
