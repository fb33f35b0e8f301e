#!/bin/bash
# The reference's OWN k-core and personalised-PageRank headers + drivers, unmodified: on the reference's operators (oracle/_ref/ref_*,
# built by `make -C oracle ref_extra` where /root/reference exists) beside the same sources on THIS repo's operators
# (tests/dropin/_build/refalg_*).  Same file, same box; both drivers validate against the reference's CPU code ("Number of errors").
# Output: gpurun_out/refalg_times.txt.  Usage: tools/bench_refalg.sh [log2 V] [entries]
set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
LV=${1:-17}; NE=${2:-2000000}
python - "$LV" "$NE" <<'PY'
import sys
import numpy as np
import gunrock_amd as gr
lv, ne = int(sys.argv[1]), int(sys.argv[2])
_, c = gr.generate("rmat_sym", 1 << lv, ne, seed=11)
ro = np.asarray(c.row_offsets); ci = np.asarray(c.column_indices)
rows = np.repeat(np.arange(len(ro) - 1), np.diff(ro))
with open("/tmp/refalg.mtx", "w") as f:
    f.write("%%%%MatrixMarket matrix coordinate pattern general\n%d %d %d\n" % (len(ro) - 1, len(ro) - 1, len(ci)))
    np.savetxt(f, np.stack([rows + 1, ci + 1], 1), fmt="%d")
print("graph: rmat_sym 2^%d vertices, %d directed edges" % (lv, len(ci)))
PY
{
echo "# $(head -c 0 /dev/null)reference algorithm headers + drivers (unmodified): reference operators vs this repo's operators, same file, one MI355X"
for a in kcore ppr; do
  for exe in oracle/_ref/ref_$a tests/dropin/_build/refalg_$a; do
    if [ -x $exe ]; then
      for rep in 1 2; do
        echo "== $exe (run $rep)"; timeout 300 $exe /tmp/refalg.mtx > /tmp/refalg.out 2>&1; echo "rc $?"; grep -i "GPU Elapsed\|Number of errors\|error" /tmp/refalg.out | head -4 || tail -3 /tmp/refalg.out
      done
    else echo "== $exe: not built"; fi
  done
done
} > gpurun_out/refalg_times.txt 2>&1
rm -f /tmp/refalg.mtx
cat gpurun_out/refalg_times.txt
