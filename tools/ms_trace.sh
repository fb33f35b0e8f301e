#!/bin/bash
# Kernel sequences of tools/ms_trace.py's searches under rocprofv3 --kernel-trace:  tools/ms_trace.sh <tag> [VAR=value ...]
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/mst_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
echo "== $TAG $*"
env "$@" timeout 150 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o p -- python tools/ms_trace.py lj > "$OUT/log" 2>&1
grep -E "^round" "$OUT/log" || tail -5 "$OUT/log"
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not f:
    print("NO TRACE"); sys.exit(0)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "reset_seed_kernel" in r["Kernel_Name"] or "_init_kernel" in r["Kernel_Name"]]
starts.append(len(rows))
for k in range(3, min(len(starts) - 1, 3 + 16)):  # rounds 0 (new rules) and 1 (old) of 8 sources each, three training searches in front
    a, b = starts[k], starts[k + 1]
    parts = []
    for r in rows[a:b]:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("grx::", "")
        n = n.replace("bfs_", "").replace("_kernel", "").split("<")[0]
        if n.startswith("at::") or n.startswith("__amd"): continue
        parts.append("%s %.0f" % (n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    span = (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
    print("  %s search %d span %.1f us | %s" % ("round0" if k - 3 < 8 else "round1", (k - 3) % 8, span, " ".join(parts)))
PY
rm -rf "$OUT"
