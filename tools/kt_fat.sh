#!/bin/bash
# Kernel durations of the last forward search of tools/fwd_loop.py under rocprofv3 --kernel-trace, one line per configuration:
#   tools/kt_fat.sh <tag> [VAR=value ...]        (environment of the traced command)
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/kt_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
env "$@" timeout 120 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o p -- python tools/fwd_loop.py ${KT_GRAPH:-lj} 10 ${KT_DIR:-fwd} > "$OUT/log" 2>&1
python - "$OUT" "$TAG $*" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not f:
    print(sys.argv[2], "NO TRACE"); sys.exit(0)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "reset_seed_kernel" in r["Kernel_Name"] or "_init_kernel" in r["Kernel_Name"] or "fwd_start_kernel" in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
parts = []
for r in rows[a:b]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("grx::", "")
    n = n.replace("bfs_", "").replace("_kernel", "").split("<")[0]
    parts.append("%s %.0f" % (n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
span = (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3
print("%-58s span %.1f us | %s" % (sys.argv[2], span, " ".join(parts)))
PY
rm -rf "$OUT"
