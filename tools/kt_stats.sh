#!/bin/bash
# Per-kernel totals of any command under rocprofv3 --kernel-trace --stats:   tools/kt_stats.sh <tag> <command ...>
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/kts_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
timeout ${KT_TIMEOUT:-600} rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o p -- "$@" > "$OUT/log" 2>&1
tail -${KT_LOG_LINES:-6} "$OUT/log"
python - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("NO STATS"); sys.exit(0)
print("| kernel | calls | total ms | avg us | min us | max us | % |")
for r in list(csv.DictReader(open(f[0])))[:24]:
    n = r["Name"].replace("void ", "").replace("grx::", "")[:78]
    print("| %s | %s | %.3f | %.2f | %.2f | %.2f | %s |" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                       float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
PY
find "$OUT" -name "*.csv" -size +2M -delete
