#!/bin/bash
# Kernel names + durations (us) of the LAST call of a looped command under rocprofv3 --kernel-trace:
#   tools/kt_last.sh <tag> <first-kernel-substring of a call> -- <command ...>
set -u
TAG=$1; FIRST=$2; shift 3
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ktl_$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
timeout 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o p -- "$@" > "$OUT/log" 2>&1
python - "$OUT" "$TAG" "$FIRST" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if not f:
    print(sys.argv[2], "NO TRACE"); sys.exit(0)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if sys.argv[3] in r["Kernel_Name"]]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
agg = collections.OrderedDict()
seq = []
for r in rows[a:b]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("grx::", "").split("<")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(n, [0, 0.0]); agg[n][0] += 1; agg[n][1] += d
    seq.append("%s %.0f" % (n.replace("_kernel", ""), d))
span = (int(rows[b - 1]["End_Timestamp"]) - t0) / 1e3
print("%s: span %.1f us, kernel time %.1f us" % (sys.argv[2], span, sum(v[1] for v in agg.values())))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-34s x%-4d %9.1f us" % (n, c, t))
print("  sequence: " + " | ".join(seq[:120]))
PY
rm -rf "$OUT"
