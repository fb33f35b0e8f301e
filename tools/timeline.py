"""Kernel timeline of the last search in a rocprofv3 --kernel-trace directory:
    python tools/timeline.py <dir with p_kernel_trace.csv> [which search, default -2]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
inits = [i for i, r in enumerate(rows) if "_init_kernel" in r["Kernel_Name"]]
a = inits[which]
b = inits[which + 1] if which + 1 < 0 or which + 1 < len(inits) else len(rows)
if which == -1:
    b = len(rows)
t0 = int(rows[a]["Start_Timestamp"])
prev = None
for r in rows[max(0, a - 1):b - 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("grx::", "").replace("void ", "")[:30]
    print("%-30s start %8.1f dur %7.1f gap %5.1f" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
