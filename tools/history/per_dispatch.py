"""Per-dispatch PMC view of one kernel:  python tools/per_dispatch.py <prof dir> <kernel substring> [last N dispatches]
Rows = dispatches in launch order (of the LAST N, default 12), columns = counters of all passes."""
import collections
import csv
import glob
import os
import sys

out, pat = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 12
cols = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, "pmc*", "p_counter_collection.csv"))):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    ids = sorted(per)[-last:]
    for c in sorted({c for i in ids for c in per[i]}):
        cols[c] = [per[i].get(c, 0.0) for i in ids]
kt = glob.glob(os.path.join(out, "kt", "p_kernel_trace.csv"))
if kt:
    rows = [r for r in csv.DictReader(open(kt[0])) if pat in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    cols["duration_us"] = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-last:]]
names = list(cols)
print("| dispatch | " + " | ".join(names) + " |")
print("|---|" + "---|" * len(names))
n = max(len(v) for v in cols.values())
for i in range(n):
    print("| %d | " % i + " | ".join("%.4g" % cols[c][i] if i < len(cols[c]) else "" for c in names) + " |")
