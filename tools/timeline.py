"""Kernel timeline of one search in a rocprofv3 --kernel-trace directory:
    python tools/timeline.py <dir with p_kernel_trace.csv> [which search, default -2] [kernel-name substring the search must contain]
A search starts at an *_init_kernel; with a substring, `which` indexes the searches that launch such a kernel."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
need = sys.argv[3] if len(sys.argv) > 3 else None
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
inits = [i for i, r in enumerate(rows) if "_init_kernel" in r["Kernel_Name"] or "reset_seed_kernel" in r["Kernel_Name"] or "fwd_start_kernel" in r["Kernel_Name"]]
spans = [(a, inits[k + 1] if k + 1 < len(inits) else len(rows)) for k, a in enumerate(inits)]
if need:
    spans = [(a, b) for a, b in spans if any(need in r["Kernel_Name"] for r in rows[a:b])]
a, b = spans[which]
t0 = int(rows[a]["Start_Timestamp"])
prev = None
busy = 0
for r in rows[max(0, a - 1):b - 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("grx::", "").replace("void ", "")[:34]
    print("%-34s start %8.1f dur %7.1f gap %5.1f" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
    busy += e - s
print("search %d of %d%s: span %.1f us, kernel time %.1f us" % (which, len(spans), " with " + need if need else "", (prev - t0) / 1e3, busy / 1e3))
