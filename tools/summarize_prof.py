"""Aggregate rocprofv3 CSV output of tools/profile.sh into one markdown table per kernel."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
print("# rocprofv3 summary:", os.path.basename(out))
kt = os.path.join(out, "kt", "p_kernel_stats.csv")
if os.path.exists(kt):
    print("\n## kernel-trace --stats\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in csv.DictReader(open(kt)):
        print("| %s | %s | %.3f | %.2f | %s |" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                 float(r["AverageNs"]) / 1e3, r["Percentage"]))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(os.path.join(out, "pmc*", "p_counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
print("\n## PMC counters (sum over all dispatches of the kernel in that pass; separate pass per group)\n")
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", agg[k].get("FETCH_SIZE", 0))):
    print("### %s\n" % k)
    print("| counter | sum | dispatches | per dispatch |")
    print("|---|---|---|---|")
    for c in sorted(agg[k]):
        n = calls[k][c]
        print("| %s | %.4g | %d | %.4g |" % (c, agg[k][c], n, agg[k][c] / max(n, 1)))
    print()
