"""per-dispatch durations of the binned-relaxation kernels from a rocprofv3 --kernel-trace run:
    python tools/relax_trace.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    n = r["Kernel_Name"]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for key, tag in (("sssp_init_kernel", "\nI"), ("sssp_rscatter", "S"), ("sssp_rsweep", "W"), ("sssp_level_kernel", "L"), ("sssp_head", "h")):
        if key in n:
            out.append("%s%.0f" % (tag, us))
print(" ".join(out))
