cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tl_road -o p -- python tools/run_algo.py bfs road 2 0 merge_path forward > gpurun_out/tl_road.log 2>&1
grep algo gpurun_out/tl_road.log | cut -c1-200
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/tl_road/**/p_kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:50], r["Calls"], r["AverageNs"], r["Percentage"])
rows=sorted(csv.DictReader(open(glob.glob("gpurun_out/tl_road/**/p_kernel_trace.csv",recursive=True)[0])), key=lambda r:int(r["Start_Timestamp"]))
# middle of the second search: print 12 consecutive kernels
mid=len(rows)*3//4
prev=None
for r in rows[mid:mid+12]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    print(r["Kernel_Name"].split("(")[0][-30:], "dur %.1f gap %.1f"%((e-s)/1e3,(s-prev)/1e3 if prev else 0)); prev=e
PY
rm -rf gpurun_out/tl_road
